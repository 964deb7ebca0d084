"""The N > 1 path of bench.py on CPU: two and four gloo ranks shard a frame into interleaved strips,
all-gather them and reassemble the image; the segment broadcast carries the records byte for byte."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import tiles

SEG_DTYPE = np.dtype([("raw", "u1", 112)])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, height, width, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the "renderer": pixel (y, x, c) = 1000*y + x + c/10, so every misplaced row shows
        yy, xx, cc = np.meshgrid(np.arange(height), np.arange(width), np.arange(3), indexing="ij")
        full = (1000.0 * yy + xx + cc / 10.0).astype(np.float32)
        rows = tiles.rows_of_rank(height, rank, world)
        first, stride, n = tiles.strips_of_rank(height, rank, world)
        assert (first, stride) == (rank, world) and len(rows) == n * 8
        mine = torch.from_numpy(full[rows].copy())
        gathered = torch.zeros((world, n * 8, width, 3), dtype=torch.float32)
        frame = torch.zeros((height, width, 3), dtype=torch.float32)
        tiles.gather_frame(mine, gathered, frame, world)
        assert np.array_equal(frame.numpy(), full), "reassembled frame differs"
        # pipelined gather: forty "frames" (frame k = full + k); over gloo the pipeline keeps one gather in
        # flight and stages it through host buffers, over RCCL two
        pipe = tiles.StripPipeline(world, n * 8, width, torch.device("cpu"), timeout_s=60)
        assert pipe.depth == 1 and pipe.staged
        seen = []
        for k in range(40):
            slot = pipe.begin(frame)
            if k >= pipe.depth:
                seen.append(frame.numpy().copy())        # frame k - depth has just been completed
            pipe.mine[slot].copy_(torch.from_numpy(full[rows] + np.float32(k)))
            pipe.submit(slot)
        pipe.drain(frame)
        assert np.array_equal(frame.numpy(), full + np.float32(39)), "last pipelined frame differs"
        for k, got in enumerate(seen):
            assert np.array_equal(got, full + np.float32(k)), f"pipelined frame {k} differs"
        # the frame loop bench.py itself runs (tiles.run_steps), 40 frames, with its per-step hook
        pipe = tiles.StripPipeline(world, n * 8, width, torch.device("cpu"), timeout_s=60)
        steps_seen = []
        tiles.run_steps(pipe, frame, lambda mine, i: mine.copy_(torch.from_numpy(full[rows] * np.float32(2) + np.float32(i))),
                        40, on_step=steps_seen.append)
        assert steps_seen == list(range(40))
        assert np.array_equal(frame.numpy(), full * np.float32(2) + np.float32(39)), "run_steps: last frame differs"
        # several frames per collective: frame k = full * 3 + k, seven to a gather (40 is no multiple: the last
        # batch is cut short); every frame of a completed batch is in image order in pipe.batch, the last one in `frame`
        K = 7
        pipe = tiles.StripPipeline(world, n * 8, width, torch.device("cpu"), timeout_s=60, frames_per_gather=K)
        batches = []

        def render_k(mine, i):
            mine.copy_(torch.from_numpy(full[rows] * np.float32(3) + np.float32(i)))
        for i in range(40):
            slot = pipe.begin(frame)
            if i and i % K == 0:
                pass
            render_k(pipe.target(slot), i)
            pipe.submit(slot)
            if pipe.sent[slot] and pipe.depth == 1:
                pipe._finish(slot, frame)             # completed here so that the batch can be looked at
                batches.append(pipe.batch[slot].numpy().copy())
        pipe.drain(frame)
        assert np.array_equal(frame.numpy(), full * np.float32(3) + np.float32(39)), "batched: last frame differs"
        assert len(batches) == 40 // K
        for b, got in enumerate(batches):
            for j in range(K):
                assert np.array_equal(got[j], full * np.float32(3) + np.float32(b * K + j)), f"batched frame {b * K + j} differs"
        pipe = tiles.StripPipeline(world, n * 8, width, torch.device("cpu"), timeout_s=60, frames_per_gather=4)
        tiles.run_steps(pipe, frame, render_k, 10)
        assert np.array_equal(frame.numpy(), full * np.float32(3) + np.float32(9)), "batched run_steps: last frame differs"
        # segment broadcast
        segs = None
        root = None
        if rank == 0:
            rng = np.random.default_rng(5)
            segs = np.zeros(777, SEG_DTYPE)
            segs["raw"] = rng.integers(0, 256, (777, 112), dtype=np.uint8)
            root = [-1, -1, 1.5, 1, 1, 3.5]
        buf, n_seg, root_box = tiles.broadcast_segments(segs, root, torch.device("cpu"))
        assert n_seg == 777 and np.allclose(root_box, [-1, -1, 1.5, 1, 1, 3.5])
        ref = np.random.default_rng(5).integers(0, 256, (777, 112), dtype=np.uint8)
        assert np.array_equal(buf.numpy()[:777 * 112].reshape(777, 112), ref)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_strips_gather_and_broadcast_over_gloo(world, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, 64, 40, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(os.path.join(str(tmp_path), f"ok{r}")) for r in range(world))


def _stall_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe = tiles.StripPipeline(world, 8, 16, torch.device("cpu"), timeout_s=3)
    frame = torch.zeros((16, 16, 3), dtype=torch.float32)
    if rank == 0:
        try:
            tiles.run_steps(pipe, frame, lambda mine, i: mine.fill_(1.0), 1)
            open(os.path.join(out_dir, "no_error"), "w").write("x")
        except tiles.StripStall as e:
            open(os.path.join(out_dir, "stalled"), "w").write(str(e))
    else:
        import time
        time.sleep(8)            # never joins the gather
    os._exit(0)                  # the group holds a gather that cannot complete: leave without tearing it down


def test_a_gather_that_stalls_raises_instead_of_blocking(tmp_path):
    port = _free_port()
    mp.spawn(_stall_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    msg = open(os.path.join(str(tmp_path), "stalled")).read()
    assert "rank 0" in msg and "frame 0" in msg and "slot 0" in msg


def test_strip_bookkeeping():
    assert tiles.strips_of_rank(512, 3, 8) == (3, 8, 8)
    rows = np.concatenate([tiles.rows_of_rank(64, r, 4) for r in range(4)])
    assert sorted(rows.tolist()) == list(range(64))
    assert tiles.rows_of_rank(64, 1, 4)[:9].tolist() == [8, 9, 10, 11, 12, 13, 14, 15, 40]
    with pytest.raises(ValueError):
        tiles.strips_of_rank(100, 0, 2)
    with pytest.raises(ValueError):
        tiles.strips_of_rank(24, 0, 2)
    g = torch.arange(2 * 16 * 3 * 1, dtype=torch.float32).view(2, 16, 3, 1)
    f = tiles.assemble(g, torch.zeros(32, 3, 1))
    assert torch.equal(f[8:16], g[1, :8]) and torch.equal(f[16:24], g[0, 8:])
