"""Run as a program by test_gpu_exchange.py (a process of its own, because it owns an RCCL process group): one rank,
but every frame goes through tiles.StripPipeline's collective path - the all-gather on the collective's stream while
the next frame is rendered - and the frames must equal the ones rendered straight into an image."""
import datetime
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as pb  # noqa: E402
from raytracer_amd import tiles  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29519")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev, timeout=datetime.timedelta(seconds=60))
    W = H = 128
    kw = dict(screen_width=W, screen_height=H, photon_number=400, bounces=1)
    ctx = pb.Context(pb.control_defaults(**kw), device=0)
    ctx.set_scene()
    ctx.cast_photon_beams()
    segs = ctx.download_segments()
    root = np.concatenate(ctx.root_box()).astype(np.float32)
    # the records travel as in a job of several ranks and the frames are rendered from what arrived
    buf, n, root_b = tiles.broadcast_segments(segs, root, dev, itemsize=pb.SEGMENT_DTYPE.itemsize)
    assert n == len(segs) and np.array_equal(root_b, root)
    assert np.array_equal(buf.cpu().numpy()[:n * pb.SEGMENT_DTYPE.itemsize], np.ascontiguousarray(segs).view(np.uint8).reshape(-1))
    ctx.upload_segments_device(buf.data_ptr(), n, root_b[:3], root_b[3:], stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream().cuda_stream
    direct = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    ctx.draw_strips_device(direct.data_ptr(), 0, 1, H // 8, mode=pb.MODE_721, cull="bins", stream=stream)
    torch.cuda.synchronize()
    assert float(direct.max()) > 0

    frame = torch.zeros((H, W, 3), dtype=torch.float32, device=dev)
    pipe = tiles.StripPipeline(1, H, W, dev, frame=frame, timeout_s=60, exchange=True)
    assert pipe.collective and pipe.overlapped and pipe.depth == 2 and not pipe.direct
    seen = []

    def render(mine, i):
        # frame i is the image times (i + 1), so a slot handed back too early or a frame put together from the wrong
        # gather shows
        ctx.draw_strips_device(mine.data_ptr(), 0, 1, H // 8, mode=pb.MODE_721, cull="bins", stream=stream)
        mine.mul_(float(i + 1))

    steps = 40
    for i in range(steps):
        slot = pipe.begin(frame)
        if i >= pipe.depth:
            seen.append((i - pipe.depth, frame.clone()))   # begin() has just completed the frame that used this slot
        render(pipe.mine[slot], i)
        pipe.submit(slot)
    pipe.drain(frame)
    torch.cuda.synchronize()
    assert torch.equal(frame, direct * float(steps)), "last frame differs"
    for k, got in seen:
        assert torch.equal(got, direct * float(k + 1)), f"frame {k} differs"
    # and through the loop bench.py runs
    pipe = tiles.StripPipeline(1, H, W, dev, frame=frame, timeout_s=60, exchange=True)
    tiles.run_steps(pipe, frame, render, 25)
    torch.cuda.synchronize()
    assert torch.equal(frame, direct * 25.0)
    # several frames to a collective (bench.py does that for frames this short): the last frame and every batch
    pipe = tiles.StripPipeline(1, H, W, dev, frame=frame, timeout_s=60, exchange=True, frames_per_gather=4)
    assert pipe.K == 4 and pipe.depth == 2
    tiles.run_steps(pipe, frame, render, 22)            # five batches and a half
    torch.cuda.synchronize()
    assert torch.equal(frame, direct * 22.0), "batched gathers: last frame differs"
    for slot, first in ((0, 16), (1, 20)):               # what the two slots last held: frames 16..19 and 20..21
        for j in range(4 if slot == 0 else 2):
            assert torch.equal(pipe.batch[slot][j], direct * float(first + j + 1)), f"batched frame {first + j} differs"
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    print("exchange rehearsal ok")


if __name__ == "__main__":
    main()
