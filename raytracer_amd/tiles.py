"""Row-tile sharding of one frame over the ranks of a torch.distributed job (one process per GPU).

Pixels are independent given the segment set (DrawStandard's loop body, skeleton.cpp:336-362, reads only
the tree, the triangles and the beams), so the image is cut into 8-row strips and strip s is rendered by
rank s % world: interleaved, because the lit cone is centred and rows differ in cost by more than 10x.
Two exchange steps exist: the segment set is broadcast once from the rank that emitted it, and the
rendered strips are all-gathered per frame.  Nothing here touches the GPU path itself, so the same code
runs over RCCL ("nccl" backend, device tensors) and over gloo (CPU tensors, used by the tests)."""
import datetime

import numpy as np
import torch
import torch.distributed as dist

STRIP_ROWS = 8


def strips_of_rank(height, rank, world):
    """(first_strip, strip_stride, n_strips) of `rank`: strips first_strip + k*strip_stride, k < n_strips."""
    if height % STRIP_ROWS:
        raise ValueError(f"image height {height} is not a multiple of {STRIP_ROWS}")
    total = height // STRIP_ROWS
    if total % world:
        raise ValueError(f"{total} strips do not split evenly over {world} ranks")
    return rank, world, total // world


def rows_of_rank(height, rank, world):
    """Image rows of `rank`, in the order its strips are packed."""
    first, stride, n = strips_of_rank(height, rank, world)
    return np.concatenate([np.arange(STRIP_ROWS) + STRIP_ROWS * (first + k * stride) for k in range(n)])


def assemble(gathered, frame):
    """gathered[r] = packed strips of rank r, shape (world, n*8, W, C)  ->  frame (H, W, C), in place."""
    world, rows, width, ch = gathered.shape
    n = rows // STRIP_ROWS
    frame.view(n, world, STRIP_ROWS, width, ch).copy_(gathered.view(world, n, STRIP_ROWS, width, ch).transpose(0, 1))
    return frame


def gather_frame(mine, gathered, frame, world):
    """All-gather the packed strips of every rank and put them in image order."""
    if world == 1:
        frame.copy_(mine)
        return frame
    # concatenated along dim 0: the one output layout every backend (RCCL, gloo) accepts
    dist.all_gather_into_tensor(gathered.view(-1, *gathered.shape[2:]), mine)
    return assemble(gathered, frame)


class StripStall(RuntimeError):
    """A strip gather did not finish within the pipeline's time limit."""


class StripPipeline:
    """All-gather of the ranks' strips, pipelined where the backend allows it.

    Over RCCL ("nccl" backend, device tensors) the gather of frame k runs on the collective's own stream
    while frame k + 1 is rendered (two slots), so the exchange is hidden behind compute; a gather that
    never finishes is ended by the process group's own time limit (init_process_group(timeout=...)).
    Over any other backend (gloo: host tensors, and device tensors only through staging copies that
    have been seen to stall with more than one gather outstanding) every gather goes through host
    buffers, one at a time, and waits at most `timeout_s` before raising StripStall with rank, slot and
    frame.  Every frame is rendered, gathered and put in image order either way; only the order of
    the waits differs.

        pipe = StripPipeline(world, rows_per_rank, width, device)
        for k in range(frames):
            slot = pipe.begin(frame)          # finishes the frame that used this slot `depth` steps ago
            render into pipe.mine[slot]
            pipe.submit(slot)
        pipe.drain(frame)
    """

    def __init__(self, world, rows, width, device, channels=3, depth=2, frame=None, timeout_s=120.0, exchange=False,
                 frames_per_gather=1):
        """exchange=True runs the collective path even in a job of one rank (a rehearsal of the N > 1 code on one
        GPU: the gather then has one participant, everything else is as with several).
        frames_per_gather = K > 1: a slot holds the strips of K consecutive frames and one collective carries them
        all (a frame of 0.03 ms is shorter than what a collective costs the host to enqueue); every frame is still
        rendered, gathered and put in image order - K at a time, into self.batch[slot] (K, H, W, C) - and `frame`
        receives the last of them.  Render frame i into target(slot), not mine[slot]."""
        self.world = world
        self.timeout_s = timeout_s
        self.K = K = max(1, int(frames_per_gather))
        self.collective = world > 1 or (exchange and dist.is_initialized())
        backend = dist.get_backend() if (self.collective and dist.is_initialized()) else None
        self.overlapped = backend == "nccl"
        self.depth = depth if (not self.collective or self.overlapped) else 1
        # one rank owns every strip: it renders straight into the frame, nothing to exchange or copy
        self.direct = not self.collective and frame is not None and tuple(frame.shape) == (rows, width, channels)
        if self.direct:
            self.K = K = 1
        lead = (K,) if K > 1 else ()
        self.mine = [frame if self.direct else torch.zeros(lead + (rows, width, channels), dtype=torch.float32, device=device)
                     for _ in range(self.depth)]
        self.gathered = [torch.zeros((world,) + lead + (rows, width, channels), dtype=torch.float32, device=device)
                         if self.collective else None for _ in range(self.depth)]
        self.staged = self.collective and not self.overlapped
        if self.staged:
            self.h_mine = torch.zeros(lead + (rows, width, channels), dtype=torch.float32)
            self.h_gathered = torch.zeros((world,) + lead + (rows, width, channels), dtype=torch.float32)
        # the views a step needs, made once: at 0.04 ms a frame the host's time per step is what bounds the rate
        n = rows // STRIP_ROWS
        self.gathered_flat = [g.view(-1, *g.shape[-3:]) if g is not None else None for g in self.gathered]
        if K == 1:
            self.gathered_strips = [g.view(world, n, STRIP_ROWS, width, channels).transpose(0, 1)
                                    if g is not None else None for g in self.gathered]
            self.batch = [None] * self.depth
        else:
            # (world, K, n, 8, W, C) -> (K, n, world, 8, W, C): all K frames in image order with one strided copy
            self.gathered_strips = [g.view(world, K, n, STRIP_ROWS, width, channels).permute(1, 2, 0, 3, 4, 5)
                                    if g is not None else None for g in self.gathered]
            self.batch = [torch.zeros((K, world * rows, width, channels), dtype=torch.float32, device=device)
                          for _ in range(self.depth)]
            self.batch_strips = [b.view(K, n, world, STRIP_ROWS, width, channels) for b in self.batch]
        self.frame_strips = None        # (frame it was made for, its view as (strips per rank, rank, 8, W, C))
        self.work = [None] * self.depth
        self.filled = [0] * self.depth  # frames rendered into the slot since it was last completed
        self.sent = [False] * self.depth
        self.k = 0

    def target(self, slot):
        """Where this rank's packed strips of the next frame go."""
        return self.mine[slot] if self.K == 1 else self.mine[slot][self.filled[slot]]

    def _assemble(self, slot, frame):
        if self.K > 1:
            self.batch_strips[slot].copy_(self.gathered_strips[slot])
            frame.copy_(self.batch[slot][self.filled[slot] - 1])
            return
        if self.frame_strips is None or self.frame_strips[0] is not frame:
            g = self.gathered[slot]
            self.frame_strips = (frame, frame.view(g.shape[1] // STRIP_ROWS, g.shape[0], STRIP_ROWS, g.shape[2], g.shape[3]))
        self.frame_strips[1].copy_(self.gathered_strips[slot])

    def _finish(self, slot, frame):
        if not self.filled[slot]:
            return
        if not self.sent[slot]:
            self._send(slot)              # a batch cut short by the end of the loop
        if not self.collective:
            if not self.direct:
                frame.copy_(self.mine[slot] if self.K == 1 else self.mine[slot][self.filled[slot] - 1])
        else:
            if self.work[slot] is not None:
                self.work[slot].wait()        # RCCL: the current stream waits for that gather; the host does not
                self.work[slot] = None
            self._assemble(slot, frame)
        self.filled[slot] = 0
        self.sent[slot] = False

    def begin(self, frame):
        slot = (self.k // self.K) % self.depth
        if self.filled[slot] == 0 or self.sent[slot]:
            self._finish(slot, frame)
        return slot

    def submit(self, slot):
        """Frame k is in target(slot): counted, and the slot's collective started once it holds K frames."""
        self.filled[slot] += 1
        self.k += 1
        if self.filled[slot] == self.K:
            self._send(slot)

    def _send(self, slot):
        self.sent[slot] = True
        if self.collective:
            g = self.gathered[slot]
            if self.overlapped:
                self.work[slot] = dist.all_gather_into_tensor(self.gathered_flat[slot], self.mine[slot].view(-1, *self.mine[slot].shape[-3:]),
                                                              async_op=True)
            else:
                # host-staged and complete before submit returns: never more than one gather outstanding
                self.h_mine.copy_(self.mine[slot])
                work = dist.all_gather_into_tensor(self.h_gathered.view(-1, *self.h_gathered.shape[-3:]),
                                                   self.h_mine.view(-1, *self.h_mine.shape[-3:]), async_op=True)
                try:
                    done = work.wait(timeout=datetime.timedelta(seconds=self.timeout_s))
                except RuntimeError as e:
                    raise StripStall(f"rank {dist.get_rank()}: strip gather of frame {self.k - 1} (slot {slot}) failed or "
                                     f"timed out after {self.timeout_s:.0f} s: {e}") from e
                if done is False:
                    raise StripStall(f"rank {dist.get_rank()}: strip gather of frame {self.k - 1} (slot {slot}) did not "
                                     f"finish within {self.timeout_s:.0f} s")
                self.gathered[slot].copy_(self.h_gathered)

    def drain(self, frame):
        first = ((self.k + self.K - 1) // self.K) % self.depth      # the oldest slot still in flight
        for d in range(self.depth):
            self._finish((first + d) % self.depth, frame)


def run_steps(pipe, frame, render, steps, first=0, on_step=None):
    """The frame loop of bench.py: `steps` frames through `pipe`, frame i rendered by render(mine, i) into this
    rank's packed strips, then every frame still in flight completed.  on_step(i) runs after frame i has been
    submitted (bench.py re-arms its standstill watchdog there)."""
    for i in range(first, first + steps):
        slot = pipe.begin(frame)
        render(pipe.target(slot), i)
        pipe.submit(slot)
        if on_step is not None:
            on_step(i)
    pipe.drain(frame)


def broadcast_segments(segs, root_box, device, src=0, itemsize=112):
    """Broadcast the segment records (numpy structured array on `src`, None elsewhere) and the 6 floats of
    the root box.  Returns (byte tensor on `device` holding the records, n_segments, root_box as numpy)."""
    rank = dist.get_rank()
    meta = torch.zeros(8, dtype=torch.float64, device=device)
    if rank == src:
        meta[:6] = torch.as_tensor(np.asarray(root_box, dtype=np.float64))
        meta[6] = float(len(segs))
    dist.broadcast(meta, src)
    n = int(meta[6].item())
    buf = torch.empty(max(n, 1) * itemsize, dtype=torch.uint8, device=device)
    if rank == src and n:
        buf[:n * itemsize].copy_(torch.from_numpy(np.ascontiguousarray(segs).view(np.uint8).reshape(-1)))
    dist.broadcast(buf, src)
    return buf, n, meta[:6].cpu().numpy().astype(np.float32)
