"""Row-tile sharding of one frame over the ranks of a torch.distributed job (one process per GPU).

Pixels are independent given the segment set (DrawStandard's loop body, skeleton.cpp:336-362, reads only
the tree, the triangles and the beams), so the image is cut into 8-row strips and strip s is rendered by
rank s % world: interleaved, because the lit cone is centred and rows differ in cost by more than 10x.
Two exchange steps exist: the segment set is broadcast once from the rank that emitted it, and the
rendered strips are all-gathered per frame.  Nothing here touches the GPU path itself, so the same code
runs over RCCL ("nccl" backend, device tensors) and over gloo (CPU tensors, used by the tests)."""
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
