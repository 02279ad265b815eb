"""Host-side sharding of the frame over GPUs: one process per GPU, rank r of k
owns the rows y with (y // strip_rows) % k == r; pixelIndex stays global so RNG keys -- and
therefore every radiance value -- are identical to the 1-GPU run.  The only collective on the
path is the sum of the zero-padded float3 accumulation buffers onto rank 0 (RCCL on GPUs, gloo
in the CPU tests); adding zeros is exact, so the reduced frame is bit-identical to a 1-GPU frame.
The same arithmetic lives in csrc/ptmi355.hip:local_to_pixel / tile_rows."""
import numpy as np


def owned_rows(rank, world, strip_rows, height):
    """Boolean mask over rows."""
    y = np.arange(height)
    if world <= 1:
        return np.ones(height, dtype=bool)
    return (y // strip_rows) % world == rank


def tile_pixel_indices(rank, world, strip_rows, width, height):
    """Global pixelIndex (x + y*W) of every pixel this rank owns, in the library's local order."""
    rows = np.nonzero(owned_rows(rank, world, strip_rows, height))[0]
    return (rows[:, None] * width + np.arange(width)[None, :]).reshape(-1).astype(np.int64)


def step_iterations(step, batch, world):
    """(iter0, count) traced by EVERY rank at `step`: per-GPU work is fixed (weak scaling), so a
    rank that owns 1/world of the pixels traces batch*world iterations of them per step."""
    count = batch * world
    return 1 + step * count, count


def reduce_frame(dist, image, frame, dst=0):
    """Sum the ranks' zero-padded accumulation buffers onto `dst`.  `frame` is a staging copy so
    that the running sums in `image` stay per-rank (a reduce is in place on the destination)."""
    frame.copy_(image)
    dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame
