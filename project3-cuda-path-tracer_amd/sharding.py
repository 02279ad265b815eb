"""Host-side sharding of the frame over GPUs: one process per GPU, rank r of k owns the rows y with
(y // strip_rows) % k == r; pixelIndex stays global so RNG keys -- and therefore every radiance value -- are
identical to the 1-GPU run.  The only collective on the path brings the tiles' running sums to rank 0:

* `TileGather` (default): every rank packs ITS rows (W*3 floats each, N/k*12 B in all -- 0.96 MB per rank for
  800x800 on 8 GPUs) and the packed tiles are gathered onto rank 0 (RCCL send/recv over the root's direct xGMI
  links), which copies them into the frame.  Double-buffered: the gather of step i runs on RCCL's stream while step
  i+1 traces.  Copies are exact, so the assembled frame is bit-identical to a 1-GPU frame.
* `reduce_frame`: sum of the zero-padded full-frame buffers (N*12 B per rank); adding zeros is exact, same result.

The same row arithmetic lives in csrc/pt_types.hpp:local_to_pixel and csrc/ptmi355.hip:tile_rows."""
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


def step_iterations(step, batch, world, scaling="weak"):
    """(iter0, count) traced by EVERY rank at `step`.  weak: per-GPU work is fixed, so a rank that owns 1/world of the
    pixels traces batch*world iterations of them per step; strong: the frame gets `batch` iterations per step whatever
    the number of GPUs (total work fixed, per-GPU work shrinks)."""
    count = batch * world if scaling == "weak" else batch
    return 1 + step * count, count


def reduce_frame(dist, image, frame, dst=0):
    """Sum the ranks' zero-padded accumulation buffers onto `dst`.  `frame` is a staging copy so
    that the running sums in `image` stay per-rank (a reduce is in place on the destination)."""
    frame.copy_(image)
    dist.reduce(frame, dst=dst, op=dist.ReduceOp.SUM)
    return frame


class TileGather:
    """Gather of the ranks' tile rows onto rank 0, two staging slots (see the module docstring).  `image` is this
    rank's full-frame accumulation buffer (only its own rows are non-zero), `frame` rank 0's assembled frame.
    torch is passed in: this module stays importable without it."""

    def __init__(self, torch, dist, rank, world, strip_rows, width, height, device, via_host=False):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.W3 = width * 3
        self.H = height
        self.via_host = via_host                 # gloo (debug / CPU tests): collectives on host tensors
        rows = [np.nonzero(owned_rows(r, world, strip_rows, height))[0] for r in range(world)]
        self.nrows = [len(r) for r in rows]
        self.max_rows = max(self.nrows)
        self.my_rows = torch.as_tensor(rows[rank], dtype=torch.int64, device=device)
        cdev = "cpu" if via_host else device
        self.staging = [torch.zeros(self.max_rows * self.W3, dtype=torch.float32, device=device) for _ in range(2)]
        self.host_staging = [torch.zeros(self.max_rows * self.W3, dtype=torch.float32) for _ in range(2)] if via_host else None
        self.recv = None
        self.all_rows = None
        if rank == 0:
            self.recv = [[torch.zeros(self.max_rows * self.W3, dtype=torch.float32, device=cdev) for _ in range(world)]
                         for _ in range(2)]
            self.all_rows = [torch.as_tensor(r, dtype=torch.int64, device=device) for r in rows]
        self.pending = [None, None]
        self.seq = [0, 0]                        # issue order of the slots' gathers (the newer sum must land last)
        self.issued = 0
        self.bytes_per_rank = self.nrows[rank] * self.W3 * 4

    def start(self, image, slot):
        """Pack this rank's rows (on the current stream, i.e. after the trace enqueued so far) and start the gather."""
        torch = self.torch
        st = self.staging[slot][: self.nrows[self.rank] * self.W3].view(self.nrows[self.rank], self.W3)
        torch.index_select(image.view(self.H, self.W3), 0, self.my_rows, out=st)
        if self.via_host:
            self.host_staging[slot].copy_(self.staging[slot])            # synchronises: debug path
            send = self.host_staging[slot]
        else:
            send = self.staging[slot]
        self.pending[slot] = self.dist.gather(send, gather_list=self.recv[slot] if self.rank == 0 else None, dst=0,
                                              async_op=True)
        self.issued += 1
        self.seq[slot] = self.issued

    def finish(self, frame, slot):
        """Wait for the slot's gather; rank 0 copies the tiles into `frame`."""
        work = self.pending[slot]
        if work is None:
            return
        work.wait()
        self.pending[slot] = None
        if self.rank == 0:
            fv = frame.view(self.H, self.W3)
            for r in range(self.world):
                t = self.recv[slot][r][: self.nrows[r] * self.W3].view(self.nrows[r], self.W3)
                if self.via_host:
                    t = t.to(frame.device)
                fv.index_copy_(0, self.all_rows[r], t)

    def drain(self, frame):
        for slot in sorted((0, 1), key=lambda k: self.seq[k]):
            self.finish(frame, slot)
