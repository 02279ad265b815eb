"""Host-side sharding of the frame over GPUs: one process per GPU, rank r of k owns the rows y with
(y // strip_rows) % k == r; pixelIndex stays global so RNG keys -- and therefore every radiance value -- are
identical to the 1-GPU run.  The only collective on the path brings the tiles' running sums to rank 0:

* `TileGather` (default): every rank packs ITS rows (W*3 floats each, N/k*12 B in all -- 0.96 MB per rank for
  800x800 on 8 GPUs) and the packed tiles are gathered onto rank 0 (RCCL send/recv over the root's direct xGMI
  links), which copies them into the frame.  Double-buffered: the gather of step i runs on RCCL's stream while step
  i+1 traces.  Copies are exact, so the assembled frame is bit-identical to a 1-GPU frame.
* `reduce_frame`: sum of the zero-padded full-frame buffers (N*12 B per rank); adding zeros is exact, same result.

The same row arithmetic lives in csrc/pt_types.hpp:local_to_pixel and csrc/pt_h_session.hpp:tile_rows."""
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

    def __init__(self, torch, dist, rank, world, strip_rows, width, height, device, via_host=False, slots=2):
        self.torch, self.dist, self.rank, self.world = torch, dist, rank, world
        self.slots = slots
        self.W3 = width * 3
        self.H = height
        self.via_host = via_host                 # gloo (debug / CPU tests): collectives on host tensors
        rows = [np.nonzero(owned_rows(r, world, strip_rows, height))[0] for r in range(world)]
        self.nrows = [len(r) for r in rows]
        self.max_rows = max(self.nrows)
        self.my_rows = torch.as_tensor(rows[rank], dtype=torch.int64, device=device)
        cdev = "cpu" if via_host else device
        self.staging = [torch.zeros(self.max_rows * self.W3, dtype=torch.float32, device=device) for _ in range(slots)]
        self.host_staging = [torch.zeros(self.max_rows * self.W3, dtype=torch.float32) for _ in range(slots)] if via_host else None
        self.recv = None
        self.recv_all = None
        self.frame_src = None
        if rank == 0:
            # one receive buffer per slot, [world][max_rows][W*3]; the gather list is its world slices.  Row y of the
            # frame is row frame_src[y] of that buffer seen as world * max_rows rows: ONE index_select assembles the
            # frame (round 3 issued one index_copy_ per rank -- eight launches per exchange on the root at N = 8)
            self.recv_all = [torch.zeros(world * self.max_rows * self.W3, dtype=torch.float32, device=cdev) for _ in range(slots)]
            self.recv = [[b[r * self.max_rows * self.W3:(r + 1) * self.max_rows * self.W3] for r in range(world)] for b in self.recv_all]
            src = np.zeros(height, dtype=np.int64)
            for r in range(world):
                src[rows[r]] = r * self.max_rows + np.arange(len(rows[r]))
            self.frame_src = torch.as_tensor(src, dtype=torch.int64, device=device)
        self.pending = [None] * slots
        self.seq = [0] * slots                   # issue order of the slots' gathers (the newer sum must land last)
        self.issued = 0
        self.bytes_per_rank = self.nrows[rank] * self.W3 * 4

    def pack(self, image, slot):
        """This rank's rows -> the slot's staging buffer, on the current stream (i.e. after the trace enqueued so far)."""
        st = self.staging[slot][: self.nrows[self.rank] * self.W3].view(self.nrows[self.rank], self.W3)
        self.torch.index_select(image.view(self.H, self.W3), 0, self.my_rows, out=st)

    def start(self, image, slot, packed=False):
        """Pack this rank's rows (unless the caller already has) and start the gather."""
        if not packed:
            self.pack(image, slot)
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
            t = self.recv_all[slot]
            if self.via_host:
                t = t.to(frame.device)
            self.torch.index_select(t.view(self.world * self.max_rows, self.W3), 0, self.frame_src, out=frame.view(self.H, self.W3))

    def drain(self, frame):
        for slot in sorted(range(self.slots), key=lambda k: self.seq[k]):
            self.finish(frame, slot)


class TileGatherThread:
    """TileGather with the collective issued from a thread of its own, one or more exchanges behind the tracing.

    At one exchange per iteration (BASELINE's north star: "reduce ... once per iteration") an iteration is 0.1 ms of GPU
    time at 800x800, while issuing a gather, waiting for the previous one and copying its tiles into the frame costs the
    calling thread about as much again in host time: the host, not the device, set the pace.  Here the tracing thread
    only packs its rows (one kernel, stream-ordered after the iteration's finalGather -- the packed tile is the running
    sum after exactly that iteration) and records an event; this thread waits for the event ON A SIDE STREAM, issues
    the gather there, and rank 0 copies the tiles into the frame behind it, in issue order.  `slots` staging buffers:
    the tracing thread runs up to `slots` exchanges ahead.  On CPU tensors (gloo, the N > 1 tests) there are no
    streams: the pack is synchronous and the thread's gather blocks, same order.

    Measured on one MI355X (bench.py --force-dist, profiles/r04/sub_exchange_check.log): once the device was no longer the
    limit this form is SLOWER than TileGather on the tracing thread (0.36-0.39 x the no-exchange rate against 0.94-0.95 x):
    two Python threads take turns at the interpreter lock.  bench.py uses it only with --exchange-thread."""

    def __init__(self, torch, dist, rank, world, strip_rows, width, height, device, via_host=False, slots=4):
        import queue
        import threading
        self.torch = torch
        self.g = TileGather(torch, dist, rank, world, strip_rows, width, height, device, via_host=via_host, slots=slots)
        self.slots = slots
        self.device = device
        self.cuda = getattr(device, "type", str(device)) == "cuda"
        self.bytes_per_rank = self.g.bytes_per_rank
        if self.cuda:
            # high priority: the pack's successors (RCCL's kernel, the frame assembly) are a few workgroups each and must not
            # queue behind the persistent grids of the batches in flight (in-library form: csrc/pt_multi.hpp, same reason)
            self.side = torch.cuda.Stream(device, priority=-1)
            self.ev_packed = [torch.cuda.Event() for _ in range(slots)]
            self.ev_free = [torch.cuda.Event() for _ in range(slots)]
        self.jobs = queue.SimpleQueue()
        self.cond = threading.Condition()
        self.issued = 0          # exchanges handed over (tracing thread)
        self.enqueued = 0        # exchanges this thread has issued
        self.error = None
        self.th = threading.Thread(target=self._loop, name="tile-exchange", daemon=True)
        self.th.start()

    def _wait_enqueued(self, n):
        with self.cond:
            while self.enqueued < n and self.error is None:
                self.cond.wait(0.05)
        if self.error is not None:
            raise self.error

    def exchange(self, image, frame):
        """Tracing thread: after the batch just enqueued, this rank's rows -> rank 0's frame."""
        k = self.issued
        slot = k % self.slots
        self._wait_enqueued(k - self.slots + 1)              # the slot's previous exchange has been issued (its event recorded)
        if self.cuda:
            if k >= self.slots:
                self.torch.cuda.current_stream().wait_event(self.ev_free[slot])    # ... and has consumed the staging buffer
            self.g.pack(image, slot)
            self.ev_packed[slot].record()
        else:
            self.g.pack(image, slot)
        self.issued += 1
        self.jobs.put((slot, image, frame))

    def _loop(self):
        torch = self.torch
        if self.cuda:
            torch.cuda.set_device(self.device)
        while True:
            job = self.jobs.get()
            if job is None:
                return
            slot, image, frame = job
            try:
                if self.cuda:
                    with torch.cuda.stream(self.side):
                        self.side.wait_event(self.ev_packed[slot])
                        self.g.start(image, slot, packed=True)
                        self.g.finish(frame, slot)                   # the side stream waits for the collective, not the host
                        self.ev_free[slot].record(self.side)
                else:
                    self.g.start(image, slot, packed=True)
                    self.g.finish(frame, slot)
            except BaseException as e:                               # surfaced by the tracing thread
                self.error = e
            with self.cond:
                self.enqueued += 1
                self.cond.notify_all()

    def drain(self, frame=None):
        self._wait_enqueued(self.issued)
        if self.cuda:
            self.side.synchronize()

    def close(self):
        self.jobs.put(None)
        self.th.join(timeout=10.0)
