"""N > 1 path on CPU: two processes over gloo shard the frame in interleaved row strips, each
renders only its own pixels (the CPU oracle stands in for the kernels here -- this test is about
the host-side sharding and the reduce, not the GPU), and the reduce(SUM) of the zero-padded
buffers on rank 0 must equal the 1-process image bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, strip_rows, out_path, mode):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    from oracle import pyoracle as po
    pt = ge.load_package()
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    geoms, mats, cam = z["cornell_64__geoms"], z["cornell_64__materials"], z["cornell_64__camera"]
    depth = int(z["cornell_64__depth"])
    W, H = [int(v) for v in cam[0]["resolution"]]
    batch = 2
    own = pt.sharding.tile_pixel_indices(rank, world, strip_rows, W, H)
    image = torch.zeros(W * H * 3, dtype=torch.float32)
    frame = torch.zeros_like(image)
    tr = po.Tracer(geoms, mats, cam, depth)
    gather = None
    nsteps = 5 if mode == "thread" else 3        # more exchanges than the thread's three staging slots
    for step in range(nsteps):
        iter0, count = pt.sharding.step_iterations(step, batch, world)
        for it in range(iter0, iter0 + count):
            tr.iterate(it)
        # this rank's accumulation buffer holds only its own pixels (zero elsewhere)
        mine = np.zeros((W * H, 3), dtype=np.float32)
        mine[own] = tr.image[own]
        image.copy_(torch.from_numpy(mine.reshape(-1)))
        if mode == "reduce":
            pt.sharding.reduce_frame(dist, image, frame, dst=0)
        elif mode == "thread":                   # the gather issued from the exchange thread, up to three exchanges behind
            if gather is None:
                gather = pt.sharding.TileGatherThread(torch, dist, rank, world, strip_rows, W, H, torch.device("cpu"), via_host=True, slots=3)
            gather.exchange(image, frame)
        else:                                    # the gather of packed tile rows, two slots in flight
            if gather is None:
                gather = pt.sharding.TileGather(torch, dist, rank, world, strip_rows, W, H, torch.device("cpu"), via_host=True)
            gather.finish(frame, step & 1)
            gather.start(image, step & 1)
    if gather is not None:
        gather.drain(frame)
        if mode == "thread":
            assert gather.enqueued == nsteps and gather.error is None
            gather.close()
    if rank == 0:
        np.save(out_path, frame.numpy().reshape(-1, 3))
        np.save(out_path + ".ref.npy", tr.image)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["reduce", "gather", "thread"])
@pytest.mark.parametrize("strip_rows", [4, 7])
def test_two_rank_reduce_equals_single(tmp_path, strip_rows, mode):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, strip_rows, out, mode), nprocs=2, join=True)
    got, want = np.load(out), np.load(out + ".ref.npy")
    assert got.tobytes() == want.tobytes()


def test_eight_ranks_uneven_strips(tmp_path):
    """What the driver launches at N = 8: eight ranks, a strip count that does not divide (64 rows in strips of 5 =
    13 strips: ranks 0-4 own two, 5-7 one; the last strip is short), the gather of packed tile rows with two slots in
    flight -- rank 0's frame equals the 1-process image."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npy")
    port = _free_port()
    mp.spawn(_worker, args=(8, port, 5, out, "gather"), nprocs=8, join=True)
    got, want = np.load(out), np.load(out + ".ref.npy")
    assert got.tobytes() == want.tobytes()


def test_eight_ranks_exchange_thread(tmp_path):
    """The per-iteration cadence of bench.py's sub-measurement at N = 8: every rank hands its exchanges to
    sharding.TileGatherThread; more exchanges than staging slots, uneven strips."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npy")
    port = _free_port()
    mp.spawn(_worker, args=(8, port, 5, out, "thread"), nprocs=8, join=True)
    got, want = np.load(out), np.load(out + ".ref.npy")
    assert got.tobytes() == want.tobytes()


def test_tiles_partition_the_frame():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    sh = ge.load_package().sharding
    for world, strip, W, H in ((2, 8, 800, 800), (8, 8, 800, 800), (3, 5, 64, 37), (8, 16, 3840, 2160)):
        seen = np.zeros(W * H, dtype=np.int32)
        sizes = []
        for r in range(world):
            idx = sh.tile_pixel_indices(r, world, strip, W, H)
            seen[idx] += 1
            sizes.append(len(idx))
            assert (np.diff(idx) > 0).all()            # local order is ascending global pixelIndex
        assert (seen == 1).all()
        assert max(sizes) - min(sizes) <= strip * W    # balanced to within one strip
    assert sh.step_iterations(0, 16, 8) == (1, 128) and sh.step_iterations(2, 16, 8) == (257, 128)
    assert sh.step_iterations(2, 16, 8, "strong") == (33, 16)
