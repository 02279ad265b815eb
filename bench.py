#!/usr/bin/env python3
"""bench.py -- Mrays/s of the wavefront path tracer on the BASELINE.json workload.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--config c2|c3|c4|c5]
                    [--scaling weak|strong] [--reduce-every I] [--collective gather|reduce] [--inproc]

A *step* is one pass of the hot path over one batch of synthetic input: B iterations (samples per pixel) of the
800x800 depth-8 Cornell box with the mirror ball (BASELINE.json configs[1], "c2"), traced as one path pool through
ray generation, the fused intersect / shade / compact bounce kernels and the final gather, with every buffer
resident in HBM.  value = rays traced (sum over bounces of live paths, counted on the device) / wall time, in Mrays/s.

N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is tiled across the ranks in interleaved row
strips.  --scaling weak (default; the contract's definition: per-GPU work fixed): every rank traces B*N iterations
of its tile per step.  --scaling strong: the frame gets B iterations per step whatever N (total work fixed).  The
tiles' running sums travel to rank 0 once per --reduce-every iterations (default: once per step; 1 = the per-iteration
exchange of BASELINE.json's north_star) as a gather of the packed tile rows (N/k*12 B per rank) or, with
--collective reduce, as a sum of the zero-padded full frames; either overlaps the next batch's tracing.
--inproc: the same tiling inside the library, one process driving all N devices (include/ptmi355.h:
pt_scene_desc::devices; RCCL send/recv from C++) -- what a host that links libptmi355.so gets; same JSON line.

Steps are enqueued back to back (pt_trace_batch_async) and the region ends with one synchronisation: consecutive
steps overlap on the device (the library runs them on alternating launch streams with their own path pools; every
gather stays in call order, the image is bit-identical), which is what a host that renders many iterations gets.

Extra JSON objects (see the task statement): `roofline` for the dominant kernel and `cpu_baseline` (the plain-C
oracle on this host's cores, rank 0, N = 1 only).  The kernel is bound by vector-instruction issue, not by HBM:
`roofline.frac` = issue cycles per launch (the EXECUTED opcode histogram of an instrumented build x the issue cost of
each opcode measured one by one: profiles/tools/isa_count.py, profiles/microbench/gen_issue_ops.py) over 1024 SIMDs x
2.4 GHz x the launch time measured here with HIP events; the measured HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE), the
fp32 rate and the active-lane fraction ride along, and `roofline.timed_pass` states the same cycles over the timed
pass's own time per step.  The counter profile (profiles/traffic.json, written by profiles/collect.py) is keyed
by a hash of the kernel sources and compile flags: after any kernel change the counter-derived fields are null until
the collection has been re-run.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
SIMDS = 256 * 4                # 256 CUs x 4 SIMDs
PEAK_CLOCK_GHZ = 2.4           # MI355X_MICROARCH.md: max clock
MFMA_F16_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense f16 / bf16
# SURVEY 8(d) algorithmic bytes: intersect 44 B/ray + shade/scatter 104 B/ray +
# compaction 4 B/ray, + 88 B per surviving path -- all done by the fused k_bounce launch
BYTES_PER_RAY = 44 + 104 + 4
BYTES_PER_SURVIVOR = 88


def build_digest():
    """What a counter profile is keyed by: the kernel sources AND how they are compiled (-fno-slp-vectorize
    changes the instruction mix as much as a source edit does).  A library loaded through PTMI355_LIB (an A/B variant)
    never matches a committed profile."""
    import hashlib
    h = hashlib.sha256()
    h.update(csrc_digest().encode())
    sys.path.insert(0, os.path.join(ROOT, "project3-cuda-path-tracer_amd"))
    try:
        import build as _b
        h.update(" ".join(_b.HIPCC_FLAGS).encode())
    except Exception:
        h.update(b"?")
    h.update((os.environ.get("PTMI355_LIB") or "").encode())
    return h.hexdigest()[:16]


def csrc_digest():
    """sha256 (first 16 hex digits) over the kernel sources -- comments and white space removed, so that only code
    changes count: counter profiles are only valid for the build they were taken on."""
    import hashlib
    import re
    h = hashlib.sha256()
    d = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip")):
            text = open(os.path.join(d, name), "r", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
            text = re.sub(r"//[^\n]*", " ", text)
            h.update(name.encode())
            h.update("".join(text.split()).encode())
    return h.hexdigest()[:16]


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port):
    """The driver's own form of an N-rank run (one process per GPU, rendezvous on 127.0.0.1)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


PHASES = ("main", "exchange", "strong", "shared")      # N > 1: what one line consists of; each can run in a launch of its own


def run_launch(cmd, env, timeout):
    """One child launch (a process group of its own): returns (exit code, the JSON objects it printed).  Lines that are
    not JSON go to stderr as they come.  `timeout` > 0: the group is ended after that many seconds and the exit code is
    124 (as timeout(1) reports it)."""
    import signal
    import subprocess
    import threading
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    expired = []
    timer = None
    if timeout and timeout > 0:
        def expire():
            expired.append(True)
            sys.stderr.write("bench.py: the ranks did not finish in %.0f s: ending their process group (exit code 124)\n" % timeout)
            try:
                os.killpg(p.pid, signal.SIGTERM)
                time.sleep(5.0)
                os.killpg(p.pid, signal.SIGKILL)
            except OSError:
                pass
        timer = threading.Timer(timeout, expire)
        timer.daemon = True
        timer.start()
    objs = []
    for line in p.stdout:
        if line.startswith("{"):
            try:
                objs.append(json.loads(line))
                continue
            except ValueError:
                pass
        sys.stderr.write(line)
    rc = p.wait()
    if timer:
        timer.cancel()
    return (124 if expired else rc), objs


def merge_phases(main_obj, subs):
    """The one line of an N > 1 run from the objects its phases printed.  `main_obj`: the main timed pass's line (None when
    that launch left none); `subs`: {phase: object or {"failed": ...}} of the sub-measurements, each from a launch of its own."""
    out = main_obj if main_obj is not None else {"value": None, "failed": "the main timed pass left no line"}
    out.setdefault("config", {})
    for name, key in (("exchange", "per_iteration_exchange"), ("strong", "strong"), ("shared", "per_iteration_shared_frame")):
        if name not in subs:
            continue
        v = subs[name]
        v = v.get(key, v) if isinstance(v, dict) else {"failed": "no object"}
        if key == "per_iteration_exchange":
            out[key] = v                                       # top-level AND where rounds 3-4 had it
        out["config"][key] = v
    out["phases"] = {"launches": "one child launch (fresh processes, rendezvous of its own) per phase, the main timed pass first",
                     "order": ["main"] + [k for k in PHASES[1:] if k in subs]}
    return out


def self_launch(args, argv):
    """`python3 bench.py --gpus N` from a plain shell (no WORLD_SIZE in the environment): start the N ranks as fresh child
    processes through torch.distributed.run and print ONE line.  This parent has imported neither torch nor the HIP library
    -- nothing here has touched the GPU, and nothing is exec'ed: every launcher is a child in a session of its own, so
    a time limit can end exactly that process group.
    ONE LAUNCH PER PHASE, the main timed pass first: `value` must not depend on a collective pattern
    that has never run with peers -- the per-iteration exchange, strong scaling and the shared host frame each get fresh
    processes and a rendezvous of their own afterwards; a phase that hangs (the ranks' own watchdog, then the time limit
    here), dies or leaves no object becomes {"failed": ...} in the line, and the line is still printed with the main
    pass's number.  Exit code: the main pass's (124: ended by the time limit)."""
    base = [a for a in argv if a != "--print-launch"]
    if args.print_launch:
        print(json.dumps({"launch": launch_command(args.gpus, base, int(os.environ.get("MASTER_PORT") or 0) or free_port())}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("MASTER_PORT", None)

    def launch(phase, timeout):
        rc, objs = 1, []
        for attempt in (1, 2):                        # (the port was free when it was looked up, not necessarily when the rendezvous binds it)
            cmd = launch_command(args.gpus, base + (["--phase", phase] if phase else []), free_port())
            sys.stderr.write("bench.py: --gpus %d without a launcher, phase '%s': starting %s\n" % (args.gpus, phase or "all", " ".join(cmd[1:10])))
            t0 = time.monotonic()
            rc, objs = run_launch(cmd, env, timeout)
            if objs or rc in (0, 3, 124) or time.monotonic() - t0 > 30.0:
                break
            sys.stderr.write("bench.py: the launch ended with code %d after %.0f s without a line: once more on another port\n" % (rc, time.monotonic() - t0))
        return rc, objs

    single = getattr(args, "phase", "all") != "all" or getattr(args, "one_launch", False)
    if single:                                         # a named phase (or --one-launch): exactly one launch, relayed as it is
        rc, objs = launch(None, args.launch_timeout)          # (a --phase on the command line is in `base` already)
        for o in objs:
            print(json.dumps(o), flush=True)
        if rc == 0 and not objs:
            sys.stderr.write("bench.py: the ranks exited without a JSON line\n")
            return 1
        return rc
    main_limit = args.launch_timeout if args.launch_timeout > 0 else 0.0
    rc_main, objs = launch("main", main_limit)
    main_obj = objs[-1] if objs else None
    subs = {}
    wanted = [] if args.no_sub else (["exchange", "strong"] + (["shared"] if args.shared_frame else []))
    for ph in wanted:
        limit = args.launch_timeout if args.launch_timeout > 0 else 3.0 * args.sub_timeout + 120.0
        rc, objs = launch(ph, limit)
        if objs and rc == 0:
            subs[ph] = objs[-1]
        else:
            why = "ended by the time limit of %.0f s" % limit if rc == 124 else "exit code %d" % rc
            subs[ph] = {"failed": "phase '%s': %s%s" % (ph, why, "" if not objs else "; its ranks said: " + json.dumps(objs[-1])[:300])}
    print(json.dumps(merge_phases(main_obj, subs)), flush=True)
    if main_obj is None and rc_main == 0:
        sys.stderr.write("bench.py: the main pass's ranks exited without a JSON line\n")
        return 1
    return rc_main


def load_scene(pt, name):
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    g = lambda k: z["%s__%s" % (name, k)]
    return pt.Scene(g("geoms"), g("materials"), g("camera"), int(g("depth")), name=name)


class Ctx:
    """What every measurement of one bench.py run shares: the process group, the scene, the flags."""


def setup(args):
    # dmabuf IPC (the pool's driver has no legacy IPC): must be in the environment BEFORE the HIP runtime initialises,
    # i.e. before torch is imported; launchers normally export it already
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    c = Ctx()
    c.args, c.torch, c.dist = args, torch, dist
    c.rank = int(os.environ.get("RANK", "0"))
    c.world = int(os.environ.get("WORLD_SIZE", "1"))
    c.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    c.inproc = args.inproc
    if c.inproc and c.world != 1:
        raise SystemExit("--inproc is one process: do not launch it through torch.distributed.run")
    if c.world != args.gpus and not c.inproc:
        args.gpus = c.world                    # (a plain `--gpus N` never gets here: main() starts the ranks itself)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.same_device:
        c.local_rank = 0
    if c.local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d of %d: local rank %d but only %d GPU(s) visible (one rank per GPU; --same-device puts "
                         "every rank on cuda:0 for a rehearsal)" % (c.rank, c.world, c.local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(c.local_rank)
    c.dist_on = (c.world > 1 or args.force_dist) and not c.inproc
    c.n_tiles = args.gpus if c.inproc else c.world          # GPUs the frame is tiled over
    if c.dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if args.backend == "nccl":      # "nccl" is RCCL on ROCm
            dist.init_process_group("nccl", rank=c.rank, world_size=c.world,
                                    device_id=torch.device("cuda", c.local_rank))
        else:
            dist.init_process_group("gloo", rank=c.rank, world_size=c.world)

    pt = c.pt = ge.load_package()
    pt.library()
    c.scene_name = {"c2": "cornell", "c3": "cornell_glass", "c4": "cornell", "c5": "cornell_4k"}[args.config]
    scene = load_scene(pt, c.scene_name)
    if args.config == "c4":       # BASELINE configs[3]: naive loop over a 100 032-triangle UV sphere, material 1
        tris = pt.meshes.uv_sphere(n_lat=97, n_lon=521)
        assert len(tris) == 100032
        geoms, tris, meshes = pt.meshes.add_mesh(scene.geoms, tris, material_id=1)
        scene = pt.Scene(geoms, scene.materials, scene.camera, scene.traceDepth, triangles=tris, meshes=meshes,
                         name="cornell+mesh")
    c.scene = scene
    c.W, c.H = scene.resolution
    c.npix = c.W * c.H
    c.flags = 0
    for f in args.flags.split(","):
        c.flags |= {"compact": pt.PT_COMPACT, "sort": pt.PT_SORT_MATERIAL, "unfused": pt.PT_UNFUSED,
                    "cache": pt.PT_CACHE_FIRST, "bvh": pt.PT_MESH_BVH, "aa": pt.PT_AA_JITTER, "": 0}[f]
    # an explicit (non-null) torch stream: the library launches on it, torch copies / RCCL order against it
    c.stream = torch.cuda.Stream()
    torch.cuda.set_stream(c.stream)
    return c


class Session:
    """One pathtraceInit ... pathtraceFree with the exchange plumbing of its cadence.

    scaling / reduce_every as on the command line; exchange=False traces at the same cadence (same batch sizes, same
    number of calls) without moving the tiles -- the reference rate an exchange is held against; threaded: the tile
    gather of the process form is issued from sharding.TileGatherThread (None: only with --exchange-thread and more than
    one exchange per step; otherwise the two-slot TileGather on the tracing thread)."""

    def __init__(self, c, scaling, reduce_every, exchange=True, threaded=None, it0=0):
        args, torch, pt = c.args, c.torch, c.pt
        self.c, self.scaling, self.do_exchange = c, scaling, exchange and (c.dist_on or c.inproc)
        self.per_step_iters = pt.sharding.step_iterations(0, args.batch, c.n_tiles, scaling)[1]
        self.every = self.per_step_iters if (reduce_every <= 0 or not (c.dist_on or c.inproc)) else min(reduce_every, self.per_step_iters)
        self.exchanges_per_step = -(-self.per_step_iters // self.every)
        self.image = torch.zeros(c.npix * 3, dtype=torch.float32, device="cuda")     # accumulation buffer (torch-owned)
        self.frame = torch.zeros_like(self.image) if c.dist_on else None           # rank 0: the assembled frame / reduce staging
        torch.cuda.synchronize()
        if c.inproc:
            # the library owns every device's stream and buffers; device 0 assembles the frame after every batch
            devices = [0] * args.gpus if args.same_device else list(range(args.gpus))
            pt.pathtraceInit(c.scene, flags=c.flags, tile=(0, 1, args.strip_rows), max_batch=self.every, devices=devices)
        else:
            pt.pathtraceInit(c.scene, flags=c.flags, device=c.local_rank, stream=c.stream.cuda_stream,
                             tile=(c.rank, c.world, args.strip_rows), max_batch=self.every,
                             device_image=self.image.data_ptr())
        self.transport = pt.exchange_transport() if c.inproc else None
        self.gather = self.gather_thread = None
        if c.dist_on and self.do_exchange and args.collective == "gather":
            if threaded is None:
                threaded = self.exchanges_per_step > 1 and args.exchange_thread and not args.no_exchange_thread
            cls = pt.sharding.TileGatherThread if threaded else pt.sharding.TileGather
            g = cls(torch, c.dist, c.rank, c.world, args.strip_rows, c.W, c.H, torch.device("cuda", c.local_rank),
                    via_host=(args.backend == "gloo"))
            if threaded:
                self.gather_thread = g
            else:
                self.gather = g
        self.bytes_per_rank = (self.gather or self.gather_thread).bytes_per_rank if (self.gather or self.gather_thread) else c.npix * 12
        self.pending = None
        self.exchanges = 0
        self.it = it0

    def exchange(self):
        """tiles' running sums -> rank 0, overlapped with whatever is traced next"""
        k = self.exchanges
        self.exchanges += 1
        if self.gather_thread is not None:
            self.gather_thread.exchange(self.image, self.frame)
        elif self.gather is not None:
            self.gather.finish(self.frame, k & 1)                  # the gather issued two exchanges ago used this slot
            self.gather.start(self.image, k & 1)
        else:
            if self.pending is not None:
                self.pending.wait()
            self.frame.copy_(self.image)
            self.pending = self.c.dist.reduce(self.frame, dst=0, op=self.c.dist.ReduceOp.SUM, async_op=True)

    def step(self):
        c, pt = self.c, self.c.pt
        iter0, count = pt.sharding.step_iterations(self.it, c.args.batch, c.n_tiles, self.scaling)
        self.it += 1
        for j in range(0, count, self.every):
            pt.trace_batch_async(iter0 + j, min(self.every, count - j))          # enqueue only
            if c.dist_on and self.do_exchange:
                self.exchange()

    def barrier(self):
        c, torch = self.c, self.c.torch
        if c.dist_on:
            if self.gather_thread is not None:
                self.gather_thread.drain()
            if self.gather is not None:
                self.gather.drain(self.frame)
            if self.pending is not None:
                self.pending.wait()
                self.pending = None
            torch.cuda.synchronize()
            c.dist.barrier()
        if c.inproc:
            c.pt.synchronize()                     # every device's launch and exchange stream
        torch.cuda.synchronize()

    def timed(self, steps, warmup):
        """W untimed steps, then exactly K steps between two barriers; max over ranks of the time, sum of the rays."""
        c, torch, pt = self.c, self.c.torch, self.c.pt
        for _ in range(warmup):
            self.step()
        self.barrier()
        rays0, first0, _ = pt.counters()
        self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step()
        self.barrier()
        dt = time.perf_counter() - t0
        rays1, first1, _ = pt.counters()
        rays, first = rays1 - rays0, first1 - first0
        rank_rays = rays
        if c.dist_on:
            tmax = torch.tensor([dt], dtype=torch.float64, device="cuda"); c.dist.all_reduce(tmax, op=c.dist.ReduceOp.MAX)
            rsum = torch.tensor([float(rays)], dtype=torch.float64, device="cuda"); c.dist.all_reduce(rsum, op=c.dist.ReduceOp.SUM)
            dt, rays = float(tmax.item()), float(rsum.item())
        return dt, rays, first, rank_rays

    def close(self):
        if self.gather_thread is not None:
            self.gather_thread.close()
        self.c.pt.pathtraceFree()


def per_call_rates(c, n_it=256):
    """The drop-in calling pattern (src/main.cpp:130-140): ONE pathtrace() per iteration.  `mrays_per_s`: the calls
    alone, enqueued back to back (pt_trace_batch_async(iter, 1): what a host gets that does not look at state.image
    between iterations); `pcie_inclusive_sync`: every call hands the running sum back in host memory before it returns
    (pathtrace.cu:389-392: the reference's exact semantics, 7.68 MB over PCIe per call at 800x800);
    `pcie_inclusive_async`: PT_ASYNC_IMAGE (the copy of call i overlaps the tracing of call i+1)."""
    pt, torch = c.pt, c.torch
    out = {}
    host = np.zeros((c.npix, 3), dtype=np.float32)
    L = pt.library()

    def masked_streams():
        # what the library did with the windows (csrc/pt_h_enqueue.hpp: ensure_la_masks): windows traced on the lanes' CU-masked
        # streams / calls whose gather ran on the compute units set aside for it
        import ctypes
        out = (ctypes.c_uint64 * 2)()
        return [int(v) for v in out] if L.ptdbg_lookahead_masked(out) == 0 else None

    def run(flags, call, max_batch=1, rays=None, first=33, probe=None):
        # (the library's own launch stream, as a host that links libptmi355.so gets it -- not torch's)
        # `rays`: the rays of iterations first .. first + n_it - 1 as counted by an earlier run over the same iterations -- a
        # PT_LOOKAHEAD session counts windows when it traces them, ahead of the calls that consume them
        pt.pathtraceInit(c.scene, flags=flags, device=c.local_rank, max_batch=max_batch)
        for k in range(first - 1):
            call(1 + k)
        pt.synchronize()
        r0 = pt.total_rays()
        t1 = time.perf_counter()
        for k in range(n_it):
            call(first + k)
        pt.synchronize()
        el = time.perf_counter() - t1
        traced = pt.total_rays() - r0
        run.probe = probe() if probe else None
        pt.pathtraceFree()
        run.rays = traced
        return round((traced if rays is None else rays) / el / 1e6, 2), round(el / n_it * 1e3, 4)

    out["mrays_per_s"], out["ms_per_call"] = run(c.flags, lambda it: pt.trace_batch_async(it, 1))
    # as host/pathtrace_shim.cpp initialised the library up to round 5: PT_PIN_IMAGE | PT_HOST_SPARSE (the reference's host
    # only reads state.image: a call writes the pixels whose sum changed), every iteration traced inside its own call ...
    out["pcie_inclusive_sync_no_lookahead"], out["pcie_inclusive_sync_no_lookahead_ms_per_call"] = run(
        c.flags | pt.PT_HOST_SPARSE, lambda it: L.pt_trace(None, 0, it, host.ctypes.data), first=85)
    rays_85 = run.rays
    if c.args.digest:
        import hashlib
        out["host_image_md5_no_lookahead"] = hashlib.md5(host.tobytes()).hexdigest()
    # ... and as the shim initialises it now: + PT_LOOKAHEAD, max_batch = 64 -- the library traces windows of 4, 16, 64, 64, ..
    # iterations ahead of the caller and every call gathers its own sample (include/ptmi355.h).  Same calls, same iteration
    # numbers, state.image complete at every return; the timed calls start at a window's first iteration (85 = 1 + 4 + 16 + 64)
    # and span whole windows, and the rate counts the rays of exactly those iterations (counted by the run above)
    la = c.flags | pt.PT_HOST_SPARSE | pt.PT_LOOKAHEAD
    la_batch = max(4, min(64, 41000000 // c.npix))       # the shim's rule (host/pathtrace_shim.cpp): windows of up to 64 iterations and ~40 M paths
    out["lookahead_max_batch"] = la_batch
    out["pcie_inclusive_sync"], out["pcie_inclusive_sync_ms_per_call"] = run(
        la, lambda it: L.pt_trace(None, 0, it, host.ctypes.data), max_batch=la_batch, rays=rays_85, first=85, probe=masked_streams)
    if run.probe:
        out["lookahead_cu_masked"] = {"windows": run.probe[0], "calls": run.probe[1],
                                      "note": "since pathtraceInit: windows traced on 232 of the 256 compute units, calls whose gather wrote the host image from the other 24"}
    if c.args.digest:
        out["host_image_md5"] = hashlib.md5(host.tobytes()).hexdigest()       # the host image after the synchronous calls
    out["lookahead_no_host_image"], out["lookahead_no_host_image_ms_per_call"] = run(
        c.flags | pt.PT_LOOKAHEAD, lambda it: L.pt_trace(None, 0, it, None), max_batch=la_batch, rays=rays_85, first=85)
    # ... and for a host that may write into the image between calls: every pixel, every call
    out["pcie_inclusive_sync_every_pixel"], out["pcie_inclusive_sync_every_pixel_ms_per_call"] = run(c.flags, lambda it: L.pt_trace(None, 0, it, host.ctypes.data))
    out["pcie_inclusive_async"], _ = run(c.flags | pt.PT_ASYNC_IMAGE | pt.PT_HOST_SPARSE, lambda it: L.pt_trace(None, 0, it, host.ctypes.data))
    out["calls"] = n_it
    out["note"] = ("one pathtrace() per iteration (src/main.cpp:130-140), 1 spp per call: mrays_per_s = calls enqueued back to back "
                   "(no host image, max_batch = 1); pcie_inclusive_sync = the running sum in host memory when each call returns "
                   "(pathtrace.cu:389-392) with PT_PIN_IMAGE | PT_HOST_SPARSE | PT_LOOKAHEAD, max_batch = lookahead_max_batch, as the drop-in shim sets "
                   "them: windows of iterations traced ahead, every call gathers its own sample; ..._no_lookahead = every iteration "
                   "traced inside its own call (the shim up to round 5); lookahead_no_host_image = the same calls without a host image; "
                   "..._every_pixel = without PT_HOST_SPARSE and without PT_LOOKAHEAD; pcie_inclusive_async = PT_ASYNC_IMAGE | PT_HOST_SPARSE")
    return out


def shared_frame_rate(c, iters):
    """The north star's cadence -- the frame complete after EVERY iteration -- without an exchange: the ranks hand
    pathtrace() ONE host frame (shared memory, page-locked by each process) and every rank's launch writes the pixels of its
    own tile into it while it traces (PT_SHARED_IMAGE, include/ptmi355.h).  One synchronous pathtrace() per iteration per
    rank, as the reference's host calls it; the frame holds the sum after iteration i once every rank's call has returned.
    Verified here: every rank compares its tile's pixels in the shared frame with its own accumulation buffer."""
    torch, dist, pt, args = c.torch, c.dist, c.pt, c.args
    L = pt.library()
    path = "/dev/shm/ptmi355_frame_%s" % os.environ.get("MASTER_PORT", "0")          # one job per rendezvous port on a node
    nbytes = c.npix * 12
    if c.rank == 0:
        with open(path, "wb") as f:
            f.truncate(nbytes)
    dist.barrier()
    frame = np.memmap(path, dtype=np.float32, mode="r+", shape=(c.npix, 3))
    out = {}
    try:
        pt.pathtraceInit(c.scene, flags=c.flags | pt.PT_PIN_IMAGE | pt.PT_SHARED_IMAGE, device=c.local_rank,
                         tile=(c.rank, c.world, args.strip_rows), max_batch=1, pin_image=False)
        ptr = frame.ctypes.data
        warm = 8
        for it in range(1, warm + 1):
            if L.pt_trace(None, 0, it, ptr) != 0:
                raise RuntimeError((L.pt_last_error() or b"pt_trace failed").decode(errors="replace"))
        torch.cuda.synchronize()
        dist.barrier()
        rays0 = pt.counters()[0]
        t0 = time.perf_counter()
        for it in range(warm + 1, warm + 1 + iters):
            if L.pt_trace(None, 0, it, ptr) != 0:
                raise RuntimeError("pt_trace failed")
        dist.barrier()
        dt = time.perf_counter() - t0
        rays = pt.counters()[0] - rays0
        mine = pt.sharding.tile_pixel_indices(c.rank, c.world, args.strip_rows, c.W, c.H)
        bad = int((pt.get_image(c.npix)[mine].view(np.uint32) != np.asarray(frame)[mine].view(np.uint32)).sum())
        pt.pathtraceFree()
        v = torch.tensor([dt, float(rays), float(bad)], dtype=torch.float64, device="cuda")
        tmax = v[:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        vsum = v[1:].clone(); dist.all_reduce(vsum, op=dist.ReduceOp.SUM)
        dt, rays, bad = float(tmax.item()), float(vsum[0].item()), int(vsum[1].item())
        out = {"mrays_per_s": round(rays / dt / 1e6, 2), "ms_per_iteration": round(dt / iters * 1e3, 4), "iterations": iters,
               "frame_equals_every_ranks_tile": bad == 0,
               "note": "no exchange: one synchronous pathtrace() per iteration per rank into ONE page-locked host frame in shared memory; "
                       "every launch writes its own tile's pixels (those whose sum changed) while it traces (PT_SHARED_IMAGE)"}
        if args.digest and c.rank == 0:
            import hashlib
            out["frame_md5"] = hashlib.md5(np.asarray(frame).tobytes()).hexdigest()
            out["frame_iterations"] = warm + iters
    finally:
        del frame
        dist.barrier()
        if c.rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
    return out


def per_iteration_exchange(c):
    """N > 1, BASELINE.json's north-star cadence: the tiles' running sums travel to rank 0 after EVERY iteration (one
    iteration of its tile per call on every rank; the reference's host has the whole image after every iteration,
    src/pathtrace.cu:389-392), next to the same calls without the exchange."""
    args = c.args
    iters = args.sub_iters
    n_steps = max(1, -(-iters // c.pt.sharding.step_iterations(0, args.batch, c.n_tiles, args.scaling)[1]))
    rates = {}
    for key, exch in (("exchange", True), ("no_exchange", False)):
        if c.inproc and not exch:
            continue                                  # the in-library form exchanges after every call by construction
        s = Session(c, args.scaling, 1, exchange=exch)
        dt, rays, _, _ = s.timed(n_steps, 1)
        rates[key] = (rays / dt / 1e6, dt / (n_steps * s.per_step_iters) * 1e3, s)
        s.close()
    r, ms, s = rates["exchange"]
    out = {"mrays_per_s": round(r, 2), "ms_per_iteration": round(ms, 4), "iterations": n_steps * s.per_step_iters,
           "paths_per_rank_per_iteration": int(c.npix / c.n_tiles),
           "transport": s.transport or ("%s %s from %s" % (args.backend, args.collective, "an exchange thread (sharding.TileGatherThread)" if s.gather_thread else "the tracing thread")),
           "mb_per_rank_per_exchange": round(s.bytes_per_rank / 1e6, 3),
           "note": "%s scaling, 1 spp per call, every rank's tile sums to rank 0 after every iteration (BASELINE north_star)" % args.scaling}
    if "no_exchange" in rates:
        out["no_exchange_mrays_per_s"] = round(rates["no_exchange"][0], 2)
        out["ratio"] = round(r / rates["no_exchange"][0], 3)
    return out


def strong_scaling(c, steps, warmup):
    """N > 1: the frame gets `batch` iterations per step whatever N (total work fixed)."""
    args = c.args
    s = Session(c, "strong", args.reduce_every)
    dt, rays, _, _ = s.timed(steps, warmup)
    s.close()
    return {"mrays_per_s": round(rays / dt / 1e6, 2), "ms_per_step": round(dt * 1e3 / steps, 4),
            "spp_per_step_per_frame": args.batch, "paths_per_rank_per_step": int(args.batch * c.npix / c.n_tiles),
            "note": "the frame gets %d spp per step whatever N: per-GPU work shrinks with N" % args.batch}


def rank_identities(c):
    """What the collective backend actually spans: world size and backend as torch.distributed reports them and, gathered
    THROUGH that backend, every rank's device (index, PCI domain:bus:device) -- so that `RCCL saw N ranks on N different
    GPUs` can be read off the line."""
    torch, dist = c.torch, c.dist
    p = torch.cuda.get_device_properties(c.local_rank)
    mine = torch.tensor([c.rank, c.local_rank, int(getattr(p, "pci_domain_id", -1)), int(getattr(p, "pci_bus_id", -1)),
                         int(getattr(p, "pci_device_id", -1))], dtype=torch.int64,
                        device="cpu" if c.args.backend == "gloo" else "cuda")      # (gloo gathers host tensors only)
    every = [torch.zeros_like(mine) for _ in range(c.world)]
    dist.all_gather(every, mine)
    torch.cuda.synchronize()
    rows = [[int(v) for v in t.cpu()] for t in every]
    devs = ["cuda:%d @ %04x:%02x:%02x" % (r[1], r[2] & 0xffff, r[3] & 0xff, r[4] & 0xff) for r in rows]
    return {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "collective": c.args.collective,
            "devices": devs, "distinct_devices": len(set(devs)), "device_name": p.name,
            "note": "gathered with all_gather over this backend ('nccl' is RCCL on ROCm)"}


class Watchdog:
    """N > 1: a collective that never returns must end the run LOUDLY.  Phases arm a deadline; when one passes, rank 0
    prints the line as far as it got (with `watchdog` saying which phase hung) and every rank leaves -- with exit code 3, or 0
    when the line already carries `value` (the main timed pass is the first phase).
    `out` is only touched under `lock`, by the main thread and by this one; once the main thread has printed the line
    (`emit`) the watchdog never prints."""

    def __init__(self, rank, out):
        import threading
        self.rank, self.out = rank, out
        self.lock = threading.Lock()
        self.deadline = None
        self.phase = None
        self.done = False
        threading.Thread(target=self._loop, name="bench-watchdog", daemon=True).start()

    def arm(self, seconds, phase):
        with self.lock:
            self.deadline, self.phase = time.monotonic() + seconds, phase

    def disarm(self):
        with self.lock:
            self.deadline = None

    def put(self, where, key, value):
        with self.lock:
            where[key] = value

    def emit(self):
        with self.lock:
            self.done = True
            if self.rank == 0:
                print(json.dumps(self.out), flush=True)

    def _loop(self):
        while True:
            time.sleep(0.5)
            with self.lock:
                if self.done:
                    return
                if self.deadline is None or time.monotonic() < self.deadline:
                    continue
                self.done = True
                # the main timed pass comes first: when it is a LATER phase that hangs, `value` is in the line already and the
                # run has delivered what the contract asks for -- every rank then leaves with exit code 0 (a launcher that sees
                # a non-zero code may throw the line away); a hang before `value` exists is exit code 3
                code = 0 if self.out.get("value") is not None else 3
                if self.rank == 0:
                    self.out["watchdog"] = "phase '%s' did not finish in time: exit code %d" % (self.phase, code)
                    print(json.dumps(self.out), flush=True)
            if self.rank != 0:
                time.sleep(2.0)            # rank 0's line first: a launcher that sees a rank exit tears the others down
            os._exit(code)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64,
                    help="iterations (spp) per step per full frame (measured on C2: 16 -> 19.6, 32 -> 20.7, 64 -> 21.3, "
                         "128 -> 20.5 Grays/s; the pool is 40 B x 800 x 800 x batch x 2)")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2: BASELINE configs[1] (the metric's workload); c3: glass ball 1280x720 depth 16; "
                         "c4: Cornell + 100k-triangle mesh; c5: 3840x2160 Cornell (parity-test cases, selectable for measurement)")
    ap.add_argument("--flags", default="compact", help="comma list: compact,sort,unfused,cache,bvh,aa")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="N = 1: skip roofline.sustained (the timed steps repeated for --sustain-seconds)")
    ap.add_argument("--sustain-seconds", type=float, default=2.0)
    ap.add_argument("--strip-rows", type=int, default=8)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank traces batch*N iterations of its tile per step (per-GPU work fixed); "
                         "strong = the frame gets `batch` iterations per step whatever N (total work fixed)")
    ap.add_argument("--reduce-every", type=int, default=0,
                    help="N > 1: iterations between two exchanges of the tiles' running sums (0 = once per step; "
                         "1 = once per iteration, the north-star semantics)")
    ap.add_argument("--collective", default="gather", choices=["gather", "reduce"],
                    help="N > 1: gather the packed tile rows onto rank 0 (N/k*12 B per rank) or reduce(SUM) the zero-padded full frames")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --same-device exercises the N>1 code path on a single GPU (debug)")
    ap.add_argument("--same-device", action="store_true", help="debug: every rank uses cuda:0")
    ap.add_argument("--inproc", action="store_true",
                    help="N > 1 in ONE process: the library tiles the frame over --gpus devices itself (one host thread, one "
                         "stream per device; tiles gathered onto device 0 over RCCL -- csrc/pt_multi.hpp); no torch.distributed")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal on a 1-GPU box: run the N > 1 code path (process group, tile exchange, max / sum over "
                         "ranks) with a world of ONE rank, so that the RCCL calls themselves execute on hardware")
    ap.add_argument("--digest", action="store_true", help="add the md5 of the (reduced) accumulation image")
    ap.add_argument("--pcie", action="store_true", help="(kept for old command lines: config.per_call is in every N = 1 line now)")
    ap.add_argument("--no-per-call", action="store_true",
                    help="N = 1: skip config.per_call (the reference's one pathtrace() per iteration, with and without the host image)")
    ap.add_argument("--no-sub", action="store_true",
                    help="N > 1: skip config.per_iteration_exchange / config.strong (the sub-measurements after the main pass)")
    ap.add_argument("--sub-iters", type=int, default=256, help="iterations of the per-iteration-exchange sub-measurement")
    ap.add_argument("--sub-timeout", type=float, default=120.0,
                    help="N > 1: seconds a phase (per-iteration exchange, strong scaling, ...) may take before the watchdog prints "
                         "the line as far as it got and every rank exits with code 3")
    ap.add_argument("--shared-frame", action="store_true",
                    help="N > 1, process form: also measure config.per_iteration_shared_frame (PT_SHARED_IMAGE: one page-locked host "
                         "frame in shared memory written by every rank's own launches; no exchange)")
    ap.add_argument("--launch-timeout", type=float, default=0.0,
                    help="plain `--gpus N` (no launcher): seconds before the self-started ranks' process group is ended (0 = never)")
    ap.add_argument("--phase", default="all", choices=["all"] + list(PHASES),
                    help="N > 1: run only this part of the line (a plain `--gpus N` starts one launch per phase itself and merges "
                         "their objects; under an external launcher `all` runs them in one set of processes, the main timed pass first)")
    ap.add_argument("--one-launch", action="store_true", help="plain `--gpus N`: every phase in ONE launch, as an external launcher gets it")
    ap.add_argument("--print-launch", action="store_true",
                    help="plain `--gpus N`: print the torch.distributed.run command line instead of running it")
    ap.add_argument("--exchange-thread", action="store_true",
                    help="process form: issue the tile gathers from sharding.TileGatherThread when there is more than one per step "
                         "(measured SLOWER than the tracing thread's own two-slot gather once the device was no longer the limit: "
                         "two Python threads take turns at the interpreter lock; profiles/r04/sub_exchange_check.log)")
    ap.add_argument("--no-exchange-thread", action="store_true", help="(the default now; kept for old command lines)")
    args = ap.parse_args()

    # ---- `python3 bench.py --gpus N` from a plain shell: the ranks are started here, as child processes, before
    # anything in this process has imported torch or the HIP library ----
    if args.gpus > 1 and not args.inproc and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))

    c = setup(args)
    pt, torch, dist, rank, world = c.pt, c.torch, c.dist, c.rank, c.world
    inproc, dist_on, n_tiles, scene = c.inproc, c.dist_on, c.n_tiles, c.scene
    W, H, npix, flags = c.W, c.H, c.npix, c.flags
    multi = dist_on or inproc
    if args.phase not in ("all", "main"):
        return sub_phase(c)

    # the line, filled in as the measurements finish (N > 1: the watchdog prints what is there if a phase hangs)
    out = {"metric": "Mrays/sec (live paths x bounces) at 800x800 Cornell depth 8" if args.config == "c2" else
                     "Mrays/sec (live paths x bounces), config %s" % args.config,
           "value": None, "unit": "Mrays/s", "n_gpus": n_tiles, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": None, "higher_is_better": True, "scaling": args.scaling if n_tiles > 1 else "weak",
           "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {}}
    guard = Watchdog(rank, out) if multi else None

    def phase(name, seconds):
        if guard:
            guard.arm(seconds, name)

    def put(where, key, value):
        if guard:
            guard.put(where, key, value)
        else:
            where[key] = value

    if dist_on:
        phase("process group: all_gather of the ranks' devices", args.sub_timeout)
        put(out, "ranks", rank_identities(c))
    elif inproc:
        put(out, "ranks", {"world_size": 1, "backend": "in-library", "devices": ["cuda:%d" % d for d in ([0] * args.gpus if args.same_device else range(args.gpus))]})

    # ---- the drop-in calling pattern, N = 1: one pathtrace() per iteration.  Measured FIRST, in the
    # state a host finds the device in (after a 64-spp session -- 20 GB of pools on five streams -- the copy engine
    # moved PT_ASYNC_IMAGE's snapshots at 60 % of the rate it reaches in a fresh process). ----
    pc = None
    if world == 1 and not inproc and not args.force_dist and not args.no_per_call and args.config in ("c2", "c3", "c5") \
            and not (flags & ~(pt.PT_COMPACT | pt.PT_SORT_MATERIAL)):
        pc = per_call_rates(c)

    # ---- the main timed pass comes FIRST: `value` must not wait behind a phase whose collective
    # pattern differs (the per-iteration exchange follows it, then strong scaling) ----
    phase("main timed pass", max(300.0, args.sub_timeout))
    s = Session(c, args.scaling, args.reduce_every)
    per_step_iters, every = s.per_step_iters, s.every
    profile_on = not args.no_roofline and args.steps * (scene.traceDepth + 2) * -(-per_step_iters // every) <= 2000
    dt, rays, first, rank_rays = s.timed(args.steps, args.warmup)
    value = rays / dt / 1e6
    put(out, "value", round(value, 2))
    put(out, "ms_per_step", round(dt * 1e3 / args.steps, 4))

    # ---- roofline of the dominant kernel: the timed region is run a second time, identically, with HIP
    # events bracketing every kernel launch on the launch stream (2 events per launch from a preallocated
    # pool, no host sync until the region ends).  The events themselves cost ~8 us per launch (25 % at
    # 1 spp per step, 3 % at 16), so they stay out of the pass `value` is computed from. ----
    roofline = None
    if profile_on:
        pt.set_profiling(True)
        rays0, first0, _ = pt.counters()
        s.barrier()
        for _ in range(args.steps):
            s.step()
        s.barrier()
        rays1, first1, _ = pt.counters()
        rank_rays, first = rays1 - rays0, first1 - first0
        prof = pt.get_profile()
        pt.set_profiling(False)
        roofline = roofline_object(args, n_tiles if inproc else world, flags, pt, prof, rank_rays, first, dt / args.steps)

    digest = None
    if args.digest:
        import hashlib
        s.barrier()
        final = s.image
        if inproc:
            final = torch.from_numpy(pt.get_image(npix))        # the frame device 0 assembled
        if dist_on:
            # the frame rank 0 holds after the last exchange IS the result (gather: copies; reduce: sum with zeros)
            final = s.frame
        torch.cuda.synchronize()
        if rank == 0:
            digest = hashlib.md5(final.cpu().numpy().tobytes()).hexdigest()
    # ---- sustained (after the digest, which is of the timed steps): the timed region is 20 steps = 62 ms; pathtrace.cu:284-393 runs 5000 times per image.
    # The same steps again for at least two seconds of wall time, with the shader clock read (one wave, cycle counter against
    # the 100-MHz counter: pt_probe_clock) while they run; `steps` / `ms_per_step` / `value` stay the short pass's.
    sustained = None
    if not args.no_sustained and args.sustain_seconds > 0 and not multi:
        n_chunk = max(1, args.steps)
        s.barrier()
        r0, _, _ = pt.counters()
        t0 = time.perf_counter()
        done, clocks = 0, []
        while True:
            for _ in range(n_chunk):
                s.step()
            done += n_chunk
            if len(clocks) < 8:
                try:
                    clocks.append(pt.probe_clock(300))      # beside the steps just enqueued
                except Exception:
                    pass
            s.barrier()
            if time.perf_counter() - t0 >= args.sustain_seconds:
                break
        el = time.perf_counter() - t0
        r1, _, _ = pt.counters()
        sustained = {"seconds": round(el, 3), "steps": done, "mrays_per_s": round((r1 - r0) / el / 1e6, 2),
                     "ms_per_step": round(el / done * 1e3, 4), "ratio_to_value": round((r1 - r0) / el / 1e6 / value, 4),
                     "shader_clock_ghz": round(sorted(clocks)[len(clocks) // 2], 3) if clocks else None,
                     "shader_clock_samples": [round(x, 3) for x in clocks],
                     "note": "the timed steps repeated for >= %.0f s of wall time in chunks of %d steps (one synchronisation per chunk); "
                             "shader clock = one wave counting its cycle counter against the 100-MHz counter for 300 us beside each of the first "
                             "chunks (include/ptmi355.h: pt_probe_clock); the issue roof of `roofline` is priced at %.1f GHz" % (args.sustain_seconds, n_chunk, PEAK_CLOCK_GHZ)}

    gather_desc = ("a gather of the packed tile rows (%.2f MB per rank)" % (s.bytes_per_rank / 1e6)) if (s.gather or s.gather_thread) \
        else "reduce(SUM) of the zero-padded frames (%.2f MB per rank)" % (npix * 12 / 1e6)
    transport = s.transport
    exchanges_per_step = s.exchanges_per_step
    s.close()

    config = {"workload": "scenes/cornell.txt (%s) %dx%d depth %d, compaction on, %d spp per step per GPU-tile"
                          % (c.scene_name, W, H, scene.traceDepth, per_step_iters),
              "batch_spp": args.batch, "flags": args.flags,
              "sharding": ("in the library (one process): interleaved %d-row strips over %d devices, one host thread and "
                           "stream per device; %s scaling; after every batch of %d iterations the tiles' running sums "
                           "travel to device 0 (%s) and are unpacked into the frame, overlapped with the next batch"
                           % (args.strip_rows, n_tiles, args.scaling, every, transport)) if inproc else
              "whole frame" if not dist_on else
              "interleaved %d-row strips over %d GPUs; %s scaling; `value`: tiles' running sums to rank 0 every %d "
              "iterations (once per %s) by %s, overlapped with the next batch; the exchange after EVERY iteration "
              "(north_star's cadence) is `per_iteration_exchange`"
              % (args.strip_rows, world, args.scaling, every, "step" if every == per_step_iters else "%d iterations" % every, gather_desc),
              "exchange_every_iterations": 0 if not multi else every,
              "exchanges_per_step": 0 if not multi else exchanges_per_step,
              "rays_per_step": int(rays / args.steps)}
    if guard:
        with guard.lock:
            out["config"].update(config)
    else:
        out["config"].update(config)
    if digest:
        put(out, "image_md5", digest)
    if roofline:
        if sustained:
            roofline["sustained"] = sustained
        put(out, "roofline", roofline)
    elif sustained:
        put(out, "sustained", sustained)

    # ---- CPU baseline: the oracle (plain-C port) on this host, rank 0, N = 1 only ----
    if rank == 0 and world == 1 and not inproc and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(scene)

    if pc:
        out["config"]["per_call"] = pc
        out["config"]["pcie_inclusive_mrays_per_s"] = pc["pcie_inclusive_sync"]           # (the names of rounds 1-3)
        out["config"]["pcie_inclusive_async_mrays_per_s"] = pc["pcie_inclusive_async"]
        if "host_image_md5" in pc:
            out["config"]["pcie_host_image_md5"] = pc["host_image_md5"]

    # ---- N > 1: strong scaling (and, on request, the frame assembled in ONE shared host buffer by the ranks' own
    # launches) beside the main number; a failure is stated in the line ----
    if multi and not args.no_sub and args.phase == "all":
        # north star's cadence -- the tiles' sums on rank 0 after EVERY iteration -- beside the same calls without the exchange
        phase("per-iteration exchange", args.sub_timeout)
        try:
            pie = per_iteration_exchange(c)
        except Exception as e:                 # stated in the line, never silently absent
            pie = {"failed": str(e)[:300]}
        put(out, "per_iteration_exchange", pie)
        put(out["config"], "per_iteration_exchange", pie)      # (where rounds 3-4 had it)
        if n_tiles > 1:
            phase("strong scaling", args.sub_timeout)
            try:
                put(out["config"], "strong", strong_scaling(c, args.steps, args.warmup))
            except Exception as e:
                put(out["config"], "strong", {"failed": str(e)[:300]})
        if dist_on and args.shared_frame:
            phase("shared host frame", args.sub_timeout)
            try:                                          # (every rank fails at the same call or none does: same build, same flags)
                put(out["config"], "per_iteration_shared_frame", shared_frame_rate(c, args.sub_iters))
            except Exception as e:
                put(out["config"], "per_iteration_shared_frame", {"failed": str(e)[:300]})
    if guard:
        guard.emit()
    elif rank == 0:
        print(json.dumps(out))
    teardown(c)


def teardown(c):
    if not c.dist_on:
        return
    # the line is out and complete; a teardown that never returns must not keep the launcher waiting
    import threading

    def teardown_expired():
        sys.stderr.write("bench.py: destroy_process_group did not return in 60 s (the line above is complete)\n")
        os._exit(0)
    t = threading.Timer(60.0, teardown_expired)
    t.daemon = True
    t.start()
    c.dist.destroy_process_group()
    t.cancel()


def sub_phase(c):
    """`--phase exchange | strong | shared`: ONE sub-measurement of an N > 1 line in processes (and a rendezvous) of its own;
    rank 0 prints {"phase": ..., <key>: {...}} -- a failure is an object with `failed`, a hang ends with the watchdog's
    object and exit code 3 -- and the launching parent (self_launch) merges it into the line."""
    args = c.args
    multi = c.dist_on or c.inproc
    key, fn = {"exchange": ("per_iteration_exchange", lambda: per_iteration_exchange(c)),
               "strong": ("strong", lambda: strong_scaling(c, args.steps, args.warmup)),
               "shared": ("per_iteration_shared_frame", lambda: shared_frame_rate(c, args.sub_iters))}[args.phase]
    out = {"phase": args.phase, key: None}
    guard = Watchdog(c.rank, out) if multi else None
    if guard:
        guard.arm(args.sub_timeout, args.phase)
    if not multi:
        v = {"failed": "phase '%s' needs more than one rank (or --force-dist / --inproc)" % args.phase}
    else:
        try:
            v = fn()
        except Exception as e:
            v = {"failed": str(e)[:300]}
    if guard:
        guard.put(out, key, v)
        guard.emit()
    elif c.rank == 0:
        out[key] = v
        print(json.dumps(out))
    teardown(c)


def roofline_object(args, world, flags, pt, prof, rank_rays, first, timed_step_s):
    """What bounds the dominant kernel(s) of THIS run: launch times measured here with HIP events on the launch stream
    (the timed steps repeated), against
      * the vector-issue roof: issue cycles per launch = the kernel's EXECUTED opcode histogram (per-basic-block counts of
        an instrumented build: profiles/tools/isa_count.py) x the issue cost of each opcode measured one by one
        (profiles/microbench/gen_issue_ops.py), over SIMDs x 2.4 GHz x time;
      * HBM: FETCH_SIZE x 2 + WRITE_SIZE per launch (separate rocprofv3 --pmc passes) over 8 TB/s;
      * fp32: the histogram's floating-point operations x the active-lane fraction (SQ_THREAD_CYCLES_VALU) over 157.3 TF.
    `bound` is whichever of vector issue and HBM is closer to its roof; `frac` is that fraction, never above 1 by
    construction.  Without a counter profile of exactly this build and command line (profiles/collect.py ->
    profiles/traffic.json) the counter-derived fields are null rather than stale.
    The event pass runs its steps one after the other on the launch stream; in the timed pass consecutive steps overlap
    on the device (csrc/pt_h_enqueue.hpp: enqueue_batch_direct), so `value` can exceed `grays_per_s_in_kernel`.
    `timed_pass` therefore states the same issue cycles -- of every profiled kernel of a step -- over the timed
    pass's own time per step: the share of the chip's issue cycles the whole pipeline used while `value` was measured."""
    steps = args.steps
    stage_ms = {k: v[0] for k, v in prof.items() if v[1]}
    launches = {k: int(v[1]) for k, v in prof.items() if v[1]}
    common = {"pass": "the %d timed steps repeated with per-launch HIP events" % steps,
              "stage_ms": {k: round(v, 3) for k, v in stage_ms.items()}, "launches": launches.get("bounce", 0)}
    t = counter_profile(args, world)
    fused = not (flags & pt.PT_UNFUSED) and not ((flags & pt.PT_SORT_MATERIAL) and not (flags & pt.PT_COMPACT))
    ms_b, n_b = prof["bounce"]
    if fused and args.config in ("c2", "c3", "c5") and ms_b > 0:
        # SURVEY 8(d)'s byte model of the UNFUSED reference pipeline, kept beside the measured figures because the
        # contract names it; the fused kernel never moves most of these bytes, so it is not a bound and never `frac`
        survivors = rank_rays - first
        algo = rank_rays * BYTES_PER_RAY + survivors * BYTES_PER_SURVIVOR
        common["hbm_algorithmic_unfused_model"] = {"gb_per_s": round(algo / (ms_b * 1e-3) / 1e9, 1), "bytes_per_launch": int(algo / max(1, n_b)),
                                                   "note": "152 B per ray + 88 B per survivor of the reference's separate kernels; not what this kernel moves"}
        # north_star's own yardstick: ">= 50 % of the HBM roofline on the intersection + compaction pass" -- 8(d)'s algorithmic
        # bytes of those two stages (intersect 44 B/ray; compaction 4 B/ray + 88 B/survivor) over the time the kernels that
        # do them take (HIP events of this run), against 8 TB/s.  Needs no counter profile: re-derivable from this line.
        contract_bytes = (rank_rays * (44 + 4) + survivors * 88) / steps
        common["contract"] = {"frac": round(contract_bytes / (ms_b * 1e-3 / steps) / (HBM_PEAK_GBS * 1e9), 4),
                              "gb_per_s": round(contract_bytes / (ms_b * 1e-3 / steps) / 1e9, 1), "peak": HBM_PEAK_GBS,
                              "bytes_per_step": int(contract_bytes), "kernel_ms_per_step": round(ms_b / steps, 4), "target": 0.5,
                              "note": "SURVEY 8(d): (intersect 44 B + compaction 4 B) per ray + 88 B per survivor, per step, over the bounce "
                                      "kernels' time per step and 8 TB/s -- BASELINE north_star's target figure; the fused kernel moves fewer "
                                      "bytes than this model (hbm_necessary / traffic)"}
    if not t:
        return dict(common, kernel=None, bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None,
                    source="no counter profile for this build and command line (python3 profiles/collect.py)")
    ks = t["kernels"]
    stages = sorted({k["stage"] for k in ks.values() if k["stage"] in stage_ms}, key=lambda s_: -stage_ms[s_])
    if not stages:
        return dict(common, kernel=None, bound="hbm", achieved=None, peak=HBM_PEAK_GBS, unit="GB/s", frac=None, traffic=None, source="profile has no stage of this run")
    st = stages[0]
    sel = [k for k in ks.values() if k["stage"] == st]
    per_step_s = stage_ms[st] * 1e-3 / steps
    nl = sum(k["launches_per_step"] for k in sel)
    cyc = sum(k["launches_per_step"] * k.get("issue_cycles_per_launch", 0.0) for k in sel)
    cyc_guide = sum(k["launches_per_step"] * k.get("issue_cycles_guide_rates_per_launch", 0.0) for k in sel)
    byt = sum(k["launches_per_step"] * k.get("hbm_bytes_per_launch", 0) for k in sel)
    flops = sum(k["launches_per_step"] * k.get("flops_fp32_per_launch_64_lanes", 0.0) * k.get("active_lane_fraction", 1.0) for k in sel)
    lanes = sum(k["launches_per_step"] * k.get("issue_cycles_per_launch", 0.0) * k.get("active_lane_fraction", 0.0) for k in sel) / cyc if cyc else None
    unpriced = max(k.get("unpriced_share_of_cycles", 0.0) for k in sel)
    issue_frac = cyc / per_step_s / (SIMDS * PEAK_CLOCK_GHZ * 1e9)
    hbm_frac = byt / per_step_s / (HBM_PEAK_GBS * 1e9)
    names = " + ".join(sorted({k["name"].split("(")[0] for k in sel}))
    r = dict(common, kernel=names, stage=st, launches_per_step=nl, avg_launch_us=round(per_step_s / nl * 1e6, 2),
             traffic=int(byt / nl) if byt else None,
             source="profiles/traffic.json@build:%s (profiles/collect.py: instrumented-build opcode histogram, rocprofv3 --pmc passes)" % t["build_sha16"],
             valu_issue={"achieved": round(cyc / per_step_s / 1e9, 1), "peak": round(SIMDS * PEAK_CLOCK_GHZ, 1), "unit": "G SIMD issue-cycles/s",
                         "frac": round(issue_frac, 4), "issue_cycles_per_launch": int(cyc / nl),
                         # the same executed-opcode histogram at the GUIDE's issue rates (full rate 2 cycles per wave-instruction
                         # per SIMD, half rate 4, transcendental 8: each opcode in the class its measured cost is nearest to) --
                         # `frac` prices a full-rate opcode at the 2.3-2.5 cycles this repository's own microbenchmark measured
                         # (ramp and tail included); and the part of it that ran lanes that were switched on
                         "frac_guide_rates": round(cyc_guide / per_step_s / (SIMDS * PEAK_CLOCK_GHZ * 1e9), 4) if cyc_guide else None,
                         "useful_frac": round(cyc_guide / per_step_s / (SIMDS * PEAK_CLOCK_GHZ * 1e9) * lanes, 4) if (cyc_guide and lanes) else None,
                         "valu_insts_per_ray": round(64.0 * sum(k["launches_per_step"] * k.get("wave_insts_per_launch", {}).get("valu", 0.0) for k in sel) /
                                                     max(1.0, rank_rays / steps), 1),
                         "unpriced_share_of_cycles": round(unpriced, 4),
                         "issue_cost_table": "profiles/r06/costs_r06.json (profiles/r05/issue_ops_r05.json: one row per opcode, measured in round 5; round 6 added the matrix-pipe stage's opcodes at the guide's figures)",
                         "sq_insts_valu_over_counted": [k.get("sq_insts_valu_over_counted") for k in sel]},
             hbm_measured={"achieved": round(byt / per_step_s / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_frac, 4),
                           "note": "FETCH_SIZE x 2 + WRITE_SIZE per launch (gfx950 correction)"},
             fp32={"achieved": round(flops / per_step_s / 1e12, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(flops / per_step_s / 157.3e12, 4),
                   "active_lane_fraction": round(lanes, 4) if lanes else None,
                   "note": "fp32 operations of the executed opcode histogram (fma = 2) x active lanes (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU / 64)"},
             grays_per_s_in_kernel=round(rank_rays / (stage_ms[st] * 1e-3) / 1e9, 3) if st == "bounce" else None)
    mf = sum(k["launches_per_step"] * k.get("mfma_flops_per_launch", 0.0) for k in sel)
    if mf:
        # the matrix pipe (C4 as stated: the sphere reject of every (ray, triangle) pair as one bilinear form, v_mfma_f32_16x16x32_f16):
        # executed MFMAs of the counted build x 16 x 16 x 32 x 2 over the kernel time, against the dense binary16 peak
        r["mfma"] = {"achieved": round(mf / per_step_s / 1e12, 1), "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(mf / per_step_s / (MFMA_F16_PEAK_TFLOPS * 1e12), 4),
                     "pairs_per_s": round(mf / (2.0 * 32) / per_step_s, 0),
                     "note": "v_mfma_f32_16x16x32_f16 wave-instructions of the instrumented build x 16384 flops, over the kernel time; peak = MI355X dense f16 (MI355X_MICROARCH.md); "
                             "one output element = one (ray, triangle) pair of the bounding-sphere reject (csrc/pt_k_trisweep.hpp)"}
    if st == "bounce" and fused and args.config in ("c2", "c3", "c5"):
        surv_step, first_step = (rank_rays - first) / steps, first / steps
        # (b) what the fused kernel MUST move: a pool row (40 B) read per ray that was not generated in registers, a pool
        # row written per survivor, 16 B per path that ends with a non-zero colour (19.5 % of the ending paths on C2:
        # profiles/r04/traffic_accounting_c2.md) -- per launch like `traffic`, averaged over the step's launches
        nz = 0.195 if args.config == "c2" else 0.25
        need = (40.0 * surv_step + 40.0 * surv_step + 16.0 * nz * first_step) / nl
        r["hbm_necessary"] = {"bytes_per_launch": int(need), "measured_over_necessary": round(byt / nl / need, 3) if byt else None,
                              "note": "40 B per pool row read (every ray but the camera rays, which are generated in registers) + 40 B per "
                                      "survivor written + 16 B per path ending with a non-zero colour (%.1f %% of the ending paths), averaged "
                                      "over the %d launches of a step like `traffic`" % (100 * nz, nl)}
    # the scalar pipe beside it (round 5): one scalar ALU / branch / scalar-load instruction per ~4.4 cycles per SIMD, a wait
    # or nop ~1.2 (profiles/r04/issue_ops_scalar.txt); executed counts from the same instrumented build.  It issues beside
    # the vector pipe, so the two fractions do not add -- but when both are near 1 a cut on one side only moves the limit
    sc_cyc = sum(k["launches_per_step"] * (4.42 * (k.get("wave_insts_per_launch", {}).get("salu", 0.0) + k.get("wave_insts_per_launch", {}).get("branch", 0.0) +
                                                   k.get("wave_insts_per_launch", {}).get("smem", 0.0)) +
                                           1.2 * k.get("wave_insts_per_launch", {}).get("sopp", 0.0)) for k in sel)
    if sc_cyc:
        r["scalar_issue"] = {"frac": round(sc_cyc / per_step_s / (SIMDS * PEAK_CLOCK_GHZ * 1e9), 4),
                             "scalar_insts_per_ray": round(64.0 * sum(k["launches_per_step"] * sum(k.get("wave_insts_per_launch", {}).get(c, 0.0) for c in ("salu", "branch", "smem", "sopp"))
                                                                      for k in sel) / max(1.0, rank_rays / steps), 1),
                             "note": "scalar ALU + branch + scalar-load instructions x 4.42 cycles, waits / nops x 1.2, over 1024 SIMDs x 2.4 GHz x the kernel time"}
    cyc_all = sum(k["launches_per_step"] * k.get("issue_cycles_per_launch", 0.0) for k in ks.values())
    byt_all = sum(k["launches_per_step"] * k.get("hbm_bytes_per_launch", 0) for k in ks.values())
    if timed_step_s > 0:
        r["timed_pass"] = {"ms_per_step": round(timed_step_s * 1e3, 4), "event_pass_ms_per_step": round(sum(stage_ms.values()) / steps, 4),
                           "valu_issue_frac": round(min(1.0, cyc_all / timed_step_s / (SIMDS * PEAK_CLOCK_GHZ * 1e9)), 4),
                           "hbm_frac": round(min(1.0, byt_all / timed_step_s / (HBM_PEAK_GBS * 1e9)), 4),
                           "note": "consecutive steps overlap on the device in the timed pass (serial in the event pass): all profiled kernels' "
                                   "issue cycles / HBM bytes per step over the timed pass's time per step"}
    # the shader clock this kernel family was MEASURED at (a diagnostic build stamps every wave's life with the cycle and the
    # real-time counter: profiles/tools/wave_clock.py): against it the same issue cycles are a larger share of what the chip
    # delivered.  A provenance figure from profiles/, like the opcode histogram; `frac` stays against the 2.4 GHz peak.
    try:
        import re
        meds = [float(m) for m in re.findall(r"median ([0-9.]+) GHz", open(os.path.join(ROOT, "profiles", "r04", "wave_clock.txt")).read())]
        if meds and args.config == "c2":
            clk = sum(meds) / len(meds)
            # (round 5's `frac_at_measured_clock` is gone: dividing by the clock the chip throttles to states a utilisation of
            # what was delivered, not an efficiency; the clock THIS run held is `sustained.shader_clock_ghz`)
            r["valu_issue"]["measured_clock_ghz"] = round(clk, 3)
            r["valu_issue"]["clock_source"] = "profiles/r04/wave_clock.txt: k_bounce on C2, per-wave cycle counter against the 100-MHz real-time counter"
    except Exception:
        pass
    if mf and r["mfma"]["frac"] >= max(issue_frac, hbm_frac):
        r.update(bound="mfma", achieved=r["mfma"]["achieved"], peak=MFMA_F16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(min(1.0, r["mfma"]["frac"]), 4))
    elif issue_frac >= hbm_frac:
        r.update(bound="valu-issue", achieved=r["valu_issue"]["achieved"], peak=r["valu_issue"]["peak"], unit=r["valu_issue"]["unit"], frac=round(min(1.0, issue_frac), 4))
    else:
        r.update(bound="hbm", achieved=r["hbm_measured"]["achieved"], peak=HBM_PEAK_GBS, unit="GB/s", frac=round(min(1.0, hbm_frac), 4))
    return r


def counter_profile(args, world):
    """The entry of profiles/traffic.json (written by profiles/collect.py) for THIS build (hash of the kernel
    sources and the compile flags) and THIS command line; None otherwise -- stale counters are never reported."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        t = json.load(open(path))["configs"]["%s|%s|%d" % (args.config, args.flags, args.batch)]
    except Exception:
        return None
    if t.get("build_sha16") != build_digest():
        return None
    return t


def _total_rays(pt):
    return pt.total_rays()            # device-side counter; synchronises the stream


def usable_cpus():
    """CPUs this process may actually use: the affinity mask, further limited by the cgroup CPU quota (the GPU
    boxes of this pool show 256 hardware threads but grant 16 CPUs of time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", ):
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, int(int(quota) / int(period))))
        except Exception:
            pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and p > 0:
            n = min(n, max(1, q // p))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(scene):
    """The oracle (plain-C port of the reference's path) on this host's CPUs, SURVEY 8(d): (i) ONE thread, (ii) every
    CPU the process may use, with the core count and the CPU model stated.  `value` is the multi-thread figure in the
    decomposition the GPU batch uses -- one whole iteration per thread (iterations are independent), images summed in
    iteration order; `path_ranges` is the other decomposition BASELINE.md section 4 describes, threads over contiguous path
    ranges of ONE iteration (a barrier per bounce).  ~0.1 GB of scratch per thread, so the thread count is also bounded
    by the free memory."""
    from oracle import pyoracle as po
    ncores = usable_cpus()
    try:
        import psutil
        ncores = max(1, min(ncores, int(psutil.virtual_memory().available / (160 << 20))))
    except Exception:
        pass
    tr = po.Tracer(scene.geoms.view(po.GEOM_DT), scene.materials.view(po.MATERIAL_DT),
                   scene.camera.view(po.CAMERA_DT), scene.traceDepth, flags=po.F_COMPACT,
                   trig=po.TRIG_SHARED)
    t0 = time.perf_counter()
    st1 = tr.iterate(1)                                     # (i) one thread, one iteration
    el1 = time.perf_counter() - t0
    one = {"value": round(st1.rays / el1 / 1e6, 3), "unit": "Mrays/s", "cores": 1,
           "sample": "iteration 1 of the same workload (%.1f s), oracle/ptoracle.c, single thread" % el1}
    rays, t0, iters = 0, time.perf_counter(), 0
    while True:
        rays += tr.iterate_parallel(2 + iters, ncores, ncores)
        iters += ncores
        el = time.perf_counter() - t0
        if el > 10.0 or iters >= 4 * ncores:
            break
    out = {"value": round(rays / el / 1e6, 3), "unit": "Mrays/s", "cores": ncores, "kind": "port", "cpu_model": cpu_model(),
           "decomposition": "one whole iteration per thread",
           "sample": "%d iterations of the same 800x800 depth-8 Cornell workload (%.1f s), oracle/ptoracle.c, one "
                     "whole iteration per thread on %d pthreads" % (iters, el, ncores),
           "one_thread": one}
    try:
        t0 = time.perf_counter()
        r2, n2 = 0, 0
        while n2 < 8 and time.perf_counter() - t0 < 5.0:
            r2 += tr.iterate(2 + iters + n2, threads=ncores).rays
            n2 += 1
        el2 = time.perf_counter() - t0
        out["path_ranges"] = {"value": round(r2 / el2 / 1e6, 3), "unit": "Mrays/s", "cores": ncores,
                              "sample": "%d iterations (%.1f s), %d pthreads over contiguous path ranges of one iteration "
                                        "(BASELINE.md section 4's decomposition)" % (n2, el2, ncores)}
    except Exception as e:
        out["path_ranges"] = {"error": str(e)[:200]}
    ref = reference_headers_baseline(scene, po)
    if ref:
        out["reference_headers"] = ref
    c1 = c1_baseline(po)
    if c1:
        out["c1"] = c1
    return out


def c1_baseline(po):
    """BASELINE configs[0] exactly as stated: scenes/cornell_diffuse.txt (400 x 400, depth 4, diffuse only), 1 spp,
    the oracle's plain single-thread loop."""
    try:
        z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
        g = lambda k: z["cornell_diffuse__" + k]
        tr = po.Tracer(g("geoms"), g("materials"), g("camera"), int(g("depth")), flags=po.F_COMPACT, trig=po.TRIG_SHARED)
        t0 = time.perf_counter()
        st = tr.iterate(1)
        el = time.perf_counter() - t0
        return {"value": round(st.rays / el / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
                "sample": "configs[0]: cornell_diffuse 400x400, 1 spp, depth 4, %d rays in %.2f s, single thread" % (st.rays, el)}
    except Exception as e:
        return {"error": str(e)[:200]}


def reference_headers_baseline(scene, po):
    """The same workload through the REFERENCE'S OWN __host__ headers (intersections.h, interactions.h, thrust RNG),
    compiled in the build container into oracle/_ref/libptref_b_shared.so (it travels with the snapshot; the
    reference sources do not): one iteration, one thread -- the plain C++ loop north_star asks to time beside the
    GPU.  None when the library is absent."""
    import ctypes as C
    try:
        if not po.ref_available():
            return None
        L = po.ref("b_shared")
        L.ref_trace_iteration.restype = C.c_longlong
        W, H = scene.resolution
        n = W * H
        img = np.zeros((n, 3), dtype=np.float32)
        live = np.zeros(64, dtype=np.int32)
        geoms = np.ascontiguousarray(scene.geoms)
        mats = np.ascontiguousarray(scene.materials)
        cam = np.ascontiguousarray(scene.camera)
        t0 = time.perf_counter()
        rays = L.ref_trace_iteration(geoms.ctypes.data_as(C.c_void_p), len(geoms), mats.ctypes.data_as(C.c_void_p),
                                     cam.ctypes.data_as(C.c_void_p), scene.traceDepth, 1, 1,
                                     img.ctypes.data_as(C.c_void_p), live.ctypes.data_as(C.c_void_p), None)
        el = time.perf_counter() - t0
        return {"value": round(rays / el / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": "iteration 1 of the same workload (%.1f s) through the reference's intersections.h / "
                          "interactions.h + rocThrust, single thread" % el}
    except Exception as e:        # a baseline must never break the bench line
        return {"error": str(e)[:200]}


if __name__ == "__main__":
    main()
