"""bench.py's contract (one JSON line, required keys) and the N > 1 launch path.  The 2-rank run
uses gloo with both ranks on cuda:0 (RCCL refuses two ranks on one device): same code path as the
driver's torch.distributed.run launch except for the collective backend; the reduced frame must be
bit-identical to the 1-rank frame over the same iterations."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(cmd, timeout=600, per_bounce=True, sustained=False):
    if not sustained:
        cmd = list(cmd) + ["--no-sustained"]          # (roofline.sustained repeats the timed steps for two seconds: one test looks at it)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if per_bounce:
        env["PTMI355_WHOLE_MAX"] = "0"       # a kernel per bounce whatever the batch size (the launch count is asserted)
    else:
        env.pop("PTMI355_WHOLE_MAX", None)
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line)


def _check_roofline(r):
    """A fraction comes from measured counters of exactly this build and command line (profiles/traffic.json) and never
    exceeds 1; without such a profile every counter-derived field is null, never stale, never a model's number."""
    assert r["bound"] in ("valu-issue", "hbm", "mfma") and "traffic" in r and "source" in r and "frac" in r
    if r["frac"] is not None:
        assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - min(1.0, r["achieved"] / r["peak"])) < 2e-3
        assert r["traffic"] > 0 and r["valu_issue"]["unpriced_share_of_cycles"] < 0.05
        assert r["valu_issue"]["frac"] <= 1.0 and r["hbm_measured"]["frac"] <= 1.0 and 0.0 < r["fp32"]["frac"] <= 1.0
        assert 0.0 < r["fp32"]["active_lane_fraction"] <= 1.0
        assert 0.0 < r["timed_pass"]["valu_issue_frac"] <= 1.0 and 0.0 < r["timed_pass"]["hbm_frac"] <= 1.0
    else:
        assert r["traffic"] is None and r["achieved"] is None


def test_default_line_and_its_counter_profile():
    """The driver's command line (defaults: C2, 64 spp per step): when profiles/traffic.json holds this build's profile
    the roofline object carries measured fractions; they are checked for consistency either way."""
    d = run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], per_bounce=False, sustained=True)
    assert d["config"]["batch_spp"] == 64 and d["value"] > 1000
    _check_roofline(d["roofline"])
    assert d["roofline"]["launches"] == 5 * 8
    # what round 5's line left out: the timed steps repeated for two seconds with the shader clock read
    # beside them, the issue fraction at the guide's rates and the part of it that ran switched-on lanes
    su = d["roofline"]["sustained"]
    assert su["seconds"] >= 2.0 and su["steps"] >= 5 and su["mrays_per_s"] > 1000 and 0.6 < su["ratio_to_value"] < 1.2
    assert abs(su["mrays_per_s"] - d["config"]["rays_per_step"] * su["steps"] / su["seconds"] / 1e6) / su["mrays_per_s"] < 0.02
    assert su["shader_clock_ghz"] is not None and 1.0 < su["shader_clock_ghz"] < 2.6
    if d["roofline"]["frac"] is not None:
        v = d["roofline"]["valu_issue"]
        assert "frac_at_measured_clock" not in v and 0.0 < v["useful_frac"] < v["frac_guide_rates"] < v["frac"] <= 1.0
        assert abs(v["useful_frac"] - v["frac_guide_rates"] * d["roofline"]["fp32"]["active_lane_fraction"]) < 2e-3
    # figures a reader can re-derive from SURVEY 8(d) and the line itself: north_star's yardstick --
    # the intersect + compaction bytes (48 B per ray + 88 B per survivor) over the bounce kernels' time and 8 TB/s ...
    c = d["roofline"]["contract"]
    rays = d["config"]["rays_per_step"]
    survivors = rays - 64 * 640000
    assert abs(c["bytes_per_step"] / (48 * rays + 88 * survivors) - 1.0) < 1e-3          # (the event pass traces other iterations than the timed pass)
    assert abs(c["frac"] - c["bytes_per_step"] / (c["kernel_ms_per_step"] * 1e-3) / 8e12) < 2e-3 and 0.3 < c["frac"] < 1.0
    # ... and, when the counter profile is current, what the fused kernel must move beside what it measurably moves
    if d["roofline"]["frac"] is not None:
        n = d["roofline"]["hbm_necessary"]
        assert 0.95 < n["measured_over_necessary"] < 1.3 and abs(n["measured_over_necessary"] - d["roofline"]["traffic"] / n["bytes_per_launch"]) < 2e-3


def test_bench_digest_equals_oracle():
    """The bench's OWN path -- the driver's command line: 64 spp per step, asynchronous batches overlapped on the lanes,
    default launch plan -- against the oracle: the accumulation image after 1 warm-up + 2 timed
    steps (iterations 1..192) is, byte for byte, the oracle's sum of the same iterations (summed in iteration order;
    16 pthreads, one whole iteration each), and so is the rays-per-step count the metric's numerator comes from."""
    import hashlib
    import numpy as np
    from oracle import pyoracle as po
    d = run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-per-call",
             "--digest"], per_bounce=False)
    assert d["config"]["batch_spp"] == 64 and d["steps"] == 2
    z = np.load(os.path.join(ROOT, "tests", "golden", "scenes.npz"))
    g = lambda k: z["cornell__" + k]
    tr = po.Tracer(g("geoms"), g("materials"), g("camera"), int(g("depth")), flags=po.F_COMPACT, trig=po.TRIG_SHARED)
    threads = max(1, min(16, len(os.sched_getaffinity(0))))
    rays = [tr.iterate_parallel(1 + 64 * k, 64, threads) for k in range(3)]
    assert hashlib.md5(tr.image.tobytes()).hexdigest() == d["image_md5"]
    assert d["config"]["rays_per_step"] == (rays[1] + rays[2]) // 2


def test_single_gpu_line():
    d = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
             "--digest", "--pcie"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    # the reference's calling pattern (one pathtrace() per iteration): without the host image, with it synchronously, PT_ASYNC_IMAGE
    assert d["config"]["pcie_inclusive_mrays_per_s"] > 100 and d["config"]["pcie_inclusive_async_mrays_per_s"] > 100
    pc = d["config"]["per_call"]
    # one pathtrace() per iteration with the host image: traced ahead of the caller (PT_LOOKAHEAD, what the shim sets) against
    # every iteration inside its own call
    assert pc["pcie_inclusive_sync"] > pc["pcie_inclusive_sync_no_lookahead"] > 100 and pc["lookahead_no_host_image"] > pc["pcie_inclusive_sync"]
    assert pc["mrays_per_s"] > pc["pcie_inclusive_sync_no_lookahead"] and pc["pcie_inclusive_async"] > 100 and pc["calls"] >= 64
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["unit"] == "Mrays/s" and d["value"] > 100
    _check_roofline(d["roofline"])
    assert d["roofline"]["launches"] == 3 * 8
    test_single_gpu_line.md5 = d["image_md5"]


def test_one_iteration_per_call_with_the_host_image():
    """bench.py --pcie at 1 spp per step under the default launch plan: the iteration runs as ONE launch whose waves
    write the running sum into the (page-locked, device-mapped) host image themselves; the frame after the timed
    steps equals the frame of the launch-per-bounce plan with a copy after every iteration."""
    args = [sys.executable, "bench.py", "--steps", "6", "--warmup", "2", "--batch", "1", "--no-cpu-baseline",
            "--no-roofline", "--digest", "--pcie"]
    one = run(args, per_bounce=False)
    per = run(args, per_bounce=True)
    assert one["config"]["pcie_inclusive_mrays_per_s"] > 100 and per["config"]["pcie_inclusive_mrays_per_s"] > 100
    assert one["image_md5"] == per["image_md5"]
    assert one["config"]["pcie_host_image_md5"] == per["config"]["pcie_host_image_md5"]     # 136 calls' running sum, on the host


def _two(port, *extra, backend="gloo", same_device=True, per_bounce=True):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3",
           "--warmup", "1", "--no-roofline", "--backend", backend, "--digest", "--sub-iters", "8", "--shared-frame"] + list(extra)
    if same_device:
        cmd.append("--same-device")
    return run(cmd, per_bounce=per_bounce)


def test_two_ranks_same_frame():
    """Two PROCESSES of the HIP library on one GPU (gloo: RCCL refuses two ranks on one device), every exchange
    mode: the frame rank 0 assembles is bit-identical to the 1-process frame over the same iterations."""
    one = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
               "--no-roofline", "--digest"])
    # weak scaling, 2 x 2 iterations per step: gather once per step; gather every iteration; full-frame reduce
    for k, extra in enumerate((["--batch", "2"], ["--batch", "2", "--reduce-every", "1"],
                               ["--batch", "2", "--collective", "reduce"],
                               ["--batch", "4", "--scaling", "strong", "--reduce-every", "3"],
                               ["--batch", "2", "--reduce-every", "1", "--exchange-thread"])):
        two = _two(29533 + k, *extra)
        assert two["n_gpus"] == 2 and two["scaling"] == ("strong" if "strong" in extra else "weak")
        assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"], extra      # same 4 iterations per step
        assert two["image_md5"] == one["image_md5"], extra
        # the north star's questions ride along in every N > 1 line: the exchange after every iteration (next to the same
        # cadence without it; issued from the tracing thread, or from sharding.TileGatherThread with --exchange-thread)
        # and strong scaling
        pie, strong = two["config"]["per_iteration_exchange"], two["config"]["strong"]
        assert pie["mrays_per_s"] > 10 and pie["no_exchange_mrays_per_s"] > 10 and pie["iterations"] >= 8, pie
        assert ("exchange thread" if "--exchange-thread" in extra else "tracing thread") in pie["transport"] \
            or extra == ["--batch", "2", "--collective", "reduce"], pie
        assert strong["mrays_per_s"] > 10 and strong["spp_per_step_per_frame"] == int(extra[1]), strong
        # the north star's cadence is a top-level object of every N > 1 line, and the line says what the backend spans
        assert two["per_iteration_exchange"] == pie and two["config"]["exchange_every_iterations"] >= 1
        assert two["ranks"]["world_size"] == 2 and two["ranks"]["backend"] == "gloo" and len(two["ranks"]["devices"]) == 2
        # ... and the frame assembled in ONE shared host buffer by the ranks' own launches, no exchange (PT_SHARED_IMAGE,
        # --shared-frame): refused under this helper's kernel-per-bounce plan (it needs one-launch iterations), measured below
        assert "one launch" in two["config"]["per_iteration_shared_frame"]["failed"], two["config"]["per_iteration_shared_frame"]
    two = _two(29539, "--batch", "2", per_bounce=False)
    assert two["image_md5"] == one["image_md5"]
    shared = two["config"]["per_iteration_shared_frame"]
    assert shared["frame_equals_every_ranks_tile"] is True and shared["mrays_per_s"] > 10 and shared["iterations"] >= 8, shared


def test_plain_gpus_2_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` the way the driver runs `--gpus 1` -- no launcher, no WORLD_SIZE: bench.py starts the two
    ranks itself (child processes of a parent that has not touched the GPU) and relays rank 0's line; same frame as the
    1-process run, the per-iteration exchange in the line, the backend's world size stated."""
    one = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
               "--no-roofline", "--digest"])
    env = {k: os.environ.pop(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT") if k in os.environ}
    try:
        two = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2", "--no-roofline",
                   "--backend", "gloo", "--same-device", "--digest", "--sub-iters", "8", "--launch-timeout", "500"])
    finally:
        os.environ.update(env)
    assert two["n_gpus"] == 2 and two["image_md5"] == one["image_md5"]
    assert two["config"]["rays_per_step"] == one["config"]["rays_per_step"]
    assert two["ranks"]["world_size"] == 2 and two["ranks"]["distinct_devices"] == 1          # (--same-device)
    pie = two["per_iteration_exchange"]
    assert pie["mrays_per_s"] > 10 and pie["no_exchange_mrays_per_s"] > 10 and pie["iterations"] >= 8, pie
    assert "watchdog" not in two and two["config"]["strong"]["mrays_per_s"] > 10


def _ranks(n, port, *extra):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(n), "--steps", "2",
           "--warmup", "1", "--no-roofline", "--backend", "gloo", "--same-device", "--digest", "--sub-iters", "8"] + list(extra)
    return run(cmd, timeout=900)


@pytest.mark.parametrize("config", ["c2", "c5"])
def test_four_ranks_same_frame(config):
    """Rehearsal of the driver's scaling run with as many PROCESSES of the HIP library as this pool lets one GPU hold
    with a margin (the box allows six processes on the GPU: four ranks, their launcher, and the test runner if an earlier
    test has already opened the device; eight ranks run in tests/test_sharding_gloo.py on the CPU): strips of 7 rows --
    115 strips for C2's 800 rows, 309 for C5's 2160 (`--config c5`), the last one short -- do not divide by four, so
    the ranks own different numbers of rows; weak scaling (4 x 1 iterations per step per tile), gather per step and every
    second iteration -- rank 0's frame is the 1-process frame."""
    one = run([sys.executable, "bench.py", "--config", config, "--steps", "2", "--warmup", "1", "--batch", "4",
               "--no-cpu-baseline", "--no-roofline", "--digest"])
    for k, extra in enumerate((["--batch", "1"], ["--batch", "1", "--reduce-every", "2"])):
        four = _ranks(4, 29833 + k + (10 if config == "c5" else 0), "--config", config, "--strip-rows", "7", *extra)
        assert four["n_gpus"] == 4 and four["scaling"] == "weak"
        assert four["config"]["rays_per_step"] == one["config"]["rays_per_step"], extra
        assert four["image_md5"] == one["image_md5"], (config, extra)


def test_one_rank_rccl():
    """The N > 1 code path over RCCL ITSELF on this one GPU: a world of one rank (bench.py --force-dist) initialises
    the RCCL process group on the device, packs and gathers its tile (the whole frame) per step / per iteration,
    reduces the full frame, and takes the max / sum over ranks -- every RCCL call the driver's multi-GPU launch makes
    executes on hardware, and the assembled frame equals the plain 1-process frame bit for bit."""
    one = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
               "--no-roofline", "--digest"])
    for k, extra in enumerate(([], ["--reduce-every", "1"], ["--collective", "reduce"], ["--scaling", "strong", "--reduce-every", "3"])):
        env_port = str(29733 + k)
        os.environ["MASTER_PORT"] = env_port
        try:
            d = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
                     "--no-roofline", "--digest", "--force-dist", "--backend", "nccl", "--sub-iters", "16"] + extra)
        finally:
            os.environ.pop("MASTER_PORT", None)
        assert d["n_gpus"] == 1 and d["config"]["exchanges_per_step"] >= 1, extra
        pie = d["config"]["per_iteration_exchange"]                    # RCCL gather per iteration
        assert pie["mrays_per_s"] > 10 and 0.05 < pie["ratio"] < 20.0 and pie["iterations"] >= 16, pie
        assert d["ranks"]["world_size"] == 1 and d["ranks"]["backend"] == "nccl" and d["ranks"]["distinct_devices"] == 1
        assert d["config"]["rays_per_step"] == one["config"]["rays_per_step"], extra
        assert d["image_md5"] == one["image_md5"], extra


def test_two_gpus_rccl():
    """The same over RCCL, one rank per GPU -- needs two GPUs (skipped on the 1-GPU boxes of this pool)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    one = run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--batch", "4", "--no-cpu-baseline",
               "--no-roofline", "--digest"])
    for k, extra in enumerate((["--batch", "2"], ["--batch", "2", "--reduce-every", "1"], ["--batch", "2", "--collective", "reduce"])):
        two = _two(29633 + k, *extra, backend="nccl", same_device=False)
        assert two["image_md5"] == one["image_md5"], extra
