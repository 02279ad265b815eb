"""Known-answer pins of the DEVICE's own arithmetic, one function at a time, through the probe entry points of the
C-ABI (include/ptmi355.h: pt_probe_rng / pt_probe_sincos / pt_probe_hemisphere): the pieces of the hot path that the
reference merely calls in third-party code -- thrust's minstd_rand + uniform_real_distribution (pathtrace.cu:41-45,
interactions.h:12-13), the sin / cos binding of interactions.h:40-41 -- against published constants, the golden
vectors generated from the reference's headers, and the oracle.  Bit-exact."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import __graft_entry__ as ge

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pt():
    p = ge.load_package()
    p.library()
    return p


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_minstd_rand_published_known_answer(pt):
    """C++ [rand.predef]: the 10 000th consecutive invocation of a default-constructed minstd_rand (seed 1) produces
    399268537 -- a pin no toolchain in this repository produced.  thrust::default_random_engine is minstd_rand."""
    st, u = pt.probe_rng([1, 1, 0, 2147483647, 2147483648, 0xffffffff], 10000)
    assert (st == 399268537).all()                    # seeds 0, m, 2^31 and 2^32 - 1 all reduce to 1 (lcg_seed)
    assert (bits(u) == bits(np.float32(np.float32(399268536) / np.float32(2147483648.0)))).all()
    # the tenth, hundredth, ... draws by exponentiation: 48271^k mod (2^31 - 1)
    for k in (1, 2, 10, 100, 1000, 99999):
        st, _ = pt.probe_rng([1], k)
        assert int(st[0]) == pow(48271, k, 2147483647), k


def test_device_rng_equals_the_golden_vectors(pt, po, golden):
    """rng.npz holds thrust's own outputs (rocThrust through the reference's headers, tests/golden/make_golden.py):
    engine states and u01 values of 256 seeds x 8 draws, and the first draw of 512 (iter, pixel, depth) keys."""
    z = golden["rng"]
    seeds = z["seeds"].astype(np.uint32)
    for k in range(8):
        st, u = pt.probe_rng(seeds, k + 1)
        assert (st == z["raw"][:, k]).all(), k
        assert (bits(u) == bits(z["u01"][:, k])).all(), k
    # u01 == 1.0f exactly is reachable: a state whose successor is m - 1
    st, u = pt.probe_rng([pow(48271, -1, 2147483647) * 2147483646 % 2147483647], 1)
    assert int(st[0]) == 2147483646 and float(u[0]) == 1.0
    # makeSeededRandomEngine: utilhash twice, xor, seed -- same keys as the oracle test
    keys = z["key"]
    engines = np.array([po.lib().pto_make_seeded_engine(int(a), int(b), int(c)) for a, b, c in keys], dtype=np.uint32)
    st, _ = pt.probe_rng(engines, 1)
    assert (st == z["first_raw"]).all()


def test_device_hemisphere_equals_the_reference_headers(pt, po, golden):
    """calculateRandomDirectionInHemisphere on the device against interactions.h itself (hemisphere.npz, the shared
    trig binding) and against the oracle on further random normals, including the helper-axis switch points."""
    z = golden["hemisphere"]
    got = pt.probe_hemisphere(z["normals"], z["seeds"].astype(np.uint32))
    assert (bits(got) == bits(z["shared"])).all()
    rng = np.random.default_rng(11)
    n = rng.normal(size=(4096, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    s3 = np.float32(0.5773502691896257645091487805019574556476)
    n[:64, 0] = np.nextafter(s3, np.float32([0, 1] * 32))                  # |n.x| just below / above 1/sqrt(3)
    n[64:128, 1] = np.nextafter(s3, np.float32([0, 1] * 32)); n[64:128, 0] = 0.9
    n = n.astype(np.float32)
    seeds = rng.integers(1, 2147483646, len(n)).astype(np.uint32)
    got = pt.probe_hemisphere(n, seeds)
    want = po.hemisphere(n, seeds, po.TRIG_SHARED)
    assert (bits(got) == bits(want)).all()


def test_shared_sincos_every_argument(pt, po):
    """The shared sin / cos on EVERY binary32 argument calculateRandomDirectionInHemisphere can
    form -- around = u01 * TWO_PI lies in [0, fl(2 pi)]: 1 086 918 620 floats -- device against oracle through the two
    position-weighted checksums of pt_probe_sincos / pto_sincos_sums, chunk by chunk (a differing chunk is narrowed down
    to its first differing argument)."""
    hi = int(np.float32(6.2831853071795864769).view(np.uint32)) + 1        # one past fl(2 pi)
    chunk = 1 << 24
    starts = list(range(0, hi, chunk))
    workers = max(1, min(16, len(os.sched_getaffinity(0))))

    def cpu(first):
        return po.sincos_sums(first, min(chunk, hi - first))
    with ThreadPoolExecutor(workers) as ex:                                # ctypes releases the GIL
        want = list(ex.map(cpu, starts))
    for first, w in zip(starts, want):
        n = min(chunk, hi - first)
        got = pt.probe_sincos_sums(first, n)
        if got != w:
            lo, cnt = first, n
            while cnt > 1:                                                 # bisect to the first differing argument
                half = cnt // 2
                if pt.probe_sincos_sums(lo, half) != po.sincos_sums(lo, half):
                    cnt = half
                else:
                    lo, cnt = lo + half, cnt - half
            x = np.uint32(lo).view(np.float32)
            raise AssertionError("sincos(%r) [bits 0x%08x]: device %r, oracle %r" % (float(x), lo, pt.probe_sincos([x]), po.sincos(float(x))))
    # values, not only sums, on a sample -- and arguments outside the range the tracer uses
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 4097), [0.0, -0.0, 1e-30, 10.0, 100.0, 1000.0, -3.0]]).astype(np.float32)
    s, c = pt.probe_sincos(xs)
    for x, sv, cv in zip(xs, s, c):
        ws, wc = po.sincos(float(x))
        assert (bits(np.float32(sv)), bits(np.float32(cv))) == (bits(np.float32(ws)), bits(np.float32(wc))), float(x)


def test_newton_sqrt_every_argument(pt):
    """glm::length and glm::normalize run sqrt and 1 / sqrt through Newton's iteration on v_rsq_f32 (csrc/pt_device.hpp:
    sqrt_newton, round 5) instead of v_sqrt_f32 + residual tests and v_rcp_f32 + refinement.  The replacement is only
    admissible if it is EXACTLY the correctly rounded sqrtf / divide: checked here on every binary32 x in [2^-102, 2^128)
    -- 1.93 * 10^9 arguments, binade by binade -- and the callers' gates (x >= 2^-96 for the root, 2^-80 <= x <= 2^80
    for the normalisation) are inside that range.  Outside it the forms DO differ (subnormal intermediates): counted
    too, so that the test would notice if the sweep were not comparing anything.  Inside [1 - 2^-12, 1 + 2^-12] the
    probe's reciprocal root is normalize_unit's four-addition form (rsqrt_near_one, for vectors that are unit vectors
    up to rounding; tests/test_arith_models_cpu.py has its CPU model): the sweep covers those 6 145 arguments with it."""
    lo, hi = 127 - 102, 254                     # biased exponents of 2^-102 and 2^127
    bad = [0, 0]
    for e in range(lo, hi + 1, 8):
        n = (min(hi + 1, e + 8) - e) << 23
        a, b = pt.probe_sqrt(e << 23, n)
        bad[0] += a; bad[1] += b
        assert (a, b) == (0, 0), "binades 2^%d..: sqrt mismatches %d, 1 / sqrt mismatches %d" % (e - 127, a, b)
    below = pt.probe_sqrt(1 << 23, 16 << 23)   # 2^-126 .. 2^-110: outside the gates
    assert below[0] > 0 and below[1] > 0
    assert pt.probe_sqrt(0x3f800000, 0) == (0, 0)
    assert pt.probe_sqrt(0x3f800000 - 4096, 4096 + 2048 + 1) == (0, 0)      # the near-one gate by itself


def test_shader_clock_probe(pt, scenes):
    """pt_probe_clock (bench.py's roofline.sustained): one wave counts its cycle counter against the 100-MHz counter -- a plausible
    shader clock on an idle device, and beside a session's launches (the wave is compiled for sixteen scalar registers so that it
    fits beside the persistent grid: it must come back while batches are still enqueued)."""
    idle = pt.probe_clock(200)
    assert 0.1 < idle < 2.6, idle
    s = scenes["cornell"]
    scene = pt.Scene(s["geoms"], s["materials"], s["camera"], s["depth"])
    pt.pathtraceInit(scene, flags=pt.PT_COMPACT, max_batch=16)
    try:
        for k in range(12):
            pt.trace_batch_async(1 + 16 * k, 16)
        busy = [pt.probe_clock(200) for _ in range(3)]
        pt.synchronize()
    finally:
        pt.pathtraceFree()
    assert all(0.5 < b < 2.6 for b in busy), busy
    with pytest.raises(pt.PtError):
        pt.probe_clock(0)

