"""profiles/tools/isa_count.py (the basic-block instrumentation behind the exact roofline) on a hand-written
kernel: blocks are found where control can enter, every block gets exactly one counter increment that touches neither
SCC / VCC nor a register of the kernel, the descriptor grows by the counter registers, the dump precedes s_endpgm, and
the histogram arithmetic (counts x opcode costs) comes out as computed by hand.  No GPU, no compiler."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "profiles", "tools", "isa_count.py")

KERNEL = """\t.text
_Z6k_toyP4Args:
\ts_load_dwordx2 s[2:3], s[0:1], 0x8
\tv_mov_b32_e32 v1, 0
\ts_waitcnt lgkmcnt(0)
\ts_cmp_eq_u32 s2, 0
\ts_cbranch_scc1 .LBB0_3
.LBB0_1:
\tv_add_f32_e32 v1, v1, v0
\tv_fma_f32 v1, v1, v0, v1
\ts_add_i32 s2, s2, -1
\ts_cmp_lg_u32 s2, 0
\ts_cbranch_scc1 .LBB0_1
\tv_mul_f32_e32 v1, v1, v1
.LBB0_3:
\tglobal_store_dword v2, v1, s[4:5]
\ts_endpgm
\t.section\t.rodata,"a",@progbits
\t.amdhsa_kernel _Z6k_toyP4Args
\t\t.amdhsa_user_sgpr_count 2
\t\t.amdhsa_user_sgpr_kernarg_segment_ptr 1
\t\t.amdhsa_next_free_vgpr 3
\t\t.amdhsa_next_free_sgpr 6
\t\t.amdhsa_accum_offset 4
\t.end_amdhsa_kernel
\t.text
.Lfunc_end0:
\t.size\t_Z6k_toyP4Args, .Lfunc_end0-_Z6k_toyP4Args
"""


def run(*args):
    return subprocess.run([sys.executable, TOOL] + list(args), capture_output=True, text=True, check=True).stdout


def test_instrument_and_hist(tmp_path):
    src, out, mp = tmp_path / "k.s", tmp_path / "k_counted.s", tmp_path / "map.json"
    src.write_text(KERNEL)
    msg = run("instrument", str(src), "k_toy", str(out), str(mp))
    m = json.loads(mp.read_text())
    # four blocks: entry .. branch | loop body .. back edge | fall-through multiply | store + end
    assert [b[0] for b in m["blocks"]] == ["s_load_dwordx2", "v_add_f32_e32", "v_mul_f32_e32", "global_store_dword"]
    assert m["blocks"][1][-1] == "s_cbranch_scc1" and m["blocks"][3][-1] == "s_endpgm"
    assert "4 blocks" in msg and "1 exits" in msg
    text = out.read_text()
    lines = text.split("\n")
    # counters live above the kernel's own registers (3 VGPRs -> v8, v9: two registers hold the kernarg pointer's halves)
    assert "\t\t.amdhsa_next_free_vgpr 10" in text and "\t\t.amdhsa_next_free_sgpr 102" in text
    incs = [i for i, l in enumerate(lines) if l.startswith("\tv_add_u32_e32 v8, 1, v8")]
    assert len(incs) == 4                                            # one increment per block
    for i in incs:                                                   # exec saved, one lane selected, exec restored
        assert lines[i - 2] == "\ts_mov_b64 s[100:101], exec" and lines[i - 1].startswith("\ts_mov_b64 exec, 0x")
        assert lines[i + 1] == "\ts_mov_b64 exec, s[100:101]"
    lanes = sorted(int(lines[i - 1].split("0x")[1], 16) for i in incs)
    assert lanes == [1, 2, 4, 8]
    # nothing the instrumentation adds writes SCC, VCC or a register below v8 / s100 (except v0, used by the dump AFTER the last use)
    added = [l for l in lines if l not in KERNEL.split("\n")]
    body_added = [l for l in added if l.startswith("\t") and not l.startswith("\t\t")]
    dump_at = next(i for i, l in enumerate(lines) if l.startswith("\tv_readlane_b32 s100"))
    assert dump_at > max(incs) and lines.index("\ts_endpgm") > dump_at
    for l in body_added:
        op = l.split()[0]
        assert not op.startswith(("s_cmp", "s_add", "s_and", "s_or", "v_cmp")) or "s[100:101]" in l or op == "s_cmp_eq_u64", l
    assert any(l.startswith("\tglobal_atomic_add v0, v8, s[100:101]") for l in lines)
    # "lanes" mode: popcount(exec) instead of 1, one temporary register more
    out2, mp2 = tmp_path / "k_lanes.s", tmp_path / "map2.json"
    run("instrument", str(src), "k_toy", str(out2), str(mp2), "lanes")
    t2 = out2.read_text()
    assert t2.count("\tv_bcnt_u32_b32 v10, s100, 0") == 4 and t2.count("\tv_bcnt_u32_b32 v10, s101, v10") == 4
    assert "\t\t.amdhsa_next_free_vgpr 11" in t2
    # histogram: the loop body ran 5 times per launch in 2 launches, the others once per launch
    counts = np.zeros(m["words"], dtype=np.uint32)
    for bid, n in enumerate((2, 10, 2, 2)):
        counts[bid] = n
    cf = tmp_path / "c.u32"
    counts.tofile(cf)
    costs = tmp_path / "costs.json"
    costs.write_text(json.dumps({"cycles": {"v_add_f32": 2.25, "v_fma_f32": 2.5, "v_mov_b32": 2.25}, "default": 4.0}))
    h = json.loads(run("hist", str(mp), str(cf), str(costs), "2").strip().split("\n")[-1])
    assert h["valu_per_launch"] == (2 * 1 + 10 * 2 + 2 * 1) / 2
    want = (2 * 2.25 + 10 * (2.25 + 2.5) + 2 * 4.0) / 2                # v_mul_f32 has no row here: the default, and it is reported
    assert abs(h["issue_cycles_per_launch"] - want) < 1e-9
    assert abs(h["unpriced_share_of_cycles"] - (2 * 4.0 / 2) / want) < 1e-9
    assert h["flops_fp32_per_launch"] == 64.0 * (10 * 1 + 10 * 2 + 2 * 1) / 2
    assert h["wave_insts_per_launch"]["branch"] == (2 + 10 + 2) / 2     # the two conditional branches and s_endpgm


def test_trace_overlap_union_and_sum(tmp_path):
    """profiles/tools/trace_overlap.py on a synthetic kernel trace: two steps of two k_bounce launches each that overlap
    pairwise (a launch of the next step starts while the last one of this step still runs), one k_gather per step, an
    unrelated kernel before the span -- the span starts at the first k_bounce of the last two steps, the union counts
    overlapped time once, the sum counts it twice."""
    rows = [("void at::native::fill(float)", 0, 50),
            ("void (anonymous namespace)::k_bounce<0, true, 0, true, true, false>((anonymous namespace)::BounceArgs)", 1000, 1400),
            ("void (anonymous namespace)::k_bounce<0, true, 0, true, false, false>((anonymous namespace)::BounceArgs)", 1400, 2000),
            ("(anonymous namespace)::k_gather(float*)", 2000, 2100),
            ("void (anonymous namespace)::k_bounce<0, true, 0, true, true, false>((anonymous namespace)::BounceArgs)", 1800, 2300),   # step 2 starts early
            ("void (anonymous namespace)::k_bounce<0, true, 0, true, false, false>((anonymous namespace)::BounceArgs)", 2300, 2900),
            ("(anonymous namespace)::k_gather(float*)", 2900, 3000)]
    p = tmp_path / "t_kernel_trace.csv"
    with open(p, "w") as f:
        f.write('"Kind","Kernel_Name","Start_Timestamp","End_Timestamp"\n')
        for name, s, e in rows:
            f.write('"KERNEL_DISPATCH","%s",%d,%d\n' % (name, s, e))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "trace_overlap.py"), str(p), "2", "2"],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = out.stdout
    assert "span 2.0 us = 0.0010 ms per step" in text                    # 1000 .. 3000 ns
    assert "some kernel of the session running: 2.0 us" in text          # no gap inside the span
    assert "sum of the kernels' own durations:   2.3 us" in text and "1.15 x the span" in text      # 400 + 600 + 100 + 500 + 600 + 100 ns
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "tools", "trace_overlap.py"), str(p), "3", "2"],
                         capture_output=True, text=True)
    assert bad.returncode != 0 and "only 4 k_bounce launches" in (bad.stderr + bad.stdout)


def test_line_cycles_by_source_line(tmp_path):
    """profiles/tools/line_cycles.py: the toy kernel again, its assembly once without and once with .file / .loc
    directives (what -gline-tables-only adds).  Executed instructions are priced like the histogram and land on the
    source line of the nearest .loc above them; an assembly whose blocks differ from map.json's is refused."""
    src, out, mp = tmp_path / "k.s", tmp_path / "k_counted.s", tmp_path / "map.json"
    src.write_text(KERNEL)
    run("instrument", str(src), "k_toy", str(out), str(mp))
    m = json.loads(mp.read_text())
    g = KERNEL.replace("\t.text\n_Z6k_toyP4Args:", '\t.text\n\t.file\t1 "/src" "toy.hip"\n_Z6k_toyP4Args:\n\t.loc\t1 10 0', 1)
    g = g.replace("\tv_add_f32_e32 v1, v1, v0\n", "\t.loc\t1 20 3\n\tv_add_f32_e32 v1, v1, v0\n")
    g = g.replace("\ts_add_i32 s2, s2, -1\n", "\t.loc\t1 21 3\n\ts_add_i32 s2, s2, -1\n")
    g = g.replace("\tglobal_store_dword", "\t.loc\t1 30 1\n\tglobal_store_dword")
    gs = tmp_path / "k_g.s"
    gs.write_text(g)
    counts = np.zeros(m["words"], dtype=np.uint32)
    for bid, n in enumerate((2, 10, 2, 2)):
        counts[bid] = n
    cf = tmp_path / "c.u32"
    counts.tofile(cf)
    costs = tmp_path / "costs.json"
    costs.write_text(json.dumps({"cycles": {"v_add_f32": 2.25, "v_fma_f32": 2.5, "v_mov_b32": 2.25}, "default": 4.0}))
    tool = os.path.join(ROOT, "profiles", "tools", "line_cycles.py")
    r = subprocess.run([sys.executable, tool, str(gs), "k_toy", str(mp), str(cf), str(costs), "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rows = {l.split()[0]: l.split() for l in r.stdout.split("\n") if l.startswith("toy.hip:")}
    # line 20: the loop's add + fma, 10 executions in 2 launches; line 21: the v_mul after the back edge inherits the last
    # .loc (2 executions, default price) and the loop's three scalar instructions (10 x 3 x 4.42)
    text = r.stdout
    assert "vector %.4g" % ((2 * 2.25 + 10 * 4.75 + 2 * 4.0) / 2) in text
    assert "scalar %.4g" % ((2 * (3 * 4.42 + 1.2) + 10 * 3 * 4.42 + 2 * 4.42) / 2) in text
    assert set(rows) == {"toy.hip:10", "toy.hip:20", "toy.hip:21", "toy.hip:30"}
    assert rows["toy.hip:20"][-2:] == ["v_add_f32_e32:0.0", "v_fma_f32:0.0"]          # (per launch, in millions)
    # another build: one instruction more in the loop
    gs.write_text(g.replace("\tv_fma_f32 v1, v1, v0, v1\n", "\tv_fma_f32 v1, v1, v0, v1\n\tv_mov_b32_e32 v2, v1\n"))
    bad = subprocess.run([sys.executable, tool, str(gs), "k_toy", str(mp), str(cf), str(costs), "2"], capture_output=True, text=True)
    assert bad.returncode != 0 and "not the same build" in (bad.stderr + bad.stdout)
