#!/usr/bin/env python3
"""Executed-instruction histogram of ONE kernel, exact: per-basic-block execution counts from an instrumented build.

  isa_count.py instrument full.s KERNEL_SUBSTR out.s map.json
        full.s = device assembly of csrc/ptmi355.hip (hipcc --cuda-device-only -S).  Rewrites the one kernel whose
        mangled name contains KERNEL_SUBSTR: every basic block gets a counter (one lane of a VGPR above the kernel's
        own registers; exec is switched to that lane for one v_add_u32 and restored: no SCC / VCC / live register is
        touched, s[100:101] are free in every kernel of this library), and every s_endpgm first adds the wave's
        counters to BounceArgs::dbg_counts (kernarg offset 0).  map.json = the blocks and their instructions.
        (mode "lanes": the counters accumulate the blocks' ACTIVE LANES, popcount(exec), instead of their executions;
        profiles/tools/hot_blocks.py divides the two)
  isa_count.py hist map.json counts.u32 costs.json [launches]
        counts.u32 = what ptdbg_counts() returned after the workload.  Prints the executed opcode histogram and the
        issue cycles per launch (opcode -> cycles from the microbenchmark table costs.json), JSON on the last line.

A block = a maximal run of instructions entered only at its top (a label, or the instruction after a branch).
"""
import json
import re
import sys

LANES = 62                      # counters per VGPR: lanes 0..30 and 32..62 (lane 31 of the first two registers parks the kernarg pointer)


def slot_lane(i):
    return i if i < 31 else i + 1


def exec_to_lane(lane):
    """exec := 1 << lane without touching SCC / VCC: a positive 32-bit literal for the low half, s_bitset1 for the high"""
    if lane < 31:
        return ['\ts_mov_b64 exec, 0x%x' % (1 << lane)]
    return ['\ts_mov_b64 exec, 0', '\ts_bitset1_b32 exec_hi, %d' % (lane - 32)]
INSTR = re.compile(r'^\t([a-z][a-z0-9_]*)\b(.*)$')
LABEL = re.compile(r'^(\.LBB\d+_\d+):')
BRANCH = ('s_branch', 's_cbranch', 's_setpc', 's_endpgm', 's_swappc', 's_call')


def is_branch(op):
    return op.startswith(BRANCH)


def kernel_span(lines, substr):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\w+):', l)
        if m and substr in m.group(1) and start is None:
            start, name = i, m.group(1)
        if start is not None and l.startswith('.Lfunc_end') and i > start:
            return start, i, name
    raise SystemExit("kernel containing %r not found" % substr)


def partition(body):
    """[(first_line_index_in_body, [(index, op, rest)])]: the blocks"""
    blocks, cur, fresh = [], None, True
    for i, l in enumerate(body):
        if LABEL.match(l):
            fresh = True
            continue
        m = INSTR.match(l)
        if not m or m.group(1).startswith('.'):
            continue
        op = m.group(1)
        if fresh:
            cur = [i, []]
            blocks.append(cur)
            fresh = False
        cur[1].append((i, op, m.group(2).split(';')[0].strip()))
        if is_branch(op):
            fresh = True
    return blocks


def instrument(path, substr, out_path, map_path, mode="full"):
    lines = open(path).read().split('\n')
    a, b, name = kernel_span(lines, substr)
    body = lines[a + 1:b]
    blocks = partition(body)
    # the kernel's descriptor
    d0 = next(i for i, l in enumerate(lines) if l.strip() == '.amdhsa_kernel ' + name)
    d1 = next(i for i in range(d0, len(lines)) if lines[i].strip() == '.end_amdhsa_kernel')
    desc = {}
    for i in range(d0, d1):
        m = re.match(r'\s*\.amdhsa_(\w+)\s+(\S+)', lines[i])
        if m:
            desc[m.group(1)] = (i, m.group(2))
    nv = int(desc['next_free_vgpr'][1])
    ns = int(desc['next_free_sgpr'][1])
    if ns > 100:
        raise SystemExit("kernel uses s100/s101 (next_free_sgpr %d)" % ns)
    if desc['user_sgpr_kernarg_segment_ptr'][1] != '1' or int(desc['user_sgpr_count'][1]) < 2 or \
            any(desc[k][1] != '0' for k in ('user_sgpr_dispatch_ptr', 'user_sgpr_queue_ptr') if k in desc):
        raise SystemExit("kernarg pointer is not in s[0:1]")
    base = (nv + 7) & ~7
    nreg = max(2, -(-len(blocks) // LANES))
    regs = [base + j for j in range(nreg)]
    extra = 1 if mode == "lanes" else 0
    edits = {desc['next_free_vgpr'][0]: '\t\t.amdhsa_next_free_vgpr %d' % (base + nreg + extra),
             desc['next_free_sgpr'][0]: '\t\t.amdhsa_next_free_sgpr 102',
             desc['accum_offset'][0]: '\t\t.amdhsa_accum_offset %d' % ((base + nreg + extra + 3) & ~3)}
    ins = {}                      # body line index -> lines to insert before it
    for bid, (first, instrs) in enumerate(blocks):
        r, lane = regs[bid // LANES], slot_lane(bid % LANES)
        seq = ['\ts_mov_b64 s[100:101], exec'] + exec_to_lane(lane) + ['\tv_add_u32_e32 v%d, 1, v%d' % (r, r), '\ts_mov_b64 exec, s[100:101]']
        if mode == "lanes":
            # the block's ACTIVE LANES instead of its executions: popcount(exec) added to the same counter (v_bcnt reads the
            # saved mask as data; one temporary VGPR above the counters; still no SCC / VCC / live register touched)
            tmp = base + nreg
            seq = ['\ts_mov_b64 s[100:101], exec'] + exec_to_lane(lane) + [
                '\tv_bcnt_u32_b32 v%d, s100, 0' % tmp, '\tv_bcnt_u32_b32 v%d, s101, v%d' % (tmp, tmp),
                '\tv_add_u32_e32 v%d, v%d, v%d' % (r, tmp, r), '\ts_mov_b64 exec, s[100:101]']
        if mode in ("full", "inc", "lanes"):
            ins.setdefault(first, []).extend(seq)
    # entry: clear the counters, park the kernarg pointer in lane 31 of the first two
    entry = ['\tv_mov_b32_e32 v%d, 0' % r for r in regs]
    entry += ['\tv_writelane_b32 v%d, s0, 31' % regs[0], '\tv_writelane_b32 v%d, s1, 31' % regs[1]]
    first0 = blocks[0][0]
    ins[first0] = entry + ins.get(first0, [])
    # exits: add this wave's counters to dbg_counts[reg * 64 + lane]
    nexit = 0
    for first, instrs in blocks:
        for (i, op, rest) in instrs:
            if op != 's_endpgm' or mode == "inc":
                continue
            nexit += 1
            lab = '.Lisa_count_skip_%d' % nexit
            seq = ['\ts_mov_b64 exec, -1', '\tv_readlane_b32 s100, v%d, 31' % regs[0], '\tv_readlane_b32 s101, v%d, 31' % regs[1],
                   '\ts_nop 4', '\ts_load_dwordx2 s[100:101], s[100:101], 0x0', '\ts_waitcnt lgkmcnt(0)',
                   '\ts_cmp_eq_u64 s[100:101], 0', '\ts_cbranch_scc1 ' + lab,
                   '\tv_mbcnt_lo_u32_b32 v0, -1, 0', '\tv_mbcnt_hi_u32_b32 v0, -1, v0', '\tv_lshlrev_b32_e32 v0, 2, v0']
            for j, r in enumerate(regs):
                if j:
                    seq.append('\tv_add_u32_e32 v0, 0x100, v0')
                seq.append('\tglobal_atomic_add v0, v%d, s[100:101]' % r)
            seq += ['\ts_waitcnt vmcnt(0)', lab + ':']
            ins.setdefault(i, []).extend(seq)
    out_body = []
    for i, l in enumerate(body):
        if i in ins:
            out_body.extend(ins[i])
        out_body.append(edits.pop(a + 1 + i, l))       # (the descriptor sits between the code and .Lfunc_end)
    lines[a + 1:b] = out_body
    for i, l in edits.items():
        lines[i + len(out_body) - len(body) if i >= b else i] = l
    open(out_path, 'w').write('\n'.join(lines))
    m = {"kernel": name, "regs": nreg, "lanes": LANES, "words": nreg * 64,
         "blocks": [[op for (_, op, _) in instrs] for (_, instrs) in blocks]}
    json.dump(m, open(map_path, 'w'))
    print("%s: %d blocks, counters in v%d..v%d, %d exits, %d words" % (name, len(blocks), regs[0], regs[-1], nexit, nreg * 64))


def op_class(op):
    if op.startswith('v_'):
        return 'valu'
    if op.startswith('s_'):
        if op.startswith(('s_load', 's_buffer_load', 's_store', 's_dcache', 's_memtime', 's_memrealtime', 's_atc')):
            return 'smem'
        if is_branch(op):
            return 'branch'
        if op in ('s_waitcnt', 's_nop', 's_barrier', 's_setprio', 's_sleep', 's_sethalt', 's_setkill', 's_trap', 's_icache_inv',
                  's_incperflevel', 's_decperflevel', 's_ttracedata', 's_sendmsg', 's_sendmsghalt'):
            return 'sopp'
        return 'salu'
    if op.startswith('ds_'):
        return 'lds'
    return 'vmem'


def hist(map_path, counts_path, cost_path, launches):
    import numpy as np
    m = json.load(open(map_path))
    c = np.fromfile(counts_path, dtype=np.uint32)
    costs = json.load(open(cost_path))
    per_op, default = costs["cycles"], costs["default"]
    execd = []
    for bid in range(len(m["blocks"])):
        execd.append(int(c[(bid // m["lanes"]) * 64 + slot_lane(bid % m["lanes"])]))
    ops = {}
    for bid, blk in enumerate(m["blocks"]):
        for op in blk:
            ops[op] = ops.get(op, 0) + execd[bid]
    cls = {}
    for op, n in ops.items():
        cls[op_class(op)] = cls.get(op_class(op), 0) + n
    valu = {op: n for op, n in ops.items() if op_class(op) == 'valu'}
    total_valu = sum(valu.values())
    cyc, unpriced, cyc_guide = 0.0, 0.0, 0.0
    rows = []
    for op, n in sorted(valu.items(), key=lambda kv: -kv[1]):
        key = op if op in per_op else re.sub(r'_(e32|e64|dpp|sdwa)$', '', op)
        if key in per_op:
            cost, known = per_op[key], True
        else:
            cost, known = default, False
        cyc += n * cost
        # the same opcode at the GUIDE's rate class (MI355X_MICROARCH.md: full rate 2 cycles per wave-instruction per SIMD,
        # half rate 4, transcendental 8): the class the measured cost is nearest to -- a roof that owes nothing to this
        # repository's own microbenchmark (its rows include ramp and tail: 2.3-2.5 for a full-rate opcode)
        cyc_guide += n * min((2.0, 4.0, 8.0), key=lambda g: abs(g - cost))
        if not known:
            unpriced += n * cost
        rows.append((op, n, cost, known))
    print("kernel %s: %d launches" % (m["kernel"], launches))
    print("wave-instructions per launch by class: " + ", ".join("%s %.0f" % (k, v / launches) for k, v in sorted(cls.items())))
    print("%-28s %14s %7s %8s" % ("VALU opcode", "per launch", "share", "cycles"))
    for op, n, cost, known in rows:
        if n / max(1, total_valu) >= 0.002:
            print("%-28s %14.0f %6.1f%% %7.2f%s" % (op, n / launches, 100.0 * n / total_valu, cost, "" if known else "  (class default)"))
    out = {"kernel": m["kernel"], "launches": launches, "wave_insts_per_launch": {k: v / launches for k, v in cls.items()},
           "valu_per_launch": total_valu / launches, "issue_cycles_per_launch": cyc / launches,
           "issue_cycles_guide_rates_per_launch": cyc_guide / launches,
           "unpriced_share_of_cycles": unpriced / max(1.0, cyc),
           "valu_opcodes_per_launch": {op: n / launches for op, n in valu.items()},
           # matrix pipe: a v_mfma_*_16x16x32_* is 16 x 16 x 32 multiply-adds per wave-instruction (the every-triangle loop's stage 1)
           "mfma_flops_per_launch": 2.0 * 16 * 16 * 32 * sum(n for op, n in valu.items() if re.match(r'v_mfma_\w*16x16x32', op)) / launches,
           "flops_fp32_per_launch": 64.0 * sum(n * (2 if re.match(r'v_(fma|fmac|mad|mac|pk_fma)_f32', op) else 1)
                                               for op, n in valu.items()
                                               if re.match(r'v_(add|sub|subrev|mul|fma|fmac|mad|mac|min|max|min3|max3|med3|rcp|rsq|sqrt|exp|log|sin|cos|fract|floor|ceil|trunc|rndne|ldexp|div_fixup|div_fmas|div_scale|cmp\w*|pk_fma|pk_mul|pk_add)_f32', op)) / launches}
    print(json.dumps(out))


if __name__ == "__main__":
    if sys.argv[1] == "instrument":
        instrument(*sys.argv[2:7])
    else:
        hist(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 1)
