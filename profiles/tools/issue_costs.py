#!/usr/bin/env python3
"""issue_ops.json (profiles/microbench/issue_ops: one row per opcode and waves-per-SIMD) -> costs.json for isa_count.py
hist: cycles per wave-instruction per SIMD at the best of the measured occupancies >= 2 waves per SIMD."""
import json, sys
d = json.load(open(sys.argv[1]))
best = {}
for r in d["rows"]:
    c = r["cycles_per_wave_inst_per_simd"]
    if r["waves_per_simd"] >= 2 and (r["op"] not in best or c < best[r["op"]]):
        best[r["op"]] = c
pair = best.pop("pair:v_cmp_lt_f32_e32+v_cndmask_b32_e32", None)
if pair is not None:
    best["v_cndmask_b32_e32"] = 2.0 * pair - best["v_cmp_lt_f32_e32"]
out = {}
for op, c in best.items():
    out[op] = round(c, 3)
    if op.endswith(("_e32", "_e64")):
        out.setdefault(op[:-4], round(c, 3))          # the form the assembler prints without a suffix
# opcodes the generator has no template for are priced like the slowest plain class (and reported as unpriced)
json.dump({"cycles": out, "default": 4.4, "source": sys.argv[1], "clock_ghz": max(r["in_kernel_clock_ghz"] for r in d["rows"])}, open(sys.argv[2], "w"), indent=0)
print("%d opcodes priced" % len(out))
