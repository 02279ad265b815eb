#!/bin/bash
# Round-1 evidence, one gpurun call:  bash profiles/collect_r01.sh
# -> gpurun_out/r01/: rocprofv3 summaries (C2 default line, C4 with the hierarchy) and the bench lines of
# every configuration; copy what should be judged into profiles/r01/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r01
mkdir -p "$OUT"
cd "$ROOT"
bash profiles/run_rocprof.sh r01 > "$OUT/rocprof_c2.log" 2>&1
cp gpurun_out/prof_r01/summary.txt "$OUT/rocprof_r01_c2_summary.txt"
cp gpurun_out/prof_r01/traffic.json "$OUT/traffic_r01.json" 2>/dev/null
cp gpurun_out/prof_r01/traffic.json profiles/traffic.json 2>/dev/null
find gpurun_out/prof_r01/trace -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprof_r01_c2_kernel_stats.csv" \;
bash profiles/run_rocprof.sh r01c4 --config c4 --flags compact,bvh > "$OUT/rocprof_c4bvh.log" 2>&1
cp gpurun_out/prof_r01c4/summary.txt "$OUT/rocprof_r01_c4bvh_summary.txt"
find gpurun_out/prof_r01c4/trace -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprof_r01_c4bvh_kernel_stats.csv" \;
b() { name=$1; shift; timeout 600 python bench.py "$@" 2>"$OUT/bench_$name.err" | tail -1 > "$OUT/bench_r01_$name.json"; cut -c1-220 "$OUT/bench_r01_$name.json"; }
b c2 --config c2
b c2_b1 --config c2 --batch 1 --pcie --no-cpu-baseline
b c3 --config c3 --flags compact,sort --no-cpu-baseline
b c3_nosort --config c3 --flags compact --no-cpu-baseline
b c4_bvh --config c4 --flags compact,bvh --steps 10 --warmup 2 --no-cpu-baseline
b c4_loop --config c4 --flags compact --steps 2 --warmup 1 --batch 1 --no-cpu-baseline
b c2_aa --config c2 --flags compact,aa --no-cpu-baseline
b c5 --config c5 --batch 4 --steps 5 --warmup 1 --no-cpu-baseline
