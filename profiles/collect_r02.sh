#!/bin/bash
# Round-2 evidence, one gpurun call:  bash profiles/collect_r02.sh
# -> gpurun_out/r02/: rocprofv3 summaries (C2 default line incl. the vector-instruction mix, C3 sorted, C4 with the
# hierarchy), the bench lines of every configuration and the issue-rate microbenchmark; copy what should be judged
# into profiles/r02/ and profiles/traffic.json.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r02
mkdir -p "$OUT"
cd "$ROOT"
bash profiles/run_rocprof.sh r02 > "$OUT/rocprof_c2.log" 2>&1
cp gpurun_out/prof_r02/summary.txt "$OUT/rocprof_r02_c2_summary.txt"
cp gpurun_out/prof_r02/traffic.json "$OUT/traffic_r02.json" 2>/dev/null
cp gpurun_out/prof_r02/traffic.json profiles/traffic.json 2>/dev/null
find gpurun_out/prof_r02/trace -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprof_r02_c2_kernel_stats.csv" \;
bash profiles/run_rocprof.sh r02c3 --config c3 --flags compact,sort > "$OUT/rocprof_c3sort.log" 2>&1
cp gpurun_out/prof_r02c3/summary.txt "$OUT/rocprof_r02_c3sort_summary.txt"
find gpurun_out/prof_r02c3/trace -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprof_r02_c3sort_kernel_stats.csv" \;
bash profiles/run_rocprof.sh r02c4 --config c4 --flags compact,bvh > "$OUT/rocprof_c4bvh.log" 2>&1
cp gpurun_out/prof_r02c4/summary.txt "$OUT/rocprof_r02_c4bvh_summary.txt"
find gpurun_out/prof_r02c4/trace -name '*kernel_stats.csv' -exec cp {} "$OUT/rocprof_r02_c4bvh_kernel_stats.csv" \;
b() { name=$1; shift; timeout 600 python bench.py "$@" 2>"$OUT/bench_$name.err" | tail -1 > "$OUT/bench_r02_$name.json"; cut -c1-220 "$OUT/bench_r02_$name.json"; }
b c2 --config c2
b c2_b1 --config c2 --batch 1 --steps 200 --warmup 20 --pcie --no-cpu-baseline
b c2_b4 --config c2 --batch 4 --steps 100 --warmup 10 --no-cpu-baseline
b c2_b16 --config c2 --batch 16 --steps 50 --warmup 5 --no-cpu-baseline
b c3 --config c3 --flags compact,sort --no-cpu-baseline
b c3_nosort --config c3 --flags compact --no-cpu-baseline
b c4_bvh --config c4 --flags compact,bvh --steps 10 --warmup 2 --no-cpu-baseline
b c4_loop --config c4 --flags compact --steps 2 --warmup 1 --batch 1 --no-cpu-baseline
b c2_aa --config c2 --flags compact,aa --no-cpu-baseline
b c5 --config c5 --batch 4 --steps 5 --warmup 1 --no-cpu-baseline
PTMI355_CULL0=0 b c2_nomask --config c2 --no-cpu-baseline
# the N > 1 code path over RCCL itself, world of one rank (every RCCL call of the multi-GPU launch executes here)
for m in "" "--reduce-every 1" "--collective reduce"; do
  MASTER_PORT=29741 python bench.py --force-dist --backend nccl --steps 10 --warmup 2 --no-roofline --no-cpu-baseline $m 2>/dev/null | tail -1 | cut -c1-900
done > "$OUT/bench_r02_one_rank_rccl.txt"
# two processes of the library on this one GPU (gloo; RCCL refuses two ranks per device): the N > 1 code path
for m in "--scaling weak" "--scaling strong" "--scaling strong --reduce-every 1"; do
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29711 bench.py --gpus 2 --backend gloo --same-device --steps 10 --warmup 2 --batch 8 --no-roofline $m 2>/dev/null | tail -1 | cut -c1-900
done > "$OUT/bench_r02_two_ranks_one_gpu.txt"
(cd profiles/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_peak valu_peak.hip && timeout 400 /tmp/valu_peak > "$OUT/valu_peak_r02.json")
# per-wave start / end times of k_bounce with and without the priority rotation (two diagnostic builds)
timeout 600 python profiles/wave_times.py > "$OUT/wave_times_r02.txt" 2>&1
# soak: random cube/sphere scenes under every pipeline + the cull-stress rays, then random meshes (hierarchy and
# every-triangle kernel) incl. grazing rays -- all against the CPU oracle, bit for bit
(timeout 900 python tests/tools/fuzz_gpu.py 1000 300; timeout 900 python tests/tools/fuzz_gpu.py mesh 1 120) 2>&1 | grep -v amdgpu.ids | tail -12 > "$OUT/fuzz.log"
