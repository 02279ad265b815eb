#!/bin/bash
# Collect the rocprofv3 evidence for one round.  Usage (on the GPU box, via gpurun):
#   bash profiles/run_rocprof.sh r01 [bench args...]
# Writes gpurun_out/prof_<tag>/ ; copy the *_stats.csv / pmc summaries into profiles/.
set -u
TAG=${1:-r01}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline $*"
# pass 1: kernel trace + stats (durations)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
# passes 2..: PMC counters, each in its own run (no trace domains combined with --pmc)
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_sq" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d "$OUT/pmc_sq2" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_sq2.log" 2>&1
# the vector-instruction mix (classes issue at different rates: profiles/microbench/valu_peak.hip)
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_mix1" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_mix1.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH --output-format csv -d "$OUT/pmc_mix2" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_mix2.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 "$ROOT/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
cd "$ROOT"
python3 profiles/summarize.py "$OUT" "$OUT/traffic.json" $* > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
