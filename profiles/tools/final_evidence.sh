#!/bin/bash
# final_evidence.sh ROUND PART  (on the GPU box, from the repository root): the evidence of a round, in parts that each fit
# one gpurun call (20 minutes).  Results under gpurun_out/ROUND; publish with profiles/tools/publish.sh ROUND in the build
# container.  Before: profiles/tools/build_all_counted.sh and build_variant.sh x WORK -DPT_EXPERIMENTS in the build container.
#   a   issue costs re-measured (every opcode of the device assembly), C2 and C2 at 1 spp per call: kernel traces, counters,
#       counted instruction histograms, bench lines; the kernel trace of the TIMED configuration (overlap on) beside
#       PTMI355_OVERLAP=0; HBM counters of the default command
#   b   C3, C3 sorted, C5
#   c   C4 (loop over every triangle, hierarchy), the lane-count runs, the experiment variants of the parity tests
#   d   pytest -m gpu, smoke, the driver's bench command
R=${1:-r05}; PART=${2:-a}; O=gpurun_out/$R; mkdir -p $O/hb
[ -f profiles/$R/costs_$R.json ] && cp profiles/$R/costs_$R.json $O/costs_$R.json
case $PART in
a)
  [ -f $O/costs_$R.json ] || export PT_REMEASURE_COSTS=1
  PT_ROUND=$R python3 profiles/collect.py c2 c2_1spp > $O/collect_a.log 2>&1
  grep -E "^==|^fn|^fg|^it|!!" $O/collect_a.log | cut -c1-230
  bash profiles/tools/overlap_trace.sh $R > /dev/null 2>&1; head -12 $O/rocprof_${R}_c2_overlap_summary.txt
  bash profiles/tools/pmc_bytes.sh work > /dev/null 2>&1; head -6 gpurun_out/pmc_work/summary.txt ;;
b)
  PT_ROUND=$R python3 profiles/collect.py c3 c3_sort c5 > $O/collect_b.log 2>&1
  grep -E "^==|^fn|^fg|^sn|^sg|!!" $O/collect_b.log | cut -c1-230 ;;
c)
  PT_ROUND=$R python3 profiles/collect.py c4_loop c4_bvh > $O/collect_c.log 2>&1
  grep -E "^==|^tn|^tg|^pn|^pg|^km|!!" $O/collect_c.log | cut -c1-230
  PTMI355_LIB=$PWD/.ab/lan_fn/libptmi355.so python profiles/tools/count_run.py c2 compact .ab/lan_fn/map.json $O/hb/c2_fn.lanes.u32 3 64 > $O/hb/c2_fn.lanes_run.json 2>/dev/null
  PTMI355_LIB=$PWD/.ab/lan_km/libptmi355.so python profiles/tools/count_run.py c4 compact,bvh .ab/lan_km/map.json $O/hb/c4_bvh_km.lanes.u32 2 64 > $O/hb/c4_bvh_km.lanes_run.json 2>/dev/null
  bash profiles/tools/experiment_tests.sh > $O/experiment_tests.log 2>&1; tail -2 $O/experiment_tests.log ;;
d)
  python -m pytest tests -q -m gpu > $O/t_final.log 2>&1; tail -2 $O/t_final.log
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
  python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 400 $O/bench_default.json ;;
esac
