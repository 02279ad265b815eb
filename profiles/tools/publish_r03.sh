#!/bin/bash
# copy a finished collection (gpurun_out/r03) into profiles/r03, rebuild profiles/traffic.json and the table in DESIGN.md
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
S=$ROOT/gpurun_out/r03; D=$ROOT/profiles/r03
mkdir -p $D/hist
cp $S/rocprof_r03_*_summary.txt $S/roofline_r03_*.json $S/bench_r03_*.json $S/issue_ops_r03.json $S/costs_r03.json $D/
for d in c2 c2_1spp c3 c3_sort c4_loop c4_bvh c5; do for f in $S/$d/*.hist.txt; do cp $f $D/hist/${d}_$(basename $f); done; done
python3 $ROOT/profiles/collect_r03.py --assemble $D
python3 $ROOT/profiles/tools/fill_design_table.py $D
for f in $D/bench_r03_*.json; do python3 -c "
import json,sys;d=json.load(open('$f'));r=d['roofline'];t=r.get('timed_pass',{});print('%-22s %9.1f Mrays/s %8.3f ms/step bound=%s frac=%s hbm=%s lanes=%s timed_issue=%s launch_us=%s'%('$f'.split('bench_r03_')[1],d['value'],d['ms_per_step'],r.get('bound'),r.get('frac'),(r.get('hbm_measured') or {}).get('frac'),(r.get('fp32') or {}).get('active_lane_fraction'),t.get('valu_issue_frac'),r.get('avg_launch_us')))"; done
