#!/bin/bash
# final_evidence.sh [ROUND]  (on the GPU box, from the repository root): the whole evidence of a round in one call --
# pytest -m gpu, smoke, profiles/collect.py (kernel traces, counters, counted instruction histograms, bench lines per
# configuration), the lane-count runs, the HBM counters of the default command, the exchange rehearsal.
# Results under gpurun_out/ROUND; publish with profiles/tools/publish.sh ROUND in the build container.
R=${1:-r04}; O=gpurun_out/$R; mkdir -p $O/hb
python -m pytest tests -x -q -m gpu > $O/t_final.log 2>&1; tail -1 $O/t_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -f $O/costs_$R.json; cp profiles/$R/costs_$R.json $O/costs_$R.json
PT_ROUND=$R python3 profiles/collect.py > $O/collect.log 2>&1
grep -E "^==|^fn|^fg|^sn|^sg|^tn|^tg|^pn|^pg|^km|^it|!!" $O/collect.log | cut -c1-230
PTMI355_LIB=$PWD/.ab/lan_fn/libptmi355.so python profiles/tools/count_run.py c2 compact .ab/lan_fn/map.json $O/hb/c2_fn.lanes.u32 3 64 > $O/hb/c2_fn.lanes_run.json 2>/dev/null
PTMI355_LIB=$PWD/.ab/lan_km/libptmi355.so python profiles/tools/count_run.py c4 compact,bvh .ab/lan_km/map.json $O/hb/c4_bvh_km.lanes.u32 2 64 > $O/hb/c4_bvh_km.lanes_run.json 2>/dev/null
bash profiles/tools/pmc_bytes.sh work > /dev/null 2>&1; head -6 gpurun_out/pmc_work/summary.txt
bash profiles/tools/exchange_rehearsal.sh $O 2>&1 | tail -6
