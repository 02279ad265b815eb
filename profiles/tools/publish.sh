#!/bin/bash
# publish.sh [ROUND]: copy a finished collection (gpurun_out/ROUND, made by PT_ROUND=ROUND python3 profiles/collect.py on the GPU box)
# into profiles/ROUND and rebuild profiles/traffic.json
R=${1:-r04}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
S=$ROOT/gpurun_out/$R; D=$ROOT/profiles/$R
mkdir -p $D/hist
cp $S/rocprof_${R}_*_summary.txt $S/roofline_${R}_*.json $S/bench_${R}_*.json $S/costs_${R}.json $D/ 2>/dev/null
for t in $S/*/; do tag=$(basename $t); for h in $t/*.hist.txt; do [ -f "$h" ] && cp $h $D/hist/${tag}_$(basename $h); done; done
PT_ROUND=$R python3 $ROOT/profiles/collect.py --assemble $D
python3 $ROOT/profiles/tools/roofline_table.py $D
