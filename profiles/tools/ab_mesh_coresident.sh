#!/bin/bash
# ab_mesh_coresident.sh: can k_mesh and the other batch's k_bounce<MESH_PRE> share a CU?  k_mesh as ONE workgroup of 8 / 12 / 16
# waves per CU (.ab/mb512e, mb768e, x: builds with -DPT_EXPERIMENTS [-DPT_MESH_BLOCK=...]) against k_bounce grids capped at
# PTMI355_WGS_PER_CU workgroups per CU, C4 + hierarchy, consecutive batches overlapped on two launch streams
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for round in 1 2; do
for v in ${VARIANTS:-x mb512e mb768e}; do
  for w in ${WGS_LIST:-0 4 3 2}; do
    if [ "$w" = 0 ]; then unset PTMI355_WGS_PER_CU; else export PTMI355_WGS_PER_CU=$w; fi
    PTMI355_LIB=$ROOT/.ab/$v/libptmi355.so python3 $ROOT/bench.py --config c4 --flags compact,bvh --steps 10 --warmup 2 --no-cpu-baseline --no-roofline --no-per-call 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s wgs/cu<=%s  value %9.1f  ms/step %8.4f' % ('$v', '$w', d['value'], d['ms_per_step']))"
  done
done
done
