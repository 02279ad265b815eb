#!/bin/bash
# .ab/ab.sh "bench args" VARIANT...   -- alternate the builds (.ab/VARIANT/libptmi355.so; "work" = the in-tree library), 2 rounds each
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = work ]; then unset PTMI355_LIB; else export PTMI355_LIB=$ROOT/.ab/$v/libptmi355.so; fi
    python $ROOT/bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-40s value %9.1f  ms/step %8.3f  %s' % ('$v', '$ARGS'[:40], d['value'], d['ms_per_step'], {k: round(x,1) for k,x in d.get('roofline',{}).get('stage_ms',{}).items()}))"
  done
done
