#!/bin/bash
# ab_r04.sh "bench args" VARIANT...  -- like ab.sh (two alternating rounds per build, .ab/VARIANT/libptmi355.so, "work" = the
# in-tree library), printing value, ms per step and -- when the line has it -- config.per_call (the drop-in call pattern)
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = work ]; then unset PTMI355_LIB; else export PTMI355_LIB=$ROOT/.ab/$v/libptmi355.so; fi
    python $ROOT/bench.py $ARGS --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
pc=d['config'].get('per_call') or {}
print('%-8s %-44s value %9.1f  ms/step %8.4f  per_call %s' % ('$v', '$ARGS'[:44], d['value'], d['ms_per_step'], {k: pc[k] for k in ('mrays_per_s','pcie_inclusive_sync','pcie_inclusive_async') if k in pc}))"
  done
done
