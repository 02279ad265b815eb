#!/bin/bash
# ab_flags.sh "bench args" VARIANT...: value, ms per step and the frame's md5 (--digest) per build (.ab/VARIANT/libptmi355.so,
# "work" = the in-tree library), two alternating rounds -- compiler-flag variants must reproduce the digest bit for bit
ARGS=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for round in 1 2; do
  for v in "$@"; do
    if [ "$v" = work ]; then unset PTMI355_LIB; else export PTMI355_LIB=$ROOT/.ab/$v/libptmi355.so; fi
    python3 $ROOT/bench.py $ARGS --digest --no-cpu-baseline --no-roofline --no-per-call 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-10s value %9.1f  ms/step %8.4f  md5 %s' % ('$v', d['value'], d['ms_per_step'], str(d['config'].get('image_md5') or d.get('image_md5'))[:12]))"
  done
done
