#!/bin/bash
# pmc_kernel.sh TAG "COUNTER GROUP 1" "COUNTER GROUP 2" ... -- bench-args...
# One rocprofv3 --pmc pass per counter group (counters never combined with trace domains), summed per kernel name and
# divided by the number of dispatches -> gpurun_out/pmck_TAG/summary.txt (per-launch averages).
TAG=$1; shift
CGS=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do CGS+=("$1"); shift; done
shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmck_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-per-call $*"
k=0
for g in "${CGS[@]}"; do
  timeout 300 rocprofv3 --pmc $g --output-format csv -d "$OUT/g$k" -o p -- python3 "$ROOT/bench.py" $ARGS > "$OUT/g$k.log" 2>&1 || echo "group $k ($g) failed: $(tail -2 $OUT/g$k.log)"
  k=$((k+1))
done
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections, re
out = sys.argv[1]
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('(anonymous namespace)::', '')
    return n[:60]
val = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(out + '/g*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        val[k][r['Counter_Name']] += float(r['Counter_Value'])
        calls[k][r['Counter_Name']].add(r['Dispatch_Id'])
for k in sorted(val, key=lambda k: -sum(val[k].values())):
    print(k)
    for c in sorted(val[k]):
        n = max(1, len(calls[k][c]))
        print('    %-36s %16.1f per launch  (%d launches)' % (c, val[k][c] / n, n))
PY
cat "$OUT/summary.txt" | head -120
