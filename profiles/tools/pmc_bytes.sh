#!/bin/bash
# pmc_bytes.sh TAG bench-args...: FETCH_SIZE / WRITE_SIZE / kernel time per kernel of one bench.py command (three
# rocprofv3 runs: trace, FETCH_SIZE, WRITE_SIZE; counters never combined with trace domains) -> gpurun_out/pmc_TAG/summary.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-roofline $*"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 "$ROOT/bench.py" $ARGS > "$OUT/trace.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o p -- python3 "$ROOT/bench.py" $ARGS > "$OUT/fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o p -- python3 "$ROOT/bench.py" $ARGS > "$OUT/write.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import csv, glob, sys, collections, re
out = sys.argv[1]
def short(n):
    n = re.sub(r'^void ', '', n); n = n.replace('(anonymous namespace)::', '')
    return n[:70]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        a = agg[short(r['Kernel_Name'])]; a[0] += 1; a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
for name, col in (('fetch', 2), ('write', 3)):
    for f in glob.glob(out + '/%s/**/*counter_collection.csv' % name, recursive=True):
        for r in csv.DictReader(open(f)):
            agg[short(r['Kernel_Name'])][col] += float(r['Counter_Value'])
print('%-70s %6s %10s %12s %12s' % ('kernel', 'calls', 'us total', 'FETCHx2 MB', 'WRITE MB'))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    # FETCH_SIZE / WRITE_SIZE are in KB of 1024 bytes, printed as MB = 1e6 bytes; FETCH is doubled (MI355X_MICROARCH.md: gfx950 reports half of wide coalesced reads)
    print('%-70s %6d %10.0f %12.1f %12.1f' % (k, a[0], a[1], a[2] * 2 * 1024 / 1e6, a[3] * 1024 / 1e6))
PY
cat "$OUT/summary.txt"
