#!/bin/bash
# overlap_trace.sh ROUND [bench args]: kernel traces of the driver's bench command in the TIMED configuration (consecutive
# steps overlapped on the device) and with PTMI355_OVERLAP=0, summarised by trace_overlap.py, next to the line's own
# ms_per_step -> gpurun_out/ROUND/rocprof_ROUND_c2_overlap_summary.txt
R=${1:-r05}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/$R; mkdir -p $O
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-per-call --no-sustained $*"
cd /tmp && export TMPDIR=/tmp
rm -rf $O/ovl_on $O/ovl_off
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ovl_on -o t -- python3 $ROOT/bench.py $ARGS > $O/ovl_on.log 2>&1
PTMI355_OVERLAP=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ovl_off -o t -- python3 $ROOT/bench.py $ARGS > $O/ovl_off.log 2>&1
python3 $ROOT/bench.py $ARGS > $O/ovl_plain.json 2>/dev/null
cd $ROOT
S=$O/rocprof_${R}_c2_overlap_summary.txt
{
  echo "bench.py $ARGS   (rocprofv3 --kernel-trace --stats; profiles/tools/overlap_trace.sh)"
  python3 -c "import json,sys; d=json.loads(open('$O/ovl_plain.json').read().strip().splitlines()[-1]); print('the same command without the profiler: value %.1f Mrays/s, ms_per_step %.4f' % (d['value'], d['ms_per_step']))"
  for m in on off; do
    echo; echo "== consecutive steps overlapped: $m $( [ $m = off ] && echo '(PTMI355_OVERLAP=0)' )"
    grep -h '^{' $O/ovl_$m.log | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('line under the profiler: value %.1f Mrays/s, ms_per_step %.4f' % (d['value'], d['ms_per_step']))"
    python3 profiles/tools/trace_overlap.py $(find $O/ovl_$m -name "*kernel_trace.csv" | head -1) 20 8
  done
} > $S
cat $S
