. "$(dirname "$0")/need_experiments.sh"      # (experiment variables: the shipped library ignores them)
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], (d['config'].get('per_call') or {}).get('mrays_per_s'))"; }
export PTMI355_OVERLAP=4 PTMI355_LANE_STREAMS=2
for g in 8 10 12 15 18 20; do
  PTMI355_ITER_WGS_ALL=$g python bench.py --steps 600 --warmup 40 --batch 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "wgs_all $g batch 1 (caller / own)"
  PTMI355_ITER_WGS_ALL=$g python bench.py --steps 300 --warmup 40 --batch 2 --no-cpu-baseline --no-roofline --no-sub --no-per-call 2>/dev/null | one "wgs_all $g batch 2"
done
for l in "3 3" "2 2" "4 2" "6 2" "8 2" "3 1" "2 1"; do set -- $l
  PTMI355_OVERLAP=$1 PTMI355_LANE_STREAMS=$2 python bench.py --steps 600 --warmup 40 --batch 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "lanes $1 streams $2 batch 1 (caller / own)"
done
