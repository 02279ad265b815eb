. "$(dirname "$0")/need_experiments.sh"      # (experiment variables: the shipped library ignores them)
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'])"; }
for r in 1 2; do
for b in 1 2 4 8; do
  python bench.py --steps $((800 / b)) --warmup 40 --batch $b --no-cpu-baseline --no-roofline --no-sub --no-per-call 2>/dev/null | one "default(6/6) batch $b"
  PTMI355_OVERLAP=4 PTMI355_LANE_STREAMS=2 python bench.py --steps $((800 / b)) --warmup 40 --batch $b --no-cpu-baseline --no-roofline --no-sub --no-per-call 2>/dev/null | one "lanes 4 streams 2 batch $b"
done
for ls in "4 4" "4 2" "6 3" "6 2"; do set -- $ls
  PTMI355_OVERLAP=$1 PTMI355_LANE_STREAMS=$2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-sub --no-per-call 2>/dev/null | one "64 spp lanes $1 streams $2"
  PTMI355_OVERLAP=$1 PTMI355_LANE_STREAMS=$2 python bench.py --config c3 --flags compact,sort --no-cpu-baseline --no-roofline --no-sub --no-per-call 2>/dev/null | one "c3 sorted lanes $1 streams $2"
done
done
