#!/bin/bash
# sweep_iter_grid2.sh: the library's defaults (work) against the previous rule (whole grid, four lanes) at 1 / 2 / 4 / 8 spp per call
. "$(dirname "$0")/need_experiments.sh"      # (experiment variables: the shipped library ignores them)
run() { env $1 python bench.py --steps $3 --warmup 40 --batch $2 --no-roofline --no-per-call --no-sub --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 batch $2:', d['value'], d['ms_per_step'])"; }
for r in 1 2; do for b in 1 2 4 8; do run "PTMI355_ITER_TPW=0 PTMI355_OVERLAP=4" $b $((800 / b)); run "X=1" $b $((800 / b)); done; done
run "PTMI355_ITER_WGS_ALL=12" 1 600; run "PTMI355_ITER_WGS_ALL=18" 1 600; run "PTMI355_ITER_TPW=6" 2 300; run "PTMI355_ITER_TPW=12" 2 300; run "PTMI355_ITER_TPW=12" 4 150
python bench.py --steps 10 --warmup 3 --no-roofline --no-sub --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k: v for k, v in d['config']['per_call'].items() if k != 'note'})"
