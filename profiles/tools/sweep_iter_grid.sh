#!/bin/bash
# sweep_iter_grid.sh: k_iteration's grid (tiles per wave, PTMI355_ITER_TPW) x lanes (PTMI355_OVERLAP) at 1 / 2 / 4 spp per call
. "$(dirname "$0")/need_experiments.sh"      # (experiment variables: the shipped library ignores them)
run() { PTMI355_ITER_TPW=$1 PTMI355_OVERLAP=$2 python bench.py --steps 400 --warmup 40 --batch $3 --no-roofline --no-per-call --no-sub --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tpw $1 lanes $2 batch $3:', d['value'], d['ms_per_step'])"; }
for b in 1 2 4; do for t in 0 3 5 8; do for ov in 4 6; do run $t $ov $b; done; done; done
run 5 5 1; run 5 7 1; run 4 6 1; run 6 6 1
for t in 0 3 5; do echo "alone tpw $t"; PTMI355_ITER_TPW_ALONE=$t python bench.py --steps 5 --warmup 2 --no-roofline --no-sub --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k: v for k, v in d['config']['per_call'].items() if k != 'note'})"; done
