#!/bin/bash
# sweep_queues.sh: 1 spp per call against hardware queues (GPU_MAX_HW_QUEUES), lanes (PTMI355_OVERLAP), the launch streams
# the lanes share (PTMI355_LANE_STREAMS) and the priority of the library's own launch stream (PTMI355_MAIN_PRIO): on the
# caller's stream (bench.py --batch 1: a torch stream) / the library's own stream (config.per_call), and through the
# library's worker threads (one context, nothing to exchange).  First the defaults, then round 4's earlier choices.
A="--steps 300 --warmup 30 --batch 1 --no-cpu-baseline --no-roofline --no-sub"
one() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], (d['config'].get('per_call') or {}).get('mrays_per_s'))"; }
for q in 2 3 4 6 8; do
  GPU_MAX_HW_QUEUES=$q python bench.py $A 2>/dev/null | one "defaults queues $q caller-stream / own-stream"
  GPU_MAX_HW_QUEUES=$q PTMI355_XCHG=peer python bench.py $A --no-per-call --inproc --reduce-every 1 2>/dev/null | one "defaults queues $q inproc"
done
