#!/bin/bash
# sub_exchange_check.sh: config.per_iteration_exchange of the default (64 spp per step) line in the one-rank rehearsal,
# with the tile gather issued from the tracing thread and from sharding.TileGatherThread
sub() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
p=d['config'].get('per_iteration_exchange') or {}
print('$1', d['value'], p.get('mrays_per_s'), p.get('no_exchange_mrays_per_s'), p.get('ratio'), p.get('transport'), d['config'].get('sub_measurements'))"; }
B="python bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --sub-iters 1024"
for r in 1 2; do
$B 2>/dev/null | sub "default"
$B --no-exchange-thread 2>/dev/null | sub "--no-exchange-thread"
$B --exchange-thread 2>/dev/null | sub "--exchange-thread"
done
