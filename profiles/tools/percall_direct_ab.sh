one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', {k: v for k, v in (d['config'].get('per_call') or {}).items() if k != 'note'})"; }
for r in 1 2; do
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "default"
# (round 5: config.per_call.pcie_inclusive_sync_every_pixel of every line is this plan -- PT_PIN_IMAGE without PT_HOST_SPARSE)
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "HOST_SPARSE=0"
PTMI355_EPI_DIRECT=0 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "EPI_DIRECT=0"
done
