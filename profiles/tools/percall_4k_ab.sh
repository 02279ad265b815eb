one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', {k: v for k, v in (d['config'].get('per_call') or {}).items() if k != 'note'})"; }
for c in c5 c3; do
python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "$c default"
PTMI355_WHOLE_MAX_HOST=6000000 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "$c one launch up to 6 M paths only"
# (round 5: config.per_call.pcie_inclusive_sync_every_pixel of every line is this plan -- PT_PIN_IMAGE without PT_HOST_SPARSE)
python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-sub 2>/dev/null | one "$c every pixel every call"
done
