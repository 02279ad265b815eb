one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
print('$1', d['value'], d['ms_per_step'])"; }
for b in 64 32 16 8; do
python bench.py --config c3 --batch $b --steps $((1280 / b)) --warmup 5 --no-cpu-baseline --no-roofline --no-per-call --no-sub 2>/dev/null | one "unsorted batch $b"
python bench.py --config c3 --flags compact,sort --batch $b --steps $((1280 / b)) --warmup 5 --no-cpu-baseline --no-roofline --no-per-call --no-sub 2>/dev/null | one "sorted   batch $b"
done
