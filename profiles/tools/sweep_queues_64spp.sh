one() { python -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])
p=d['config'].get('per_iteration_exchange') or {}
print('$1', d['value'], d['ms_per_step'], p.get('mrays_per_s'), p.get('ratio'))"; }
for r in 1 2; do for q in 4 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-roofline --no-per-call 2>/dev/null | one "queues $q c2 64spp"
GPU_MAX_HW_QUEUES=$q python bench.py --config c5 --no-cpu-baseline --no-roofline --no-per-call 2>/dev/null | one "queues $q c5"
GPU_MAX_HW_QUEUES=$q python bench.py --config c3 --flags compact,sort --no-cpu-baseline --no-roofline --no-per-call 2>/dev/null | one "queues $q c3 sorted"
GPU_MAX_HW_QUEUES=$q python bench.py --config c4 --flags compact,bvh --no-cpu-baseline --no-roofline --no-per-call 2>/dev/null | one "queues $q c4 bvh"
GPU_MAX_HW_QUEUES=$q python bench.py --gpus 1 --force-dist --no-cpu-baseline --no-roofline --sub-iters 1024 2>/dev/null | one "queues $q force-dist"
done; done
