cd project3-cuda-path-tracer_amd
for v in "4 0" "5 0" "6 0" "8 0" "4 1" "5 1" "6 1"; do
  set -- $v
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DPT_MIN_WAVES=$1 -DPT_DEFER=$2 -Rpass-analysis=kernel-resource-usage -o libptmi355.so csrc/ptmi355.hip 2>&1 | grep -A12 "k_bounceILi0ELb1" | grep -E "VGPRs:|Scratch|Occupancy" | tr '\n' ' '
  echo " <- waves=$1 defer=$2"
  (cd .. && timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   fused  value', d['value'], 'kernel Grays/s', d['roofline']['grays_per_s_in_kernel'])")
  (cd .. && timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --flags compact,unfused 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   unfused value', d['value'], d['roofline']['stage_ms'])")
done
