# occupancy / geom-source sweep (round 1).  usage: bash profiles/variants_r01.sh
cd project3-cuda-path-tracer_amd
for v in "0 0" "0 36000" "0 28000" "1 0" "1 36000" "1 28000"; do
  set -- $v
  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 -DPT_GEOM_LDS=$1 -Rpass-analysis=kernel-resource-usage -o libptmi355.so csrc/ptmi355.hip 2>&1 | grep -A12 "k_bounceILi0ELb1ELb0" | grep -E "VGPRs:|Scratch|Occupancy" | awk '{print $3,$4,$5}' | tr '\n' ' '
  echo " <- geom_lds=$1 lds_pad=$2"
  (cd .. && PTMI355_LDS_PAD=$2 timeout 120 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   fused  value', d['value'], 'kernel Grays/s', d['roofline']['grays_per_s_in_kernel'])")
done
