# k_mesh sweep (round 2): bash profiles/variants_mesh_r02.sh "<flags>" ...
cd project3-cuda-path-tracer_amd
cp libptmi355.so /tmp/keep.so
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 $v -Rpass-analysis=kernel-resource-usage -o libptmi355.so csrc/ptmi355.hip 2>&1 | grep -A12 "6k_meshILb1E" | grep -E "VGPRs:|Scratch|Occupancy" | awk '{print $3,$4,$5}' | tr '\n' ' '
  echo " <- [$v]"
  (cd .. && timeout 120 python bench.py --config c4 --flags compact,bvh --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', d['value'], d['roofline']['stage_ms'])")
done
cp /tmp/keep.so libptmi355.so
