# k_mesh sweep (round 2): bash profiles/variants_mesh_r02.sh "<flags>" ...
cd project3-cuda-path-tracer_amd
KEEP=$(mktemp /tmp/keep.XXXXXX.so); cp libptmi355.so "$KEEP"
# whatever happens (a failed build, an interrupted sweep), the in-tree library is put back; new sweeps use
# profiles/tools/build_variant.sh + ab.sh, which never touch it (PTMI355_LIB)
trap 'cp "$KEEP" libptmi355.so; rm -f "$KEEP"' EXIT
for v in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 $v -Rpass-analysis=kernel-resource-usage -o libptmi355.so csrc/ptmi355.hip 2>&1 | grep -A12 "6k_meshILb1E" | grep -E "VGPRs:|Scratch|Occupancy" | awk '{print $3,$4,$5}' | tr '\n' ' '
  echo " <- [$v]"
  (cd .. && timeout 120 python bench.py --config c4 --flags compact,bvh --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', d['value'], d['roofline']['stage_ms'])")
done
