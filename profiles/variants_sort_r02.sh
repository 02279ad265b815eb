# material-sort kernel sweep (round 2): bash profiles/variants_sort_r02.sh "<flags>|<env>" ...
cd project3-cuda-path-tracer_amd
cp libptmi355.so /tmp/keep.so
for v in "$@"; do
  fl="${v%%|*}"; en="${v#*|}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 $fl -o libptmi355.so csrc/ptmi355.hip 2>&1 | grep error
  echo " <- [$fl] [$en]"
  (cd .. && env $en timeout 120 python bench.py --config c3 --flags compact,sort --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', d['value'], d['roofline']['stage_ms'])")
done
cp /tmp/keep.so libptmi355.so
