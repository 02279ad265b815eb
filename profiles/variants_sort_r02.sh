# material-sort kernel sweep (round 2): bash profiles/variants_sort_r02.sh "<flags>|<env>" ...
cd project3-cuda-path-tracer_amd
KEEP=$(mktemp /tmp/keep.XXXXXX.so); cp libptmi355.so "$KEEP"
# whatever happens (a failed build, an interrupted sweep), the in-tree library is put back; new sweeps use
# profiles/tools/build_variant.sh + ab.sh, which never touch it (PTMI355_LIB)
trap 'cp "$KEEP" libptmi355.so; rm -f "$KEEP"' EXIT
for v in "$@"; do
  fl="${v%%|*}"; en="${v#*|}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 $fl -o libptmi355.so csrc/ptmi355.hip 2>&1 | grep error
  echo " <- [$fl] [$en]"
  (cd .. && env $en timeout 120 python bench.py --config c3 --flags compact,sort --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   value', d['value'], d['roofline']['stage_ms'])")
done
