# bench lines at the grid the library picks (and PTMI355_WGS_PER_CU caps, for comparison)
. "$(dirname "$0")/need_experiments.sh"      # (experiment variables: the shipped library ignores them)
for w in ${WGS_LIST:-8 5}; do
  for cfg in "--config c3 --flags compact,sort" "--config c2 --batch 1 --steps 200 --warmup 20" "--config c2" "--config c3" "--config c5 --batch 4 --steps 5 --warmup 1" "--config c4 --flags compact,bvh --steps 10 --warmup 2"; do
    echo "WGS<=$w $cfg: $(PTMI355_WGS_PER_CU=$w python3 bench.py $cfg --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])')"
  done
done
