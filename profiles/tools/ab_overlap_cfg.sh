# every bench configuration with consecutive batches overlapped on n lanes (PTMI355_OVERLAP; 0 = one stream)
for ov in ${OV_LIST:-0 2 4}; do
  for cfg in "--config c2" "--config c3" "--config c3 --flags compact,sort" "--config c5 --batch 4 --steps 10 --warmup 2" "--config c4 --flags compact,bvh --steps 10 --warmup 2" "--config c2 --batch 1 --steps 400 --warmup 40" "--config c2 --batch 16 --steps 40 --warmup 4"; do
    echo "lanes $ov $cfg: $(PTMI355_OVERLAP=$ov python3 bench.py $cfg --no-cpu-baseline --no-roofline 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(d["value"], d["ms_per_step"])')"
  done
done
