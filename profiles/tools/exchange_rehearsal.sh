#!/bin/bash
# exchange_rehearsal.sh [OUTDIR]: the per-iteration exchange on ONE GPU (the whole frame, 7.68 MB, sent to itself after every
# iteration -- what the root of eight receives), 800x800, 1 spp per call: no exchange / in the library over RCCL and over
# peer copies / one process per GPU through torch.distributed (nccl).  One JSON line each into OUTDIR/x_*.json.
OUT=${1:-gpurun_out/r04}; mkdir -p $OUT
A="--steps 300 --warmup 30 --batch 1 --no-cpu-baseline --no-roofline --no-per-call --no-sub"
python bench.py $A 2>$OUT/x_none.err | tail -1 > $OUT/x_none.json
PTMI355_XCHG=rccl python bench.py $A --inproc --reduce-every 1 2>$OUT/x_inproc.err | tail -1 > $OUT/x_inproc_rccl.json
# one context, transport "none": what the library's worker / exchange threads cost with nothing to move
PTMI355_XCHG=peer python bench.py $A --inproc --reduce-every 1 2>>$OUT/x_inproc.err | tail -1 > $OUT/x_inproc_none.json
# two contexts on the one device, copies on the exchange streams (half the frame moves)
python bench.py $A --gpus 2 --same-device --inproc --reduce-every 1 2>>$OUT/x_inproc.err | tail -1 > $OUT/x_inproc_peer2.json
python bench.py $A --force-dist --reduce-every 1 2>$OUT/x_dist.err | tail -1 > $OUT/x_dist.json
python - $OUT <<'PY'
import json, sys
for n in ("x_none", "x_inproc_none", "x_inproc_rccl", "x_inproc_peer2", "x_dist"):
    try:
        d = json.loads(open("%s/%s.json" % (sys.argv[1], n)).read().strip().splitlines()[-1])
        print("%-14s %9.1f Mrays/s  %.4f ms per iteration" % (n, d["value"], d["ms_per_step"]))
    except Exception as e:
        print(n, "failed:", e)
PY
