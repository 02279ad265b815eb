# on the GPU box, from the repository root: the reference host (oracle/_ref/refhost) and ptbench on scenes/cornell.txt at full size, wall time + PNG comparison -> profiles/r06/refhost_full_size.txt
set -e
O=$PWD/gpurun_out/r06/refhost; mkdir -p $O; cd $O
cp $GRAFT_REPO_ROOT/scenes/cornell.txt .
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/project3-cuda-path-tracer_amd:$LD_LIBRARY_PATH
for rep in 1 2 3; do s=$(date +%s.%N); $GRAFT_REPO_ROOT/oracle/_ref/refhost cornell.txt T$rep > refhost_$rep.log 2>&1 || true; e=$(date +%s.%N); echo "refhost: $(python3 -c "print(round($e-$s,3))") s wall"; tail -2 refhost_$rep.log; done
md5sum cornell.T1.5000samp.png
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as ge
pt = ge.load_package()
b = pt.build_ptbench()
import subprocess, time
t=time.time(); p = subprocess.run([b, "cornell.txt", "--out", "ptb"], capture_output=True, text=True); print("ptbench: %.2f s wall" % (time.time()-t), p.returncode, p.stdout[-300:], p.stderr[-300:])
PY
md5sum ptb*.png | head -3
python3 -c "
from PIL import Image; import numpy as np
a=np.asarray(Image.open('cornell.T1.5000samp.png').convert('RGB')); import glob
b=np.asarray(Image.open(glob.glob('ptb*.png')[0]).convert('RGB')); print('pixels equal:', (a==b).all(), a.shape)"
rm -f *.png
