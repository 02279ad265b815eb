#!/usr/bin/env python3
"""c5_render.py [OUTDIR]  (on the GPU box): BASELINE config C5 as stated -- scenes/cornell_4k.txt, 3840x2160, 5000
iterations, the frame tiled over eight contexts (on this pool: eight contexts on the one GPU) -- rendered once through
the headless host; what is kept (the 4K PNG is 25 MB, the raw sum 100 MB): a 960x540 box-filtered PNG of the tonemapped
frame, the pooled statistic against the reference's 5000-sample PNG, md5 of the raw running sum, rays, seconds.
    OUTDIR/c5_5000spp_960x540.png   OUTDIR/c5_5000spp.json   OUTDIR/c5_5000spp.log"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "r05")
    devices = sys.argv[2] if len(sys.argv) > 2 else "0,0,0,0,0,0,0,0"
    os.makedirs(out, exist_ok=True)
    pt = ge.load_package()
    exe = pt.build_ptbench()
    tmp = tempfile.mkdtemp()
    t0 = time.time()
    p = subprocess.run([exe, os.path.join(ROOT, "scenes", "cornell_4k.txt"), "--devices", devices, "--batch", "4",
                        "--out", os.path.join(tmp, "c5"), "--save-sum"], capture_output=True, text=True, timeout=1100)
    wall = time.time() - t0
    open(os.path.join(out, "c5_5000spp.log"), "w").write(p.stdout + p.stderr)
    if p.returncode != 0:
        raise SystemExit(p.stdout + p.stderr)
    W, H = 3840, 2160
    full = pt.load_pfm(os.path.join(tmp, "c5.5000samp.sum.pfm"), W, H)
    rgb = pt.image_to_rgb8(full, W, H, 5000.0)                              # saveImage's pipeline: (H, W, 3) uint8
    small = rgb.reshape(H // 4, 4, W // 4, 4, 3).astype(np.float32).mean(axis=(1, 3))
    from PIL import Image
    Image.fromarray(np.clip(np.round(small), 0, 255).astype(np.uint8)).save(os.path.join(out, "c5_5000spp_960x540.png"), optimize=True)
    golden = {"png_stat": np.load(os.path.join(ROOT, "tests", "golden", "png_stat.npz"))}
    from test_gpu_camera_tiles import c5_pooled_statistic
    line = [l for l in p.stdout.splitlines() if "Mrays/s" in l][-1]
    rec = {"scene": "scenes/cornell_4k.txt", "resolution": [W, H], "iterations": 5000, "devices": devices, "batch": 4,
           "ptbench": line, "wall_s": round(wall, 2), "sum_md5": hashlib.md5(full.tobytes()).hexdigest(),
           "png_md5_full_frame": hashlib.md5(rgb.tobytes()).hexdigest(),
           "pooled_rel_l2_vs_reference_png": round(c5_pooled_statistic(full, 5000, golden), 5),
           "mean_radiance": [round(float(v), 6) for v in (full / 5000.0).mean(axis=0)]}
    json.dump(rec, open(os.path.join(out, "c5_5000spp.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
