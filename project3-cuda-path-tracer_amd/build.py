"""Build libptmi355.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libptmi355.so")
SOURCES = [os.path.join(HERE, "csrc", f) for f in ("ptmi355.hip", "pt_device.hpp", "pt_types.hpp", "pt_kernels.hpp", "pt_bvh.hpp", "pt_cull.hpp")] + \
          [os.path.join(ROOT, "include", "ptmi355.h")]
# -ffp-contract=off: the reference arithmetic (GLM, no FMA) must be reproduced bit for bit.
# -fno-slp-vectorize: the SLP vectoriser pairs fp32 multiplies / adds into v_pk_* and pays for it in register
# shuffles; v_pk_mul_f32 issues in 4.3 cycles against 2 x 2.45 (profiles/r02/valu_peak_r02.json): +3 % without it.
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17"]


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    if force or stale(LIB, SOURCES):
        cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB, SOURCES[0]]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
