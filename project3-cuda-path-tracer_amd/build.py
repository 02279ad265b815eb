"""Build libptmi355.so (HIP, gfx950) in-tree.  hipcc cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libptmi355.so")
# the translation unit first, then every header it includes (all of csrc/) and the C-ABI header
SOURCES = [os.path.join(HERE, "csrc", "ptmi355.hip")] + \
          sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc")) if f.endswith(".hpp")) + \
          [os.path.join(ROOT, "include", "ptmi355.h")]
STAMP = LIB + ".src-sha256"        # what the library in the tree was built from (sources + flags)
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


def source_digest():
    """sha256 over the sources' bytes and the compile flags: a library is current when it was built from exactly these
    (modification times do not survive a snapshot copy, and a prebuilt .so that travels with the tree must not be taken
    for the build of sources that changed afterwards)."""
    import hashlib
    h = hashlib.sha256(" ".join(HIPCC_FLAGS).encode())
    for path in SOURCES:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    want = source_digest()
    have = open(STAMP).read().strip() if os.path.exists(STAMP) else ""
    if force or not os.path.exists(LIB) or have != want:
        # built beside the target and renamed into place: a process that loads the library while another one builds it
        # (several ranks starting at once) sees the old file or the new one, never half of one
        tmp = "%s.tmp%d" % (LIB, os.getpid())
        cmd = [hipcc()] + HIPCC_FLAGS + ["-o", tmp, SOURCES[0]]
        if verbose:
            print(" ".join(cmd))
        try:
            subprocess.run(cmd, check=True, cwd=HERE)
            os.replace(tmp, LIB)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        with open(STAMP, "w") as f:
            f.write(want + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
