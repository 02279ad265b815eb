"""The reference-side binding (host/pathtrace_shim.cpp) must compile against the
reference's own headers.  Runs only where /root/reference exists (the build
container); the GPU box skips it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
CUDAINC = "/usr/local/lib/python3.10/dist-packages/triton/backends/nvidia/include"


@pytest.mark.skipif(not (os.path.isdir(REF) and os.path.isdir(CUDAINC)), reason="reference tree / CUDA headers absent")
def test_shim_compiles_against_reference_headers(tmp_path):
    src = os.path.join(ROOT, "project3-cuda-path-tracer_amd", "host", "pathtrace_shim.cpp")
    out = str(tmp_path / "shim.o")
    cmd = ["g++", "-std=c++11", "-c", "-w", "-I" + CUDAINC, "-I" + os.path.join(REF, "src"),
           "-I" + os.path.join(REF, "external", "include"), "-I" + os.path.join(ROOT, "include"), src, "-o", out]
    subprocess.run(cmd, check=True)
    syms = subprocess.run(["nm", "-C", out], check=True, capture_output=True, text=True).stdout
    for want in ("pathtraceInit(Scene*)", "pathtraceFree()", "pathtrace(uchar4*, int, int)"):
        assert (" T " + want) in syms, want
    for used in ("pt_init", "pt_free", "pt_trace", "pt_set_camera", "pt_last_error"):
        assert (" U " + used) in syms, used
