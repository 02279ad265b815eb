# sourced by the A/B scripts that switch EXPERIMENT variables (PTMI355_WGS_PER_CU, _LANE_STREAMS, _ITER_TPW, _MULTI_DIRECT, ...):
# the shipped library ignores those (include/ptmi355.h, "Environment"), so such a script would compare identical
# configurations without a word.  Point PTMI355_LIB at a -DPT_EXPERIMENTS build (profiles/tools/build_variant.sh x WORK
# -DPT_EXPERIMENTS) or stop here.
ROOT_NE="$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)"
if [ -z "$PTMI355_LIB" ] && [ -f "$ROOT_NE/.ab/x/libptmi355.so" ]; then export PTMI355_LIB="$ROOT_NE/.ab/x/libptmi355.so"; fi
python3 - <<'PY' || { echo "$(basename "$0"): needs a -DPT_EXPERIMENTS build: profiles/tools/build_variant.sh x WORK -DPT_EXPERIMENTS (then PTMI355_LIB=.ab/x/libptmi355.so)" >&2; exit 2; }
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.getcwd())
import ctypes
lib = os.environ.get("PTMI355_LIB")
if not lib or not os.path.exists(lib):
    sys.exit(1)
L = ctypes.CDLL(lib)
L.pt_version.restype = ctypes.c_char_p
sys.exit(0 if b"+experiments" in L.pt_version() else 1)
PY
