#!/bin/bash
# experiment_tests.sh: the -m gpu tests that carry experiment / test-hook variants (PTMI355_FIN_SERIAL's stamp wrap,
# lane-stream layouts, grid sizes, PTMI355_EPI_DIRECT / HOST_EPILOGUE / ASYNC_DIRECT / MULTI_DIRECT / XCHG_THREAD ...),
# run against a -DPT_EXPERIMENTS build of the working tree (.ab/x, built in the build container beforehand:
#   profiles/tools/build_variant.sh x WORK -DPT_EXPERIMENTS).  The shipped library reads none of those variables, so the
# same tests skip these variants in the driver's run (pt.has_experiments() is False there).
set -e
cd "$(dirname "$0")/../.."
test -f .ab/x/libptmi355.so || { echo "build .ab/x first: profiles/tools/build_variant.sh x WORK -DPT_EXPERIMENTS"; exit 2; }
export PTMI355_LIB=$PWD/.ab/x/libptmi355.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_overlap.py tests/test_gpu_host_image.py tests/test_multi_device.py -m gpu -q -x \
  -k "one_iteration_per_call_overlapped or overlapped_small_batches or host_image_kept_current or async_image_written or final_colour_stamps or frame_over_two_contexts or frame_tiled_over" "$@"
