#!/bin/bash
# .ab/build_variant.sh NAME [REV] [extra hipcc flags...]: libptmi355.so of git revision REV (default: working tree) -> .ab/NAME/libptmi355.so
set -e
NAME=$1; REV=${2:-WORK}; shift; shift || true
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/.ab/$NAME; mkdir -p "$OUT/src/csrc" "$OUT/src/include"
if [ "$REV" = WORK ]; then
  cp $ROOT/project3-cuda-path-tracer_amd/csrc/* "$OUT/src/csrc/"; cp $ROOT/include/ptmi355.h "$OUT/src/include/"
else
  for f in $(git -C $ROOT ls-tree --name-only $REV project3-cuda-path-tracer_amd/csrc/); do git -C $ROOT show $REV:$f > "$OUT/src/csrc/$(basename $f)"; done
  git -C $ROOT show $REV:include/ptmi355.h > "$OUT/src/include/ptmi355.h"
fi
# the sources include "../../include/ptmi355.h"-style paths relative to csrc: mirror the tree depth
mkdir -p "$OUT/t/pkg/csrc" "$OUT/t/include"; cp "$OUT/src/csrc/"* "$OUT/t/pkg/csrc/"; cp "$OUT/src/include/ptmi355.h" "$OUT/t/include/"
(cd "$OUT/t/pkg" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -std=c++17 "$@" -o "$OUT/libptmi355.so" csrc/ptmi355.hip)
echo "$OUT/libptmi355.so"
