#!/bin/bash
# build_counted.sh NAME KERNEL_SUBSTR [extra hipcc flags]: libptmi355.so whose kernel KERNEL_SUBSTR (mangled-name substring) counts how
# often each of its basic blocks executes (profiles/tools/isa_count.py) -> .ab/NAME/{libptmi355.so, map.json, plain.s}
# (host side with -DPT_EXPERIMENTS: the count buffer is switched on by PTMI355_DBG_COUNTS, an experiment variable since round 5)
set -e
NAME=$1; KERNEL=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/.ab/$NAME; mkdir -p "$OUT"
B=/opt/rocm/lib/llvm/bin
SRC=$ROOT/project3-cuda-path-tracer_amd/csrc/ptmi355.hip
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fPIC -std=c++17 $*"
cd "$OUT"
# the device assembly is the same for every NAME: made once per state of csrc/ and flags
KEY=$( (cat "$ROOT"/project3-cuda-path-tracer_amd/csrc/*; echo "$FLAGS") | sha256sum | cut -c1-16)
PLAIN=$ROOT/.ab/plain_$KEY.s
if [ ! -s "$PLAIN" ]; then /opt/rocm/bin/hipcc $FLAGS --cuda-device-only -S -o "$PLAIN.tmp.$$" "$SRC" && mv "$PLAIN.tmp.$$" "$PLAIN"; fi
ln -sf "$PLAIN" plain.s
if [ "${COUNT_MODE:-full}" = none ]; then cp plain.s counted.s; echo "{\"words\": 64}" > map.json; else python3 "$ROOT/profiles/tools/isa_count.py" instrument plain.s "$KERNEL" counted.s map.json ${COUNT_MODE:-full}; fi
$B/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c counted.s -o dev.o
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -o dev.out dev.o
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev.out -output=dev.hipfb
/opt/rocm/bin/hipcc $FLAGS -DPT_EXPERIMENTS --cuda-host-only -c "$SRC" -o host.o -Xclang -fcuda-include-gpubinary -Xclang dev.hipfb
/opt/rocm/bin/hipcc -shared -fPIC host.o -o libptmi355.so
rm -f dev.o dev.out dev.hipfb host.o counted.s
echo "$OUT/libptmi355.so"
