#!/bin/bash
# Development aid: builds libvacancy_hip.so with extra compiler flags for carve_fused.hip into
# build/variants/<name>/ (A/B runs of kernel variants in one GPU session: VCY_HIP_LIB=<path> python bench.py).
#   profiles/tools/build_variant.sh <name> [extra hipcc flags...]
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/vacancy_amd/csrc
OUT=$ROOT/build/variants/$NAME
mkdir -p "$OUT"
make -C "$SRC" -s all
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
 -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -I$ROOT/include -I$SRC -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c "$SRC/carve_fused.hip" -o "$OUT/carve_fused.o"
OBJS=$(cd "$SRC" && ls *.o | grep -v '^carve_fused.o$' | sed "s#^#$SRC/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvacancy_hip.so" $OBJS "$OUT/carve_fused.o" -ldl -Wl,-rpath,/opt/rocm/lib
rm -f "$OUT/carve_fused.o"
echo "$OUT/libvacancy_hip.so"
