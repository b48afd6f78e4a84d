#!/bin/bash
# Development aid: builds libvacancy_hip.so with extra compiler flags for one source file (carve_fused.hip, or
# $VARIANT_SRC) into build/variants/<name>/ (A/B runs of kernel variants in one GPU session:
# VCY_HIP_LIB=<path> python bench.py).
#   [VARIANT_SRC=mc_kernels.hip] profiles/tools/build_variant.sh <name> [extra hipcc flags...]
set -eu
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
SRC=$ROOT/vacancy_amd/csrc
OUT=$ROOT/build/variants/$NAME
VSRC=${VARIANT_SRC:-carve_fused.hip}
VOBJ=${VSRC%.hip}.o
mkdir -p "$OUT"
# every object but the variant's own
make -C "$SRC" -s $(cd "$SRC" && ls *.hip | grep -v -e "^$VSRC\$" $([ "$VSRC" = carve_fused.hip ] && echo "-e ^carve_fused_u8.hip\$ -e ^carve_fused_u16.hip\$") | sed 's/\.hip$/.o/')
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
 -fno-gpu-flush-denormals-to-zero -Wall -Wno-unused-function -I$ROOT/include -I$SRC -fno-slp-vectorize"
# (carve_fused.hip is compiled three times: by itself -- host side and small kernels -- and included by the two units that
# hold the halves of the carve kernel's instances; a variant of it is a variant of all three)
VSRCS="$VSRC"
[ "$VSRC" = carve_fused.hip ] && VSRCS="carve_fused.hip carve_fused_u8.hip carve_fused_u16.hip"
VOBJS=""; EXCL=""
for f in $VSRCS; do
  o=${f%.hip}.o
  /opt/rocm/bin/hipcc $FLAGS "$@" -c "$SRC/$f" -o "$OUT/$o" &
  VOBJS="$VOBJS $OUT/$o"; EXCL="$EXCL -e ^$o\$"
done
wait
OBJS=$(cd "$SRC" && ls *.o | grep -v $EXCL | sed "s#^#$SRC/#")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/libvacancy_hip.so" $OBJS $VOBJS -ldl -Wl,-rpath,/opt/rocm/lib
rm -f $VOBJS
echo "$OUT/libvacancy_hip.so"
