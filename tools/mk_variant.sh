#!/bin/bash
# tools/mk_variant.sh NAME "DEFINES" [files...] -- a variant build of the library under ntt_amd/variants/ for A/B
# runs (tools/ab.sh): the listed kernel files (default: the three hot-path files) are recompiled with DEFINES,
# everything else comes from the regular build.
set -e
cd "$(dirname "$0")/../ntt_amd/csrc"
NAME=$1; DEFS=$2; shift 2 || true
FILES=${@:-kernels_col.hip kernels_col16.hip kernels_row.hip}
make -j8 >/dev/null
mkdir -p ../variants build/var_$NAME
OBJS=""
for f in kernels_col kernels_col16 kernels_row kernels_fused kernels_misc kernels_crt kernels_u64 kernels_dist engine crt direct64 stream harness multidev comm; do
  src=$f.hip; [ -f $src ] || src=$f.cpp
  if echo " $FILES " | grep -q " $src "; then
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $DEFS -x hip -c $src -o build/var_$NAME/$f.o ) &
    OBJS="$OBJS build/var_$NAME/$f.o"
  else
    OBJS="$OBJS build/$f.o"
  fi
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS -lhsa-runtime64 -ldl -o ../variants/libntt_amd_$NAME.so
ls -la ../variants/libntt_amd_$NAME.so
