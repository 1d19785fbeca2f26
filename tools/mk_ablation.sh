#!/bin/bash
# tools/mk_ablation.sh -- timing-only builds of the row kernel (row_body.hpp NTT_ABL_* switches) as variant
# libraries under ntt_amd/variants/ (results are wrong on purpose: run bench.py with --no-check).
set -e
cd "$(dirname "$0")/../ntt_amd/csrc"
make -j8 >/dev/null
mkdir -p ../variants
OBJS="build/kernels_col.o build/kernels_col16.o build/kernels_misc.o build/kernels_u64.o build/engine.o build/crt.o build/direct64.o build/stream.o build/harness.o"
build() {  # name, defines
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function $2 -c kernels_row.hip -o build/kernels_row_$1.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJS build/kernels_row_$1.o -lhsa-runtime64 -o ../variants/libntt_amd_$1.so
}
build full ""
build nodata "-DNTT_ABL_NODATA"
build nolds "-DNTT_ABL_NOLDS"
build notw "-DNTT_ABL_NOTW"
build arith "-DNTT_ABL_NODATA -DNTT_ABL_NOLDS -DNTT_ABL_NOTW"
ls -la ../variants
