#!/bin/bash
# tools/mk_ablation.sh -- timing-only builds of the row kernel (row_body.hpp NTT_ABL_* switches) as variant
# libraries under ntt_amd/variants/ (results are wrong on purpose: run bench.py with --no-check).
set -e
cd "$(dirname "$0")"
bash mk_variant.sh full "" kernels_row.hip
bash mk_variant.sh nodata "-DNTT_ABL_NODATA" kernels_row.hip
bash mk_variant.sh nolds "-DNTT_ABL_NOLDS" kernels_row.hip
bash mk_variant.sh notw "-DNTT_ABL_NOTW" kernels_row.hip
bash mk_variant.sh arith "-DNTT_ABL_NODATA -DNTT_ABL_NOLDS -DNTT_ABL_NOTW" kernels_row.hip
