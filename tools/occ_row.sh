#!/bin/bash
# row kernel time vs occupancy: extra dynamic LDS per workgroup -> workgroups (= waves per SIMD) per CU
for dyn in 0 3072 9000 15000 23000 36000 64000; do
  echo -n "dynlds=$dyn wg_per_cu=$((163840/(17408+dyn))) "
  NTT_AMD_ROW_DYNLDS=$dyn NTT_AMD_STREAMS=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f'%d['value'], {k:round(v,2) for k,v in d['roofline']['kernel_ms'].items()})" || exit 1
done
