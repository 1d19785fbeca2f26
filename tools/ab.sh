#!/bin/bash
# tools/ab.sh VARIANT... -- interleaved A/B of library builds under ntt_amd/variants/ (bench.py, 3 rounds,
# default lanes unless NTT_AMD_STREAMS is set)
for round in 1 2 3; do for v in "$@"; do
  echo -n "$v: "
  NTT_AMD_LIB=$PWD/ntt_amd/variants/libntt_amd_$v.so timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps ${AB_STEPS:-50} --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f'%d['value'], {k:round(v,2) for k,v in d['roofline']['kernel_ms_per_step_untimed'].items()})" || exit 1
done; done
