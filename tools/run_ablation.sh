#!/bin/bash
# one lane (kernels run alone), per-kernel time of each row-kernel ablation variant (tools/mk_ablation.sh)
for v in full nodata nolds notw arith; do
  echo -n "$v: "
  NTT_AMD_STREAMS=1 NTT_AMD_LIB=$PWD/ntt_amd/variants/libntt_amd_$v.so timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-check --steps 20 --warmup 5 2>/dev/null \
    | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f'%d['value'], 'row_mul avg launch us %.1f'%(d['roofline']['avg_launch_ms']*1e3), {k:round(v,3) for k,v in d['roofline']['kernel_ms_per_step_untimed'].items()})" || exit 1
done
