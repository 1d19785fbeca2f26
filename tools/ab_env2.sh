#!/bin/bash
# interleaved A/B of environment settings with the per-kernel one-lane table: tools/ab_env2.sh "VAR=val ..." "VAR=val ..."
for round in 1 2 3; do for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-also --sustained-seconds 0 --steps ${AB_STEPS:-50} --warmup 10 ${AB_ARGS} 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.0f'%d['value'], 'one-lane us/product:', {k:round(v['us_per_product'],3) for k,v in r['kernels_one_lane'].items()})" || exit 1
done; done
