#!/bin/bash
# interleaved A/B of environment settings: tools/ab_env.sh "VAR=val ..." "VAR=val ..."
for round in 1 2 3; do for cfg in "$@"; do
  echo -n "[$cfg] "
  env $cfg timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps ${AB_STEPS:-50} --warmup 10 2>/dev/null || exit 1
done; done
