#!/bin/bash
for mask in 1 2 4 3 5 6 7; do
  echo -n "mask=$mask "
  NTT_AMD_PIPE_MASK=$mask NTT_AMD_PIPE=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 --chunk 64 2>/dev/null || exit 1
done
