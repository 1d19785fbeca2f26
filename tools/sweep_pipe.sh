#!/bin/bash
for c in 16 32 64 128; do
  echo -n "pipe=1 chunk=$c "
  NTT_AMD_PIPE=1 NTT_AMD_STREAMS=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 --chunk $c 2>/dev/null || exit 1
done
echo -n "pipe=0 default "; timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 2>/dev/null
