#!/bin/bash
for pipe in 0 1; do for c in 16 32 64 128 256; do
  echo -n "pipe=$pipe chunk=$c "
  NTT_AMD_PIPE=$pipe NTT_AMD_STREAMS=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 --chunk $c 2>/dev/null || exit 1
done; done
