#!/bin/bash
for s in 1 2; do for c in 64 128 256; do for k in 0 2 3; do
  echo -n "streams=$s chunk=$c col16=$k "
  NTT_AMD_COL16=$k NTT_AMD_STREAMS=$s timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 --chunk $c 2>/dev/null || exit 1
done; done; done
