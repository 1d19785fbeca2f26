#!/bin/bash
# lanes x products-per-launch sweep of the headline configuration (rate only): tools/sweep_lanes_chunk.sh
for round in 1 2; do for s in 2 3; do for c in 40 48 56 64; do
  echo -n "streams=$s chunk=$c: "
  NTT_AMD_STREAMS=$s timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 --chunk $c 2>/dev/null | python3 -c "import json,sys; print('%.0f' % json.loads(sys.stdin.read())['value'])" || exit 1
done; done; done
