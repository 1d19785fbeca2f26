#!/bin/bash
# Barrett kernels: LDS-staged forward column pass (the default since round 3, NTT_AMD_COL16=3) against the register kernel
# (NTT_AMD_COL16=2) over transform lengths; products/s, interleaved.
cd "$(dirname "$0")/.."
for cfg in "65536 2048" "32768 4096" "131072 1024" "262144 512" "1048576 128"; do
  set -- $cfg
  for v in 2 3 2 3; do
    echo -n "n=$1 batch=$2 NTT_AMD_COL16=$v: "
    NTT_AMD_COL16=$v timeout -k 10 120 python3 bench.py --barrett --n $1 --batch $2 --no-cpu-baseline --no-also --no-one-lane --sustained-seconds 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "import json,sys; print(round(json.loads(sys.stdin.read())['value']))" || exit 1
  done
done
