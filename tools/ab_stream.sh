#!/bin/bash
# env experiments for tools/bench_stream.py (which copy engines / hardware queues the runtime picks)
for cfg in "A=1" "GPU_MAX_HW_QUEUES=8" "HSA_ENABLE_SDMA=1" "HSA_ENABLE_SDMA=0" "NTT_AMD_STREAMS=1"; do
  echo "[$cfg]"
  env $cfg timeout -k 10 200 python3 tools/bench_stream.py --slice 64 --count 512 2>&1 | grep -E "^\{|rror" | cut -c1-170 || exit 1
done
