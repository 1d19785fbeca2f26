#!/bin/bash
# chunk-size sweep for bench.py (products per kernel launch); run through gpurun
for c in 28 32 56 64 112 128 256 512 1024; do
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 2 --chunk $c 2>/dev/null || exit 1
done
timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 10 --warmup 2 --chunk 128 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms_per_step_untimed'])"
