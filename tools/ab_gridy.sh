for round in 1 2 3; do
for gy in 0 16; do
  echo -n "[headline COL_GRIDY=$gy] "
  NTT_AMD_COL_GRIDY=$gy timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-one-lane --sustained-seconds 0 --steps 100 --warmup 10 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.0f'%d['value'], {k:round(v,2) for k,v in d['roofline']['kernel_ms_per_step_untimed'].items()})" || exit 1
done; done
for gy in 0 16 0 16; do echo "[crt COL_GRIDY=$gy]"; NTT_AMD_COL_GRIDY=$gy timeout -k 10 120 python3 tools/time_crt.py 2>&1 | grep "^timing"; done
