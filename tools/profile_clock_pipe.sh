#!/bin/bash
# Effective clock (GRBM_GUI_ACTIVE / 8 / duration) of the fused pipeline kernel with subsets of roles
# enabled: does running memory-bound and VALU-bound roles together lower the clock (power limit)?
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp; cd /tmp
for mask in 1 2 4 7; do
  OUT=$REPO/gpurun_out/prof_clockpipe_$mask
  mkdir -p "$OUT"
  NTT_AMD_PIPE=1 NTT_AMD_PIPE_MASK=$mask timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc" -- python3 $REPO/bench.py --no-cpu-baseline --no-kernel-timing --steps 3 --warmup 1 --chunk 256 > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
  find "$OUT" -type f ! -name '*.csv' ! -name '*.txt' -delete
  python3 - "$OUT" $mask <<'PY'
import csv,glob,sys,collections
rows=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'pipe_kernel' not in r['Kernel_Name']: continue
        k=r['Dispatch_Id']
        rows[k][r['Counter_Name']]=float(r['Counter_Value'])
        rows[k]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
vals=[v for v in rows.values() if v['dur']>200]
if vals:
    d=sum(v['dur'] for v in vals)/len(vals); clk=sum(v['GRBM_GUI_ACTIVE']/8/v['dur']/1e3 for v in vals)/len(vals)
    print('mask',sys.argv[2],'dispatches',len(vals),'avg_dur_us %.1f'%d,'clock_GHz %.3f'%clk)
PY
done
