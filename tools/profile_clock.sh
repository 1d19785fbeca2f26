#!/bin/bash
# effective clock under load: GRBM_GUI_ACTIVE / 8 / duration on long dispatches (chunk = 1024)
# usage: profile_clock.sh [tag]   (NTT_AMD_LIB selects a variant library, CLOCK_STEPS the run length)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_clock${1:+_$1}
mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
NTT_AMD_STREAMS=1 timeout -s KILL 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/pmc" -- python3 $REPO/bench.py --no-cpu-baseline --no-check --no-kernel-timing --steps ${CLOCK_STEPS:-3} --warmup 1 --chunk 1024 > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
find "$OUT" -type f ! -name '*.csv' ! -name '*.txt' -delete
python3 - "$OUT" <<'PY'
import csv,glob,sys,collections
rows=collections.defaultdict(dict)
for f in glob.glob(sys.argv[1]+'/pmc/**/*counter_collection.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        if 'ntt::' not in r['Kernel_Name']: continue
        k=(r['Dispatch_Id'],r['Kernel_Name'].split('<')[0].split('::')[-1]+('_inv' if ', true, false>' in r['Kernel_Name'] else ''))
        rows[k][r['Counter_Name']]=float(r['Counter_Value'])
        rows[k]['dur']=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
for k,v in sorted(rows.items(), key=lambda kv:int(kv[0][0]))[-9:]:
    print(k[1], 'dur_us=%.1f'%v['dur'], 'clock_GHz=%.3f'%(v['GRBM_GUI_ACTIVE']/8/v['dur']/1e3), 'wave_cyc_per_wave=%.0f'%(4*v['SQ_WAVE_CYCLES']/v['SQ_WAVES']), 'valu/wave=%.0f'%(v['SQ_INSTS_VALU']/v['SQ_WAVES']), 'wait_inst=%.2f'%(v['SQ_WAIT_INST_ANY']/v['SQ_WAVE_CYCLES']), 'wait_any=%.2f'%(v['SQ_WAIT_ANY']/v['SQ_WAVE_CYCLES']), 'active=%.2f'%(v['SQ_ACTIVE_INST_ANY']/v['SQ_WAVE_CYCLES']))
PY
