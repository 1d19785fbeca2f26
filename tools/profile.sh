#!/bin/bash
# tools/profile.sh TAG -- rocprofv3 evidence for bench.py on the GPU box (run through gpurun).
# Pass 1: --kernel-trace --stats (per-kernel time).  Passes 2-4: PMC counters, each in a run of
# its own (FETCH_SIZE and WRITE_SIZE cannot share a pass; SQ counters fill their own 8 slots).
# Raw output goes to gpurun_out/prof_TAG/; tools/summarize_profile.py condenses it for profiles/.
set -u
TAG=${1:-r01}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# default lanes (NTT_AMD_STREAMS unset): the trace then sees the same overlapped launches as bench.py
cd /tmp
BENCH="python3 $REPO/bench.py --no-cpu-baseline --no-one-lane --no-also --sustained-seconds 0"

timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $BENCH --steps 5 --warmup 2 \
    > "$OUT/trace.log" 2>&1 || { echo "kernel-trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
echo "trace pass done"
for PASS in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" \
            "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    NAME=$(echo $PASS | cut -d' ' -f1)
    timeout -s KILL 300 rocprofv3 --pmc $PASS --output-format csv -d "$OUT/pmc_$NAME" -- $BENCH --steps 1 --warmup 1 --settle-seconds 0 \
        > "$OUT/pmc_$NAME.log" 2>&1 || { echo "pmc pass $NAME failed"; tail -5 "$OUT/pmc_$NAME.log"; exit 1; }
    echo "pmc pass $NAME done"
done
# keep the merge small: only csv files travel back
find "$OUT" -type f ! -name '*.csv' ! -name '*.log' -delete
du -sh "$OUT"
