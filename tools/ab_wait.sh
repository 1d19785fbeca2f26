#!/bin/bash
# Host-side wait policy of the HIP / HSA runtimes against the single-product latency (tools/time_single.py,
# tools/run_cli_cases.py): default, active wait in the HIP runtime, polling signals in the HSA runtime.
set -u
cd "$(dirname "$0")/.."
run() {
    echo "== $*"
    env "$@" timeout -k 10 120 python tools/time_single.py 2>&1 | grep -v amdgpu.ids || return 1
    env "$@" timeout -k 10 120 python tools/run_cli_cases.py 2>&1 | grep "average latency" || return 1
}
run NTT_AMD_NOP=1 && run ROC_ACTIVE_WAIT_TIMEOUT=100 && run HSA_ENABLE_INTERRUPT=0 && run NTT_AMD_NOP=1
