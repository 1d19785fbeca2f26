#!/bin/bash
# sweep of tools/ubench_xcdsync.hip variants (built into ntt_amd/variants/); stops at the first failure
cd ntt_amd/variants || exit 1
for v in 0 1 2; do for m in 0 1; do for pc in 1 2 4; do for kib in 2048; do
  echo -n "variant $v: "
  timeout -k 5 30 ./ubench_xcdsync_v$v $m $pc $kib 100 | tail -1 || exit 1
done; done; done; done
