#!/bin/bash
# round 5: grid kernel with sorted hand-out + wave-synchronous interaction, A/B on hugeArray
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout -k 10 500 python scripts/ablate_c4_sorted.py 1.25e8 > gpurun_out/r05_c4_sorted_ab.log 2>gpurun_out/r05_c4_sorted_ab.err
echo rc=$?
tail -30 gpurun_out/r05_c4_sorted_ab.log; tail -5 gpurun_out/r05_c4_sorted_ab.err
