#!/bin/bash
# round 3: calibration of the VALU issue peak (scripts/valu_peak.hip) + its counters
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
./scripts/valu_peak > gpurun_out/r03/valu_peak_streams.json
tail -c 600 gpurun_out/r03/valu_peak_streams.json
rocprofv3 -L > gpurun_out/r03/counters_list.txt 2>&1 || true
grep -c "" gpurun_out/r03/counters_list.txt
# counters of the same streams at 4 waves per SIMD (one counter set per run)
for k in 0 1; do
  if [ $k = 0 ]; then set_="SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU"; else set_="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY"; fi
  rocprofv3 --pmc $set_ --kernel-trace -d gpurun_out/r03/vp_pmc$k -o vp --output-format csv -- ./scripts/valu_peak --w4 > gpurun_out/r03/vp_pmc$k.log 2>&1
  find gpurun_out/r03/vp_pmc$k -name "*counter_collection.csv" | head -2
done
