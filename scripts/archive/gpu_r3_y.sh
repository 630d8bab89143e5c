#!/bin/bash
# round 3: SQ_INSTS_VALU of the compiled kernel against the launch size (GettingStarted): instructions per ray and per wave
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
rm -rf gpurun_out/r03y_pmc
ODW_SL_SIZES=1e6,3e6,1e7,3e7,1e8 ODW_SL_REPS=2 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace -d gpurun_out/r03y_pmc -o sl --output-format csv -- python3 scripts/short_launch.py > gpurun_out/r03/r03y_pmc.log 2>&1 || { tail -20 gpurun_out/r03/r03y_pmc.log; exit 1; }
f=$(find gpurun_out/r03y_pmc -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'PY' | tee gpurun_out/r03/r03y_insts_by_size.log
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'odw_spec_kernel' in r['Kernel_Name']]
by = collections.OrderedDict()
for r in rows:
  by.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
  by[r['Dispatch_Id']]['grid'] = int(r['Grid_Size'])
for d, v in by.items():
  print('dispatch', d, 'grid threads', v['grid'], 'waves', v.get('SQ_WAVES'), 'VALU insts %.4g' % v.get('SQ_INSTS_VALU', 0))
PY
rm -rf gpurun_out/r03y_pmc
