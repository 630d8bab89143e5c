#!/bin/bash
# round 3, call L: mesh kernel -- parity, rates (leaf 8 / 4), rocprof summaries at 6.5e4 and 1e6 facets
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
timeout -k 10 600 python -m pytest tests/test_mesh.py tests/test_brep.py tests/test_gpu_fuzz.py tests/test_surface_source.py -m gpu -x -q > $O/r03l_tests.log 2>&1 || { tail -60 $O/r03l_tests.log; exit 1; }
tail -2 $O/r03l_tests.log
for leaf in 8 4; do
  echo "leaf $leaf"
  ODW_BVH_LEAF=$leaf timeout -k 10 300 python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03l_err.log | tee $O/r03l_mesh_leaf$leaf.jsonl
done
timeout -k 10 300 python scripts/bench_mesh.py --segments 256 1024 --sigma 0.12 2>$O/r03l_err.log | tee $O/r03l_meshwide.jsonl
python scripts/profile_round.py r03l_mesh65k --script scripts/bench_mesh.py --script-args "--segments 256 --steps 3 --warmup 1 --plain" --kernel "odw_mesh_kernel" --rays 1e7 > $O/r03l_prof65k.log 2>&1 || { tail -30 $O/r03l_prof65k.log; exit 1; }
python scripts/profile_round.py r03l_mesh1m --script scripts/bench_mesh.py --script-args "--segments 1024 --steps 3 --warmup 1 --plain" --kernel "odw_mesh_kernel" --rays 1e7 > $O/r03l_prof1m.log 2>&1 || { tail -30 $O/r03l_prof1m.log; exit 1; }
cp gpurun_out/r03l_mesh*_pmc.json gpurun_out/r03l_mesh*_kernel_stats.csv gpurun_out/r03l_mesh*_pmc_current.json $O/ 2>/dev/null || true
python - <<'PY'
import json
for t in ('65k','1m'):
  d=json.load(open(f'gpurun_out/r03l_mesh{t}_pmc.json'))
  a=d['counters_avg_per_dispatch']
  print(t, 'ms', d['kernel_ms_rocprof'], {k: ('%.4g' % v) for k, v in a.items()})
PY
