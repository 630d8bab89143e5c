#!/bin/bash
# round 3, call F: mesh (BVH) kernels -- baseline rates, counters at 6.5e4 and 1e6 facets, occupancy / leaf-size A/B
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
python scripts/bench_mesh.py --segments 0 64 256 1024 > $O/r03f_mesh_base.jsonl 2>$O/r03f_err.log || { tail -20 $O/r03f_err.log; exit 1; }
cat $O/r03f_mesh_base.jsonl
python scripts/bench_mesh.py --segments 256 1024 --sigma 0.12 > $O/r03f_mesh_wide.jsonl 2>$O/r03f_err.log || { tail -20 $O/r03f_err.log; exit 1; }
cat $O/r03f_mesh_wide.jsonl
for leaf in 2 4 16; do
  echo "leaf $leaf"
  ODW_BVH_LEAF=$leaf python scripts/bench_mesh.py --segments 256 1024 2>$O/r03f_err.log | tee $O/r03f_mesh_leaf$leaf.jsonl
done
for v in bvhw2 bvhw3; do
  echo "variant $v"
  ODW_TRACE_LIB=$PWD/build/libodw_$v.so python scripts/bench_mesh.py --segments 64 256 1024 2>$O/r03f_err.log | tee $O/r03f_mesh_$v.jsonl
done
python scripts/profile_round.py r03f_mesh65k --script scripts/bench_mesh.py --script-args "--segments 256 --steps 3 --warmup 1" --kernel "odw_trace_kernel<true" --rays 1e7 > $O/r03f_prof65k.log 2>&1 || { tail -30 $O/r03f_prof65k.log; exit 1; }
python scripts/profile_round.py r03f_mesh1m --script scripts/bench_mesh.py --script-args "--segments 1024 --steps 3 --warmup 1" --kernel "odw_trace_kernel<true" --rays 1e7 > $O/r03f_prof1m.log 2>&1 || { tail -30 $O/r03f_prof1m.log; exit 1; }
cp gpurun_out/r03f_mesh*_pmc.json gpurun_out/r03f_mesh*_kernel_stats.csv gpurun_out/r03f_mesh*_pmc_current.json $O/ 2>/dev/null || true
python - <<'PY'
import json
for t in ('65k','1m'):
  d=json.load(open(f'gpurun_out/r03f_mesh{t}_pmc.json'))
  a=d['counters_avg_per_dispatch']
  print(t, 'ms', d['kernel_ms_rocprof'], {k: ('%.4g' % v) for k, v in a.items()})
PY
