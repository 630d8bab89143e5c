#!/bin/bash
# round 3, final pass, part 2: counter passes (profiles/r03/<tag>_*)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
O=gpurun_out/r03
T=${1:-r03z}
for c in c3 c4 c5; do
  python scripts/profile_round.py ${T}_$c --config $c > $O/${T}_prof_$c.log 2>&1 || { tail -30 $O/${T}_prof_$c.log; exit 1; }
done
python scripts/profile_round.py ${T}_mesh65k --script scripts/bench_mesh.py --script-args "--segments 256 --steps 3 --warmup 1 --plain" --kernel "odw_mesh_kernel" --rays 1e7 > $O/${T}_prof65k.log 2>&1 || { tail -30 $O/${T}_prof65k.log; exit 1; }
python scripts/profile_round.py ${T}_mesh1m --script scripts/bench_mesh.py --script-args "--segments 1024 --steps 3 --warmup 1 --plain" --kernel "odw_mesh_kernel" --rays 1e7 > $O/${T}_prof1m.log 2>&1 || { tail -30 $O/${T}_prof1m.log; exit 1; }
python scripts/bench_mesh.py --segments 0 64 256 1024 > $O/${T}_mesh.jsonl 2>$O/${T}_err.log
python scripts/bench_mesh.py --segments 256 1024 --sigma 0.12 >> $O/${T}_mesh.jsonl 2>$O/${T}_err.log
cp gpurun_out/${T}_*_pmc.json gpurun_out/${T}_*_kernel_stats.csv gpurun_out/${T}_*_pmc_current.json gpurun_out/${T}_*_bench.json $O/ 2>/dev/null || true
ls $O | grep ${T} | head -40
