#!/bin/bash
# round 4: counters and rates of the mesh kernel with the sorted hand-out order (and without it, for the record)
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp ODW_PROFILE_ROUND=r04
T=${1:-r04b}
O=gpurun_out
python scripts/profile_round.py ${T}_mesh65k --script scripts/bench_mesh.py --script-args "--segments 256 --steps 3 --warmup 1 --plain" --kernel "odw_mesh_kernel" --rays 1e7 > $O/${T}_prof65k.log 2>&1 || { tail -30 $O/${T}_prof65k.log; exit 1; }
python scripts/profile_round.py ${T}_mesh1m --script scripts/bench_mesh.py --script-args "--segments 1024 --steps 3 --warmup 1 --plain" --kernel "odw_mesh_kernel" --rays 1e7 > $O/${T}_prof1m.log 2>&1 || { tail -30 $O/${T}_prof1m.log; exit 1; }
python scripts/bench_mesh.py --segments 0 64 256 1024 > $O/${T}_mesh.jsonl 2>$O/${T}_err.log
python scripts/bench_mesh.py --segments 256 1024 --sigma 0.12 >> $O/${T}_mesh.jsonl 2>>$O/${T}_err.log
ODW_MESH_PRESORT=0 python scripts/bench_mesh.py --segments 64 256 1024 > $O/${T}_mesh_unsorted.jsonl 2>>$O/${T}_err.log
python scripts/bench_facet_scenes.py > $O/${T}_facet_scenes.log 2>&1 || true
rm -f $O/${T}_*_pmc?.log $O/${T}_*_trace.log
tail -3 $O/${T}_prof1m.log | cut -c1-400
cat $O/${T}_mesh.jsonl | cut -c1-200
tail -5 $O/${T}_facet_scenes.log
