"""N>1 path on CPU: two ranks (gloo) shard the global ray index range, trace
their shards (the CPU oracle stands in for the device here), sum histogram
and counters with the product's reduce helper; rank 0 must hold exactly the
single-process result (integer sums: independent of the rank count)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT, project

WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import numpy as np, torch, torch.distributed as dist
from conftest import project
from oracle import capi
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation import parallel
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
pr = project('lensesAndMirrors')
det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=64, ny=64, toward=pr.source.xform[[3, 7, 11]])
from oracle_tracer import OracleTracer
first, n = parallel.shardRange(1000, 30001, rank, world)
tr = OracleTracer(nthreads=2)
tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(det)
tr.trace(first, n, 77)
calls = []
reduce_, sync_ = dist.reduce, tr.sync
dist.reduce = lambda *a, **k: (calls.append('reduce'), reduce_(*a, **k))[1]
tr.sync = lambda *a, **k: (calls.append('sync'), sync_(*a, **k))[1]
parallel.reduceResults(tr, dist, torch)            # the product's helper: ONE collective for counters + histogram
# ... and the tracer's own stream is waited for BEFORE the collective reads the block (the launches run on a stream torch
# knows nothing about: without this wait the reduce could sum a block that is still being written)
assert calls == ['sync', 'reduce'], calls
if rank == 0:
  c = tr.counters()
  np.savez(sys.argv[2], hist=tr.histogram().astype(np.int64), cnt=np.array([c[k] for k in capi.CNT_NAMES]))
dist.barrier()
dist.destroy_process_group()
'''


def test_eight_rank_reduce_matches_single(tmp_path, oracle):
  """the rank arithmetic of a whole node (BASELINE configs[3] / [4] run on 8 GPUs), on CPU: eight gloo ranks, each its
  shard of the index range on the oracle, ONE reduce -- rank 0 holds the single-process histogram and counters"""
  from freecad.optics_design_workbench_amd import scenes
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  out = tmp_path / 'r0.npz'
  s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
  env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1')
  res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), str(script), ROOT, str(out)], env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stderr[-3000:]
  pr = project('lensesAndMirrors')
  det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=64, ny=64, toward=pr.source.xform[[3, 7, 11]])
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 1000, 30001, 77, det=det)
  got = np.load(out)
  assert np.array_equal(got['hist'], ref['hist'].astype(np.int64))
  assert [int(v) for v in got['cnt']] == [ref['counters'][k] for k in oracle.CNT_NAMES]
  # the sweep's deal over eight ranks: every value to exactly one rank, loads within one of each other
  from freecad.optics_design_workbench_amd.simulation import sweep
  shares = [sweep.shareOfRank(64, r, 8) for r in range(8)]
  assert sorted(k for sh in shares for k in sh) == list(range(64)) and all(len(sh) == 8 for sh in shares)
  shares = [sweep.shareOfRank(30, r, 8) for r in range(8)]
  assert sorted(k for sh in shares for k in sh) == list(range(30)) and max(map(len, shares)) - min(map(len, shares)) == 1


def test_shard_range_partitions_exactly():
  from freecad.optics_design_workbench_amd.simulation import parallel
  for n in (0, 1, 7, 1000, 10**9 + 3):
    for world in (1, 2, 3, 8):
      parts = [parallel.shardRange(5, n, r, world) for r in range(world)]
      assert sum(p[1] for p in parts) == n
      pos = 5
      for f, c in parts:
        assert f == pos
        pos += c
      assert max(p[1] for p in parts) - min(p[1] for p in parts) <= 1
  # weak-scaling step addressing never overlaps between ranks/steps
  seen = set()
  for s in range(3):
    for r in range(4):
      f = parallel.shardFirst(s, r, 4, 100)
      assert f not in seen
      seen.add(f)
  assert parallel.shardFirst(0, 0, 4, 100, warm=True) >= parallel.WARM_BASE


def test_two_rank_reduce_matches_single(tmp_path, oracle):
  from freecad.optics_design_workbench_amd import scenes
  script = tmp_path / 'worker.py'
  script.write_text(WORKER)
  out = tmp_path / 'rank0.npz'
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ, OMP_NUM_THREADS='2')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), ROOT, str(out)]
  res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout + res.stderr
  got = np.load(out)
  pr = project('lensesAndMirrors')
  det = scenes.planeDetector(pr.scene, 'OpticalAbsorberGroup', nx=64, ny=64, toward=pr.source.xform[[3, 7, 11]])
  ref = oracle.trace(pr.scene, pr.source, pr.limits, 1000, 30001, 77, det=det)
  assert np.array_equal(got['hist'].reshape(64, 64), ref['hist'].astype(np.int64))
  assert [int(v) for v in got['cnt']] == [ref['counters'][k] for k in oracle.CNT_NAMES]


RUN_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tests'))
import numpy as np, torch, torch.distributed as dist
from oracle_tracer import OracleTracer
from freecad.optics_design_workbench_amd.scene import open_fcstd
from freecad.optics_design_workbench_amd.simulation import runSimulation, resultsFolderPath
dist.init_process_group('gloo')
doc = open_fcstd(sys.argv[2])
st = doc.OpticalSimulationSettings
st.EndAfterRays, st.EndAfterHits = 'inf', '1500'
st.StoreHitInitPhi = True
store = runSimulation(doc, 'true', seed=42, resultsPath=resultsFolderPath(sys.argv[2]), raysPerLaunch=700,
                      tracer=OracleTracer(nthreads=1))              # torch.distributed is picked up (WORLD_SIZE=2)
assert store.totalRecordedHits > 1500, store.totalRecordedHits     # the job's total, on every rank
local = len(store.hits())
total = torch.tensor([local]); dist.all_reduce(total)
assert int(total) == store.totalRecordedHits and 0 < local < store.totalRecordedHits
st.EndAfterHits = 'inf'
fans = runSimulation(doc, 'fans', resultsPath=resultsFolderPath(sys.argv[2]), tracer=OracleTracer(nthreads=1), dist=dist)
assert fans.totalTracedRays == 40
dist.barrier()
dist.destroy_process_group()
'''


def test_two_rank_run_simulation(tmp_path, oracle):
  """runSimulation under a 2-rank launcher: launches are sharded by ray index,
  both ranks write into ONE run folder, the end criterion sees the job's totals;
  the merged result is the single-process result, row for row"""
  import shutil
  from conftest import SCENES
  from oracle_tracer import OracleTracer
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from freecad.optics_design_workbench_amd.simulation import rawFolders, resultsFolderPath, runSimulation
  path = str(tmp_path / 'GettingStarted.FCStd')
  shutil.copy(os.path.join(SCENES, 'GettingStarted.FCStd'), path)
  script = tmp_path / 'run_worker.py'
  script.write_text(RUN_WORKER)
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script), ROOT, path]
  res = subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS='1'), capture_output=True, text=True, timeout=600)
  assert res.returncode == 0, res.stdout + res.stderr
  folders = rawFolders(resultsFolderPath(path))
  assert len(folders) == 2                                    # one folder per run, shared by the ranks
  merged = folders[0].loadHits('*').hits
  files = [f for _, _, fs in os.walk(folders[0]._path) for f in fs if f.endswith('-hits.pkl')]
  assert len({f.split('-pid')[1].split('-')[0] for f in files}) == 2      # both processes wrote hit files
  doc = open_fcstd(path)
  st = doc.OpticalSimulationSettings
  st.EndAfterRays, st.EndAfterHits = 'inf', '1500'
  st.StoreHitInitPhi = True
  single = runSimulation(doc, 'true', seed=42, raysPerLaunch=700, tracer=OracleTracer()).hits().hits

  def rows(h):
    a = np.concatenate([h['points'], h['directions'], h['powers'][:, None], h['initPhi'][:, None]], axis=1)
    return a[np.lexsort(a.T[::-1])]

  assert len(merged['points']) == len(single['points']) > 1500
  assert np.array_equal(rows(merged), rows(single))
  fan_hits = folders[1].loadHits('*').hits
  assert len(fan_hits['points']) > 30
  assert folders[0].loadProgress()['totalRecordedHits'] == len(single['points'])
