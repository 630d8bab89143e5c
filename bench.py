#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo rays/second on benchmark/lensesAndMirrors.

  python bench.py --gpus N --steps K --warmup W
(N>1: launched by torch.distributed.run, one rank per GPU, RCCL.)

A step = one pass of the hot path (generate -> trace to termination -> record
64-B hit rows + detector histogram) over one batch of RAYS_PER_STEP rays per
GPU; inputs (scene tables, CDF tables) are resident in HBM before the timed
region.  Rays are addressed by a global Philox index, so every rank traces its
own disjoint index range (weak scaling, no data-path collective); one RCCL
sum-reduce of the detector histogram + counters to rank 0 closes the timed
region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / cross-process device memory on this pool needs dmabuf IPC (the image exports it already)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

SCENE = os.path.join(ROOT, 'tests', 'golden', 'scenes', 'lensesAndMirrors.FCStd')
SEED = 0x0D15EA5E
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
RAY_STATE_BYTES = 64    # S, read + written once per segment
HIT_BYTES = 64          # H, one row per recorded hit


def host_cores():
  """the cores this process may really use: its affinity mask, cut by the cgroup's CPU quota
  (a GPU box hands a 16-CPU share of a 128-thread host to a one-GPU job)"""
  n = len(os.sched_getaffinity(0))
  for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
    try:
      with open(path) as f:
        words = f.read().split()
      if path.endswith('cpu.max'):
        quota, period = words[0], float(words[1])
      else:
        quota = words[0]
        with open('/sys/fs/cgroup/cpu/cpu.cfs_period_us') as f:
          period = float(f.read())
      if quota not in ('max', '-1'):
        n = min(n, max(1, int(float(quota) / period + 0.5)))
      break
    except (OSError, ValueError, IndexError):
      continue
  return n


def cpu_baseline(proj, det, seconds=12.0):
  """the CPU oracle (a port: the reference's own CPU path needs FreeCAD/OCC,
  absent here) on all host cores, bounded sample of the same workload"""
  from oracle import capi
  threads = min(capi.threads(), host_cores())
  # the oracle hands out 4096 rays at a time to its threads: a call has to hold several such
  # pieces per thread to keep all cores busy; one untimed call first (thread pool start-up,
  # first touch of the 80 MB table)
  chunk = max(200_000, threads * 4096 * 4)
  capi.trace(proj.scene, proj.source, proj.limits, 0, threads * 4096, SEED, det=det, nthreads=threads,
             hit_capacity=threads * 4096 + 16)
  done, t0 = 0, time.perf_counter()
  while True:
    capi.trace(proj.scene, proj.source, proj.limits, done, chunk, SEED, det=det, nthreads=threads,
               hit_capacity=chunk + 16)
    done += chunk
    dt = time.perf_counter() - t0
    if dt >= seconds:
      break
  return dict(value=done / dt, unit='rays/s', cores=threads, kind='port',
              sample=f'{done} rays of the same workload (oracle/odw_oracle.c, OpenMP, {dt:.1f} s)')


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=5)
  ap.add_argument('--warmup', type=int, default=1)
  ap.add_argument('--rays-per-step', type=float, default=1e8)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-hits', action='store_true', help='histogram only (diagnostic, not the metric)')
  args = ap.parse_args()

  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  n_per = int(args.rays_per_step)

  import torch
  dist = None
  if world > 1 or 'RANK' in os.environ:   # launched by torch.distributed.run (also with one rank)
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.simulation import parallel
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

  proj = scenes.bakeProject(SCENE)
  det = scenes.planeDetector(proj.scene, 'OpticalAbsorberGroup', nx=1024, ny=1024,
                             toward=proj.source.xform[[3, 7, 11]])
  tr = Tracer(local_rank)
  tr.setScene(proj.scene)
  tr.setSource(proj.source)
  tr.setLimits(proj.limits)
  tr.setDetector(det)
  record_hits = not args.no_hits
  if record_hits:
    tr.reserveHits(n_per + 1024)   # <= 1 recorded hit per ray in this scene; reused every step

  def step(s):
    # hit rows of one step are the step's output; the buffer is recycled
    tr.reset() if s < 0 else tr.resetHits()
    first = parallel.shardFirst(s if s >= 0 else -s - 1, rank, world, n_per, warm=s < 0)
    tr.trace(first, n_per, SEED, record_hits=record_hits, histogram=True)

  def barrier():
    tr.sync()
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()

  for w in range(args.warmup):
    step(-1 - w)
  barrier()
  tr.reset()
  tr.timingEnable(True)
  tr.timingRead()
  barrier()
  t0 = time.perf_counter()
  for s in range(args.steps):
    step(s)
  tr.sync()
  if dist is not None:
    parallel.reduceResults(tr, dist, torch)
  barrier()
  dt = time.perf_counter() - t0
  kernel_ms, launches = tr.timingRead()

  if dist is not None:
    t = torch.tensor([dt], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

  if rank == 0:
    cnt = tr.counters()       # after the reduce: whole job on rank 0
    total_rays = n_per * args.steps * world
    assert cnt['traced_rays'] == total_rays, (cnt, total_rays)
    hist_total = int(tr.histogram().sum())
    assert hist_total + cnt['hist_overflow'] == cnt['recorded_hits'], (hist_total, cnt)
    kbar = cnt['segments'] / cnt['traced_rays']
    hbar = cnt['recorded_hits'] / cnt['traced_rays']
    bytes_per_ray = kbar * 2 * RAY_STATE_BYTES + hbar * HIT_BYTES
    avg_kernel_s = kernel_ms / 1e3 / max(1, launches)
    achieved = bytes_per_ray * n_per / avg_kernel_s / 1e9
    # HBM traffic of one launch from the committed rocprofv3 PMC passes of this
    # same command (FETCH_SIZE / WRITE_SIZE, separate --pmc runs, gfx950
    # correction applied; see profiles/traffic.json).  PMC cannot be collected
    # from inside the process, so the figure is only attached when the
    # workload matches the profiled one.
    traffic = traffic_src = traffic_bytes = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
      tj = json.load(open(tpath))
      if tj.get('rays_per_launch') == n_per and record_hits:
        traffic_bytes = tj['hbm_bytes_per_launch']
        traffic = traffic_bytes / avg_kernel_s / 1e9       # same unit as `achieved`
        traffic_src = tj.get('source')
    out = {
        'metric': 'Monte-Carlo rays/sec (whole node), lensesAndMirrors.FCStd',
        'value': total_rays / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': 'benchmark/lensesAndMirrors.FCStd, %.0e Monte-Carlo rays per step per GPU '
                               '(BASELINE configs[2]), Gaussian point source sigma=1e-2, Philox4x32-10 seed 0x0D15EA5E'
                               % n_per,
                   'rays_per_step_per_gpu': n_per, 'segments_per_ray': kbar, 'hits_per_ray': hbar,
                   'record_hit_rows': record_hits, 'histogram': '1024x1024 u64',
                   'parallelism': f'ray-index sharding x{world}, one RCCL reduce'},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_bytes_per_launch': traffic_bytes,
                     'algorithmic_bytes_per_launch': bytes_per_ray * n_per, 'traffic_source': traffic_src,
                     'kernel': 'odw_trace_kernel', 'avg_kernel_ms': avg_kernel_s * 1e3,
                     'algorithmic_bytes_per_ray': bytes_per_ray},
    }
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(proj, det)
    print(json.dumps(out), flush=True)
  tr.close()
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
