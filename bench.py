#!/usr/bin/env python3
"""Headline benchmark: Monte-Carlo rays/second on benchmark/lensesAndMirrors.

  python bench.py --gpus N --steps K --warmup W [--config c3|c4|c5]

N > 1 without a launcher: this script starts `python -m torch.distributed.run
--nproc-per-node N bench.py ...` itself as a CHILD process -- before anything
here touches the GPU -- relays its output and exits with its code.  Under a
launcher (RANK / WORLD_SIZE in the environment) it is one rank of N, one rank
per GPU, RCCL.  Fewer than N devices, or a launcher whose world size is not N,
is an error: the line never reports another `n_gpus` than the one asked for.

Configs (BASELINE.json `configs`):
  c3 (default, the metric's config)  benchmark/lensesAndMirrors, 1e8 rays per step per GPU
  c4  benchmark/hugeArray, 1e9 rays per step sharded over 8 GPUs = 1.25e8 per step per GPU
      (weak scaling: the per-GPU share stays 1.25e8 for any N), one RCCL histogram reduce
  c5  examples/1-getting-started radius sweep: 64 radii x 1e7 rays per step, the radii dealt
      out over the ranks, spot size (the notebook's calcFwhm) per radius, table all-reduced

A step (c3, c4) = one pass of the hot path (generate -> trace to termination ->
record 64-B hit rows + detector histogram) over one batch of rays per GPU;
inputs (scene tables, CDF tables) are resident in HBM before the timed region.
Rays are addressed by a global Philox index, so every rank traces its own
disjoint index range (weak scaling, no data-path collective); one RCCL
sum-reduce of the detector histogram + counters to rank 0 closes the timed
region.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# RCCL / cross-process device memory on this pool needs dmabuf IPC (the image exports it already)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

SCENES = os.path.join(ROOT, 'tests', 'golden', 'scenes')
SEED = 0x0D15EA5E
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 f64 FMA lanes x 2 flop x 2.4 GHz
SIMDS, CLOCK_GHZ = 1024, 2.4     # MI355X_MICROARCH.md: 256 CUs x 4 SIMD-32, 2400 MHz max clock
RAY_STATE_BYTES = 64    # S, read + written once per segment
HIT_BYTES = 64          # H, one row per recorded hit

CONFIGS = {
    'c3': dict(scene='lensesAndMirrors', rays=1e8, steps=20, warmup=3, kernel='odw_trace_kernel<false, false, false, true>',
               workload='benchmark/lensesAndMirrors.FCStd, %.0e Monte-Carlo rays per step per GPU (BASELINE configs[2]), '
                        'Gaussian point source sigma=1e-2, Philox4x32-10 seed 0x0D15EA5E'),
    'c4': dict(scene='hugeArray', rays=1.25e8, steps=3, warmup=1, kernel='odw_grid_kernel<true, true>',
               workload='benchmark/hugeArray.FCStd (1500 spheres, grid kernel), %.3e Monte-Carlo rays per step per GPU = '
                        'the 1/8 share of 1e9 (BASELINE configs[3]), Gaussian point source sigma=0.2, one RCCL '
                        'histogram reduce'),
    'c5': dict(scene='GettingStarted', rays=1e7, steps=1, warmup=0, kernel='odw_trace_kernel<false, false, false, true>',
               workload='examples/1-getting-started/GettingStarted.FCStd, spherical-lens radius sweep: 64 radii '
                        'linspace(9, 11, 64) x %.0e Monte-Carlo rays each per step (BASELINE configs[4]), spot size = '
                        'calcFwhm of optimize-spotsize.ipynb per radius'),
}
N_RADII = 64


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
  big = proj.scene.n_prims > 64          # brute force over 1500 spheres: ~100x slower per ray
  chunk = threads * 4096 * (1 if big else 4)
  if not big:
    chunk = max(200_000, chunk)
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


# ODW_BENCH_REHEARSE=1: the N ranks of `--gpus N` all work on GPU 0 and talk through gloo -- the launcher, the shards,
# the reduce, the max-over-ranks clock and rank 0's checks run as on a node, on a box with one GPU.  The line says so
# ("rehearsal"); its value is not a measurement.
REHEARSE = os.environ.get('ODW_BENCH_REHEARSE') == '1'


def spawn_ranks(args, argv):
  """--gpus N > 1 without a launcher: start the N ranks as a CHILD process.  Counting the devices may
  initialise the HIP runtime in this process (torch.cuda.device_count() calls hipGetDeviceCount); that is
  harmless here because this process is never replaced by another program and runs no kernel: it starts
  the launcher as a child, waits for it and exits with its code.  The rendezvous port is the launcher's to
  pick (--standalone on 127.0.0.1: no bind-then-close race)."""
  import torch
  have = torch.cuda.device_count()
  if have < args.gpus and not REHEARSE:
    sys.stderr.write(f'bench.py: --gpus {args.gpus} asked for, {have} device(s) present\n')
    return 2
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
         f'--nproc-per-node={args.gpus}', os.path.abspath(__file__)] + argv
  return subprocess.call(cmd, env=dict(os.environ, MASTER_ADDR='127.0.0.1'))


def pmc_figures(cfg_name, n_per, record_hits, kernel, sources_sha256=None, path=None):
  """HBM traffic and VALU figures of the dominant kernel from the committed rocprofv3 PMC passes of
  this same command (profiles/pmc_current.json: FETCH_SIZE / WRITE_SIZE / SQ_* in separate --pmc
  runs, gfx950 correction applied).  PMC cannot be collected from inside the process, so the
  figures are attached only when the workload matches the profiled one -- and only when they were taken on THIS code:
  every entry carries the sha256 of the library's sources at the time of its pass (scripts/profile_round.py,
  _native.sources_hash); on a mismatch the entry comes back as {'pmc_stale': True, ...} and roofline_block() reports
  `pmc_stale` instead of a fraction computed from another kernel's instruction counts."""
  path = path or os.path.join(ROOT, 'profiles', 'pmc_current.json')
  if not os.path.exists(path):
    return None
  with open(path) as f:
    allcfg = json.load(f)
  tj = allcfg.get(cfg_name)
  if not tj or tj.get('rays_per_launch') != n_per or not record_hits:
    return None
  if tj.get('kernel') and not kernel.startswith(tj['kernel']):     # profiled on another kernel (e.g. --compile off)
    return None
  if sources_sha256 is None:
    from freecad.optics_design_workbench_amd import _native
    sources_sha256 = _native.sources_hash()
  if tj.get('sources_sha256') != sources_sha256:
    return dict(pmc_stale=True, source=tj.get('source'), profiled_sources_sha256=tj.get('sources_sha256'),
                sources_sha256=sources_sha256)
  return tj


def roofline_block(kernel_name, avg_kernel_s, rays_per_launch, bytes_per_ray, pmc, note=None):
  """The bound of these kernels is VECTOR-INSTRUCTION ISSUE, not HBM: rays live in registers from generation to
  termination, the only HBM traffic is the hit rows.  frac = wave64 VALU instructions per second (SQ_INSTS_VALU of
  the committed rocprofv3 pass of this same command / the kernel time measured live with HIP events) / the peak
  issue rate FOR THE KERNEL'S OWN INSTRUCTION MIX (per-class costs measured by scripts/valu_peak.hip at the
  kernels' occupancy, profiles/r03/valu_peak.json).  Beside it: float64 flop/s against the 78.6 TF vector peak,
  the counters' HBM bytes against 8 TB/s, and SURVEY 8(d)'s wavefront-formulation figure (what the same work would
  move if ray state lived in HBM between bounces) -- kept for comparability, never as `frac`."""
  alg = bytes_per_ray * rays_per_launch
  r = {'bound': 'valu_issue', 'achieved': None, 'peak': None, 'unit': 'G wave64 VALU instructions/s', 'frac': None,
       'traffic': None, 'kernel': kernel_name, 'avg_kernel_ms': avg_kernel_s * 1e3,
       'wavefront_equivalent_frac': alg / avg_kernel_s / 1e9 / HBM_PEAK_GBS,
       'wavefront_equivalent_note': 'SURVEY 8(d): (2 x 64 B ray state per segment + 64 B per hit row) x rays / kernel time / 8 TB/s; '
                                    'a register-resident megakernel never moves the ray state: not a bound',
       'algorithmic_bytes_per_ray': bytes_per_ray, 'algorithmic_bytes_per_launch': alg}
  if note:
    r['note'] = note
  if pmc and pmc.get('pmc_stale'):
    # the committed counter pass was taken on other sources than the library that ran: no fraction from it
    r['pmc_stale'] = True
    r['note'] = ((note + '; ') if note else '') + ('profiles/pmc_current.json was taken on other sources (sha256 %s...) than this build (%s...): '
                                                   'run scripts/profile_round.py again; frac not computed'
                                                   % (str(pmc.get('profiled_sources_sha256'))[:12], str(pmc.get('sources_sha256'))[:12]))
    return r
  v = pmc.get('valu') if pmc else None
  if pmc:
    r['traffic'] = pmc['hbm_bytes_per_launch'] / avg_kernel_s / 1e9            # GB/s, PMC bytes per launch / live kernel time
    r['traffic_bytes_per_launch'] = pmc['hbm_bytes_per_launch']
    r['hbm_counter_frac'] = r['traffic'] / HBM_PEAK_GBS
    r['traffic_source'] = pmc.get('source')
  if v and v.get('cyc_per_inst_calibrated'):
    r['achieved'] = v['insts_per_launch'] / avg_kernel_s / 1e9
    r['peak'] = SIMDS * CLOCK_GHZ / v['cyc_per_inst_calibrated']
    r['frac'] = r['achieved'] / r['peak']
    r['valu'] = {k: v[k] for k in ('insts_per_launch', 'salu_insts_per_launch', 'cyc_per_inst_calibrated', 'insts_by_class',
                                   'class_cycles', 'unclassified_split', 'active_lanes_per_inst', 'wait_any_frac',
                                   'lds_bank_conflict_frac', 'kernel_ms_profiled', 'source', 'definition') if k in v}
    r['valu']['calibration'] = 'profiles/r03/valu_peak.json'
    if v.get('fp64_flop_per_launch'):
      r['fp64_flops_frac'] = v['fp64_flop_per_launch'] / avg_kernel_s / 1e12 / FP64_VECTOR_PEAK_TFLOPS
  else:
    r['note'] = ((note + '; ') if note else '') + 'no committed counter pass for this workload: instruction counts unknown, frac not computed'
  return r


def summary_scalars(name, line):
  """what one nested config line says, as flat numbers: value, ms per step, the roofline fraction of its kernel,
  lanes per vector instruction (grid kernel), the sweep's answer"""
  if 'error' in line or 'value' not in line:
    return {f'{name}_error': line.get('error', 'no line')}
  rf = line.get('roofline') or {}
  out = {f'{name}_value': line['value'], f'{name}_unit': line['unit'], f'{name}_ms_per_step': line['ms_per_step'],
         f'{name}_steps': line['steps'], f'{name}_roofline_frac': rf.get('frac'), f'{name}_avg_kernel_ms': rf.get('avg_kernel_ms'),
         f'{name}_kernel': rf.get('kernel')}
  if rf.get('frac_kernel_alone') is not None:
    out[f'{name}_roofline_frac_kernel_alone'] = rf['frac_kernel_alone']
    out[f'{name}_kernel_alone_ms'] = rf.get('kernel_alone_ms')
  lanes = (rf.get('valu') or {}).get('active_lanes_per_inst')
  if lanes is not None:
    out[f'{name}_active_lanes_per_inst'] = lanes
  spot = (line.get('config') or {}).get('spot_size') or {}
  for k in ('best_radius_mm', 'best_fwhm_mm', 'best_radius_by_rms_mm', 'best_radius_by_fwhm_1e3_mm', 'fwhm_1e3_at_best_mm'):
    if k in spot:
      out[f'{name}_{k}'] = spot[k]
  return out


def compact(line, nested=False):
  """the stdout line: every number of the record, without the long explanatory blocks (they are in the detail
  record): per-class instruction tables, notes, per-radius tables of nested lines"""
  import copy
  out = copy.deepcopy(line)
  rf = out.get('roofline')
  if isinstance(rf, dict):
    for k in ('wavefront_equivalent_note', 'traffic_source'):
      rf.pop(k, None)
    v = rf.get('valu')
    if isinstance(v, dict):
      rf['valu'] = {k: v[k] for k in ('insts_per_launch', 'cyc_per_inst_calibrated', 'active_lanes_per_inst', 'wait_any_frac',
                                      'lds_bank_conflict_frac', 'kernel_ms_profiled', 'source') if k in v}
  if isinstance(out.get('end_to_end'), dict):
    out['end_to_end'].pop('note', None)
  cfg = out.get('config')
  if isinstance(cfg, dict):
    if isinstance(cfg.get('scene_compiled'), dict):
      cfg['scene_compiled'].pop('note', None)
    spot = cfg.get('spot_size')
    if isinstance(spot, dict):
      for k in ('fwhm_note', 'fwhm_1e3_note'):
        spot.pop(k, None)
      if nested:
        for k in ('radii', 'fwhm_mm', 'rms_spot_mm', 'fwhm_1e3_mm'):
          spot.pop(k, None)
  if isinstance(out.get('extra_configs'), dict):
    out['extra_configs'] = {k: compact(v, nested=True) for k, v in out['extra_configs'].items()}
  return out


def detector_of(cfg_name, proj):
  """the 1024 x 1024 detector window a config's launches bin into"""
  from freecad.optics_design_workbench_amd import scenes
  if cfg_name == 'c4':
    # the absorbers are spheres (no planar face): their hits are binned in projection along the
    # array's z axis, window = the footprint of the 10 x 10 x 5 array (pitch 5 mm) + margin
    gi = proj.scene.group_index('OpticalAbsorberGroup')
    return dict(group=gi, origin=[-0.5, -0.5, 61.0], ex=[1.0, 0.0, 0.0], ey=[0.0, 1.0, 0.0],
                x_lo=-25.0, x_hi=25.0, y_lo=-25.0, y_hi=25.0, nx=1024, ny=1024)
  return scenes.planeDetector(proj.scene, 'OpticalAbsorberGroup', nx=1024, ny=1024, toward=proj.source.xform[[3, 7, 11]])


def run_trace_config(args, cfg_name, cfg, rank, local_rank, world, dist, torch):
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.simulation import parallel
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

  n_per = int(args.rays_per_step if args.rays_per_step else cfg['rays'])
  proj = scenes.bakeProject(os.path.join(SCENES, cfg['scene'] + '.FCStd'))
  det = detector_of(cfg_name, proj)
  tr = Tracer(local_rank)
  tr.setScene(proj.scene)
  tr.setSource(proj.source)
  tr.setLimits(proj.limits)
  tr.setDetector(det)
  # scene-compiled kernel (odw_compile_scene): part of the scene's preparation, like its upload -- done
  # before the clock starts; hugeArray (grid kernel) is outside its domain and keeps the generic kernel
  try:
    compiled = tr.compileScene(args.compile)
  except Exception as e:            # no hiprtc / compiler trouble: the generic kernels run, the line says so
    compiled = dict(mode=0, seconds=0.0, cache=0, error=str(e)[:300])
  record_hits = not args.no_hits
  if record_hits:
    # c3: <= 1 recorded hit per ray; c4: 0.16 per ray (3 absorber layers of 15); reused every step
    tr.reserveHits(n_per + 1024 if cfg_name != 'c4' else n_per // 2)

  def step(s):
    # hit rows of one step are the step's output; the buffer is recycled
    tr.reset() if s < 0 else tr.resetHits()
    first = parallel.shardFirst(s if s >= 0 else -s - 1, rank, world, n_per, warm=s < 0)
    tr.trace(first, n_per, SEED, record_hits=record_hits, histogram=not args.no_histogram)

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
    t = torch.tensor([dt], dtype=torch.float64, device='cpu' if REHEARSE else 'cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())

  out = None
  if rank == 0:
    cnt = tr.counters()       # after the reduce: whole job on rank 0
    total_rays = n_per * args.steps * world
    assert cnt['traced_rays'] == total_rays, (cnt, total_rays)
    assert cnt['hits_dropped'] == 0, cnt
    hist = tr.histogram()
    hist_total = int(hist.sum())
    if args.dump_results:      # (tests: the reduced histogram and counters of the whole job, to compare with the shards' sum)
      import numpy as np
      np.savez(args.dump_results, hist=hist, counters=np.array([cnt[k] for k in sorted(cnt)], dtype=np.int64),
               names=np.array(sorted(cnt)))
    assert args.no_histogram or hist_total + cnt['hist_overflow'] == cnt['recorded_hits'], (hist_total, cnt)
    kbar = cnt['segments'] / cnt['traced_rays']
    hbar = cnt['recorded_hits'] / cnt['traced_rays']
    bytes_per_ray = kbar * 2 * RAY_STATE_BYTES + hbar * HIT_BYTES
    avg_kernel_s = kernel_ms / 1e3 / max(1, launches)
    kernel_name = 'odw_spec_kernel' if compiled['mode'] else cfg['kernel']
    pmc = pmc_figures(cfg_name, n_per, record_hits, kernel_name)
    out = {
        'metric': 'Monte-Carlo rays/sec (whole node), lensesAndMirrors.FCStd' if cfg_name == 'c3'
                  else f'Monte-Carlo rays/sec (whole node), {cfg["scene"]}.FCStd',
        'value': total_rays / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': cfg['workload'] % n_per, 'name': cfg_name,
                   'rays_per_step_per_gpu': n_per, 'segments_per_ray': kbar, 'hits_per_ray': hbar,
                   'record_hit_rows': record_hits, 'histogram': '1024x1024 u64',
                   'scene_compiled': {'mode': {0: 'off', 1: 'structure'}[compiled['mode']],
                                      'compile_seconds': compiled['seconds'], 'cache': compiled['cache'],
                                      'note': 'hiprtc compile of the ray loop against the scene, before the timed region',
                                      **({'error': compiled['error']} if 'error' in compiled else {})},
                   'parallelism': f'ray-index sharding x{world}, one RCCL reduce'},
        'roofline': roofline_block(kernel_name, avg_kernel_s, n_per, bytes_per_ray, pmc),
    }
    if world == 1 and record_hits and cfg_name == 'c3' and not args.no_end_to_end:
      out['end_to_end'] = end_to_end(tr, n_per, max(2, args.steps))
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(proj, det)
  tr.close()
  return out


def end_to_end(tr, n_per, steps):
  """PCIe-inclusive rate, reported beside `value` (never as it): every step's hit rows cross to
  host memory, the copy of step k overlapping the trace of step k+1 (two row buffers, a copy
  stream of its own: `Tracer.traceStreaming`)"""
  if not hasattr(tr, 'traceStreaming'):
    return None
  from freecad.optics_design_workbench_amd.simulation import parallel
  # host buffers and their first touch are not part of the rate: allocated before the clock starts,
  # one untimed pass over both
  buffers = [tr.hostRows(n_per + 1024) for _ in range(2)]
  for chunk in tr.traceStreaming(((parallel.shardFirst(s, 0, 1, n_per, warm=True), n_per) for s in range(2)), SEED,
                                 capacity=n_per + 1024, buffers=buffers):
    pass
  tr.reset()
  tr.sync()
  t0 = time.perf_counter()
  rows = 0
  for chunk in tr.traceStreaming(((parallel.shardFirst(s, 0, 1, n_per), n_per) for s in range(steps)), SEED,
                                 capacity=n_per + 1024, buffers=buffers):
    rows += len(chunk)
  dt = time.perf_counter() - t0
  return dict(rays_per_s=n_per * steps / dt, rows=rows, seconds=dt, gb_per_s=rows * 64 / dt / 1e9,
              note='hit rows of every step copied to page-locked host memory (append order), copy of step k '
                   'overlapped with the trace of step k+1')


def run_sweep_config(args, cfg, rank, local_rank, world, dist, torch):
  import numpy as np
  from freecad.optics_design_workbench_amd import scenes
  from freecad.optics_design_workbench_amd.scene import open_fcstd
  from freecad.optics_design_workbench_amd.simulation import sweep
  from freecad.optics_design_workbench_amd.simulation.tracer import Tracer

  n_per = int(args.rays_per_step if args.rays_per_step else cfg['rays'])
  radii = np.linspace(9, 11, args.radii)
  doc = open_fcstd(os.path.join(SCENES, cfg['scene'] + '.FCStd'))

  def setRadius(d, r):
    d.Sphere.Radius = float(r)
  tr = Tracer(local_rank)
  # one compiled kernel serves the whole sweep (every radius has the same structure); it is built here, before
  # the clock starts, from the scene of the first radius (sticky mode: the other radii find it in the cache)
  setRadius(doc, radii[0])
  first = scenes.bakeProject(doc)
  tr.setScene(first.scene)
  tr.setLimits(first.limits)
  try:
    compiled = tr.compileScene(args.compile)
  except Exception as e:               # no hiprtc / compiler trouble: the generic kernels run, the line says so
    compiled = dict(mode=0, seconds=0.0, cache=0, error=str(e)[:300])

  def barrier():
    tr.sync()
    torch.cuda.synchronize()
    if dist is not None:
      dist.barrier()

  def run():
    # keepSample: every radius also hands back its rows [::n // 1000] -- the sample size the notebook itself works on
    # (optimize-spotsize.ipynb cell 9: EndAfterRays = 1e3); gathered inside the timed sweep, evaluated after it
    return sweep.parameterSweep(doc, setRadius, radii, rays=n_per, seed=SEED, tracer=tr, dist=dist,
                                measure=dict(fwhm=sweep.calcFwhm, rms=sweep.rmsSpot), keepSample=1000)

  for _ in range(args.warmup):
    run()
  barrier()

  def contexts():       # the tracer and the contexts the sweep keeps beside it (groups of radii take turns on them)
    return [tr] + list(getattr(tr, '_sweepLanes', None) or [])
  for t in contexts():
    t.timingEnable(True)
    t.timingRead()
  barrier()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    res = run()
  barrier()
  dt = time.perf_counter() - t0
  # trace kernels of this rank, all contexts: a launch traces a GROUP of radii (batch launches); per radius = per 1e7 rays
  kernel_ms, launches = 0.0, 0
  for t in contexts():
    ms, n = t.timingRead()
    kernel_ms += ms
    launches += n
  radii_of_rank = len(sweep.shareOfRank(len(radii), rank, world)) * args.steps
  if dist is not None:
    t = torch.tensor([dt], dtype=torch.float64, device='cpu' if REHEARSE else 'cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
  # the batch launch by itself (after the timed region, nothing else on the GPU): one launch of up to 16 radii, three times,
  # the last two timed -- inside a sweep the kernel shares the GPU with the post-hoc chains of other groups, which stretches
  # its own time; `frac_kernel_alone` is the same instruction count against this time
  solo_ms = None
  if rank == 0:
    try:
      ks = radii[:min(16, len(radii))]
      prs = []
      for r in ks:
        setRadius(doc, r)
        prs.append(scenes.bakeProject(doc))
      tr.setLimits(prs[0].limits); tr.setSource(prs[0].source); tr.setSceneBatch([p.scene for p in prs]); tr.setDetector(None)
      for rep in range(3):
        tr.reset()
        tr.traceBatch(0, n_per, SEED, int(n_per * 1.25) + 1024)
        tr.sync()
        if rep == 0:
          tr.timingRead()
      ms, n = tr.timingRead()
      solo_ms = ms / max(n, 1) / len(ks)
    except Exception as e:
      solo_ms = None
      print(f'[bench] the solo batch launch failed: {e}', file=sys.stderr)
  # the notebook's figure of merit on the notebook's sample size: calcFwhm of the ~1000 thinned rows per radius the
  # last sweep kept (host arithmetic of 64 small clouds, after the timed region; one more all-reduce of the column)
  fwhm_1e3 = sweep.fwhmOfSamples(res, dist=dist, device=local_rank)
  out = None
  if rank == 0:
    total_rays = n_per * len(radii) * args.steps
    assert res.tracedRays == n_per * len(radii), (res.tracedRays, n_per, len(radii))
    kbar = res.segments / res.tracedRays
    hbar = res.recordedHits / res.tracedRays
    bytes_per_ray = kbar * 2 * RAY_STATE_BYTES + hbar * HIT_BYTES
    avg_kernel_s = kernel_ms / 1e3 / max(1, radii_of_rank)        # per radius (a launch holds several)
    info = tr.compiledInfo()             # the kernel the last radius ran
    kernel_name = 'odw_spec_kernel' if info['mode'] == 1 else cfg['kernel']
    pmc = pmc_figures('c5', n_per, True, kernel_name)
    best_r, best_f = res.best('fwhm') if np.isfinite(res.columns['fwhm']).any() else (float('nan'), float('nan'))
    rms_r, rms_v = res.best('rms')
    out = {
        'metric': 'Monte-Carlo rays/sec (whole node), GettingStarted.FCStd radius sweep',
        'value': total_rays / dt, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': cfg['workload'] % n_per, 'name': 'c5', 'radii': len(radii),
                   'rays_per_radius': n_per, 'segments_per_ray': kbar, 'hits_per_ray': hbar,
                   'scene_compiled': {'mode': {0: 'off', 1: 'structure', 2: 'auto'}[info['mode']],
                                      'compile_seconds': compiled.get('seconds', 0.0), 'cache': compiled.get('cache', 0),
                                      'note': 'one kernel for the whole sweep (every radius has the same structure), compiled before the timed region',
                                      **({'error': compiled['error']} if 'error' in compiled else {})},
                   'parallelism': f'radii dealt out over {world} rank(s), one RCCL all-reduce of the table',
                   'spot_size': {'kind': 'calcFwhm (optimize-spotsize.ipynb cell 8)', 'radii': radii.tolist(),
                                 'fwhm_mm': [None if np.isnan(v) else float(v) for v in res.columns['fwhm']],
                                 'best_radius_mm': best_r, 'best_fwhm_mm': best_f,
                                 'fwhm_note': 'null = the notebook\'s fit finds no half-maximum (its except clause '
                                              'skips such azimuth bins): with 1e7 hits the innermost radial bins are '
                                              'flat near the focus; the rms spot radius below stays defined',
                                 'rms_spot_mm': [float(v) for v in res.columns['rms']],
                                 'best_radius_by_rms_mm': rms_r, 'best_rms_spot_mm': rms_v,
                                 'fwhm_1e3_mm': [None if np.isnan(v) else float(v) for v in fwhm_1e3],
                                 'best_radius_by_fwhm_1e3_mm': float(radii[int(np.nanargmin(fwhm_1e3))]) if np.isfinite(fwhm_1e3).any() else None,
                                 'fwhm_1e3_at_best_mm': float(np.nanmin(fwhm_1e3)) if np.isfinite(fwhm_1e3).any() else None,
                                 'fwhm_1e3_note': 'calcFwhm on points[::n // 1000] of every radius (the ~1e3 hits per run the notebook '
                                                  'itself traces, cell 9): the estimator in its own regime; the rows are gathered '
                                                  'inside the timed sweep, the 64 small fits run after it'}},
        'roofline': roofline_block(kernel_name, avg_kernel_s, n_per, bytes_per_ray, pmc,
                                   note=f'trace kernels of rank 0: {launches} batch launches for {radii_of_rank} radii, kernel time per radius '
                                        '(1e7 rays), the ray pass of the launch included, measured while the post-hoc kernels of other groups share the GPU (the batch '
                                        'kernel by itself: kernel_alone_ms / frac_kernel_alone); per step the host '
                                        'also re-bakes the scenes and searches the detector plane per radius'),
    }
    if solo_ms:
      rf = out['roofline']
      rf['kernel_alone_ms'] = solo_ms
      rf['kernel_alone_note'] = 'one batch launch of up to 16 radii with nothing else on the GPU, per radius, after the timed region'
      if rf.get('frac') is not None:
        rf['frac_kernel_alone'] = rf['frac'] * rf['avg_kernel_ms'] / solo_ms
    if world == 1 and not args.no_cpu_baseline:
      proj = scenes.bakeProject(doc)
      out['cpu_baseline'] = cpu_baseline(proj, None, seconds=8.0)
  tr.close()
  return out


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=None)
  ap.add_argument('--warmup', type=int, default=None)
  ap.add_argument('--config', choices=sorted(CONFIGS), default='c3')
  ap.add_argument('--rays-per-step', type=float, default=None,
                  help='rays per step per GPU (c5: per radius); default: the BASELINE size of the config')
  ap.add_argument('--radii', type=int, default=N_RADII, help='c5: number of radii of the sweep')
  ap.add_argument('--compile', choices=['off', 'structure'], default='structure',
                  help='scene-compiled kernels (odw_compile_scene); scenes outside their domain (c4) run the generic ones')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-end-to-end', action='store_true')
  ap.add_argument('--no-hits', action='store_true', help='histogram only (diagnostic, not the metric)')
  ap.add_argument('--no-histogram', action='store_true', help='hit rows only (diagnostic, not the metric)')
  ap.add_argument('--dump-results', default=None, help='c3 / c4: rank 0 writes the job\'s histogram and counters (after the reduce) to this .npz')
  ap.add_argument('--no-extra', action='store_true',
                  help='c3 only: leave out the c4 (3 steps) and c5 (4 sweeps) lines nested under "extra_configs"')
  args = ap.parse_args()
  cfg = CONFIGS[args.config]
  if args.steps is None:
    args.steps = cfg['steps']
  if args.warmup is None:
    args.warmup = cfg['warmup']
  if args.gpus < 1:
    ap.error('--gpus must be >= 1')

  launched = 'RANK' in os.environ
  if args.gpus > 1 and not launched:
    sys.exit(spawn_ranks(args, sys.argv[1:]))
  rank = int(os.environ.get('RANK', 0))
  local_rank = int(os.environ.get('LOCAL_RANK', 0))
  world = int(os.environ.get('WORLD_SIZE', 1))
  if world != args.gpus:
    sys.stderr.write(f'bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)\n')
    sys.exit(2)

  import torch
  if torch.cuda.device_count() < world and not REHEARSE:
    sys.stderr.write(f'bench.py: {world} rank(s) but {torch.cuda.device_count()} device(s) present\n')
    sys.exit(2)
  dist = None
  if REHEARSE:
    local_rank = 0                       # (from here on: the device a rank works on)
    os.environ.setdefault('ODW_RANKS_PER_DEVICE', str(world))      # (the ranks share GPU 0's memory: simulation/sweep.py)
  if launched:   # launched by torch.distributed.run (also with one rank)
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if REHEARSE:
      dist.init_process_group('gloo')
    else:
      dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

  if args.config == 'c5':
    out = run_sweep_config(args, cfg, rank, local_rank, world, dist, torch)
  else:
    out = run_trace_config(args, args.config, cfg, rank, local_rank, world, dist, torch)
  if args.config == 'c3' and world == 1 and not args.no_extra and not args.no_hits and not args.rays_per_step:
    # the other two GPU configs of BASELINE.json in the same driver-timed record: c4 (hugeArray, 3 steps) and
    # c5 (the radius sweep, 4 sweeps after 2 untimed ones: the first sweep after the contexts are set up runs ~10 % slower than the
    # steady state, and single sweeps vary box to box), each measured exactly like its own `--config` line
    # (one GPU only: a scaling run needs the headline per N, and an extra that fails on one rank would take the others' line with it)
    import copy
    extra = {}
    for name, steps, warmup in (('c4', 3, 1), ('c5', 4, 2)):
      sub = copy.copy(args)
      sub.config, sub.steps, sub.warmup = name, steps, warmup
      sub.no_cpu_baseline = sub.no_end_to_end = True
      try:
        if name == 'c5':
          line = run_sweep_config(sub, CONFIGS[name], rank, local_rank, world, dist, torch)
        else:
          line = run_trace_config(sub, name, CONFIGS[name], rank, local_rank, world, dist, torch)
      except Exception as e:            # (a failing extra must not take the headline line with it; it is reported)
        line = {'error': f'{type(e).__name__}: {e}'[:500]}
      if rank == 0:
        if name == 'c5' and 'config' in line:          # (the per-radius table belongs to the c5 line of its own)
          line['config'].get('spot_size', {}).pop('radii', None)
        extra[name] = line
    if rank == 0:
      out['extra_configs'] = extra
      # the same figures as plain numbers where a reader that keeps only the scalars of `config` still finds them
      for name, line in extra.items():
        out['config'].update(summary_scalars(name, line))
  if rank == 0 and REHEARSE:
    out['rehearsal'] = f'{world} rank(s) on GPU 0 over gloo (ODW_BENCH_REHEARSE=1): the flow of a node, not a measurement'
  if rank == 0:
    detail = json.dumps(out)
    # the full record (instruction mixes, per-radius tables) goes to stderr and, where the folder exists, to
    # gpurun_out/bench_detail.json; stdout carries ONE compact line
    sys.stderr.write('[bench detail] ' + detail + '\n')
    try:
      if os.path.isdir(os.path.join(ROOT, 'gpurun_out')):
        with open(os.path.join(ROOT, 'gpurun_out', 'bench_detail.json'), 'w') as f:
          f.write(detail + '\n')
    except OSError:
      pass
    print(json.dumps(compact(out)), flush=True)
  if dist is not None:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
