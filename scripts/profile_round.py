#!/usr/bin/env python3
"""Collect the round's profile evidence on the GPU box (run from the repo root):

  python scripts/profile_round.py r02_c3 [--config c3|c4] [--no-bench]

1. `python bench.py --config C`                                      -> gpurun_out/<tag>_bench.json
2. `rocprofv3 --kernel-trace --stats -- python3 bench.py --config C --steps 16 --warmup 4` -> <tag>_kernel_stats.csv
3. `rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py ...`    -> <tag>_pmc.json, one run per counter set
   (FETCH_SIZE and WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes; no trace domains besides
   --kernel-trace next to --pmc)
rocprofv3 writes SQLite databases; the summaries are read from their views.  Also writes
gpurun_out/<tag>_pmc_current.json = the entry of profiles/pmc_current.json for this config
(HBM bytes per launch with the gfx950 correction, VALU figures).
"""
import argparse
import glob
import shutil
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('tag')
ap.add_argument('--config', default='c3')
ap.add_argument('--no-bench', action='store_true')
ap.add_argument('--kernel', default=None, help='substring of the kernel name the counters are taken from')
ap.add_argument('--script', default=None, help='profile `python3 <script> <script-args>` instead of bench.py (needs --kernel)')
ap.add_argument('--script-args', default='')
ap.add_argument('--rays', type=float, default=None, help='with --script: rays per launch')
args = ap.parse_args()
tag = args.tag
if args.script:
  args.no_bench = True
KERNEL_NAME = args.kernel or {'c3': 'odw_spec_kernel', 'c4': 'odw_grid_kernel', 'c5': 'odw_spec_kernel'}[args.config]
KERNEL_LIKE = '%' + KERNEL_NAME + '%'
out = os.path.join(ROOT, 'gpurun_out')
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp')
# c5: the sweep as it runs -- batch launches of 2 - 16 radii each, their rays generated once per launch by
# odw_batch_rays_kernel.  Counters and kernel time are SUMMED over every dispatch of the two kernels in the process
# and divided by the radii the process traced (64 per sweep x (warmup + steps) sweeps): per radius = per 1e7 rays,
# what bench.py compares with.
C5_RADII, C5_SWEEPS = 64, 2
KERNELS = [KERNEL_NAME] + (['odw_batch_rays_kernel'] if args.config == 'c5' and not args.script else [])
BENCH = ['python3', os.path.join(ROOT, 'bench.py'), '--config', args.config, '--steps', '1' if args.config == 'c5' else '3', '--warmup', '1',
         '--no-cpu-baseline', '--no-end-to-end', '--no-extra']
if args.script:
  import shlex
  BENCH = ['python3', os.path.join(ROOT, args.script)] + shlex.split(args.script_args)


def run(cmd, log):
  print('+', ' '.join(cmd), flush=True)
  with open(os.path.join(out, log), 'w') as f:
    subprocess.run(cmd, cwd=ROOT, env=env, stdout=f, stderr=subprocess.STDOUT, check=True)


def db_of(folder):
  files = glob.glob(os.path.join(folder, '**', '*.db'), recursive=True)
  if not files:
    raise SystemExit(f'no rocprofv3 database under {folder}')
  return sqlite3.connect(sorted(files)[-1])


def tables(con):
  return [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]


# 1. the bench line
if not args.no_bench:
  res = subprocess.run(['python3', os.path.join(ROOT, 'bench.py'), '--config', args.config, '--no-extra'], cwd=ROOT, env=env,
                       capture_output=True, text=True, check=True)
  line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
  open(os.path.join(out, f'{tag}_bench.json'), 'w').write(line + '\n')
  print(line[:200], flush=True)

# 2. kernel trace
d = os.path.join(out, f'{tag}_trace')
# (more launches than the counter passes: the first two or three launches of a process run 5 - 15 %
#  slower -- clocks, first touch -- and the average should be the steady state bench.py reports)
TRACE_BENCH = [x for x in BENCH]
if args.config != 'c5' and not args.script:
  TRACE_BENCH[TRACE_BENCH.index('--steps') + 1] = '16'
  TRACE_BENCH[TRACE_BENCH.index('--warmup') + 1] = '4'
run(['rocprofv3', '--kernel-trace', '--stats', '-d', d, '--'] + TRACE_BENCH, f'{tag}_trace.log')
con = db_of(d)
name = [t for t in tables(con) if t.startswith('top_kernels')]
rows = con.execute(f'select * from {name[0]}').fetchall() if name else []
cols = [c[1] for c in con.execute(f'pragma table_info({name[0]})')] if name else []
with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w') as f:
  f.write(','.join(cols) + '\n')
  for r in rows:
    f.write(','.join(str(x) for x in r) + '\n')
print(open(os.path.join(out, f'{tag}_kernel_stats.csv')).read()[:600], flush=True)
con.close()
shutil.rmtree(d, ignore_errors=True)        # the raw databases are large; gpurun copies back <= 64 MiB
kernel_ms = None
if rows and cols:
  ci = {c.lower(): k for k, c in enumerate(cols)}
  total_us = 0.0
  for r in rows:
    nm = str(r[ci.get('name', 0)])
    if KERNEL_NAME in nm and kernel_ms is None:
      for key in ('average', 'avg', 'averagens', 'average_ns'):
        if key in ci:
          kernel_ms = float(r[ci[key]]) / 1e3          # the view reports microseconds
          break
    if any(k in nm for k in KERNELS):
      calls = [ci[k] for k in ('calls', 'count', 'total_calls') if k in ci]
      avgc = [ci[k] for k in ('average', 'avg', 'averagens', 'average_ns') if k in ci]
      if calls and avgc:
        total_us += float(r[calls[0]]) * float(r[avgc[0]])
  if len(KERNELS) > 1 and total_us:
    kernel_ms = total_us / 1e3 / (C5_RADII * C5_SWEEPS)        # per radius

# 3. counters, one set per run
sets = [['FETCH_SIZE'], ['WRITE_SIZE'], ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_WAVE_CYCLES'],
        ['SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY'],
        ['SQ_ACTIVE_INST_VALU', 'SQ_THREAD_CYCLES_VALU', 'SQ_INSTS_LDS', 'SQ_ACTIVE_INST_LDS'],
        ['SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_TRANS_F64'],
        ['SQ_INSTS_VALU_INT32', 'SQ_INSTS_VALU_INT64', 'SQ_INSTS_VALU_CVT', 'SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_ADD_F32',
         'SQ_INSTS_VALU_MUL_F32'],
        ['GRBM_GUI_ACTIVE', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'SQ_INSTS_VMEM']]
avg = {}
for k, cs in enumerate(sets):
  d = os.path.join(out, f'{tag}_pmc{k}')
  try:
    run(['rocprofv3', '--pmc'] + cs + ['--kernel-trace', '-d', d, '--'] + BENCH, f'{tag}_pmc{k}.log')
  except subprocess.CalledProcessError as e:
    print('counter set failed:', cs, e, flush=True)
    continue
  con = db_of(d)
  view = [t for t in tables(con) if t.startswith('counters_collection')]
  cols = [c[1] for c in con.execute(f'pragma table_info({view[0]})')]
  kcol = [c for c in cols if 'kernel' in c.lower() and 'name' in c.lower()] or [c for c in cols if c.lower() in ('name', 'kernel')]
  ncol = [c for c in cols if c.lower() in ('counter_name', 'counter')]
  vcol = [c for c in cols if c.lower() in ('value', 'counter_value')]
  if not (kcol and ncol and vcol):
    raise SystemExit(f'unexpected columns in {view[0]}: {cols}')
  like = ' or '.join(f"{kcol[0]} like '%{k}%'" for k in KERNELS)
  q = (f'select {ncol[0]}, avg({vcol[0]}), count(*), sum({vcol[0]}) from {view[0]} '
       f"where {like} group by {ncol[0]}")
  for cname, value, count, total in con.execute(q):
    if len(KERNELS) > 1:
      value = total / (C5_RADII * C5_SWEEPS)            # per radius
    avg[cname] = value
    print(cname, value, f'({count} dispatches)', flush=True)
  con.close()
  shutil.rmtree(d, ignore_errors=True)
fetch_raw = avg.get('FETCH_SIZE', 0.0) * 1024          # the counters are in KiB
write = avg.get('WRITE_SIZE', 0.0) * 1024
line = json.loads(open(os.path.join(out, f'{tag}_bench.json')).read()) if os.path.exists(os.path.join(out, f'{tag}_bench.json')) else {}
n_per = line.get('config', {}).get('rays_per_step_per_gpu') or line.get('config', {}).get('rays_per_radius') or args.rays
summary = dict(command='rocprofv3 --pmc <COUNTERS> --kernel-trace -- ' + ' '.join(BENCH[:1] + ['bench.py'] + BENCH[2:]) +
                       ' (one counter set per run)',
               kernel=KERNEL_LIKE.strip('%'), rays_per_launch=n_per,
               counters_unit=('per radius: summed over every dispatch of ' + ' + '.join(KERNELS) + f' and divided by {C5_RADII} radii x {C5_SWEEPS} sweeps') if len(KERNELS) > 1 else 'average per dispatch',
               counters_avg_per_dispatch=avg, fetch_bytes_raw=fetch_raw, fetch_bytes_corrected=2 * fetch_raw,
               write_bytes=write, hbm_bytes_per_launch=2 * fetch_raw + write, kernel_ms_rocprof=kernel_ms)
json.dump(summary, open(os.path.join(out, f'{tag}_pmc.json'), 'w'), indent=1)
ms = kernel_ms or line.get('roofline', {}).get('avg_kernel_ms')
ROUND = os.environ.get('ODW_PROFILE_ROUND', 'r05')


def _native_sources_hash():
  sys.path.insert(0, ROOT)
  from freecad.optics_design_workbench_amd import _native
  return _native.sources_hash()

valu = None
if 'SQ_INSTS_VALU' in avg and ms:
  valu = dict(insts_per_launch=avg['SQ_INSTS_VALU'], salu_insts_per_launch=avg.get('SQ_INSTS_SALU'),
              kernel_ms_profiled=ms, source=f'profiles/{ROUND}/{tag}_pmc.json, profiles/{ROUND}/{tag}_kernel_stats.csv')
  if avg.get('SQ_THREAD_CYCLES_VALU') and avg.get('SQ_ACTIVE_INST_VALU'):
    # lanes doing work per issued VALU instruction (64 = every lane)
    valu['active_lanes_per_inst'] = avg['SQ_THREAD_CYCLES_VALU'] / avg['SQ_ACTIVE_INST_VALU']
    valu['active_lanes_definition'] = 'SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (64 = every lane of every VALU instruction does work)'
  f64 = [avg.get(k) for k in ('SQ_INSTS_VALU_FMA_F64', 'SQ_INSTS_VALU_MUL_F64', 'SQ_INSTS_VALU_ADD_F64', 'SQ_INSTS_VALU_TRANS_F64')]
  if all(v is not None for v in f64):
    valu['f64_arith_insts'] = sum(f64)
    # flop per launch: an FMA counts 2, 64 lanes x the active-lane share
    lanes = valu.get('active_lanes_per_inst', 64.0)
    valu['fp64_flop_per_launch'] = (2 * f64[0] + f64[1] + f64[2] + f64[3]) * lanes
  if avg.get('SQ_WAIT_ANY') and avg.get('SQ_WAVE_CYCLES'):
    valu['wait_any_frac'] = avg['SQ_WAIT_ANY'] / avg['SQ_WAVE_CYCLES']
  if avg.get('SQ_LDS_BANK_CONFLICT') is not None and avg.get('SQ_LDS_IDX_ACTIVE'):
    valu['lds_bank_conflict_frac'] = avg['SQ_LDS_BANK_CONFLICT'] / avg['SQ_LDS_IDX_ACTIVE']
  # the calibrated issue floor (profiles/r03/valu_peak.json): dynamic counts by class where the hardware
  # classifies (f64 fma / mul / add / transcendental, int32, int64, cvt), the rest split like the static
  # census of the kernel (compares, v_cndmask, moves, min / max)
  peak_path = os.path.join(ROOT, 'profiles', 'r03', 'valu_peak.json')
  census_path = os.path.join(ROOT, 'profiles', 'r03', f'{args.config}_census.json')
  if os.path.exists(peak_path) and all(v is not None for v in f64):
    cyc = json.load(open(peak_path))['class_cycles']
    dyn = dict(f64_fma=f64[0], f64_mul=f64[1], f64_add=f64[2], f64_trans=f64[3],
               int32=avg.get('SQ_INSTS_VALU_INT32', 0.0), mov_b64=avg.get('SQ_INSTS_VALU_INT64', 0.0),
               cvt=avg.get('SQ_INSTS_VALU_CVT', 0.0),
               f32=sum(avg.get(k, 0.0) for k in ('SQ_INSTS_VALU_FMA_F32', 'SQ_INSTS_VALU_ADD_F32', 'SQ_INSTS_VALU_MUL_F32')))
    rest = avg['SQ_INSTS_VALU'] - sum(dyn.values())
    split = dict(cmp=0.35, cndmask=0.2, f64_minmax=0.08, mov_b32=0.25, mov_b64=0.12)      # (no census: a typical flat kernel)
    if os.path.exists(census_path):
      st = json.load(open(census_path))['valu_by_class']
      keys = ('cmp', 'cndmask', 'f64_minmax', 'mov_b32', 'mov_b64', 'f64_other')
      # (int64 / mov_b64 of the census: whatever the INT64 counter has not already taken)
      tot = sum(st.get(k, 0) for k in keys)
      split = {k: st.get(k, 0) / tot for k in keys}
    by_class = dict(dyn)
    for k, f in split.items():
      by_class[k] = by_class.get(k, 0.0) + rest * f
    cost = dict(f64_fma=cyc['f64_fma'], f64_mul=cyc['f64_mul'], f64_add=cyc['f64_add'], f64_trans=cyc['f64_trans'],
                f64_minmax=cyc['f64_minmax'], f64_other=cyc['f64_fma'], cmp=cyc['cmp'], cndmask=cyc['cndmask'],
                mov_b32=cyc['mov_b32'], mov_b64=cyc['mov_b64'], int32=cyc['int32'], cvt=cyc['f64_fma'], f32=cyc['f32_fma'])
    floor = sum(by_class[k] * cost[k] for k in by_class)
    valu['insts_by_class'] = by_class
    valu['class_cycles'] = cost
    valu['unclassified_split'] = dict(split, source=(f'static census profiles/r03/{args.config}_census.json' if os.path.exists(census_path)
                                                      else 'assumed'))
    valu['issue_cycles_per_launch'] = floor
    valu['cyc_per_inst_calibrated'] = floor / avg['SQ_INSTS_VALU']
    valu['definition'] = ('calibrated issue floor: sum over instruction classes of (wave-instructions per launch x cycles one SIMD '
                          'spends per instruction of that class at 4 waves per SIMD, profiles/r03/valu_peak.json); peak rate for this '
                          'mix = 1024 SIMDs x 2.4 GHz / cyc_per_inst_calibrated')
    valu['frac_profiled'] = floor / (1024 * ms * 1e-3 * 2.4e9)
entry = dict(kernel=KERNEL_LIKE.strip('%'), rays_per_launch=n_per, fetch_bytes_corrected=2 * fetch_raw, write_bytes=write,
             hbm_bytes_per_launch=2 * fetch_raw + write,
             correction='FETCH_SIZE x2 (gfx950 under-count of wide reads, MI355X_MICROARCH.md HBM section; upper bound for '
                        'this access mix), WRITE_SIZE as is; separate --pmc passes',
             source=f'profiles/{ROUND}/{tag}_pmc.json', valu=valu,
             # the code these counters were taken on: bench.py attaches them only to a build of the same sources
             sources_sha256=_native_sources_hash())
json.dump({(args.config if not args.script else tag): entry}, open(os.path.join(out, f'{tag}_pmc_current.json'), 'w'), indent=1)
print(json.dumps(entry)[:600])
