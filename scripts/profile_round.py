#!/usr/bin/env python3
"""Collect the round's profile evidence on the GPU box (run from the repo root):

  python scripts/profile_round.py v8

1. `python bench.py`                                         -> gpurun_out/<tag>_bench.json
2. `rocprofv3 --kernel-trace --stats -- python3 bench.py`     -> <tag>_kernel_stats.csv
3. `rocprofv3 --pmc <set> --kernel-trace -- python3 bench.py` -> <tag>_pmc.json, one run per counter set
   (FETCH_SIZE and WRITE_SIZE alone, as MI355X_MICROARCH.md prescribes)
rocprofv3 writes SQLite databases; the summaries are read from their views.
"""
import glob
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'vX'
out = os.path.join(ROOT, 'gpurun_out')
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR='/tmp')
BENCH = ['python3', os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu-baseline']


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
res = subprocess.run(['python3', os.path.join(ROOT, 'bench.py')], cwd=ROOT, env=env, capture_output=True, text=True, check=True)
line = [l for l in res.stdout.splitlines() if l.startswith('{')][-1]
open(os.path.join(out, f'{tag}_bench.json'), 'w').write(line + '\n')
print(line[:200], flush=True)

# 2. kernel trace
d = os.path.join(out, f'{tag}_trace')
run(['rocprofv3', '--kernel-trace', '--stats', '-d', d, '--'] + BENCH, f'{tag}_trace.log')
con = db_of(d)
name = [t for t in tables(con) if t.startswith('top_kernels')]
rows = con.execute(f'select * from {name[0]}').fetchall() if name else []
cols = [c[1] for c in con.execute(f'pragma table_info({name[0]})')] if name else []
with open(os.path.join(out, f'{tag}_kernel_stats.csv'), 'w') as f:
  f.write(','.join(cols) + '\n')
  for r in rows:
    f.write(','.join(str(x) for x in r) + '\n')
print(open(os.path.join(out, f'{tag}_kernel_stats.csv')).read()[:600], flush=True)

# 3. counters, one set per run
sets = [['FETCH_SIZE'], ['WRITE_SIZE'], ['SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_SMEM', 'SQ_WAVE_CYCLES'],
        ['SQ_BUSY_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY']]
avg = {}
for k, cs in enumerate(sets):
  d = os.path.join(out, f'{tag}_pmc{k}')
  run(['rocprofv3', '--pmc'] + cs + ['--kernel-trace', '-d', d, '--'] + BENCH, f'{tag}_pmc{k}.log')
  con = db_of(d)
  view = [t for t in tables(con) if t.startswith('counters_collection')]
  cols = [c[1] for c in con.execute(f'pragma table_info({view[0]})')]
  kcol = [c for c in cols if 'kernel' in c.lower() and 'name' in c.lower()] or [c for c in cols if c.lower() in ('name', 'kernel')]
  ncol = [c for c in cols if c.lower() in ('counter_name', 'counter')]
  vcol = [c for c in cols if c.lower() in ('value', 'counter_value')]
  if not (kcol and ncol and vcol):
    raise SystemExit(f'unexpected columns in {view[0]}: {cols}')
  q = (f'select {ncol[0]}, avg({vcol[0]}), count(*) from {view[0]} '
       f"where {kcol[0]} like '%odw_trace_kernel%' group by {ncol[0]}")
  for cname, value, count in con.execute(q):
    avg[cname] = value
    print(cname, value, f'({count} dispatches)', flush=True)
fetch_raw = avg.get('FETCH_SIZE', 0.0) * 1024          # the counters are in KiB
write = avg.get('WRITE_SIZE', 0.0) * 1024
summary = dict(command='rocprofv3 --pmc <COUNTERS> --kernel-trace -- python3 bench.py --steps 3 --warmup 1 '
                       '--no-cpu-baseline (one counter set per run)',
               kernel='odw::odw_trace_kernel<false, false, false>', rays_per_launch=100000000,
               counters_avg_per_dispatch=avg, fetch_bytes_raw=fetch_raw, fetch_bytes_corrected=2 * fetch_raw,
               write_bytes=write, hbm_bytes_per_launch=2 * fetch_raw + write)
json.dump(summary, open(os.path.join(out, f'{tag}_pmc.json'), 'w'), indent=1)
print(json.dumps(summary)[:400])
