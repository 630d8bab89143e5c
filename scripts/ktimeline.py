#!/usr/bin/env python3
"""GPU timeline summary of a rocprofv3 --kernel-trace database: for the last `window_ms` of kernel activity (or all),
busy time (union of kernel intervals), concurrency, per-kernel totals, idle gaps > 50 us"""
import glob, os, sqlite3, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*.db'), recursive=True))
con = sqlite3.connect(files[-1])
names = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
view = [n for n in names if n == 'kernels'] or [n for n in names if 'kernel' in n.lower()]
cols = [c[1] for c in con.execute(f'pragma table_info({view[0]})')]
if '--schema' in sys.argv:
  print(names); print(view[0], cols); sys.exit(0)
rows = con.execute(f'select name, start, end, queue_id, stream_id from {view[0]} order by start').fetchall() if 'stream_id' in cols else \
       con.execute(f'select name, start, end, queue_id, 0 from {view[0]} order by start').fetchall()
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else None
t_end = max(r[2] for r in rows)
if window:
  rows = [r for r in rows if r[1] >= t_end - window]
t0 = min(r[1] for r in rows)
span = (t_end - t0) / 1e6
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
busy = 0; conc_time = {}; cur = 0; last = ev[0][0]; gaps = []
for t, d in ev:
  if cur > 0: busy += t - last
  elif t - last > 50e3: gaps.append(((last - t0) / 1e6, (t - last) / 1e6))
  conc_time[cur] = conc_time.get(cur, 0) + t - last
  cur += d; last = t
print(f'span {span:.2f} ms, busy {busy / 1e6:.2f} ms ({100 * busy / 1e6 / span:.1f} %), kernels {len(rows)}')
print('time at concurrency k (ms):', {k: round(v / 1e6, 2) for k, v in sorted(conc_time.items())})
tot = {}
for n, s, e, q, st in rows:
  import re
  mm = re.search(r'([A-Za-z_][A-Za-z_0-9]*)\s*(<[^()]*>)?\s*\(', n.replace('(anonymous namespace)::', ''))
  k = (mm.group(1) if mm else n)[-48:]
  a = tot.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e6
for k, (c, ms) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:18]:
  print(f'  {k:48s} {c:5d} calls {ms:9.2f} ms  avg {1e3 * ms / c:9.1f} us')
print('idle gaps > 50 us:', len(gaps), 'total', round(sum(g[1] for g in gaps), 2), 'ms; largest', sorted(gaps, key=lambda g: -g[1])[:8])
