#!/usr/bin/env python3
"""print the kernel summary (top_kernels view) of the rocprofv3 database under a folder as CSV"""
import glob, os, sqlite3, sys
files = sorted(glob.glob(os.path.join(sys.argv[1], '**', '*.db'), recursive=True))
if not files:
  sys.exit(f'no rocprofv3 database under {sys.argv[1]}')
con = sqlite3.connect(files[-1])
views = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')") if r[0].startswith('top_kernels')]
if not views:
  sys.exit('no top_kernels view')
cols = [c[1] for c in con.execute(f'pragma table_info({views[0]})')]
print(','.join(cols))
for r in con.execute(f'select * from {views[0]}').fetchall()[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
  print(','.join(str(x) for x in r))
