#!/usr/bin/env python3
"""Per-kernel register / spill / scratch figures of a build: parses hipcc's
-Rpass-analysis=kernel-resource-usage remarks from stdin (or a file)."""
import re
import subprocess
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
cur = None
rows = {}
for line in text.splitlines():
  m = re.search(r'remark:\s+Function Name: (\S+)', line)
  if m:
    cur = m.group(1)
    rows[cur] = {}
    continue
  m = re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)', line)
  if m and cur:
    rows[cur][m.group(1).strip()] = m.group(2)
names = list(rows)
try:
  dem = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + names, capture_output=True, text=True).stdout.splitlines()
except OSError:
  dem = names
for n, d in zip(names, dem):
  r = rows[n]
  print(f"{d[:90]:90s} VGPR {r.get('VGPRs','?'):>4} spill {r.get('VGPRs Spill','?'):>3} SGPRspill {r.get('SGPRs Spill','?'):>3} "
        f"scratch {r.get('ScratchSize','?'):>4} occ {r.get('Occupancy','?')} LDS {r.get('LDS Size','?')}")
