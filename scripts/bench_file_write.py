#!/usr/bin/env python3
"""what the box's file systems take: N threads, each writing 256 MB arrays (os.write of a numpy buffer, one call per array,
outside the interpreter lock) into fresh files under a folder -- the rate that bounds runSimulation's run-folder route.
   python scripts/bench_file_write.py [folder ...]"""
import os, sys, threading, time, tempfile, shutil
import numpy as np
folders = sys.argv[1:] or ['/tmp', '/dev/shm']
chunk = np.ones(32 * 1024 * 1024, dtype=np.float64)          # 256 MB
per_thread = 6                                                # files per thread
for base in folders:
  for n in (1, 2, 4, 6, 8, 12, 16):
    d = tempfile.mkdtemp(prefix='odw_w_', dir=base)
    def work(k):
      for j in range(per_thread):
        path = os.path.join(d, f't{k}_{j}.bin')
        fd = os.open(path + '.tmp', os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
        try:
          mv = memoryview(chunk).cast('B')
          off = 0
          while off < len(mv):
            off += os.write(fd, mv[off:])
        finally:
          os.close(fd)
        os.replace(path + '.tmp', path)
        os.unlink(path)                                       # (bounded footprint: the page cache holds what is in flight)
    ts = [threading.Thread(target=work, args=(k,)) for k in range(n)]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    dt = time.perf_counter() - t0
    print(f'{base:10s} threads {n:2d}  {n * per_thread * chunk.nbytes / dt / 1e9:6.1f} GB/s', flush=True)
    shutil.rmtree(d, ignore_errors=True)
