"""bench.py's entry logic that needs no GPU: it never reports fewer GPUs than asked for."""
import os
import subprocess
import sys

from conftest import ROOT


def _run(*argv, env=None):
  e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
  e.update(env or {})
  return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(argv), env=e, capture_output=True,
                        text=True, timeout=300)


def test_more_gpus_than_present_is_an_error():
  import torch
  n = torch.cuda.device_count() + 2
  res = _run('--gpus', str(n), '--steps', '1', '--warmup', '0')
  assert res.returncode != 0 and f'--gpus {n}' in res.stderr
  assert not [l for l in res.stdout.splitlines() if l.startswith('{')]


def test_launcher_world_size_must_match():
  res = _run('--gpus', '4', env=dict(RANK='0', LOCAL_RANK='0', WORLD_SIZE='2'))
  assert res.returncode != 0 and 'launcher started 2' in res.stderr
  res = _run('--gpus', '0')
  assert res.returncode != 0
