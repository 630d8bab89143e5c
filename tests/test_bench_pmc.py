"""bench.py attaches the committed counter pass (profiles/pmc_current.json) to its roofline block only when the pass
was taken on the sources the loaded library was built from: every entry carries `_native.sources_hash()` of its day
(scripts/profile_round.py); one byte of difference and the line says `pmc_stale` instead of a fraction."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from freecad.optics_design_workbench_amd import _native


def _entry(sha):
  return dict(kernel='odw_spec_kernel', rays_per_launch=1000, fetch_bytes_corrected=10.0, write_bytes=20.0, hbm_bytes_per_launch=30.0,
              source='profiles/rXX/x_pmc.json', sources_sha256=sha,
              valu=dict(insts_per_launch=1.0e6, cyc_per_inst_calibrated=4.0, kernel_ms_profiled=1.0, active_lanes_per_inst=60.0))


def test_sources_hash_follows_every_byte(tmp_path):
  csrc = tmp_path / 'csrc'
  shutil.copytree(_native.CSRC, csrc, ignore=shutil.ignore_patterns('*.so', '*.tmp', '*.o'))
  header = tmp_path / 'odw_trace.h'
  shutil.copy(_native._HEADER, header)
  same = _native.sources_hash(str(csrc), str(header))
  assert same == _native.sources_hash() and len(same) == 64
  target = csrc / 'odw_grid.hip'
  data = bytearray(target.read_bytes())
  data[len(data) // 2] ^= 1
  target.write_bytes(bytes(data))
  assert _native.sources_hash(str(csrc), str(header)) != same
  target.write_bytes(bytes(data[:len(data) // 2]) + bytes([data[len(data) // 2] ^ 1]) + bytes(data[len(data) // 2 + 1:]))
  assert _native.sources_hash(str(csrc), str(header)) == same
  header.write_bytes(header.read_bytes() + b' ')
  assert _native.sources_hash(str(csrc), str(header)) != same


def test_a_counter_pass_of_other_sources_is_reported_stale(tmp_path):
  import bench
  now = _native.sources_hash()
  path = tmp_path / 'pmc.json'
  path.write_text(json.dumps(dict(c3=_entry(now))))
  got = bench.pmc_figures('c3', 1000, True, 'odw_spec_kernel', path=str(path))
  assert got and not got.get('pmc_stale') and got['valu']['insts_per_launch'] == 1.0e6
  r = bench.roofline_block('odw_spec_kernel', 1e-3, 1000, 960.0, got)
  assert r['frac'] is not None and 'pmc_stale' not in r
  # one byte flipped in the sources = another hash
  other = ('0' if now[0] != '0' else '1') + now[1:]
  for stale in (bench.pmc_figures('c3', 1000, True, 'odw_spec_kernel', sources_sha256=other, path=str(path)),):
    assert stale == dict(pmc_stale=True, source='profiles/rXX/x_pmc.json', profiled_sources_sha256=now, sources_sha256=other)
    r = bench.roofline_block('odw_spec_kernel', 1e-3, 1000, 960.0, stale)
    assert r['pmc_stale'] is True and r['frac'] is None and r['achieved'] is None and r['traffic'] is None
    assert 'other sources' in r['note']
  # an entry without a hash (passes of rounds 1 - 4) is stale by definition
  legacy = _entry(now)
  del legacy['sources_sha256']
  path.write_text(json.dumps(dict(c3=legacy)))
  assert bench.pmc_figures('c3', 1000, True, 'odw_spec_kernel', path=str(path))['pmc_stale'] is True
  # other workload / kernel: no entry at all, as before
  assert bench.pmc_figures('c3', 999, True, 'odw_spec_kernel', path=str(path)) is None
  assert bench.pmc_figures('c3', 1000, True, 'odw_trace_kernel<false>', path=str(path)) is None


def test_the_committed_pass_belongs_to_the_committed_sources():
  """profiles/pmc_current.json as committed: every entry carries a hash; whether it is HEAD's is what bench.py reports
  at run time (a source change after the round's last profile pass shows as pmc_stale in the line, not as a number
  computed from another kernel)"""
  path = os.path.join(ROOT, 'profiles', 'pmc_current.json')
  entries = json.load(open(path))
  assert set(entries) >= {'c3', 'c4', 'c5'}
  for name, e in entries.items():
    assert isinstance(e.get('sources_sha256'), str) and len(e['sources_sha256']) == 64, name
