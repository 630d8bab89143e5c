#!/usr/bin/env python3
"""Static census of the vector instructions of a scene-compiled flat kernel (odw_spec_kernel), by the
instruction classes of profiles/r03/valu_peak.json:

  python scripts/kernel_census.py lensesAndMirrors [out.json]

The header of the scene is written by the library (odw_compile_check), the kernel source is compiled with hipcc
to assembly with the options odw_spec.hip hands hiprtc.  The census is STATIC (one count per instruction of the
binary, hot and cold paths alike): bench.py uses it only to split the part of SQ_INSTS_VALU the hardware counters
do not classify (everything but f64 fma / mul / add / transcendental, int32, int64, cvt) into compares,
v_cndmask, moves and min / max."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, 'freecad', 'optics_design_workbench_amd', 'csrc')


def classify(op):
  if op.startswith('v_cndmask'):
    return 'cndmask'
  if op.startswith('v_cmp') or op.startswith('v_cmpx'):
    return 'cmp'
  if re.match(r'v_(fma|fmac|mad)_f64', op):
    return 'f64_fma'
  if op.startswith('v_mul_f64'):
    return 'f64_mul'
  if op.startswith('v_add_f64'):
    return 'f64_add'
  if re.match(r'v_(min|max)_f64', op):
    return 'f64_minmax'
  if re.match(r'v_(rcp|rsq|sqrt)_f64', op):
    return 'f64_trans'
  if op.startswith('v_mov_b64') or op.startswith('v_lshl_add_u64') or re.match(r'v_\w+_(u|i|b)64', op):
    return 'mov_b64'
  if re.match(r'v_(cvt|ldexp|frexp|fract|floor|ceil|trunc|rndne|div_|fixup)', op) or op.endswith('_f64') or op.endswith('_f64_e32') or op.endswith('_f64_e64'):
    return 'f64_other'
  if op.startswith('v_mov_b32') or op.startswith('v_accvgpr') or op.startswith('v_readlane') or op.startswith('v_readfirstlane') or op.startswith('v_writelane'):
    return 'mov_b32'
  return 'int32'


def census(asm, kernel='odw_spec_kernel'):
  body, on = [], False
  for line in asm.splitlines():
    if line.startswith(kernel + ':'):
      on = True
      continue
    if on:
      if re.match(r'\s+s_endpgm', line):
        break
      m = re.match(r'\s+([vs]_[a-z0-9_]+|ds_[a-z0-9_]+|global_[a-z0-9_]+|buffer_[a-z0-9_]+|flat_[a-z0-9_]+|scratch_[a-z0-9_]+)', line)
      if m:
        body.append(m.group(1))
  counts, other = {}, {}
  for op in body:
    if op.startswith('v_'):
      c = classify(op)
      counts[c] = counts.get(c, 0) + 1
    else:
      k = op.split('_')[0]
      other[k] = other.get(k, 0) + 1
  return counts, other, len(body)


def main_lib():
  """--lib <substring of the mangled kernel name> [out.json]: a kernel of the library build (one translation unit:
  odw_capi.hip includes the kernel sources)"""
  from freecad.optics_design_workbench_amd import _native
  srcname, pattern = 'odw_capi.hip', sys.argv[2]
  out_path = sys.argv[3] if len(sys.argv) > 3 else None
  d = tempfile.mkdtemp()
  try:
    cmd = [_native.hipcc(), '--offload-arch=gfx950', '-std=c++17', '-O3', '-ffp-contract=on', '-S', '--cuda-device-only',
           '-o', os.path.join(d, 'k.s'), os.path.join(CSRC, srcname)]
    subprocess.run(cmd, check=True, capture_output=True, cwd=CSRC)
    asm = open(os.path.join(d, 'k.s')).read()
  finally:
    shutil.rmtree(d, ignore_errors=True)
  names = [m.group(1) for m in re.finditer(r'^(\S+):\s+; @', asm, re.M) if pattern in m.group(1)]
  if len(names) != 1:
    raise SystemExit(f'{pattern!r} matches {names}')
  counts, other, n = census(asm, names[0])
  res = dict(kernel=names[0], source=srcname, instructions=n, valu=sum(counts.values()), valu_by_class=dict(sorted(counts.items())),
             non_valu=dict(sorted(other.items())), note='static counts of the code object (hot and cold paths alike)')
  text = json.dumps(res, indent=1)
  if out_path:
    open(out_path, 'w').write(text + '\n')
  print(text)


def main():
  if len(sys.argv) > 1 and sys.argv[1] == '--lib':
    return main_lib()
  scene = sys.argv[1] if len(sys.argv) > 1 else 'lensesAndMirrors'
  out_path = sys.argv[2] if len(sys.argv) > 2 else None
  from freecad.optics_design_workbench_amd import _native, scenes
  proj = scenes.bakeProject(os.path.join(ROOT, 'tests', 'golden', 'scenes', scene + '.FCStd'))
  header, code_bytes = _native.compile_check(proj.scene, proj.limits, 'structure')
  d = tempfile.mkdtemp()
  try:
    open(os.path.join(d, 'odw_spec.h'), 'w').write(header)
    shutil.copy(os.path.join(CSRC, 'odw_kernels.hip'), d)
    shutil.copy(os.path.join(ROOT, 'include', 'odw_trace.h'), d)
    text = open(os.path.join(CSRC, 'odw_device.h')).read().replace('"../../../include/odw_trace.h"', '"odw_trace.h"')
    open(os.path.join(d, 'odw_device.h'), 'w').write(text)
    cmd = [_native.hipcc(), '--offload-arch=gfx950', '-std=c++17', '-O3', '-ffp-contract=on', '-DODW_SPEC_HEADER="odw_spec.h"',
           *os.environ.get('ODW_SPEC_OPTS', '').split(),      # (the experiment switches odw_spec.hip hands hiprtc)
           '-I' + d, '-S', '--cuda-device-only', '-o', os.path.join(d, 'k.s'), os.path.join(d, 'odw_kernels.hip')]
    subprocess.run(cmd, check=True, capture_output=True)
    asm = open(os.path.join(d, 'k.s')).read()
    if os.environ.get('ODW_KEEP_ASM'):      # the assembly itself, to read
      shutil.copy(os.path.join(d, 'k.s'), os.environ['ODW_KEEP_ASM'])
      shutil.copy(os.path.join(d, 'odw_spec.h'), os.environ['ODW_KEEP_ASM'] + '.spec.h')
  finally:
    shutil.rmtree(d, ignore_errors=True)
  counts, other, n = census(asm)
  res = dict(scene=scene, kernel='odw_spec_kernel', code_bytes_hiprtc=code_bytes, instructions=n, valu=sum(counts.values()),
             valu_by_class=dict(sorted(counts.items())), non_valu=dict(sorted(other.items())),
             note='static counts of the code object (hot and cold paths alike)')
  text = json.dumps(res, indent=1)
  if out_path:
    open(out_path, 'w').write(text + '\n')
  print(text)


if __name__ == '__main__':
  main()
