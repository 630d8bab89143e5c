#!/usr/bin/env python3
"""profiles/r03/valu_peak.json from the outputs of scripts/gpu_r3_valu.sh (gpurun_out/r03/):
the instruction-class costs of the vector ALU of gfx950 at 4 waves per SIMD, 256-thread blocks, every CU busy
(scripts/valu_peak.hip, scripts/valu_sel.hip), with the SQ counters of the same streams beside them.

cycles per instruction per SIMD = wall time (hipEvents) x in-kernel shader clock (d s_memtime / d s_memrealtime)
/ (W x instructions per wave): what one SIMD spends per wave64 instruction with W waves resident.  (The waves of a
SIMD do not progress evenly -- the oldest wave wins the arbitration, MI355X_MICROARCH.md "Two waves per SIMD" --, so a
per-wave cycle count says nothing about the SIMD; the launch's wall time does.)"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'r03')
t = re.sub(r'\bnan\b|\binf\b', 'null', open(os.path.join(src, 'valu_peak_streams.json')).read())
streams = json.loads(t)
out = dict(device=streams['arch'], cus=streams['cus'], simds=streams['cus'] * 4, block=256,
           definition=__doc__.split('\n\n')[1].replace('\n', ' '),
           command='./scripts/valu_peak ; ./scripts/valu_sel ; rocprofv3 --pmc <set> --kernel-trace -- ./scripts/valu_peak --w4',
           streams=[])
pmc = {}
for k in (0, 1):
  path = os.path.join(src, f'vp_pmc{k}', 'vp_counter_collection.csv')
  if os.path.exists(path):
    for r in csv.DictReader(open(path)):
      m = re.match(r'void stream<(\d+)>', r['Kernel_Name'])
      if m:
        # two dispatches per stream (warm-up + measured): keep the last
        pmc.setdefault(int(m.group(1)), {})[r['Counter_Name']] = float(r['Counter_Value'])
for s in streams['streams']:
  if s['clock_ghz'] is None or s['wall_ms'] < 0.05:
    continue
  w, n = s['waves_per_simd'], s['inst_per_wave']
  e = dict(stream=s['stream'], waves_per_simd=w, inst_per_wave=n, wall_ms=s['wall_ms'], clock_ghz=s['clock_ghz'],
           cyc_per_inst_simd=round(s['wall_ms'] * 1e6 * s['clock_ghz'] / (w * n), 3))
  if w == 4 and s['id'] in pmc:
    c = pmc[s['id']]
    e['counters'] = {k: c[k] for k in sorted(c)}
    if c.get('SQ_INSTS_VALU'):
      e['SQ_ACTIVE_INST_VALU_per_SQ_INSTS_VALU'] = round(c.get('SQ_ACTIVE_INST_VALU', 0) / c['SQ_INSTS_VALU'], 3)
      if c.get('GRBM_GUI_ACTIVE'):
        e['cyc_per_inst_simd_by_GRBM_GUI_ACTIVE'] = round(c['GRBM_GUI_ACTIVE'] / 8 / (c['SQ_INSTS_VALU'] / out['simds']), 3)
  out['streams'].append(e)
sel = json.load(open(os.path.join(src, 'valu_sel.json')))
out['select_in_context'] = dict(pattern=sel['pattern'], groups=[g for g in sel['groups'] if g['waves_per_simd'] == 4],
                                note='cycles per group per SIMD at 4 waves per SIMD; minus (fma_between + fma_after) x cost of '
                                     'v_fma_f64 = the cost of the select itself')
w4 = {e['stream']: e['cyc_per_inst_simd'] for e in out['streams'] if e['waves_per_simd'] == 4}
fma = w4['v_fma_f64']
sel_cost = [g['cyc_per_group_simd'] - fma * (g['fma_between'] + g['fma_after']) for g in out['select_in_context']['groups']
            if g['mask'] == 'sgpr pair' or (g['mask'] == 'vcc' and g['fma_between'] + g['fma_after'] >= 2)]
cmp_cost = w4['v_cmp_lt_f64 vcc']
cnd = (sum(sel_cost) / len(sel_cost) - cmp_cost) / 2
out['class_cycles'] = dict(
    f64_fma=w4['v_fma_f64'], f64_mul=w4['v_mul_f64'], f64_add=w4['v_add_f64'], f64_minmax=w4['v_min_f64'],
    f64_trans=(w4['v_rcp_f64'] + w4['v_rsq_f64']) / 2, cmp=cmp_cost, cndmask=round(cnd, 3),
    cndmask_isolated=round(4 * w4['1 v_cndmask_b32 + 3 v_add_u32'] - 3 * w4['v_add_u32'], 3),
    mov_b32=w4['v_mov_b32'], mov_b64=w4['v_mov_b64'], int32=w4['v_add_u32'], f32_fma=w4['v_fma_f32'],
    f64_select_cmp_2cndmask=round(sum(sel_cost) / len(sel_cost), 3),
    note='cycles one SIMD spends per wave64 instruction at 4 waves per SIMD; cndmask = (select in context - cmp) / 2: a '
         'v_cndmask_b32 between float64 work costs about as much as a float64 instruction, twice what it costs between '
         '32-bit integer work (cndmask_isolated); back to back with the mask in VCC it costs 23 (stream "v_cndmask_b32 x, x, y, vcc")')
out['findings'] = [
    'SQ_ACTIVE_INST_VALU = SQ_INSTS_VALU for every non-transcendental stream (4 x for v_rcp_f64 / v_rsq_f64): it counts '
    'instructions, not busy cycles -- "VALU busy" derived from it (round 2) was the instruction count x 4',
    'every VOPC compare costs a float64 slot (4.2 - 4.3 cycles), also v_cmp_lt_f32; v_mov_b64 4.2; v_mov_b32 / v_add_u32 2.1 - 2.3',
    'a float64 select (v_cmp + 2 v_cndmask_b32) costs 13 - 13.5 cycles in context, v_min_f64 / v_max_f64 4.3 - 4.9',
    'a stream of float64 FMAs on every SIMD lowers the clock to 1.6 GHz (the other float64 streams hold 2.1 - 2.4)']
json.dump(out, open(os.path.join(ROOT, 'profiles', 'r03', 'valu_peak.json'), 'w'), indent=1)
print(json.dumps(out['class_cycles'], indent=1))
