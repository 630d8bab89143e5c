#!/bin/bash
# what the box lets an ordinary user read about the GPU's clocks (for the bench line's box-to-box note)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r03
O=gpurun_out/r03/clock_probe.log
{
for d in /sys/class/drm/card*/device; do
  echo "== $d"; for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk power_dpm_force_performance_level current_link_speed; do
    [ -r $d/$f ] && { echo "-- $f"; cat $d/$f; }; done
  for h in $d/hwmon/hwmon*; do for f in freq1_input freq2_input power1_average power1_input temp1_input power1_cap; do [ -r $h/$f ] && echo "$h/$f $(cat $h/$f)"; done; done
done
echo "== rocm-smi"; timeout 30 rocm-smi --showclocks --showperflevel --showpower 2>&1 | head -40
} > $O 2>&1
python - <<'PY' >> $O 2>&1
import sys, os, time, threading, glob
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from freecad.optics_design_workbench_amd import scenes
from freecad.optics_design_workbench_amd.simulation.tracer import Tracer
pr = scenes.bakeProject('tests/golden/scenes/GettingStarted.FCStd')
tr = Tracer(0); tr.setScene(pr.scene); tr.setSource(pr.source); tr.setLimits(pr.limits); tr.setDetector(None)
tr.compileScene('structure'); n = 100_000_000; tr.reserveHits(int(n*1.25)+1024)
files = glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input') + glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_average') + glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/power1_input')
print('files', files)
def read():
  out = []
  for f in files:
    try: out.append(int(open(f).read()))
    except Exception as e: out.append(None)
  return out
print('idle', read())
stop = False; samples = []
def sampler():
  while not stop:
    samples.append((time.time(), read())); time.sleep(0.02)
th = threading.Thread(target=sampler); th.start()
t0 = time.time()
for s in range(40):
  tr.reset(); tr.trace(s*n, n, 1, histogram=False)
tr.sync(); t1 = time.time(); stop = True; th.join()
print('40 launches of 1e8 rays: %.1f ms each' % ((t1-t0)/40*1e3))
for t, v in samples[::3]: print('%.3f' % (t-t0), v)
PY
tail -50 $O
