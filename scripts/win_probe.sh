#!/bin/bash
# does the compiled kernel's time depend on the LDS the histogram window takes? (slow-box hunt)
cd "$(dirname "$0")/.."
python scripts/launch_times.py 2>&1 | tail -1 | cut -c1-110
ODW_SPEC_OPTS="-DODW_HIST_WIN=64" python scripts/launch_times.py 2>&1 | tail -1 | cut -c1-110
ODW_SPEC_OPTS="-DODW_HIST_WIN=40" python scripts/launch_times.py 2>&1 | tail -1 | cut -c1-110
