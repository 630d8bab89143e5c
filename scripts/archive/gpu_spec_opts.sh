#!/bin/bash
# compiler options on the scene-compiled kernel (ODW_SPEC_OPTS), steady-state launch times of C3
cd "$(dirname "$0")/.."
try() { echo "== $1"; ODW_SPEC_OPTS="$1" python scripts/launch_times.py 2>&1 | tail -1 | cut -c1-100; }
try ""
try "-mllvm -amdgpu-sched-strategy=max-ilp"
try "-mllvm -amdgpu-sched-strategy=max-memory-clause"
try "-mllvm -amdgpu-sched-strategy=iterative-minreg"
try "-O2"
try "-mllvm -amdgpu-early-inline-all=true"
try "-mllvm -enable-post-misched=0"
try "-DODW_SPEC_WAVES=3"
try "-DODW_REFILL_MIN=24"
try "-DODW_CHUNK=4096ull"
