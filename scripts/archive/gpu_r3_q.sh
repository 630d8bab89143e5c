#!/bin/bash
# round 3: launch length against ray count (compiled kernel): slots per hit-list reservation, smallest list with reservations
#   bash scripts/gpu_r3_q.sh sweep   -> the A/B over block sizes (profiles/r03/r03q_hit_blocks.log)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r03
O=gpurun_out/r03/r03q_short.log
export ODW_SL_SIZES=1e6,3e6,1e7,3e7,1e8
echo "default" > $O; python scripts/short_launch.py >> $O 2>&1
echo "default, generic kernel" >> $O; ODW_SL_COMPILE=off python scripts/short_launch.py >> $O 2>&1
if [ "$1" = sweep ]; then
for b in 128 256 512 1024 2048 4096; do
  echo "min rows 0, block $b" >> $O; ODW_HIT_BLOCK_MIN_ROWS=0 ODW_HIT_BLOCK=$b python scripts/short_launch.py >> $O 2>&1
done
fi
cat $O
