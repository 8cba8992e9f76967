#!/bin/bash
# One-at-a-time sweep of the device's environment knobs around their defaults (diamond_scene 1080p, 32 steps).
# Usage (GPU box): bash tools/sweep_knobs.sh > gpurun_out/sweep.txt
run() {
  local label="$1"; shift
  local v
  v=$(env "$@" timeout 120 python bench.py --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic --steps ${SWEEP_STEPS:-32} --warmup ${SWEEP_WARMUP:-2} 2>/dev/null | tail -1 | python -c 'import sys,json; print(json.loads(sys.stdin.read())["value"])' 2>/dev/null)
  echo "$label $v"
}
run default X=1
run default X=1
for t in 524288 786432 1572864 2097152; do run "IGD_TAIL_THRESHOLD=$t" IGD_TAIL_THRESHOLD=$t; done
for t in 3 4 8 10; do run "IGD_TAIL_SPLIT=$t" IGD_TAIL_SPLIT=$t; done
for t in 4 6 10 12; do run "IGD_TAIL_WAVES=$t" IGD_TAIL_WAVES=$t; done
for t in 2 3 6 8; do run "IGD_FLIGHTS=$t" IGD_FLIGHTS=$t; done
for t in 32 48 96 128; do run "IGD_SHADE_GRID=$t" IGD_SHADE_GRID=$t; done
for t in 33554432 67108864; do run "IGD_BATCH_RAYS=$t" IGD_BATCH_RAYS=$t; done
run default X=1
