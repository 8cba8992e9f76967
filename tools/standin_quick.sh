#!/bin/bash
# Runs ON the GPU box: the HBM-regime stand-in (tools/make_standin_scene.py) through bench.py for each given environment setting.
# usage: tools/standin_quick.sh <tag> <triangles> <steps> "ENV1=a ENV2=b" "ENV1=c" ...      ("-" = no extra environment)
TAG=$1; TRIS=$2; STEPS=$3; shift; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
MATS=${STANDIN_MATERIALS:-lean}
DIR=/tmp/standin_${TRIS}_$MATS
mkdir -p "$ROOT/gpurun_out/$TAG"
cd "$ROOT"
[ -f "$DIR/standin.json" ] || python tools/make_standin_scene.py "$DIR" --triangles "$TRIS" --materials "$MATS" > "$ROOT/gpurun_out/$TAG/standin_make.log" 2>&1
for E in "$@"; do
  [ "$E" = "-" ] && E=""
  env $E timeout 900 python bench.py --scene "$DIR/standin.json" --steps "$STEPS" --warmup "$STEPS" --no-cpu-baseline --no-live-traffic --no-literal-config 2> "$ROOT/gpurun_out/$TAG/standin_quick.err" | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
s = d['stage_ms_rank0']
print('%-40s %8.1f Mrays/s   trav1 %7.1f  shade %7.1f  trav2 %7.1f  tail %6.1f  sort %6.1f  launch %.3f ms  frac %s' % ('$E', d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail'], s.get('ms_ray_sort', 0.0), d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
done
