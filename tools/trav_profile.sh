#!/bin/bash
# Runs ON the GPU box: event counts (tprof1 / tprof2) and phase clocks (tclocks) of k_traverse on diamond_scene and, optionally, the stand-in.
# usage: tools/trav_profile.sh <tag> [standin triangles]      (variants built before by tools/build_variant.sh)
TAG=$1; TRIS=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
V=$ROOT/ignis_amd/lib/var
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"; cd "$ROOT"
IGD_LIBRARY=$V/libig_device_hip_tprof1.so python tools/trav_events.py > "$OUT/trav_events_closest.json" 2> "$OUT/trav_events.err"
IGD_LIBRARY=$V/libig_device_hip_tprof2.so python tools/trav_events.py > "$OUT/trav_events_any.json" 2>> "$OUT/trav_events.err"
IGD_LIBRARY=$V/libig_device_hip_tclocks.so python tools/trav_clocks.py > "$OUT/trav_clocks.json" 2>> "$OUT/trav_events.err"
if [ -n "$TRIS" ]; then
  DIR=/tmp/standin_$TRIS
  [ -f "$DIR/standin.json" ] || python tools/make_standin_scene.py "$DIR" --triangles "$TRIS" > "$OUT/standin_make.log" 2>&1
  IGD_LIBRARY=$V/libig_device_hip_tprof1.so python tools/trav_events.py "$DIR/standin.json" 1920 1080 8 2 > "$OUT/trav_events_closest_standin.json" 2>> "$OUT/trav_events.err"
  IGD_LIBRARY=$V/libig_device_hip_tprof2.so python tools/trav_events.py "$DIR/standin.json" 1920 1080 8 2 > "$OUT/trav_events_any_standin.json" 2>> "$OUT/trav_events.err"
  IGD_LIBRARY=$V/libig_device_hip_tclocks.so python tools/trav_clocks.py "$DIR/standin.json" 1920 1080 8 2 > "$OUT/trav_clocks_standin.json" 2>> "$OUT/trav_events.err"
fi
tail -5 "$OUT/trav_events.err"
