#!/bin/bash
# Runs ON the GPU box: the HBM-regime workload. Generates the seeded procedural stand-in scene of BASELINE configs 3 / 5 with
# <triangles> unique triangles (16 M -> more than 1 GB of BVH, beyond the 256 MB Infinity Cache), then the bench line, the
# rocprofv3 kernel statistics and the PMC passes of `bench.py --scene` on it (tools/collect_profiles.sh, suffix _standin).
# usage: tools/run_standin.sh <tag> [triangles] [steps] [materials: lean (the traversal workload of rounds 2 - 4, default) | divergent] [suffix]
TAG=${1:-r02}
TRIS=${2:-16000000}
STEPS=${3:-16}
MATS=${4:-lean}
SUF=${5:-_standin}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
DIR=/tmp/standin_${TRIS}_$MATS
mkdir -p "$ROOT/gpurun_out/$TAG"
cd "$ROOT"
python tools/make_standin_scene.py "$DIR" --triangles "$TRIS" --materials "$MATS" > "$ROOT/gpurun_out/$TAG/standin_make.log" 2>&1
du -sh "$DIR" >> "$ROOT/gpurun_out/$TAG/standin_make.log"
bash tools/collect_profiles.sh "$TAG" "$SUF" --scene "$DIR/standin.json" --steps "$STEPS" --warmup "$STEPS"
