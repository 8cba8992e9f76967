#!/bin/bash
# usage: run_variants.sh "<num> <shift>" ...   (on the GPU box)
cd $GRAFT_REPO_ROOT
python tools/make_standin_scene.py /tmp/standin --triangles 1000000 --instances 96 > /dev/null
for v in "$@"; do
  set -- $v; num=$1; shift_=$2
  sed -i "s/^constexpr int kPostponeNum   = [0-9]*;/constexpr int kPostponeNum   = $num;/; s/^constexpr int kPostponeShift = [0-9]*;/constexpr int kPostponeShift = $shift_;/" ignis_amd/csrc/device/traverse_core.h
  (cd ignis_amd/csrc && make -j16 2>&1 | grep -i "error" )
  echo "== num=$num shift=$shift_"
  timeout 200 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('diamond', d['value'], d['ms_per_step'], d['stage_ms_rank0'])"
  timeout 200 python bench.py --scene /tmp/standin/standin.json --steps 4 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('standin', d['value'], d['ms_per_step'], d['stage_ms_rank0'])"
done
