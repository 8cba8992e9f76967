cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
# 1. occupancy A/B: 5 workgroups per CU with a 12-entry LDS stack, and the 12-entry stack alone
bash tools/ab.sh 20 base occ5s12 occ4s12 2>&1 | tee $O/ab_occ.txt
# 2. whole-wave traversal threshold of the tail as rank 0 of 8
for W in 4 2 6 8 16; do
  for rep in 1 2; do
  IGD_TAIL_WIDE=$W timeout 300 python bench.py --steps 20 --warmup 5 --as-rank-of 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('IGD_TAIL_WIDE=$W as rank 0 of 8: %8.1f Mrays/s  ms/step %.3f  trav1 %6.1f shade %6.1f trav2 %6.1f tail %5.1f' % (d['value'], d['ms_per_step'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
  done
done 2>&1 | tee $O/ab_tail_wide.txt
# 3. config 5's shape: the 1 M divergent stand-in at 4096 x 4096, rows of rank 0 of 8, next to the whole film
D=/tmp/standin_1m_div
python tools/make_standin_scene.py $D --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
timeout 900 python bench.py --scene $D/standin.json --width 4096 --height 4096 --steps 32 --warmup 8 --as-rank-of 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic --profile-key standin_divergent 2> $O/c5_rank.err | tail -1 > $O/r06_bench_config5_rank0of8.json
timeout 900 python bench.py --scene $D/standin.json --width 4096 --height 4096 --steps 32 --warmup 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic --profile-key standin_divergent 2> $O/c5_whole.err | tail -1 > $O/r06_bench_config5_whole_film.json
python - <<'PY'
import json
a=json.load(open("gpurun_out/r06h/r06_bench_config5_rank0of8.json")); b=json.load(open("gpurun_out/r06h/r06_bench_config5_whole_film.json"))
print("config 5 proxy: rank 0 of 8 %.1f Mrays/s (%.3f ms/step), whole film %.1f Mrays/s (%.3f ms/step): per-rank efficiency %.3f, projected %.2fx at 8 GPUs before the gather" % (a["value"], a["ms_per_step"], b["value"], b["ms_per_step"], a["value"]/b["value"], 8*a["value"]/b["value"]))
PY
