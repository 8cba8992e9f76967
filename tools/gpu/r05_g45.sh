cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ak; mkdir -p $O
run() { # $1 = env string or "-", rest = bench args
  E=$1; shift; EE=$E; [ "$E" = "-" ] && EE=""
  env $EE python bench.py "$@" --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('%-40s %8.1f Mrays/s  %.3f ms/step trav1 %.1f shade %.1f trav2 %.1f tail %.1f' % ('$E', d['value'], d['ms_per_step'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
}
echo "== as rank 0 of 8, 20 steps"
for rep in 1 2; do for e in - IGD_SIDE_PRIORITY=normal IGD_SIDE_PRIORITY=high IGD_TAIL_THRESHOLD=524288 IGD_TAIL_THRESHOLD=2097152 IGD_TAIL_THRESHOLD=4194304; do run $e --steps 20 --warmup 5 --as-rank-of 8; done; done 2>&1 | tee $O/sweep_rankof8.log
echo "== whole film, 20 steps"
for rep in 1 2; do for e in - IGD_SIDE_PRIORITY=normal IGD_TAIL_THRESHOLD=524288 IGD_TAIL_THRESHOLD=2097152 IGD_TAIL_THRESHOLD=4194304; do run $e --steps 20 --warmup 5; done; done 2>&1 | tee $O/sweep_whole20.log
echo "== whole film, 256 steps"
for e in - IGD_SIDE_PRIORITY=normal IGD_TAIL_THRESHOLD=2097152; do run $e --steps 256 --warmup 32; done 2>&1 | tee $O/sweep_whole256.log
