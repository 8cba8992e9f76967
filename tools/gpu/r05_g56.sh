cd $GRAFT_REPO_ROOT
O=gpurun_out/r05as; mkdir -p $O
run() { E=$1; shift; EE=$E; [ "$E" = "-" ] && EE=""
  env $EE python bench.py "$@" --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('%-28s %8.1f Mrays/s  %.3f ms/step trav1 %.1f shade %.1f trav2 %.1f tail %.1f' % ('$E', d['value'], d['ms_per_step'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
}
for rep in 1 2; do for e in - IGD_FLIGHTS=2 IGD_FLIGHTS=3 IGD_FLIGHTS=8 IGD_TAIL_WAVES=8 IGD_TAIL_WAVES=16; do run $e --steps 256 --warmup 32; done; done 2>&1 | tee $O/sweep_flights256.log
