cd $GRAFT_REPO_ROOT
O=gpurun_out/r06i; mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tail-wide8 or schedule" ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for W in 0 8 16 24 32 64; do
  for rep in 1 2; do
  IGD_TAIL_WIDE8=$W timeout 300 python bench.py --steps 20 --warmup 5 --as-rank-of 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('IGD_TAIL_WIDE8=$W as rank 0 of 8: %8.1f Mrays/s  ms/step %.3f  trav1 %6.1f shade %6.1f trav2 %6.1f tail %5.1f' % (d['value'], d['ms_per_step'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
  done
done 2>&1 | tee $O/ab_tail_wide8.txt
