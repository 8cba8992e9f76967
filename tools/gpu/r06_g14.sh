cd $GRAFT_REPO_ROOT
O=gpurun_out/r06n; mkdir -p $O
for rep in 1 2; do for E in "IGD_SORT_SINGLE_CLASS=1" "IGD_SORT_SINGLE_CLASS=0"; do
env $E timeout 600 python bench.py --scene scenes/many_point_lights.json --steps 32 --warmup 32 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); s = d['stage_ms_rank0']
print('%-26s %8.1f Mrays/s   trav1 %7.1f  shade %7.1f  trav2 %7.1f  tail %6.1f' % ('$E', d['value'], s['ms_traverse_primary'], s['ms_shade'], s['ms_traverse_secondary'], s['ms_tail']))"
done; done 2>&1 | tee $O/ab_single_class.txt
IGD_SORT_SINGLE_CLASS=0 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "many_point or config4 or selectors or analytic" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
