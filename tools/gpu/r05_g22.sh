cd $GRAFT_REPO_ROOT
O=gpurun_out/r05t; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_live.json 2> $O/bench_live.err; tail -3 $O/bench_live.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05t/bench_live.json').read().strip().splitlines()[-1])
r=d['roofline']; print(d['value'], r['traffic'], r['traffic_from_profiles'], r['measured_frac'], r['traffic_source'][:160])
PY
