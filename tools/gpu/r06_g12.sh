cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
timeout 600 python bench.py --scene scenes/many_point_lights.json --steps 32 --warmup 32 --profile-key many_point_lights --no-cpu-baseline --no-literal-config --no-extra-configs 2> $O/mpl.err | tail -1 > $O/r06_bench_many_point_lights.json
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
IGD_RAY_SORT=1 timeout 600 python bench.py --scene /tmp/standin_1m_div/standin.json --steps 16 --warmup 16 --profile-key standin_divergent_sorted --no-cpu-baseline --no-literal-config --no-extra-configs 2> $O/sorted.err | tail -1 > $O/r06_bench_standin_divergent_sorted.json
bash tools/ab_scene.sh scenes/many_point_lights.json 32 base basic4 basic2 2>&1 | tee $O/ab_mpl_occ.txt
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/r06_bench_driver20.json 2> $O/bench_driver20.err; tail -3 $O/bench_driver20.err
python - <<'PY'
import json
for f in ("r06_bench_many_point_lights","r06_bench_standin_divergent_sorted","r06_bench_driver20"):
    d=json.loads(open("gpurun_out/r06l/%s.json"%f).read().strip().splitlines()[-1]); r=d["roofline"]
    print(f, d["value"], r["kernel"], r["frac"], r["traffic"], r["measured_frac"], r.get("limiter_class"), (r.get("valu") or {}).get("issue_frac"), r.get("limiter"))
    for c in d.get("configs", []):
        rr=c["roofline"]; print("   ", c["name"][:28], c["value"], rr["kernel"], rr["bound"], rr["frac"], rr["traffic"], rr["measured_frac"], (rr.get("valu") or {}).get("issue_frac"))
PY
