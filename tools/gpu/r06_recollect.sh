cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > gpurun_out/r06/pytest_final.log 2>&1; grep -n "passed\|failed" gpurun_out/r06/pytest_final.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
bash tools/collect_profiles.sh r06 _many_point_lights --scene scenes/many_point_lights.json --steps 32 --warmup 32 > gpurun_out/r06_collect_mpl.log 2>&1
bash tools/collect_profiles.sh r06 _principled --scene scenes/diamond_scene_principled.json --steps 32 --warmup 32 > gpurun_out/r06_collect_principled.log 2>&1
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/collect_profiles.sh r06 _standin_divergent --scene /tmp/standin_1m_div/standin.json --steps 16 --warmup 16 > gpurun_out/r06_collect_standin_div.log 2>&1
python tools/kernel_resources.py ignis_amd/lib/libig_device_hip.so > gpurun_out/r06/r06_kernel_resources.txt 2>/dev/null
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r06/r06_bench_driver20.json 2> gpurun_out/r06/bench_driver20.err; tail -3 gpurun_out/r06/bench_driver20.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06/r06_bench_driver20.json").read().strip().splitlines()[-1]); r=d["roofline"]
print(d["value"], d["literal_config"]["value"], r["frac"], r["measured_frac"], r["limiter_class"], d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
for c in d["configs"]:
    rr=c["roofline"]; print("  ", c["name"][:28], c["value"], rr["kernel"], rr["bound"], rr["frac"], rr["measured_frac"], (rr.get("valu") or {}).get("issue_frac"), (rr.get("limiter") or {}).get("wave_wait_share"))
PY
