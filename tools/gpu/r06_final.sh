# Runs ON the GPU box: the round's committed evidence (profiles/r06_*), collected by tools/collect_profiles.sh per workload.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
bash tools/collect_profiles.sh r06 "" --steps 256 --warmup 32 > gpurun_out/r06_collect_headline.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/r06_bench_driver20.json 2> gpurun_out/r06/bench_driver20.err
for n in 2 4 8; do timeout 600 python bench.py --steps 20 --warmup 5 --as-rank-of $n --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 > gpurun_out/r06/r06_bench_as_rank_of_$n.json; done
bash tools/collect_profiles.sh r06 _many_point_lights --scene scenes/many_point_lights.json --steps 32 --warmup 32 > gpurun_out/r06_collect_mpl.log 2>&1
bash tools/collect_profiles.sh r06 _principled --scene scenes/diamond_scene_principled.json --steps 32 --warmup 32 > gpurun_out/r06_collect_principled.log 2>&1
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/collect_profiles.sh r06 _standin_divergent --scene /tmp/standin_1m_div/standin.json --steps 16 --warmup 16 > gpurun_out/r06_collect_standin_div.log 2>&1
IGD_RAY_SORT=1 bash tools/collect_profiles.sh r06 _standin_divergent_sorted --scene /tmp/standin_1m_div/standin.json --steps 16 --warmup 16 > gpurun_out/r06_collect_standin_div_sorted.log 2>&1
bash tools/run_standin.sh r06 16000000 16 lean _standin > gpurun_out/r06_collect_standin.log 2>&1
bash tools/trav_profile.sh r06 > gpurun_out/r06_travprof.log 2>&1
python tools/kernel_resources.py ignis_amd/lib/libig_device_hip.so > gpurun_out/r06/r06_kernel_resources.txt 2>/dev/null
tail -c 600 gpurun_out/r06/r06_bench.json; echo; tail -c 400 gpurun_out/r06/r06_bench_driver20.json
