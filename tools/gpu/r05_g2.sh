cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
# 1. bench.py as the driver runs it (20 steps) with the configs array
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver20.json 2> $O/bench_driver20.err
# 2. kernel stats of the full-variant scenes
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_principled -o stats -- python bench.py --scene scenes/diamond_scene_principled.json --steps 32 --warmup 32 --no-cpu-baseline --no-literal-config > $O/bench_principled.json 2> $O/stats_principled.err
python tools/prof_summary.py stats "$(find $O/stats_principled -name '*.db' | head -1)" > $O/r05_rocprofv3_stats_principled.txt 2>> $O/summary.err
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > $O/standin_make.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_div -o stats -- python bench.py --scene /tmp/standin_1m_div/standin.json --steps 16 --warmup 16 --no-cpu-baseline --no-literal-config > $O/bench_standin_div.json 2> $O/stats_div.err
python tools/prof_summary.py stats "$(find $O/stats_div -name '*.db' | head -1)" > $O/r05_rocprofv3_stats_standin_divergent.txt 2>> $O/summary.err
find $O -name "*.db" -delete
tail -c 3000 $O/bench_driver20.json; cat $O/r05_rocprofv3_stats_principled.txt $O/r05_rocprofv3_stats_standin_divergent.txt
