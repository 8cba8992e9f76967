cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "standin or principled or blend or plastic or quantised" 2>&1 | tail -2
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base > $O/ab_principled.log 2>&1; cat $O/ab_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base > $O/ab_standin.log 2>&1; cat $O/ab_standin.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/stats_principled -o stats -- python bench.py --scene scenes/diamond_scene_principled.json --steps 32 --warmup 32 --no-cpu-baseline --no-literal-config > /dev/null 2> $O/stats.err
python tools/prof_summary.py stats "$(find $O/stats_principled -name '*.db' | head -1)" | grep "k_bin\|k_shade" ; find $O -name "*.db" -delete
