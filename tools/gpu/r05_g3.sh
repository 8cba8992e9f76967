cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r05 _principled --scene scenes/diamond_scene_principled.json --steps 32 --warmup 32 > gpurun_out/r05_collect_principled.log 2>&1
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/collect_profiles.sh r05 _standin_divergent --scene /tmp/standin_1m_div/standin.json --steps 16 --warmup 16 > gpurun_out/r05_collect_standin_div.log 2>&1
tail -5 gpurun_out/r05_collect_principled.log gpurun_out/r05_collect_standin_div.log
