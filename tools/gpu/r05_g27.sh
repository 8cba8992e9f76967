cd $GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -2
bash tools/ab_env.sh 20 "IGD_CLEAR_IN_GENERATE=0" "-" > $O/ab_clear_headline.log 2>&1; cat $O/ab_clear_headline.log
for e in "IGD_CLEAR_IN_GENERATE=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/many_point_lights.json 32 base 2>&1 | sed "s/^/[$e] /"; done
