cd $GRAFT_REPO_ROOT
O=gpurun_out/r05s; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -2
bash tools/ab_env.sh 20 "IGD_CAMERA_COMPACT=0" "-" > $O/ab_compact_headline.log 2>&1; cat $O/ab_compact_headline.log
for e in "IGD_CAMERA_COMPACT=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base 2>&1 | sed "s/^/[$e] /"; done; 
for e in "IGD_CAMERA_COMPACT=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/many_point_lights.json 32 base 2>&1 | sed "s/^/[$e] /"; done
