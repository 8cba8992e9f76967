cd $GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -2
bash tools/ab_env.sh 20 "IGD_HIT_PACK=0" "-" > $O/ab_pack_headline.log 2>&1; cat $O/ab_pack_headline.log
for e in "IGD_HIT_PACK=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base 2>&1 | sed "s/^/[$e] /"; done > $O/ab_pack_principled.log; cat $O/ab_pack_principled.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
for e in "IGD_HIT_PACK=0" "-"; do E=$e; [ "$e" = "-" ] && E=""; env $E bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base 2>&1 | sed "s/^/[$e] /"; done > $O/ab_pack_standin.log; cat $O/ab_pack_standin.log
