cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base basic4 2>&1 | tee $O/ab_basic4.txt
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 base basic4 2>&1 | tee -a $O/ab_basic4.txt
for s in cbox/cbox.json; do [ -f scenes/$s ] && bash tools/ab_scene.sh scenes/$s 32 base basic4 2>&1 | tee -a $O/ab_basic4.txt; done
