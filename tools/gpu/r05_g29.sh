cd $GRAFT_REPO_ROOT
O=gpurun_out/r05ab; mkdir -p $O
bash tools/ab.sh 20 nodefer base > $O/ab_defer2_headline.log 2>&1; cat $O/ab_defer2_headline.log
python tools/make_standin_scene.py /tmp/standin_1m_div --triangles 1000000 --instances 96 --materials divergent > /dev/null 2>&1
bash tools/ab_scene.sh /tmp/standin_1m_div/standin.json 16 nodefer base > $O/ab_defer2_standin.log 2>&1; cat $O/ab_defer2_standin.log
timeout 1500 python -m pytest tests -x -q -m gpu -n 4 2>&1 | tail -3
