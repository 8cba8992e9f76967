cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
D=/tmp/standin_1000000_divergent
python tools/make_standin_scene.py $D --triangles 1000000 --materials divergent > $O/make.log 2>&1
timeout 600 python tools/bench_ray_order.py $D/standin.json 1920 1080 4 2>&1 | tee $O/ray_order_1m.txt
D=/tmp/standin_16000000_lean
python tools/make_standin_scene.py $D --triangles 16000000 --materials lean > $O/make16.log 2>&1
timeout 900 python tools/bench_ray_order.py $D/standin.json 1920 1080 4 2>&1 | tee $O/ray_order_16m.txt
