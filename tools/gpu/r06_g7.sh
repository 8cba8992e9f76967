cd $GRAFT_REPO_ROOT
O=gpurun_out/r06g; mkdir -p $O
bash tools/ab.sh 20 base nosortidx 2>&1 | tee $O/ab_sortidx.txt
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "schedule or reproducible or spheres" ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
