cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
( time timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/bench_driver20.json 2> $O/bench_driver20.err; tail -c 600 $O/bench_driver20.json; tail -5 $O/bench_driver20.err
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
