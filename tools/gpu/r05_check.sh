cd $GRAFT_REPO_ROOT
O=gpurun_out/r05z; mkdir -p $O
( time timeout 3000 python -m pytest tests -x -q -m gpu ) > $O/pytest_seq.log 2>&1; echo "rc=$?" >> $O/pytest_seq.log; tail -6 $O/pytest_seq.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; tail -4 $O/bench_default.err
