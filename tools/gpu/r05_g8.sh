cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
( time timeout 2400 python -m pytest tests -x -q -m gpu -n 4 ) > $O/pytest_all.log 2>&1; echo "rc=$?" >> $O/pytest_all.log; tail -8 $O/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
bash tools/ab.sh 20 base > $O/ab_headline20.log 2>&1; cat $O/ab_headline20.log
