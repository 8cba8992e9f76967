cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "rccl or quantised_node_records_vs" -n 4 > $O/pytest.log 2>&1; echo "rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 8 --warmup 4 --no-extra-configs --no-cpu-baseline --no-literal-config > $O/bench_forcedist.json 2> $O/bench_forcedist.err; tail -c 600 $O/bench_forcedist.json; tail -3 $O/bench_forcedist.err
IGNIS_CLI_FORCE_DIST=1 timeout 300 python -m ignis_amd.cli scenes/diamond_scene.json --spp 16 --width 256 --height 256 -o /tmp/o.exr --stats > $O/cli_forcedist.log 2>&1; tail -5 $O/cli_forcedist.log
for e in "-" "IGD_TAIL_THRESHOLD=0" "IGD_TAIL_THRESHOLD=262144"; do
  [ "$e" = "-" ] && e=""
  env $e bash tools/ab_scene.sh scenes/diamond_scene_principled.json 32 base 2>&1 | head -1 | sed "s/^/[$e] /"
done > $O/principled_tail.log; cat $O/principled_tail.log
