cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
bash tools/collect_profiles.sh r05 "" --steps 256 --warmup 32 > gpurun_out/r05_collect_headline.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/r05_bench_driver20.json 2> gpurun_out/r05/bench_driver20.err
for n in 2 4 8; do timeout 600 python bench.py --steps 20 --warmup 5 --as-rank-of $n --no-cpu-baseline --no-literal-config --no-extra-configs 2>/dev/null | tail -1 > gpurun_out/r05/r05_bench_as_rank_of_$n.json; done
bash tools/trav_profile.sh r05 > gpurun_out/r05_travprof.log 2>&1
tail -c 600 gpurun_out/r05/r05_bench.json; echo; tail -c 400 gpurun_out/r05/r05_bench_driver20.json
