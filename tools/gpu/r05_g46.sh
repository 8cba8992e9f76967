cd $GRAFT_REPO_ROOT
IGD_TAIL_DEBUG=1 python bench.py --steps 20 --warmup 5 --as-rank-of 8 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>&1 | grep "\[tail\]" | tail -12
IGD_TAIL_DEBUG=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-literal-config --no-extra-configs --no-live-traffic 2>&1 | grep "\[tail\]" | tail -12
