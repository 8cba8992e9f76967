cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
bash tools/ab_env.sh 20 "-" "IGD_TAIL_WIDE8=8" 2>&1 | tee $O/ab_wide8_headline.txt
bash tools/ab_env.sh 256 "-" "IGD_TAIL_WIDE8=8" 2>&1 | tee -a $O/ab_wide8_headline.txt
for W in 0 8; do for rep in 1 2 3; do
IGD_TAIL_WIDE8=$W timeout 300 python bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-extra-configs --no-live-traffic 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('IGD_TAIL_WIDE8=$W literal config 2: %.1f Mrays/s' % d['literal_config']['value'])"
done; done 2>&1 | tee -a $O/ab_wide8_headline.txt
