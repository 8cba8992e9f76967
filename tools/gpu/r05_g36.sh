cd $GRAFT_REPO_ROOT
OUT=gpurun_out/roundtrace; mkdir -p $OUT
IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_mb256.so python tools/traverse_sizes.py 23 2>&1 | grep -v "^any" | tail -9
for v in base shk2 shk8; do
  if [ $v = base ]; then unset IGD_LIBRARY; else export IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_$v.so; fi
  TRACE_NAME=shade_$v timeout 300 bash tools/round_trace.sh
  echo "== $v"; grep -m3 "k_shade" $OUT/shade_$v.txt
done
