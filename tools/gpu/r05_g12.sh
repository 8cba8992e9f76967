cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
bash tools/ab.sh 20 base kargs > $O/ab_kargs20.log 2>&1; cat $O/ab_kargs20.log
bash tools/ab.sh 64 base kargs > $O/ab_kargs64.log 2>&1; cat $O/ab_kargs64.log
IGD_LIBRARY=$GRAFT_REPO_ROOT/ignis_amd/lib/var/libig_device_hip_kargs.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -n 4 -k "oracle or golden or counters" 2>&1 | tail -2
