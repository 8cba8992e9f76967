cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
bash tools/ab.sh 20 base lean3 lean5 > $O/ab_lean_occ.log 2>&1; cat $O/ab_lean_occ.log
bash tools/ab_env.sh 20 "IGD_SHADE_GRID=16" "IGD_SHADE_GRID=32" "IGD_SHADE_GRID=128" "-" > $O/ab_shade_grid.log 2>&1; cat $O/ab_shade_grid.log
