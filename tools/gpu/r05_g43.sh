cd $GRAFT_REPO_ROOT
bash tools/trav_timeline.sh
awk 'NR%8==1' gpurun_out/roundtrace/timeline_sizes.txt | grep closest | tail -6 | cut -c1-420
echo; head -24 gpurun_out/roundtrace/timeline_rankof8.txt | cut -c1-420
