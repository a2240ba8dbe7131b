cd $GRAFT_REPO_ROOT
FULL=1 ROUND=r04 bash tools/collect_profiles.sh
