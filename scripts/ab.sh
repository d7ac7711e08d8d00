cd $GRAFT_REPO_ROOT
for V in "PCT_LDS_SORT=0" "PCT_LDS_SORT=1" "PCT_LDS_SORT=1 PCT_GRID_COOP=0" "PCT_LDS_SORT=0" "PCT_LDS_SORT=1"; do
  echo "== $V"; env $V python3 scripts/probe.py grid 2>&1 | grep -E "Q= 1048576|Q=   65536" | grep -E "ppc= 2.0 shift=1"
done
