#!/bin/bash
# A/B 9: launch floor of the public ops on small systems; tile pipeline (6 launches) vs atomic spread + per-atom gather (2 launches) below a threshold.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for thr in tile atomic auto; do
  echo "== NVALCHEMIOPS_SPREAD_PATH=$thr"
  NVALCHEMIOPS_SPREAD_PATH=$thr python tools/floor_bench.py 2>&1 | grep "^N="
done
