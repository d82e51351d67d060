#!/bin/bash
# A/B 10: which spread / gather path for which (mesh, atom count)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for thr in tile atomic auto; do
  echo "== NVALCHEMIOPS_SPREAD_PATH=$thr"
  NVALCHEMIOPS_SPREAD_PATH=$thr python tools/probe/spread_path_sweep.py 2>&1 | grep "^mesh"
done
