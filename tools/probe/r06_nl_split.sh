# work items per cell of the tiled search (NVALCHEMIOPS_NL_SPLIT: 1 = one item per cell as before, unset = device policy): config 4, config 2, headline
R=${GRAFT_REPO_ROOT:-$(pwd)}
one() { python $R/bench.py "$@" --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); k=r['kernels']
print('   ', round(r['ms_per_step'],4), {n:round(v.get('isolated_median_ms') or v['avg_ms_timed_region'],4) for n,v in k.items() if n.startswith('nl_query')})"; }
for s in 1 "" 2 3 4 6; do
  echo "split=${s:-auto}"
  export NVALCHEMIOPS_NL_SPLIT=$s; [ -z "$s" ] && unset NVALCHEMIOPS_NL_SPLIT
  echo "  c4"; one --workload c4 --steps 100
  echo "  c2"; one --workload c2 --steps 100
  echo "  headline"; one --processes 1 --steps 50
done
