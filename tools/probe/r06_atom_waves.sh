# waves per block of the wave-per-atom search (NVALCHEMIOPS_NL_ATOM_WAVES = 4 | 8 | 16): the reference's cell_list benchmark rows (where it is the working kernel), config 2, config 4 (where it is the idle one)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for w in 4 8 16 4 8 16; do
  export NVALCHEMIOPS_NL_ATOM_WAVES=$w
  python $R/bench.py --workload ref-nlist 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('waves $w ref-nlist', [round(x['median_ms'],4) for x in r['rows']])"
  for c in c2 c4; do python $R/bench.py --workload $c --steps 100 --cpu-sample 0 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.readlines()[-1]); print('waves $w $c', round(r['ms_per_step'],4))"; done
done
