#!/bin/bash
# Same-box A/B of library builds on one of the reference-configuration workloads:  tools/ab_ref.sh <rounds> <workload> <name1> <name2> ...
rounds=$1; shift; wl=$1; shift
L=nvalchemi-toolkit-ops_amd/nvalchemiops/lib
for r in $(seq $rounds); do for v in "$@"; do
  cp $L/alt_$v.so $L/libnvalchemiops_hip.so
  timeout 300 python bench.py --workload $wl 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('%-10s' % '$v', ' '.join('%d:%.4f' % (r['atoms'], r['median_ms']) for r in d['rows']))"
done; done
