export TMPDIR=/tmp BENCH_CALIB=0
for r in 1 2 3; do for o in 1 3 2; do
timeout 300 python bench.py --processes 1 --steps 40 --warmup 5 --cpu-sample 0 --overlap $o 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('overlap $o step %.3f median %.3f serial %.3f  e_pme %.6f' % (d['ms_per_step'], d['stats']['step_ms_median'], d['stats']['step_ms_median_serial_untimed'], d['energies']['e_pme']))"
done; done
