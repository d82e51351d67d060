#!/bin/bash
# tuning aid: step time with / without the two-stream overlap and with a prioritised main stream
for o in 0 1; do for mp in "" -1; do
  echo -n "overlap $o main-priority '${mp}': "
  BENCH_MAIN_PRIORITY=$mp python bench.py --steps 20 --warmup 3 --cpu-sample 0 --overlap $o 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done; done
