#!/bin/bash
# tuning aid: step time for the stream schedules of bench.py --processes 1 (0 = one stream, 1 = both branches together, 2 = PME after the D3 list)
for o in 0 1 2 1 2; do
  echo -n "overlap $o: "
  python bench.py --processes 1 --steps 20 --warmup 3 --cpu-sample 0 --overlap $o 2>/dev/null | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done
