#!/bin/bash
# developer tool: C5 step time for a few settings of the walk kernel's tuning knobs
for W in 3 4 5 6 7; do for B in 8 16 32; do
  echo -n "wgs_per_cu=$W batch=$B: "
  GSL_SINTERP_WALK_WGS_PER_CU=$W GSL_SINTERP_WALK_BATCH=$B timeout -k 10 120 python bench.py --config C5 --only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(d['ms_per_step'])"
done; done
