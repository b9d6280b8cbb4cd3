#!/bin/bash
# Same-box A/B of one environment switch on the default bench: tools/ab_env.sh VAR [workload] (two alternations, 30 steps each).
set -e
VAR=$1; WL=${2:-xattn60m}
B="python bench.py --workload $WL --no-cpu-baseline --no-roofline --no-also --steps 30 --warmup 5"
for i in 1 2; do
  for v in 0 1; do
    env $VAR=$v timeout -k 10 300 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', d['ms_per_step'])"
  done
done
