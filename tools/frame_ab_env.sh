#!/bin/bash
# same-box A/B of the default bench line under an environment switch: tools/frame_ab_env.sh VAR   (VAR=0 vs default, interleaved twice)
V=$1; shift
for r in 1 2; do
  for val in ${VALS:-0 1}; do
    env $V=$val python bench.py --no-configs --no-extras --no-cpu-baseline --steps 16 --warmup 3 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('$V=$val', d['value'], 'frames/s', d['ms_per_step'], 'ms', 'utd5', d['roofline'].get('avg_ms'))"
  done
done
