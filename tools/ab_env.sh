#!/bin/bash
# A/B of environment knobs on the headline: tools/ab_env.sh [rounds] "VAR=value" ["VAR2=value" ...]   (each knob against the default, interleaved)
R=${1:-3}; shift
for i in $(seq $R); do
  for kv in "X_DEFAULT=1" "$@"; do
    echo -n "$kv: "; env $kv timeout 200 python bench.py --cpu-frames 0 --no-extras --no-stress --no-causal --steps 199 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
  done
done
