#!/bin/bash
# The driver's short run repeated: frames/s, Solves / depth jobs redone, the first and the slowest timed step.
#   gpurun -- 'bash tools/repeat_headline.sh [runs=6] [steps=20]'
N=${1:-6}; K=${2:-20}
for i in $(seq $N); do timeout 200 python bench.py --cpu-frames 0 --no-extras --details /tmp/repeat_details.json --steps $K --warmup 5 >/dev/null 2>&1; python -c "
import sys, json
d = json.load(open('/tmp/repeat_details.json')); p=d['roofline']['persistent_launch']; s=d['step_us']
print(d['value'], d['ms_per_step'], 'redone', p['solves_redone_on_step_launches'], p['depth_jobs_redone_on_step_launches'], 'first step', s['first_step'], 'median', s['median'], 'max', s['max'], 'at', s['slowest_step'])"
done
