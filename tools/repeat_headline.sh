for i in $(seq 12); do timeout 200 python bench.py --cpu-frames 0 --no-extras --no-stress --steps 199 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['roofline']['persistent_launch']; print(d['value'], p['solves_redone_on_step_launches'], p['depth_jobs_redone_on_step_launches'], d['step_us']['max'], d['step_us']['slowest_step'])"
done
