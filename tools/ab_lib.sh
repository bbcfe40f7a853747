for i in 1 2 3 4 5; do
for v in base new; do
  if [ $v = base ]; then export ODOMETRY_HIP_LIB=$PWD/odometry_amd/lib/ab_base.so; else unset ODOMETRY_HIP_LIB; fi
  echo -n "$v: "; timeout 200 python bench.py --cpu-frames 0 --no-extras --no-stress --steps 199 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
