#!/bin/bash
# A/B of builds of the library on the dense 1080p leg (configs[2]): frames/s, level-0 launch us, frac — shipped library against var_*.so
for i in $(seq ${1:-2}); do
for lib in odometry_amd/lib/libodometry_hip.so odometry_amd/lib/var_*.so; do
  [ -f "$lib" ] || continue
  export ODOMETRY_HIP_LIB=$PWD/$lib
  echo -n "$(basename $lib): "; python bench.py --cpu-frames 0 --no-stress --no-causal --no-child-processes --extras dense --steps 20 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('dense_1080p_fps'), d.get('dense_1080p_launch_us'), d.get('dense_1080p_frac'))"
done; done
