#!/bin/bash
# A/B of builds of the library on the headline drive: the shipped odometry_amd/lib/libodometry_hip.so against every
# odometry_amd/lib/var_*.so (built with ODO_EXTRA_HIPCC_FLAGS=... python -m odometry_amd.build and copied there), interleaved.
#   tools/ab_lib.sh [rounds=3]        (on the GPU box: gpurun -- 'bash tools/ab_lib.sh')
R=${1:-3}
for i in $(seq $R); do
for lib in odometry_amd/lib/libodometry_hip.so odometry_amd/lib/var_*.so; do
  [ -f "$lib" ] || continue
  export ODOMETRY_HIP_LIB=$PWD/$lib
  echo -n "$(basename $lib): "; timeout 200 python bench.py --cpu-frames 0 --no-extras --no-stress --steps 199 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
done; done
