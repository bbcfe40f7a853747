for lib in odometry_amd/lib/libodometry_hip.so odometry_amd/lib/var_*.so; do
  export ODOMETRY_HIP_LIB=$PWD/$lib
  echo -n "$(basename $lib): "; python bench.py --cpu-frames 0 --no-stress --no-causal --extras batched --steps 199 --warmup 5 --details /tmp/d.json 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], [d.get('batched_s%d_fps'%k) for k in (1,2,4,8)])"
done
