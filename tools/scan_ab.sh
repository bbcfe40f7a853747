# A/B of the epipolar scan on one box: wave-per-slot kernel against the row-workgroup kernel (ODO_SCAN_ROWS / ODO_SCAN_SPLIT), each in
# the shipped build and in a build of the main translation unit with -fno-slp-vectorize (packed fp32 math off), if
# odometry_amd/lib/libodometry_hip_noslp.so exists.   gpurun -- 'bash tools/scan_ab.sh > gpurun_out/scan_ab.txt 2>&1'
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for lib in odometry_amd/lib/libodometry_hip.so odometry_amd/lib/libodometry_hip_noslp.so; do
  [ -f $lib ] || continue
  for cfg in "ODO_SCAN_ROWS=0" "ODO_SCAN_ROWS=1 ODO_SCAN_SPLIT=1" "ODO_SCAN_ROWS=1 ODO_SCAN_SPLIT=2" "ODO_SCAN_ROWS=1 ODO_SCAN_SPLIT=3" "ODO_SCAN_ROWS=1 ODO_SCAN_SPLIT=4"; do
    for drive in natural dense; do
      echo -n "$(basename $lib) $cfg $drive: "; env ODOMETRY_HIP_LIB=$ROOT/$lib $cfg timeout 120 python3 tools/scan_probe.py $drive 2>&1 | tail -1
    done
  done
done
