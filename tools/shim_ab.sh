set -u
ROOT=$GRAFT_REPO_ROOT
cd /tmp
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT")
from odometry_amd import synth
N=200
seq = synth.make_sequence(N, seed=0, drive="natural")
with open("/tmp/frames_ab.bin", "wb") as f:
    np.array([N, synth.KITTI_ROWS, synth.KITTI_COLS], np.int32).tofile(f)
    for l, r in zip(seq["left"], seq["right"]):
        l.astype(np.float32).tofile(f); r.astype(np.float32).tofile(f)
PY
g++ -O2 -std=c++17 -I$ROOT/include $ROOT/examples/run_odometry_synth.cpp -o /tmp/ros -L$ROOT/odometry_amd/lib -lodometry_hip -Wl,-rpath,$ROOT/odometry_amd/lib
for i in 1 2 3; do
  echo -n "ahead pyramid on:  "; /tmp/ros /tmp/frames_ab.bin --time 3 2>&1 >/dev/null | grep SHIM_FPS
  echo -n "ahead pyramid off: "; ODOMETRY_SHIM_NO_AHEAD_PYRAMID=1 /tmp/ros /tmp/frames_ab.bin --time 3 2>&1 >/dev/null | grep SHIM_FPS
done
echo -n "lookahead off:     "; ODOMETRY_SHIM_NO_LOOKAHEAD=1 /tmp/ros /tmp/frames_ab.bin --time 3 2>&1 >/dev/null | grep SHIM_FPS
ODO_RUNNER_PHASES=1 /tmp/ros /tmp/frames_ab.bin --time 1 2>&1 >/dev/null | tail -2
