# Soak of the drop-in runner: many passes of the 200-frame drive in one process, every pass compared with the first (SHIM_MISMATCH on a
# difference), in the shapes that matter.   gpurun -- 'bash tools/shim_soak.sh [passes=100] > gpurun_out/shim_soak.txt 2>&1'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
P=${1:-100}
cd /tmp
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT")
from odometry_amd import synth
N=200
seq = synth.make_sequence(N, seed=0, drive="natural")
with open("/tmp/frames_soak.bin", "wb") as f:
    np.array([N, synth.KITTI_ROWS, synth.KITTI_COLS], np.int32).tofile(f)
    for l, r in zip(seq["left"], seq["right"]):
        l.astype(np.float32).tofile(f); r.astype(np.float32).tofile(f)
PY
L="-L$ROOT/odometry_amd/lib -lodometry_hip -Wl,-rpath,$ROOT/odometry_amd/lib"
g++ -O2 -std=c++17 -I$ROOT/include $ROOT/examples/run_odometry_synth.cpp -o /tmp/ros_std $L || exit 1
g++ -O2 -std=c++17 -DODOMETRY_SHIM_WITH_OPENCV -DODOMETRY_SHIM_WITH_EIGEN -I$ROOT/tests/stubs -I$ROOT/include $ROOT/examples/run_odometry_synth.cpp -o /tmp/ros_cv $L || exit 1
for cfg in "ros_std --load-per-frame" "ros_cv --load-per-frame" "ros_cv" "ros_std"; do
  set -- $cfg
  echo -n "$cfg, $P passes: "; timeout 1200 /tmp/$1 /tmp/frames_soak.bin ${2:-} --time $P 2>&1 >/dev/null | grep -E "SHIM_FPS|SHIM_MISMATCH|SHIM_STATS" | tr '\n' ' '; echo "rc=$?"
done
echo -n "ros_cv --load-per-frame, lazy outputs: "; ODOMETRY_SHIM_LAZY_OUTPUTS=1 timeout 1200 /tmp/ros_cv /tmp/frames_soak.bin --load-per-frame --time $P 2>&1 >/dev/null | grep -E "SHIM_FPS|SHIM_MISMATCH" | tr '\n' ' '; echo
echo -n "ros_cv --load-per-frame, verify mirrors, 10 passes: "; ODOMETRY_SHIM_VERIFY_MIRRORS=1 timeout 1200 /tmp/ros_cv /tmp/frames_soak.bin --load-per-frame --time 10 2>&1 >/dev/null | grep -E "SHIM_FPS|SHIM_MISMATCH|SHIM_STATS" | tr '\n' ' '; echo
