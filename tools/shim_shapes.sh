# The drop-in classes over host images in the runner shapes that matter, on one box:
#   stand-in Mat / cv::Mat (tests/stubs)  x  every frame preloaded in its own Mat / the two Mats of a frame refilled inside the loop
#   gpurun -- 'bash tools/shim_shapes.sh [frames] [passes] > gpurun_out/shim_shapes.txt 2>&1'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
N=${1:-200}; PASSES=${2:-3}
cd /tmp
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT")
from odometry_amd import synth
N=$N
seq = synth.make_sequence(N, seed=0, drive="natural")
with open("/tmp/frames_shapes.bin", "wb") as f:
    np.array([N, synth.KITTI_ROWS, synth.KITTI_COLS], np.int32).tofile(f)
    for l, r in zip(seq["left"], seq["right"]):
        l.astype(np.float32).tofile(f); r.astype(np.float32).tofile(f)
PY
L="-L$ROOT/odometry_amd/lib -lodometry_hip -Wl,-rpath,$ROOT/odometry_amd/lib"
g++ -O2 -std=c++17 -I$ROOT/include $ROOT/examples/run_odometry_synth.cpp -o /tmp/ros_std $L || exit 1
g++ -O2 -std=c++17 -DODOMETRY_SHIM_WITH_OPENCV -DODOMETRY_SHIM_WITH_EIGEN -I$ROOT/tests/stubs -I$ROOT/include $ROOT/examples/run_odometry_synth.cpp -o /tmp/ros_cv $L || exit 1
/tmp/ros_std /tmp/frames_shapes.bin --rel-bin /tmp/rel_ref.bin > /dev/null || exit 1
run() {  # label, exe, extra args, env...
  local label=$1 exe=$2 extra=$3; shift 3
  env "$@" $exe /tmp/frames_shapes.bin $extra --rel-bin /tmp/rel.bin > /dev/null 2>&1
  if cmp -s /tmp/rel.bin /tmp/rel_ref.bin; then same=bit-identical; else same=POSES-DIFFER; fi
  for i in 1 2; do
    echo -n "$label [$same] "; env "$@" $exe /tmp/frames_shapes.bin $extra --time $PASSES 2>&1 >/dev/null | grep -E "SHIM_FPS|SHIM_MISMATCH" | tr '\n' ' '; echo
  done
}
run "stand-in preloaded           " /tmp/ros_std "" X=1
run "stand-in load-per-frame      " /tmp/ros_std "--load-per-frame" X=1
run "stand-in preloaded, no ahead " /tmp/ros_std "" ODOMETRY_SHIM_NO_LOOKAHEAD=1
run "cv::Mat preloaded            " /tmp/ros_cv "" X=1
run "cv::Mat load-per-frame       " /tmp/ros_cv "--load-per-frame" X=1
run "cv::Mat load-per-frame, swapped outputs" /tmp/ros_cv "--load-per-frame" ODOMETRY_SHIM_SWAP_OUTPUTS=1
run "cv::Mat load-per-frame, lazy " /tmp/ros_cv "--load-per-frame" ODOMETRY_SHIM_LAZY_OUTPUTS=1
run "cv::Mat preloaded, lazy      " /tmp/ros_cv "" ODOMETRY_SHIM_LAZY_OUTPUTS=1
M="ODOMETRY_SHIM_TUNE_MALLOC=1"   # opt-in: glibc keeps the pages of the runner's per-frame Mats (default thresholds: lost every frame)
run "cv::Mat load-per-frame, tuned malloc      " /tmp/ros_cv "--load-per-frame" $M
run "cv::Mat load-per-frame, tuned malloc + swapped outputs" /tmp/ros_cv "--load-per-frame" $M ODOMETRY_SHIM_SWAP_OUTPUTS=1
run "cv::Mat preloaded, tuned malloc           " /tmp/ros_cv "" $M
run "cv::Mat preloaded, tuned malloc + swapped outputs" /tmp/ros_cv "" $M ODOMETRY_SHIM_SWAP_OUTPUTS=1
run "cv::Mat load-per-frame, lazy, tuned malloc" /tmp/ros_cv "--load-per-frame" ODOMETRY_SHIM_LAZY_OUTPUTS=1 $M
run "cv::Mat load-per-frame, no ahead" /tmp/ros_cv "--load-per-frame" ODOMETRY_SHIM_NO_LOOKAHEAD=1
run "cv::Mat load-per-frame, verify  " /tmp/ros_cv "--load-per-frame" ODOMETRY_SHIM_VERIFY_MIRRORS=1
echo "== /tmp/ros_cv --load-per-frame, tuned malloc"; env $M ODO_RUNNER_PHASES=1 /tmp/ros_cv /tmp/frames_shapes.bin --load-per-frame --time 1 2>&1 >/dev/null | grep -E "runner phases|SHIM_STATS" | tail -2
echo "== /tmp/ros_cv, tuned malloc"; env $M ODO_RUNNER_PHASES=1 /tmp/ros_cv /tmp/frames_shapes.bin --time 1 2>&1 >/dev/null | grep -E "runner phases|SHIM_STATS" | tail -2
for exe in /tmp/ros_std /tmp/ros_cv; do
  for extra in "" "--load-per-frame"; do
    echo "== $exe $extra"; ODO_RUNNER_PHASES=1 $exe /tmp/frames_shapes.bin $extra --time 1 2>&1 >/dev/null | grep -E "runner phases|SHIM_STATS" | tail -3
  done
done
