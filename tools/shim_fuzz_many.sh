# Many seeds of tests/shim_fuzz_harness.cpp: stand-in build without look-ahead against the cv::Mat build (look-ahead on; every tenth seed
# also with mirror verification).   gpurun -- 'bash tools/shim_fuzz_many.sh [first seed] [last seed] [operations]'
set -u
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
A=${1:-10}; B=${2:-60}; OPS=${3:-150}
cd /tmp
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT")
from odometry_amd import synth
seq = synth.make_sequence(6, seed=4)
with open("/tmp/frames_fuzz.bin", "wb") as f:
    np.array([6, synth.KITTI_ROWS, synth.KITTI_COLS], np.int32).tofile(f)
    for l, r in zip(seq["left"], seq["right"]):
        l.astype(np.float32).tofile(f); r.astype(np.float32).tofile(f)
PY
L="-L$ROOT/odometry_amd/lib -lodometry_hip -Wl,-rpath,$ROOT/odometry_amd/lib"
g++ -O2 -std=c++17 -I$ROOT/include $ROOT/tests/shim_fuzz_harness.cpp -o /tmp/fz_std $L || exit 1
g++ -O2 -std=c++17 -DODOMETRY_SHIM_WITH_OPENCV -DODOMETRY_SHIM_WITH_EIGEN -I$ROOT/tests/stubs -I$ROOT/include $ROOT/tests/shim_fuzz_harness.cpp -o /tmp/fz_cv $L || exit 1
bad=0
for s in $(seq $A $B); do
  ODOMETRY_SHIM_NO_LOOKAHEAD=1 timeout 120 /tmp/fz_std /tmp/frames_fuzz.bin $s $OPS > /tmp/fz_want.txt 2>/dev/null || { echo "seed $s: reference run failed"; bad=1; continue; }
  timeout 120 /tmp/fz_cv /tmp/frames_fuzz.bin $s $OPS > /tmp/fz_got.txt 2>/tmp/fz_err.txt || { echo "seed $s: cv run failed"; bad=1; continue; }
  cmp -s /tmp/fz_want.txt /tmp/fz_got.txt || { echo "seed $s: DIFFERENT"; diff /tmp/fz_want.txt /tmp/fz_got.txt | head -4; bad=1; }
  if [ $((s % 10)) = 0 ]; then
    ODOMETRY_SHIM_VERIFY_MIRRORS=1 timeout 300 /tmp/fz_cv /tmp/frames_fuzz.bin $s $OPS > /tmp/fz_got.txt 2>/tmp/fz_err.txt
    cmp -s /tmp/fz_want.txt /tmp/fz_got.txt || { echo "seed $s (verify): DIFFERENT"; bad=1; }
    grep -o "verify_failures [0-9]*" /tmp/fz_err.txt | sed "s/^/seed $s: /"
  fi
done
echo "seeds $A..$B x $OPS operations: $([ $bad = 0 ] && echo all identical || echo FAILURES)"
