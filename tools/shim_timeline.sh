#!/bin/bash
# Timeline of the drop-in frame: rocprofv3 --kernel-trace --memory-copy-trace of examples/run_odometry_synth
# (the reference runner's loop over include/odometry_shim.hpp, host-resident frames), summarised per frame by tools/shim_timeline.py.
#   bash tools/shim_timeline.sh <tag> [n_frames]        (on the GPU box through gpurun; output under gpurun_out/<tag>/)
set -u
TAG=${1:-shim_tl}
N=${2:-60}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 - <<PY
import sys, numpy as np
sys.path.insert(0, "$ROOT")
from odometry_amd import synth
seq = synth.make_sequence($N, seed=0, drive="natural")
with open("/tmp/frames_tl.bin", "wb") as f:
    np.array([$N, synth.KITTI_ROWS, synth.KITTI_COLS], np.int32).tofile(f)
    for l, r in zip(seq["left"], seq["right"]):
        l.astype(np.float32).tofile(f); r.astype(np.float32).tofile(f)
PY
g++ -O2 -std=c++17 -I$ROOT/include $ROOT/examples/run_odometry_synth.cpp -o /tmp/run_odometry_synth -L$ROOT/odometry_amd/lib -lodometry_hip -Wl,-rpath,$ROOT/odometry_amd/lib
/tmp/run_odometry_synth /tmp/frames_tl.bin --time 3 > /dev/null 2> $OUT/plain.err
timeout -k 5 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -- /tmp/run_odometry_synth /tmp/frames_tl.bin --time 1 > /dev/null 2> $OUT/traced.err < /dev/null
K=$(find $OUT/trace -name "*_kernel_trace.csv" | head -1)
M=$(find $OUT/trace -name "*_memory_copy_trace.csv" | head -1)
python3 $ROOT/tools/shim_timeline.py "$K" "$M" $N > $OUT/timeline.md 2> $OUT/timeline.err
cat $OUT/plain.err | tail -2; tail -3 $OUT/traced.err; cat $OUT/timeline.md
find $OUT/trace -name "*.csv" -size +2M -delete
