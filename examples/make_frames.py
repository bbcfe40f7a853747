"""Writes N synthetic KITTI-shaped stereo frames (fp32) for examples/run_odometry_synth.cpp."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odometry_amd import synth  # noqa: E402

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 8
seq = synth.make_sequence(n, seed=0)
with open(path, "wb") as f:
    np.array([n, synth.KITTI_ROWS, synth.KITTI_COLS], np.int32).tofile(f)
    for l, r in zip(seq["left"], seq["right"]):
        l.astype(np.float32).tofile(f)
        r.astype(np.float32).tofile(f)
print("wrote", path)
