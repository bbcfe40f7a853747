import sys
sys.path.insert(0, '.')
import numpy as np
from odometry_amd import api, synth
import bench
seq = synth.make_sequence(1, seed=0)
print(bench.disparity_leg(api, seq, None))
