import sys, time
sys.path.insert(0, '.')
import numpy as np
from odometry_amd import api, synth
import bench
print(bench.dense_1080p_leg(api, synth))
