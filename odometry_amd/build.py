"""Builds the in-tree HIP library (gfx950) that implements include/odometry_hip.h.

hipcc cross-compiles without a GPU. Flags that matter for parity:
  -ffp-contract=off                          every fp32 op rounds once (arithmetic spec)
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE fp32 divide / sqrt on the device
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "odometry_hip.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "kernels.hip.h"), os.path.join(_HERE, "csrc", "odo_math.h"), os.path.join(_HERE, "csrc", "tracker.hip.h"), os.path.join(_HERE, "csrc", "camera.hip.h"),
        os.path.join(_HERE, "csrc", "dense.hip.h"),
        os.path.join(os.path.dirname(_HERE), "include", "odometry_hip.h")]
LIB = os.path.join(_HERE, "lib", "libodometry_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-shared", "-fPIC"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    cmd = [HIPCC] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
