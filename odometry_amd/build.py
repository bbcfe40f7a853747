"""Builds the in-tree HIP library (gfx950) that implements include/odometry_hip.h.

hipcc cross-compiles without a GPU. Flags that matter for parity:
  -ffp-contract=off                          every fp32 op rounds once (arithmetic spec)
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE fp32 divide / sqrt on the device
and one that matters for speed:
  -fno-slp-vectorize                         no packed fp32 (v_pk_mul_f32 / v_pk_add_f32 issue at 8 cycles against 3 for the scalar
                                             forms on gfx950, tools/microbench/valu_rates.hip, and need their operands in register
                                             pairs): lm_fine_kernel 208 -> 168 VGPRs, the pose-LM chain 4 % shorter (round 5)
  -mllvm -amdgpu-sched-strategy=iterative-ilp
                                             the LM state machine is ONE wave working through ~1 200 dependent instructions per
                                             evaluation: the ILP-first list scheduler orders them 3.5 % faster over the whole frame than
                                             the default occupancy-first one (3 465 -> 3 595 frames/s over 199 steps; max-ilp: - 0.5 %,
                                             iterative-minreg: - 2 %, iterative-maxocc: + 1.3 %, -O2: +- 0; tools/ab_lib.sh). Batched
                                             tracker: S = 2 + 1 %, S = 4 - 1.8 %, S = 8 - 0.6 % (lm_fine_kernel_batch 208 -> 229 VGPRs)
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "odometry_hip.hip")
SRC_DENSE = os.path.join(_HERE, "csrc", "dense_kernels.hip")   # its own translation unit (see dense.hip.h)
SRC_BATCH = os.path.join(_HERE, "csrc", "lm_batch_kernels.hip")   # the batched LM kernels: their own unit AND their own scheduler
DEPS = [SRC, SRC_DENSE, SRC_BATCH, os.path.join(_HERE, "csrc", "kernels.hip.h"), os.path.join(_HERE, "csrc", "odo_math.h"), os.path.join(_HERE, "csrc", "tracker.hip.h"), os.path.join(_HERE, "csrc", "batch.hip.h"), os.path.join(_HERE, "csrc", "gather.hip.h"), os.path.join(_HERE, "csrc", "camera.hip.h"),
        os.path.join(_HERE, "csrc", "dense.hip.h"), os.path.join(_HERE, "csrc", "host_fp.h"),
        os.path.join(os.path.dirname(_HERE), "include", "odometry_hip.h"), os.path.abspath(__file__)]   # (this file: the flags)
LIB = os.path.join(_HERE, "lib", "libodometry_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-mllvm",
         "-amdgpu-sched-strategy=" + os.environ.get("ODO_SCHED", "iterative-ilp"), "-fPIC"] + os.environ.get("ODO_EXTRA_HIPCC_FLAGS", "").split()   # A/B builds
# lm_batch_kernels.hip: the occupancy-first list scheduler. Round 6, batched tracker at S = 1 / 2 / 4 / 8 (frames/s, two runs):
#   iterative-ilp 3 340-3 380 / 5 820-5 860 / 9 090-9 100 / 13 970-14 040 | default 3 690 / 6 500-6 520 / 10 510-10 630 / 13 820-13 840 |
#   iterative-maxocc 3 680-3 690 / 6 480-6 490 / 10 390-10 410 / 14 290-14 430 — and the single tracker's headline under each of them
#   3 772-3 779 | 3 664-3 736 | 3 725-3 730: hence two units.
SCHED_BATCH = os.environ.get("ODO_SCHED_BATCH", "iterative-maxocc")


SCHED_DENSE = os.environ.get("ODO_SCHED_DENSE", "")   # dense_kernels.hip: "" = the main unit's


def _flags_for(src, flags):
    sched = SCHED_BATCH if src == SRC_BATCH else SCHED_DENSE if src == SRC_DENSE else ""
    if not sched:
        return flags
    if sched == "default":   # the compiler's own choice: drop the option (and the -mllvm in front of it)
        out = []
        for f in flags:
            if f.startswith("-amdgpu-sched-strategy="):
                out.pop()
                continue
            out.append(f)
        return out
    return [("-amdgpu-sched-strategy=" + sched) if f.startswith("-amdgpu-sched-strategy=") else f for f in flags]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


LIB_STAMPS = os.path.join(_HERE, "lib", "libodometry_hip_stamps.so")   # diagnostic build: phase stamps compiled in (kernels.hip.h ODO_DBG)


def build(force=False, verbose=False, stamps=False):
    """stamps=True: the diagnostic twin lib/libodometry_hip_stamps.so (-DODO_PHASE_STAMPS=1; use it through ODOMETRY_HIP_LIB with
    ODO_COARSE_STAMPS=1 / ODO_DEPTH_STAMPS=1). The product library has every stamp compiled out."""
    if not stamps and not force and not needs_build():
        return LIB
    lib = LIB_STAMPS if stamps else LIB
    flags = FLAGS + (["-DODO_PHASE_STAMPS=1"] if stamps else [])
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    objdir = os.path.join(os.path.dirname(lib), "obj_stamps" if stamps else "obj")
    os.makedirs(objdir, exist_ok=True)
    o_main, o_dense, o_batch = (os.path.join(objdir, n) for n in ("odometry_hip.o", "dense_kernels.o", "lm_batch_kernels.o"))
    cmds = [[HIPCC] + _flags_for(SRC, flags) + ["-c", "-o", o_main, SRC],
            [HIPCC] + _flags_for(SRC_DENSE, flags) + ["-c", "-o", o_dense, SRC_DENSE],
            [HIPCC] + _flags_for(SRC_BATCH, flags) + ["-c", "-o", o_batch, SRC_BATCH],
            [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, o_main, o_dense, o_batch]]
    procs = []
    for cmd in cmds[:3]:   # the three translation units compile side by side
        if verbose:
            print(" ".join(cmd))
        procs.append(subprocess.Popen(cmd))
    for pr, cmd in zip(procs, cmds[:3]):
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    if verbose:
        print(" ".join(cmds[3]))
    subprocess.check_call(cmds[3])
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, stamps="--stamps" in sys.argv))
