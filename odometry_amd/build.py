"""Builds the in-tree HIP library (gfx950) that implements include/odometry_hip.h.

hipcc cross-compiles without a GPU. Flags that matter for parity:
  -ffp-contract=off                          every fp32 op rounds once (arithmetic spec)
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE fp32 divide / sqrt on the device
and one that matters for speed:
  -fno-slp-vectorize                         no packed fp32 (v_pk_mul_f32 / v_pk_add_f32 issue at 8 cycles against 3 for the scalar
                                             forms on gfx950, tools/microbench/valu_rates.hip, and need their operands in register
                                             pairs): lm_fine_kernel 208 -> 168 VGPRs, the pose-LM chain 4 % shorter (round 5)
  -mllvm -amdgpu-sched-strategy=...          per translation unit (round 6, see SCHED_CHAIN below): iterative-ilp for the single tracker's
                                             LM chain (lm_chain_kernels.hip: ONE wave working through ~1 200 dependent instructions per
                                             evaluation; round 5: 3 465 -> 3 595 frames/s against the default), iterative-maxocc for
                                             every other kernel (throughput kernels)
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "odometry_hip.hip")
SRC_DENSE = os.path.join(_HERE, "csrc", "dense_kernels.hip")   # its own translation unit (see dense.hip.h)
SRC_CHAIN = os.path.join(_HERE, "csrc", "lm_chain_kernels.hip")   # the single tracker's LM chain: its own unit AND its own scheduler
DEPS = [SRC, SRC_DENSE, SRC_CHAIN, os.path.join(_HERE, "csrc", "kernels.hip.h"), os.path.join(_HERE, "csrc", "odo_math.h"), os.path.join(_HERE, "csrc", "tracker.hip.h"), os.path.join(_HERE, "csrc", "batch.hip.h"), os.path.join(_HERE, "csrc", "gather.hip.h"), os.path.join(_HERE, "csrc", "camera.hip.h"),
        os.path.join(_HERE, "csrc", "dense.hip.h"), os.path.join(_HERE, "csrc", "host_fp.h"),
        os.path.join(os.path.dirname(_HERE), "include", "odometry_hip.h"), os.path.abspath(__file__)]   # (this file: the flags)
LIB = os.path.join(_HERE, "lib", "libodometry_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "-mllvm",
         "-amdgpu-sched-strategy=" + os.environ.get("ODO_SCHED", "iterative-maxocc"), "-fPIC"] + os.environ.get("ODO_EXTRA_HIPCC_FLAGS", "").split()   # A/B builds
# Two machine schedulers (profiles/r06_state_machine_ab.md section 5; frames/s, whole library under ONE scheduler, round-6 source):
#                       single tracker (headline)    batched tracker S = 1 / 2 / 4 / 8
#   iterative-ilp       3 772-3 779                  3 340-3 380 / 5 820-5 860 / 9 090-9 100 / 13 970-14 040
#   (compiler default)  3 664-3 736                  3 690 / 6 500-6 520 / 10 510-10 630 / 13 820-13 840
#   iterative-maxocc    3 725-3 730                  3 680-3 690 / 6 480-6 490 / 10 390-10 410 / 14 290-14 430
#   max-ilp             3 554-3 556                  3 290-3 300 / 5 710-5 770 / 8 680-8 890 / 13 610
# The single tracker's chain — lm_coarse_kernel + lm_fine_kernel: ONE wave working through ~1 200 dependent instructions per
# evaluation — wants the ILP-first list scheduler; every other kernel is a throughput kernel and wants the occupancy-first one.
# Hence lm_chain_kernels.hip (ODO_SCHED_CHAIN, iterative-ilp) beside the main unit (ODO_SCHED, iterative-maxocc).
SCHED_CHAIN = os.environ.get("ODO_SCHED_CHAIN", "iterative-ilp")
SCHED_DENSE = os.environ.get("ODO_SCHED_DENSE", "")   # dense_kernels.hip: "" = the main unit's


def _flags_for(src, flags):
    sched = SCHED_CHAIN if src == SRC_CHAIN else SCHED_DENSE if src == SRC_DENSE else ""
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
    o_main, o_dense, o_batch = (os.path.join(objdir, n) for n in ("odometry_hip.o", "dense_kernels.o", "lm_chain_kernels.o"))
    cmds = [[HIPCC] + _flags_for(SRC, flags) + ["-c", "-o", o_main, SRC],
            [HIPCC] + _flags_for(SRC_DENSE, flags) + ["-c", "-o", o_dense, SRC_DENSE],
            [HIPCC] + _flags_for(SRC_CHAIN, flags) + ["-c", "-o", o_batch, SRC_CHAIN],
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
