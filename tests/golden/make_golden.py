"""Generates the committed golden fixtures from the CPU oracle (run in the build container; outputs are data only).

The reference ships no golden vectors or known-answer tests for this path, and it cannot be
built or imported here, so these fixtures pin the ORACLE (oracle/odo_oracle.c) against accidental change and give the
GPU tests a fixed target that does not depend on the oracle library being rebuilt identically.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402
from odometry_amd import synth  # noqa: E402

K = dict(f0=150.0, cx0=80.0, cy0=60.0)


def small_scene():
    scene = synth.Scene(3, texels_per_m=12.0, tile_texels=(5, 11, 23))
    poses = synth.trajectory(2, 3, fwd_range=(0.15, 0.25))
    out = []
    for T in poses:
        L, Z = scene.render(T, 120, 160, K["f0"], K["cx0"], K["cy0"], 0.0)
        R, _ = scene.render(T, 120, 160, K["f0"], K["cx0"], K["cy0"], 0.5)
        out.append((L, R, Z))
    return out


def main():
    rng = np.random.default_rng(20240601)
    # (i) pyramids of a 64x48 u8 image
    img = rng.integers(0, 256, (48, 64)).astype(np.uint8)
    pyr_s = O.image_pyramid(img.astype(np.float32), 3, True)
    pyr_n = O.image_pyramid(img.astype(np.float32), 3, False)
    dep = (rng.random((48, 64)) * (rng.random((48, 64)) < 0.3)).astype(np.float32)
    dpyr = O.depth_pyramid(dep, 3)
    np.savez_compressed(os.path.join(HERE, "pyramid_64x48.npz"), img=img, dep=dep,
                        **{f"smooth_l{l}": p for l, p in enumerate(pyr_s)},
                        **{f"plain_l{l}": p for l, p in enumerate(pyr_n)},
                        **{f"depth_l{l}": p for l, p in enumerate(dpyr)})
    # (ii)+(iii) LM: accumulators at a fixed pose for each robust mode, and a full Solve with its trace
    (L0, R0, Z0), (L1, R1, Z1) = small_scene()
    inv = synth.semi_dense_inverse_depth(Z0, L0, grad_th=6.0)
    p0, pd, p1 = O.image_pyramid(L0, 3, True), O.depth_pyramid(inv, 3), O.image_pyramid(L1, 3, True)
    T = O.se3_exp(np.array([0.01, -0.005, -0.18, 0.001, 0.004, -0.002], np.float32))
    accs = np.zeros((3, 3, 29))
    sig = np.zeros((3,))
    for robust in range(3):
        for l in range(3):
            r = O.lm_accumulate(p0[l], p1[l], pd[l], l, T, robust=robust, huber_delta=28.0, K=K)
            accs[robust, l] = r["acc"]
            if robust == 2 and l == 0:
                sig[0] = r["sigma"]
    sol = {}
    for robust in range(3):
        s = O.lm_solve(O.image_pyramid(L0, 3, flat=True), O.depth_pyramid(inv, 3, flat=True),
                       O.image_pyramid(L1, 3, flat=True), 120, 160, O.lm_params(max_iters=(10, 20, 30), robust=robust, K=K))
        sol[f"pose_r{robust}"] = s["pose"]
        sol[f"trace_r{robust}"] = np.array([[t["level"], t["iter"], t["n_res"], t["accepted"], t["stop"]] for t in s["trace"]],
                                           np.int32)
        sol[f"err_r{robust}"] = np.array([t["err"] for t in s["trace"]], np.float32)
        sol[f"delta_r{robust}"] = np.array([t["delta"] for t in s["trace"]], np.float32)
    np.savez_compressed(os.path.join(HERE, "lm_120x160.npz"), L0=L0.astype(np.uint8), L1=L1.astype(np.uint8), inv=inv, T=T,
                        K=np.array([K["f0"], K["cx0"], K["cy0"]], np.float32), accs=accs, sigma=sig, **sol)
    # (iv) SSD tree KAT + (vi) exp / left-update vectors
    s8 = (rng.random((64, 8)) * np.array([1e8, 1, 1e8, 1, 3, 1e-3, 5, 7])).astype(np.float32)
    tree = np.array([O.lib().orc_ssd8_tree(r.ctypes.data_as(O._fp)) for r in np.ascontiguousarray(s8)], np.float32)
    a = np.concatenate([rng.normal(0, 1, (40, 3)), rng.normal(0, 1, (40, 3)) * np.repeat([1e-6, 0.01, 0.3, 2.5], 10)[:, None]],
                       1).astype(np.float32)
    exps = np.stack([O.se3_exp(v) for v in a])
    upd = np.stack([O.se3_left_update((0.05 * v).astype(np.float32), exps[(i + 1) % 40]) for i, v in enumerate(a)])
    np.savez_compressed(os.path.join(HERE, "se3_ssd.npz"), s8=s8, tree=tree, a=a, exps=exps, upd=upd)
    # (v) depth estimator on the small pair (any_size), disparity stage and full ComputeDepth
    dp = O.depth_params(grad_th=4.0, baseline=0.5, f0=K["f0"], any_size=1)
    d1 = O.compute_depth(L0, R0, dp, stage=1)
    d2 = O.compute_depth(L0, R0, dp, stage=2)
    np.savez_compressed(os.path.join(HERE, "depth_120x160.npz"), L=L0.astype(np.uint8), R=R0.astype(np.uint8),
                        val1=d1["val"], disp1=d1["disp"].astype(np.int16), dep1=d1["dep"],
                        val2=d2["val"], dep2=d2["dep"],
                        stats=np.array([d1["n_selected"], d1["n_matched"], d2["n_valid"], d2["iters"], d2["status"]], np.int32),
                        cost=np.float32(d2["cost"]))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


def rot_xyz(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def camera():
    """(vii) camera model: a 96x128 sensor with radial + tangential distortion, a small rectifying rotation and a new
    projection matrix; lookup maps, the intrinsic pyramid and one remapped image."""
    rng = np.random.default_rng(20240917)
    raw = np.array([91.7, 91.4, 0.0, 66.3, 47.9])
    dist = np.array([-0.283, 0.074, 1.9e-4, -1.8e-5])
    R = rot_xyz(0.004, -0.011, 0.007)
    P = np.array([[88.0, 0.0, 64.5, -9.7], [0.0, 88.0, 48.25, 0.0], [0.0, 0.0, 1.0, 0.0]])
    mx, my = O.camera_init_maps(raw, dist, R, P, 96, 128)
    src = rng.integers(0, 256, (96, 128)).astype(np.uint8)
    dst = O.camera_remap(src.astype(np.float32), mx, my, 0.0)
    np.savez_compressed(os.path.join(HERE, "camera_96x128.npz"), raw=raw, dist=dist, R=R, P=P, mapx=mx, mapy=my, src=src,
                        dst=dst, intr=O.camera_intrinsics(P, 4))
    print("camera_96x128.npz", os.path.getsize(os.path.join(HERE, "camera_96x128.npz")), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "camera":
        camera()
    else:
        main()
        camera()
