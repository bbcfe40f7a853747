"""CPU runner over the oracle: the reference's frame loop (ref: run_odometry_kitti_offline.cpp:95-145, 198-271)
stepped with oracle calls. TEST INFRASTRUCTURE ONLY (parity tests and bench.py's cpu_baseline leg)."""
import numpy as np

from . import oracle as O

KEYFRAME_WEIGHT = (np.array([0.1, 1.0, 0.1, 1.0, 0.1, 1.0], np.float32) / np.float32(3.3)).astype(np.float32)


def motion_angles(T):
    """Sophus SO3::angleX/Y/Z of the R -> q -> R round trip (ref: third_party/Sophus/sophus/so3.hpp:127-154)."""
    R = O.se3_roundtrip(np.asarray(T, np.float32))[:3, :3]
    f = np.float32
    return np.array([np.arctan2(f(R[2, 1] - R[1, 2]), f(R[1, 1] + R[2, 2])),
                     np.arctan2(f(R[0, 2] - R[2, 0]), f(R[0, 0] + R[2, 2])),
                     np.arctan2(f(R[1, 0] - R[0, 1]), f(R[0, 0] + R[1, 1]))], np.float32)


def matmul4_f32(A, B):
    """4x4 fp32 product, every operation rounded once, k ascending: ((a0 b0 + a1 b1) + a2 b2) + a3 b3 — the association of the
    plain triple loop (ref: run_odometry_kitti_offline.cpp:218, Eigen's 4x4 float product is unpinned: this is the stated order)."""
    A = np.asarray(A, np.float32)
    B = np.asarray(B, np.float32)
    t = [(A[:, k:k + 1] * B[k:k + 1, :]).astype(np.float32) for k in range(4)]   # outer products: one rounding per entry
    return (((t[0] + t[1]).astype(np.float32) + t[2]).astype(np.float32) + t[3]).astype(np.float32)


class OracleRunner:
    def __init__(self, lm_params=None, depth_params=None, motion_th=1.1):
        self.lp = lm_params or O.lm_params()
        self.dp = depth_params or O.depth_params()
        self.motion_th = np.float32(motion_th)

    def init(self, left, right, abs_pose0=None):
        d = O.compute_depth(left, right, self.dp)
        if d["status"] != 0:
            raise RuntimeError("Init 0-th frame failed!")
        self.kf_img = O.image_pyramid(left, self.lp.n_levels, True, flat=True)
        self.kf_dep = O.depth_pyramid(d["dep"], self.lp.n_levels, flat=True)
        self.kf_abs = np.eye(4, dtype=np.float32) if abs_pose0 is None else np.asarray(abs_pose0, np.float32)
        self.init_pose = np.eye(4, dtype=np.float32)
        self.n_keyframes = 1
        return d

    def track(self, left, right):
        r = O.track_frame(self.kf_img, self.kf_dep, left, right, self.lp, self.dp, self.init_pose)
        if r["status"] == -2:
            raise RuntimeError("    depth failed!")
        T = r["pose"]
        inv = np.linalg.inv(T.astype(np.float64)).astype(np.float32)
        cur = matmul4_f32(self.kf_abs, inv)   # (numpy's own float32 `@` leaves the association to BLAS: a chain of ~25 keyframes
        #                                        then drifts by ulps of a 100 m translation, more than the pose tolerance)
        mot = np.concatenate([np.abs(motion_angles(T)), np.abs(T[:3, 3])]).astype(np.float32)
        mag = np.float32(0)
        for m, w in zip(mot, KEYFRAME_WEIGHT):
            mag = np.float32(mag + np.float32(m * w))
        new_kf = bool(mag > self.motion_th)
        if new_kf:
            self.kf_img, self.kf_dep, self.kf_abs = r["img_pyr"], r["dep_pyr"], cur
            self.n_keyframes += 1
        self.init_pose = T
        return dict(pose_to_keyframe=T, abs_pose=cur, new_keyframe=new_kf, motion=float(mag), solve_status=r["status"],
                    val=r["val"], disp=r["disp"], dep=r["dep"], n_valid=r["n_valid"])
