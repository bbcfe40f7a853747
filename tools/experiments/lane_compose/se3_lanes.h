// se3_lanes.h — exp() and the left update of the LM state machine (ref: src/lm_optimizer.cpp:152-153; se3.hpp:272-278,495-502,
// 765-786; so3.hpp:302-304,577-611) with their 3x3 / 4x4 matrix algebra spread over sixteen lanes of the ONE wavefront that runs
// the state machine, instead of computed entry by entry by all 64 lanes redundantly.
//
// Why: that wavefront is alone on its SIMD, so it issues one instruction per 4-5 cycles whatever its EXEC mask — the state machine's
// time IS its instruction count (profiles/r05_phase_stamps.txt) — and odo::se3_exp + se3_left_update_mat + se3_to_colmajor spend ~270
// of their ~400 instructions on matrix entries: Omega^2 (45), V (36), V * upsilon (15), delta.matrix() (24), the 4x4 product (84),
// inc.matrix() (24), three fp32 divisions one after the other (33).
//
// Layout: a 4x4 matrix lives ROW-MAJOR in lanes L = 4 i + j of every 16-lane group (i = L >> 2: the quad = matrix row; j = L & 3: the
// position inside the quad = matrix column); all four groups of the wavefront hold the same values. Moving data then needs
//   rowb<K>(v)   the value of lane (i, K) in every lane of quad i       — DPP quad_perm [K K K K], a modifier of a VALU instruction
//   colb<K>(v)   the value of lane (K, j) in every lane at position j   — ds_swizzle (bit mode), a trip through the LDS crossbar
// and the algorithm is arranged so that colb is applied to STATE only (cur.matrix(), known a whole evaluation ahead: four swizzles
// issued when cur changes, long finished when the product needs them); everything between the solve's step and the new pose uses
// selects, DPP and twelve v_readlane.
//
// BIT-IDENTICAL to the scalar functions of odo_math.h by construction: every matrix entry is computed by ONE lane with exactly the
// operations, operands and association order of the scalar code (x - y written as x + (-y), products commuted: both exact), divisions
// are IEEE divisions per lane. The algorithm is a template over a lane policy X so that tests/hostemu.cpp runs THIS source on the CPU
// with a sixteen-float emulation of the lanes and compares it with the scalar chain bit for bit (tests/test_hostemu_parity.py); the
// device policy is LaneHw in kernels.hip.h.
//
// Policy X:  typedef V (per-lane float), M (lane mask);  splat(float) -> V;  mask<bits16>() -> M (lane L mod 16 in the set iff bit L of the constant);
//            sel(M, V a, V b) -> per lane m ? a : b;  rowb<K>(V), colb<K>(V) as above;  lane(V, l) -> float (wave-uniform, l < 16);
//            sincos2(xa, xb, &sa, &ca, &sb, &cb) (odo::sincos_f of two wave-uniform arguments);  V supports + - * / and unary -.
#pragma once
#include "odo_math.h"

namespace odo {

constexpr unsigned lane_bit(int i, int j) { return 1u << (4 * i + j); }
constexpr unsigned kLanesRow3 = 0xf000u, kLanesCol3 = 0x8888u, kLanesDiag3 = lane_bit(0, 0) | lane_bit(1, 1) | lane_bit(2, 2);

// Eigen Quaternion::toRotationMatrix (odo::quat_to_rot) of a wave-uniform quaternion: R[i][j] in lane (i, j), i, j < 3 (the other
// seven lanes of a group hold finite don't-cares). Entry (i, j) = diag ? 1 - (A + B) : A +- B with A = (2 u1) u2, B = (2 u3) u4:
//        A              B (sign)                       u1     u2     u3     u4
//   row 0: tyy txy txz | tzz  -twz +twy          y y z  y x x  z z y  z w w
//   row 1: txy txx tyz | +twz tzz  -twx          y x z  x x y  z z x  w z w
//   row 2: txz tyz txx | -twy +twx tyy           z z x  x y x  y x y  w w y
template <class X>
ODO_HD typename X::V quat_rot_lanes(float qx, float qy, float qz, float qw) {
  typedef typename X::V V;
  const V x = X::splat(qx), y = X::splat(qy), z = X::splat(qz), w = X::splat(qw);
  const V u1 = X::sel(X::template mask<lane_bit(1, 1) | lane_bit(2, 2)>(), x, X::sel(X::template mask<lane_bit(0, 0) | lane_bit(0, 1) | lane_bit(1, 0)>(), y, z));
  const V u2 = X::sel(X::template mask<lane_bit(0, 0) | lane_bit(1, 2) | lane_bit(2, 1)>(), y, x);
  const V u3 = X::sel(X::template mask<lane_bit(1, 2) | lane_bit(2, 1)>(), x, X::sel(X::template mask<lane_bit(0, 2) | lane_bit(2, 0) | lane_bit(2, 2)>(), y, z));
  const V u4 = X::sel(X::template mask<lane_bit(0, 0) | lane_bit(1, 1)>(), z, X::sel(X::template mask<lane_bit(2, 2)>(), y, w));
  const V two = X::splat(2.0f), one = X::splat(1.0f);
  const V A = (two * u1) * u2;
  const V B = (two * u3) * u4;
  const V Bs = X::sel(X::template mask<lane_bit(0, 1) | lane_bit(1, 2) | lane_bit(2, 0)>(), -B, B);
  const V S = A + Bs;
  return X::sel(X::template mask<kLanesDiag3>(), one - S, S);
}

// [R | t; 0 0 0 1] from the rotation lanes and a translation that quad i holds as t_i (any position).
template <class X>
ODO_HD typename X::V affine_lanes(typename X::V R, typename X::V t_by_quad) {
  const typename X::V e3 = X::sel(X::template mask<lane_bit(3, 3)>(), X::splat(1.0f), X::splat(0.0f));
  return X::sel(X::template mask<kLanesRow3>(), e3, X::sel(X::template mask<kLanesCol3>(), t_by_quad, R));
}
template <class X>
ODO_HD typename X::V by_quad3(float a0, float a1, float a2) {   // quad i holds a_i (quad 3: 0)
  return X::sel(X::template mask<0x000fu>(), X::splat(a0), X::sel(X::template mask<0x00f0u>(), X::splat(a1), X::sel(X::template mask<0x0f00u>(), X::splat(a2), X::splat(0.0f))));
}
template <class X>
ODO_HD typename X::V by_pos3(float a0, float a1, float a2) {    // position j holds a_j (position 3: 0)
  return X::sel(X::template mask<0x1111u>(), X::splat(a0), X::sel(X::template mask<0x2222u>(), X::splat(a1), X::sel(X::template mask<0x4444u>(), X::splat(a2), X::splat(0.0f))));
}
// se3_to_colmajor(s) as lanes: s.matrix(), row-major in the lanes.
template <class X>
ODO_HD typename X::V se3_matrix_lanes(const Se3& s) {
  return affine_lanes<X>(quat_rot_lanes<X>(s.qx, s.qy, s.qz, s.qw), by_quad3<X>(s.tx, s.ty, s.tz));
}

// cur.matrix() as the product wants it: C[k] = colb<k>(C) — row k of the matrix in every quad. Computed when cur changes.
template <class X> struct CurLanes {
  typename X::V row[4];
};
template <class X>
ODO_HD void cur_lanes_set(CurLanes<X>& c, typename X::V C) {
  c.row[0] = X::template colb<0>(C);
  c.row[1] = X::template colb<1>(C);
  c.row[2] = X::template colb<2>(C);
  c.row[3] = X::template colb<3>(C);
}

// inc = SE3(exp(a).matrix() * cur.matrix()): odo::se3_exp followed by odo::se3_left_update_mat, a = [upsilon; omega] wave-uniform.
// Returns inc (wave-uniform).
template <class X>
ODO_HD void se3_exp_left_update_lanes(const float a[6], const CurLanes<X>& cur, Se3* inc) {
  typedef typename X::V V;
  // ---- so3.hpp:577-611 (wave-uniform up to the three divisions) ----
  const float ox = a[3], oy = a[4], oz = a[5];
  const float theta_sq = (ox * ox + oy * oy) + oz * oz;
  const float theta = sqrtf(theta_sq);
  const float half_theta = 0.5f * theta;
  const bool small = theta < 1e-5f;
  float imag, real, ca = 0.0f, cb = 0.0f;
  if (small) {
    const float theta_po4 = theta_sq * theta_sq;
    imag = (0.5f - (float)(1.0 / 48.0) * theta_sq) + (float)(1.0 / 3840.0) * theta_po4;
    real = (1.0f - (float)(1.0 / 8.0) * theta_sq) + (float)(1.0 / 384.0) * theta_po4;
  } else {
    float sh, ch, st, ct;
    X::sincos2(half_theta, theta, &sh, &ch, &st, &ct);
    real = ch;
    // imag = sh / theta, ca = (1 - ct) / tsq, cb = (theta - st) / (tsq * theta): three IEEE divisions, side by side in lanes 0, 1, 2
    const float tsq = theta * theta;
    const V num = X::sel(X::template mask<0x1111u>(), X::splat(sh), X::sel(X::template mask<0x2222u>(), X::splat(1.0f - ct), X::splat(theta - st)));
    const V den = X::sel(X::template mask<0x1111u>(), X::splat(theta), X::sel(X::template mask<0x2222u>(), X::splat(tsq), X::splat(tsq * theta)));
    const V quo = num / den;
    imag = X::lane(quo, 0); ca = X::lane(quo, 1); cb = X::lane(quo, 2);
  }
  const float dqx = imag * ox, dqy = imag * oy, dqz = imag * oz, dqw = real;
  // ---- delta.matrix() rotation part (se3.hpp:272-278) ----
  const V Rd = quat_rot_lanes<X>(dqx, dqy, dqz, dqw);
  // ---- V = I + ca Omega + cb Omega^2 (se3.hpp:765-786), or R for a tiny angle ----
  V Vm;
  if (small) {
    Vm = Rd;
  } else {
    // Omega = [0 -oz oy; oz 0 -ox; -oy ox 0]: column k by quad (Omega[i][k] in quad i) and row k by position (Omega[k][j] at position j)
    const V c0 = by_quad3<X>(0.0f, oz, -oy), c1 = by_quad3<X>(-oz, 0.0f, ox), c2 = by_quad3<X>(oy, -ox, 0.0f);
    const V r0 = by_pos3<X>(0.0f, -oz, oy), r1 = by_pos3<X>(oz, 0.0f, -ox), r2 = by_pos3<X>(-oy, ox, 0.0f);
    const V Om2 = (c0 * r0 + c1 * r1) + c2 * r2;
    const V Om = X::sel(X::template mask<0x000fu>(), r0, X::sel(X::template mask<0x00f0u>(), r1, r2));
    const V id = X::sel(X::template mask<kLanesDiag3>(), X::splat(1.0f), X::splat(0.0f));
    Vm = (id + X::splat(ca) * Om) + X::splat(cb) * Om2;
  }
  // ---- t_i = (V[i][0] a0 + V[i][1] a1) + V[i][2] a2: products per lane, summed inside the quad ----
  const V P = Vm * by_pos3<X>(a[0], a[1], a[2]);
  const V tq = (X::template rowb<0>(P) + X::template rowb<1>(P)) + X::template rowb<2>(P);
  const V D = affine_lanes<X>(Rd, tq);
  // ---- M = delta.matrix() * cur.matrix(), k ascending (se3_left_update_mat) ----
  const V Mm = ((X::template rowb<0>(D) * cur.row[0] + X::template rowb<1>(D) * cur.row[1]) + X::template rowb<2>(D) * cur.row[2]) +
               X::template rowb<3>(D) * cur.row[3];
  // ---- SE3(M): Eigen's matrix -> quaternion on wave-uniform values (se3_from_colmajor) ----
  float R[9];
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) R[i * 3 + j] = X::lane(Mm, 4 * i + j);
  rot_to_quat(R, inc);
  inc->tx = X::lane(Mm, 3); inc->ty = X::lane(Mm, 7); inc->tz = X::lane(Mm, 11);
}

}  // namespace odo
