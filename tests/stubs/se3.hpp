// tests/stubs/se3.hpp — TEST SCAFFOLDING ONLY: stands where the reference's callers include Sophus (`#include <se3.hpp>`,
// ref: run_odometry_kitti_offline.cpp:20, CMakeLists.txt:38). This image has no Eigen, so the vendored Sophus cannot be compiled; the
// stub carries the one thing the callers use — Sophus::SO3<float>(R).angleX() / angleY() / angleZ() (ref: run_odometry_kitti_offline.cpp:
// 254-255) — written from Sophus' documented behaviour (third_party/Sophus/sophus/so3.hpp:127-154, 463-468, 302-304: the constructor
// keeps a unit quaternion of R, matrix() rebuilds R from it, angle* is the SO(2) log of the 2x2 block made a rotation matrix). A
// maintainer of the reference has the real Sophus and does not need this file. Pins no arithmetic.
#pragma once
#include <cmath>
#include <Eigen/Core>

namespace Sophus {
template <class T> class SO3 {
 public:
  template <class D> explicit SO3(const Eigen::DenseBase<D>& R_) {
    const Eigen::Matrix<T, 3, 3> R(R_);
    T t = R(0, 0) + R(1, 1) + R(2, 2);   // Eigen's rotation matrix -> quaternion (trace based), no renormalisation
    if (t > T(0)) {
      t = std::sqrt(t + T(1));
      q_[3] = T(0.5) * t;
      t = T(0.5) / t;
      q_[0] = (R(2, 1) - R(1, 2)) * t; q_[1] = (R(0, 2) - R(2, 0)) * t; q_[2] = (R(1, 0) - R(0, 1)) * t;
    } else {
      int i = 0;
      if (R(1, 1) > R(0, 0)) i = 1;
      if (R(2, 2) > R(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + T(1));
      q_[i] = T(0.5) * t;
      t = T(0.5) / t;
      q_[3] = (R(k, j) - R(j, k)) * t; q_[j] = (R(j, i) + R(i, j)) * t; q_[k] = (R(k, i) + R(i, k)) * t;
    }
  }
  Eigen::Matrix<T, 3, 3> matrix() const {   // Eigen's quaternion -> rotation matrix
    const T x = q_[0], y = q_[1], z = q_[2], w = q_[3];
    const T tx = 2 * x, ty = 2 * y, tz = 2 * z, twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x,
            tyy = ty * y, tyz = tz * y, tzz = tz * z;
    Eigen::Matrix<T, 3, 3> R;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
    return R;
  }
  T angleX() const { const Eigen::Matrix<T, 3, 3> R = matrix(); return std::atan2(R(2, 1) - R(1, 2), R(1, 1) + R(2, 2)); }
  T angleY() const { const Eigen::Matrix<T, 3, 3> R = matrix(); return std::atan2(R(0, 2) - R(2, 0), R(0, 0) + R(2, 2)); }
  T angleZ() const { const Eigen::Matrix<T, 3, 3> R = matrix(); return std::atan2(R(1, 0) - R(0, 1), R(0, 0) + R(1, 1)); }
 private:
  T q_[4];   // x, y, z, w
};
}  // namespace Sophus
