// verify_camera.h -- Camera::ImageToWorld / ImageToWorldThreshold / CalibrationMatrix on the device for all
// eleven camera models of the reference (/root/reference/src/base/camera_models.h:187-349), FP64, operation
// for operation in the order the reference writes them.
//
//   id  model                   params                                          focal idxs  extra from
//    0  SIMPLE_PINHOLE          f cx cy                                         {0}         -
//    1  PINHOLE                 fx fy cx cy                                     {0,1}       -
//    2  SIMPLE_RADIAL           f cx cy k                                       {0}         3
//    3  RADIAL                  f cx cy k1 k2                                   {0}         3
//    4  OPENCV                  fx fy cx cy k1 k2 p1 p2                         {0,1}       4
//    5  OPENCV_FISHEYE          fx fy cx cy k1 k2 k3 k4                         {0,1}       4
//    6  FULL_OPENCV             fx fy cx cy k1 k2 p1 p2 k3 k4 k5 k6             {0,1}       4
//    7  FOV                     fx fy cx cy omega                               {0,1}       4
//    8  SIMPLE_RADIAL_FISHEYE   f cx cy k                                       {0}         3
//    9  RADIAL_FISHEYE          f cx cy k1 k2                                   {0}         3
//   10  THIN_PRISM_FISHEYE      fx fy cx cy k1 k2 p1 p2 k3 k4 sx1 sy1           {0,1}       4
//
// Models 0-4 and 6 use + - * / only.  Models 5, 7, 8, 9, 10 call atan / tan / sin / cos: those are the correctly
// rounded dsm_atan / dsm_tan / dsm_sin / dsm_cos of exact_trig.h (plain double operations, the same code in the CPU
// oracle), not ocml's -- so all eleven models are bit-identical to the CPU path.
#ifndef DAGSFM_AMD_CSRC_VERIFY_CAMERA_H_
#define DAGSFM_AMD_CSRC_VERIFY_CAMERA_H_

#include <float.h>
#include <math.h>

#include "../../include/dagsfm_mi355x.h"

#ifndef DSM_DEV
#define DSM_DEV __device__ __forceinline__
#endif

#define DSM_XT static __device__ __noinline__
#define DSM_XT_CONST __device__ const
#include "exact_trig.h"

#define DSM_NUM_CAMERA_MODELS 11

// ExistsCameraModelWithId, camera_models.h:352 (host and device)
__host__ __device__ inline bool cam_model_exists(int model_id) { return model_id >= 0 && model_id < DSM_NUM_CAMERA_MODELS; }
// focal_length_idxs.size() == 2
__host__ __device__ inline bool cam_two_focal(int model_id) {
  return model_id == 1 || model_id == 4 || model_id == 5 || model_id == 6 || model_id == 7 || model_id == 10;
}
__host__ __device__ inline int cam_num_params(int model_id) {
  const int n[DSM_NUM_CAMERA_MODELS] = {3, 4, 4, 5, 8, 8, 12, 5, 4, 5, 12};
  return cam_model_exists(model_id) ? n[model_id] : 0;
}

// CameraModel::Distortion(extra_params, u, v, &du, &dv) of the models that go through IterativeUndistortion
DSM_DEV void cam_distortion(int model_id, const double* e, double u, double v, double* du, double* dv) {
  switch (model_id) {
    case 2: {  // SIMPLE_RADIAL, camera_models.h:747-757
      const double k = e[0];
      const double u2 = u * u, v2 = v * v;
      const double r2 = u2 + v2;
      const double radial = k * r2;
      *du = u * radial;
      *dv = v * radial;
      return;
    }
    case 3: {  // RADIAL, :810-822
      const double k1 = e[0], k2 = e[1];
      const double u2 = u * u, v2 = v * v;
      const double r2 = u2 + v2;
      const double radial = k1 * r2 + k2 * r2 * r2;
      *du = u * radial;
      *dv = v * radial;
      return;
    }
    case 4: {  // OPENCV, :881-897
      const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3];
      const double u2 = u * u, uv = u * v, v2 = v * v;
      const double r2 = u2 + v2;
      const double radial = k1 * r2 + k2 * r2 * r2;
      *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2);
      *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2);
      return;
    }
    case 5: {  // OPENCV_FISHEYE, :957-982
      const double k1 = e[0], k2 = e[1], k3 = e[2], k4 = e[3];
      const double r = sqrt(u * u + v * v);
      if (r > DBL_EPSILON) {
        const double theta = dsm_atan(r);
        const double theta2 = theta * theta;
        const double theta4 = theta2 * theta2;
        const double theta6 = theta4 * theta2;
        const double theta8 = theta4 * theta4;
        const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4 + k3 * theta6 + k4 * theta8);
        *du = u * thetad / r - u;
        *dv = v * thetad / r - v;
      } else {
        *du = 0.0;
        *dv = 0.0;
      }
      return;
    }
    case 6: {  // FULL_OPENCV, :1053-1077
      const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3], k3 = e[4], k4 = e[5], k5 = e[6], k6 = e[7];
      const double u2 = u * u, uv = u * v, v2 = v * v;
      const double r2 = u2 + v2;
      const double r4 = r2 * r2;
      const double r6 = r4 * r2;
      const double radial = (1.0 + k1 * r2 + k2 * r4 + k3 * r6) / (1.0 + k4 * r2 + k5 * r4 + k6 * r6);
      *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) - u;
      *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) - v;
      return;
    }
    case 8: {  // SIMPLE_RADIAL_FISHEYE, :1278-1297
      const double k = e[0];
      const double r = sqrt(u * u + v * v);
      if (r > DBL_EPSILON) {
        const double theta = dsm_atan(r);
        const double theta2 = theta * theta;
        const double thetad = theta * (1.0 + k * theta2);
        *du = u * thetad / r - u;
        *dv = v * thetad / r - v;
      } else {
        *du = 0.0;
        *dv = 0.0;
      }
      return;
    }
    case 9: {  // RADIAL_FISHEYE, :1358-1380
      const double k1 = e[0], k2 = e[1];
      const double r = sqrt(u * u + v * v);
      if (r > DBL_EPSILON) {
        const double theta = dsm_atan(r);
        const double theta2 = theta * theta;
        const double theta4 = theta2 * theta2;
        const double thetad = theta * (1.0 + k1 * theta2 + k2 * theta4);
        *du = u * thetad / r - u;
        *dv = v * thetad / r - v;
      } else {
        *du = 0.0;
        *dv = 0.0;
      }
      return;
    }
    case 10: {  // THIN_PRISM_FISHEYE, :1459-1481
      const double k1 = e[0], k2 = e[1], p1 = e[2], p2 = e[3], k3 = e[4], k4 = e[5], sx1 = e[6], sy1 = e[7];
      const double u2 = u * u, uv = u * v, v2 = v * v;
      const double r2 = u2 + v2;
      const double r4 = r2 * r2;
      const double r6 = r4 * r2;
      const double r8 = r6 * r2;
      const double radial = k1 * r2 + k2 * r4 + k3 * r6 + k4 * r8;
      *du = u * radial + 2.0 * p1 * uv + p2 * (r2 + 2.0 * u2) + sx1 * r2;
      *dv = v * radial + 2.0 * p2 * uv + p1 * (r2 + 2.0 * v2) + sy1 * r2;
      return;
    }
    default:
      *du = 0.0;
      *dv = 0.0;
      return;
  }
}

// BaseCameraModel::IterativeUndistortion, camera_models.h:547-587: Newton iteration with central differences
DSM_DEV void cam_iterative_undistortion(int model_id, const double* e, double* u, double* v) {
  const double x0_0 = *u, x0_1 = *v;
  double x_0 = *u, x_1 = *v;
  for (int it = 0; it < 100; ++it) {
    const double a0 = fabs(1e-6 * x_0), a1 = fabs(1e-6 * x_1);
    const double step0 = DBL_EPSILON > a0 ? DBL_EPSILON : a0;
    const double step1 = DBL_EPSILON > a1 ? DBL_EPSILON : a1;
    double dx0, dx1, b00, b01, f00, f01, b10, b11, f10, f11;
    cam_distortion(model_id, e, x_0, x_1, &dx0, &dx1);
    cam_distortion(model_id, e, x_0 - step0, x_1, &b00, &b01);
    cam_distortion(model_id, e, x_0 + step0, x_1, &f00, &f01);
    cam_distortion(model_id, e, x_0, x_1 - step1, &b10, &b11);
    cam_distortion(model_id, e, x_0, x_1 + step1, &f10, &f11);
    const double J00 = 1 + (f00 - b00) / (2 * step0);
    const double J01 = (f10 - b10) / (2 * step1);
    const double J10 = (f01 - b01) / (2 * step0);
    const double J11 = 1 + (f11 - b11) / (2 * step1);
    const double invdet = 1.0 / (J00 * J11 - J10 * J01);
    const double i00 = J11 * invdet, i01 = -J01 * invdet, i10 = -J10 * invdet, i11 = J00 * invdet;
    const double r0 = x_0 + dx0 - x0_0, r1 = x_1 + dx1 - x0_1;
    const double s0 = i00 * r0 + i01 * r1;
    const double s1 = i10 * r0 + i11 * r1;
    x_0 -= s0;
    x_1 -= s1;
    if (s0 * s0 + s1 * s1 < 1e-10) break;
  }
  *u = x_0;
  *v = x_1;
}

// FOVCameraModel::Undistortion, camera_models.h:1179-1218
DSM_DEV void cam_fov_undistortion(const double* e, double u, double v, double* du, double* dv) {
  const double omega = e[0];
  const double kEpsilon = 1e-4;
  const double radius2 = u * u + v * v;
  const double omega2 = omega * omega;
  double factor;
  if (omega2 < kEpsilon) {
    factor = (omega2 * radius2) / 3.0 - omega2 / 12.0 + 1.0;
  } else if (radius2 < kEpsilon) {
    factor = (omega * (omega * omega * radius2 + 3.0)) / (6.0 * dsm_tan(omega / 2.0));
  } else {
    const double radius = sqrt(radius2);
    const double numerator = dsm_tan(radius * omega);
    factor = numerator / (radius * 2.0 * dsm_tan(omega / 2.0));
  }
  *du = u * factor;
  *dv = v * factor;
}

// Camera::ImageToWorld (camera.cc:210-214 -> CameraModelImageToWorld)
DSM_DEV void image_to_world(const dsm_camera& cam, double x, double y, double* u, double* v) {
  const int id = cam.model_id;
  if (cam_two_focal(id)) {
    const double f1 = cam.params[0], f2 = cam.params[1], c1 = cam.params[2], c2 = cam.params[3];
    if (id == 7) {  // FOV, :1126-1141
      const double uu = (x - c1) / f1;
      const double vv = (y - c2) / f2;
      cam_fov_undistortion(&cam.params[4], uu, vv, u, v);
      return;
    }
    *u = (x - c1) / f1;
    *v = (y - c2) / f2;
    if (id == 1) return;  // PINHOLE, :679-689
    cam_iterative_undistortion(id, &cam.params[4], u, v);
    if (id == 10) {  // THIN_PRISM_FISHEYE, :1434-1456
      const double theta = sqrt(*u * *u + *v * *v);
      const double theta_cos_theta = theta * dsm_cos(theta);
      if (theta_cos_theta > DBL_EPSILON) {
        const double scale = dsm_sin(theta) / theta_cos_theta;
        *u *= scale;
        *v *= scale;
      }
    }
    return;
  }
  const double f = cam.params[0], c1 = cam.params[1], c2 = cam.params[2];
  *u = (x - c1) / f;
  *v = (y - c2) / f;
  if (id == 0) return;  // SIMPLE_PINHOLE, :629-637
  cam_iterative_undistortion(id, &cam.params[3], u, v);
}

// BaseCameraModel::ImageToWorldThreshold, camera_models.h:535-543
DSM_DEV double image_to_world_threshold(const dsm_camera& cam, double threshold) {
  double mean_focal_length = 0;
  if (cam_two_focal(cam.model_id)) {
    mean_focal_length += cam.params[0];
    mean_focal_length += cam.params[1];
    mean_focal_length /= 2;
  } else {
    mean_focal_length += cam.params[0];
    mean_focal_length /= 1;
  }
  return threshold / mean_focal_length;
}

// Camera::CalibrationMatrix, camera.cc:75-93 (row-major 3x3)
DSM_DEV void calibration_matrix(const dsm_camera& cam, double* K) {
  for (int i = 0; i < 9; ++i) K[i] = 0.0;
  K[0] = K[4] = K[8] = 1.0;
  if (cam_two_focal(cam.model_id)) {
    K[0] = cam.params[0]; K[4] = cam.params[1];
    K[2] = cam.params[2]; K[5] = cam.params[3];
  } else {
    K[0] = cam.params[0]; K[4] = cam.params[0];
    K[2] = cam.params[1]; K[5] = cam.params[2];
  }
}

#endif  // DAGSFM_AMD_CSRC_VERIFY_CAMERA_H_
