// ndt_tsdf_device.h -- per-voxel update maths for the NDT and TSDF mappers on gfx950, following the CPU
// instantiation of the reference's shared compute headers (CovReal = double, Vec3 = dvec3) operation by operation.
// Compiled with -ffp-contract=off.  exp()/log() are the device libm's fp64 versions: within an ulp or so of glibc,
// well inside the 1e-5 relative parity bar for NDT values.  Citations: reference file:line.
#ifndef OHMHIP_NDT_TSDF_DEVICE_H
#define OHMHIP_NDT_TSDF_DEVICE_H

#include "walk_device.h"

namespace ohmhip
{
struct D3
{
  double x, y, z;
};

__device__ inline D3 d3(double x, double y, double z)
{
  D3 v;
  v.x = x;
  v.y = y;
  v.z = z;
  return v;
}
__device__ inline D3 operator-(const D3 &a, const D3 &b) { return d3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline D3 operator+(const D3 &a, const D3 &b) { return d3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline D3 operator*(const D3 &a, double s) { return d3(a.x * s, a.y * s, a.z * s); }
/// glm::dot evaluation order: (x*x' + y*y') + z*z'
__device__ inline double dot(const D3 &a, const D3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
/// glm::normalize: v * inversesqrt(dot(v, v)), inversesqrt(x) = 1 / sqrt(x)
__device__ inline D3 normalize(const D3 &v) { return v * (1.0 / sqrt(dot(v, v))); }

/// ohm/VoxelMeanCompute.h:102-122 (always decodes: the reference tests the constant used_bit)
__device__ inline D3 subVoxelToLocal(uint32_t pattern, double resolution)
{
  const int mean_positions = (1 << 10) - 1;
  const double mean_resolution = resolution / double(mean_positions);
  const double offset = double(0.5f) * resolution;
  return d3(int(pattern & mean_positions) * mean_resolution - offset,
            int((pattern >> 10) & mean_positions) * mean_resolution - offset,
            int((pattern >> 20) & mean_positions) * mean_resolution - offset);
}

/// ohm/VoxelMeanCompute.h:134-152 + :69-92
__device__ inline uint32_t subVoxelUpdateD3(uint32_t coord, uint32_t point_count, const D3 &v, double resolution)
{
  const int mean_positions = (1 << 10) - 1;
  const double mean_resolution = resolution / double(mean_positions);
  const double offset = double(0.5f) * resolution;
  D3 mean = subVoxelToLocal(coord, resolution);
  const double one_on_count_plus_one = double(1) / double(point_count + 1);
  mean.x += (v.x - mean.x) * one_on_count_plus_one;
  mean.y += (v.y - mean.y) * one_on_count_plus_one;
  mean.z += (v.z - mean.z) * one_on_count_plus_one;
  int px = pointToRegionCoord(mean.x + offset, mean_resolution);
  int py = pointToRegionCoord(mean.y + offset, mean_resolution);
  int pz = pointToRegionCoord(mean.z + offset, mean_resolution);
  px = (px >= 0 ? (px < (1 << 10) ? px : mean_positions) : 0);
  py = (py >= 0 ? (py < (1 << 10) ? py : mean_positions) : 0);
  pz = (pz >= 0 ? (pz < (1 << 10) ? pz : mean_positions) : 0);
  return uint32_t(px) | (uint32_t(py) << 10) | (uint32_t(pz) << 20) | (1u << 31);
}

struct Cov6
{
  float c0, c1, c2, c3, c4, c5;
};

/// ohm/CovarianceVoxelCompute.h:183-204
__device__ inline D3 solveTriangular(const Cov6 &c, const D3 &y)
{
  D3 x;
  double d;
  d = y.x;
  x.x = d / c.c0;
  d = y.y;
  d -= c.c1 * x.x;
  x.y = d / c.c2;
  d = y.z;
  d -= c.c3 * x.x;
  d -= c.c4 * x.y;
  x.z = d / c.c5;
  return x;
}

/// ohm/CovarianceVoxelCompute.h:227-267
__device__ inline void calculateSampleLikelihoods(const Cov6 &cov, const D3 &sensor, const D3 &sample,
                                                  const D3 &voxel_mean, float sensor_noise, double &p_voxel,
                                                  double &p_sample)
{
  const D3 sensor_to_sample = sample - sensor;
  const D3 sensor_ray = normalize(sensor_to_sample);
  const D3 mean_to_sensor = sensor - voxel_mean;
  const D3 a = solveTriangular(cov, sensor_ray);
  const D3 b_norm = solveTriangular(cov, mean_to_sensor);
  const double t = -dot(a, b_norm) / dot(a, a);
  const D3 voxel_ml = sensor_ray * t + sensor;
  const D3 s1 = solveTriangular(cov, voxel_ml - voxel_mean);
  p_voxel = exp(-0.5 * dot(s1, s1));
  const double sensor_noise_variance = double(sensor_noise * sensor_noise);
  const D3 d2 = voxel_ml - sample;
  p_sample = exp(-0.5 * dot(d2, d2) / sensor_noise_variance);
}

/// ohm/CovarianceVoxelCompute.h:542-635.  Returns the adjusted value; is_miss for NDT-TM.
__device__ inline float calculateMissNdt(const MapConst &mc, const Cov6 &cov, float voxel_value, bool &is_miss,
                                         const D3 &sensor, const D3 &sample, const D3 &voxel_mean,
                                         uint32_t point_count)
{
  const float inf = __int_as_float(0x7f800000);
  if (voxel_value == inf)
  {
    is_miss = true;
    return mc.miss_value;
  }
  if (point_count < mc.sample_threshold)
  {
    is_miss = true;
    return voxel_value + mc.miss_value;
  }
  double p_voxel, p_sample;
  calculateSampleLikelihoods(cov, sensor, sample, voxel_mean, mc.sensor_noise, p_voxel, p_sample);
  const double scaling_factor = 0.5 * mc.adaptation_rate;
  const double prod = p_voxel * (1.0 - p_sample);
  const double probability_update = 0.5 - scaling_factor * prod;
  is_miss = prod < scaling_factor;
  if (probability_update == probability_update)
  {
    voxel_value += float(log(probability_update / (1.0 - probability_update)));
  }
  return voxel_value;
}

/// ohm/CovarianceVoxelCompute.h:107-120
__device__ inline double packedDot(const double *A, int j, int k)
{
  const int indj = (j == 0) ? 0 : ((j == 1) ? 1 : 3);
  const int indk = (k == 0) ? 0 : ((k == 1) ? 1 : 3);
  const int m = (j <= k) ? j : k;
  double d = A[6 + k] * A[6 + j];
  for (int i = 0; i <= m; ++i)
  {
    d += A[indj + i] * A[indk + i];
  }
  return d;
}

/// ohm/CovarianceVoxelCompute.h:301-375.  Returns true when the covariance (and so the mean) was reinitialised.
/// The loops are fully unrolled so the 9-element work matrix stays in registers.
__device__ inline bool calculateHitWithCovariance(const MapConst &mc, Cov6 &cov, float &voxel_value, const D3 &sample,
                                                  const D3 &voxel_mean, uint32_t point_count)
{
  const float inf = __int_as_float(0x7f800000);
  const float initial_value = voxel_value;
  const bool was_uncertain = initial_value == inf;
  bool initialised = false;
  if (point_count == 0 || (initial_value < mc.reinit_threshold && point_count >= mc.reinit_count))
  {
    // initialiseCovariance :90-98
    const float s = 0.1f * float(mc.resolution);
    cov.c0 = cov.c2 = cov.c5 = s;
    cov.c1 = cov.c3 = cov.c4 = 0;
    initialised = true;
    point_count = 0;
  }
  voxel_value = (!was_uncertain) ? mc.hit_value + initial_value : mc.hit_value;

  const D3 sample_to_mean = (!initialised) ? sample - voxel_mean : d3(0, 0, 0);
  // unpackCovariance :152-170
  const double one_on_num_pt_plus_one = double(1) / (point_count + double(1));
  const double sc_1 = point_count ? sqrt(point_count * one_on_num_pt_plus_one) : double(1);
  const double sc_2 = one_on_num_pt_plus_one * sqrt(double(point_count));
  double A[9];
  A[0] = sc_1 * cov.c0;
  A[1] = sc_1 * cov.c1;
  A[2] = sc_1 * cov.c2;
  A[3] = sc_1 * cov.c3;
  A[4] = sc_1 * cov.c4;
  A[5] = sc_1 * cov.c5;
  A[6] = sc_2 * sample_to_mean.x;
  A[7] = sc_2 * sample_to_mean.y;
  A[8] = sc_2 * sample_to_mean.z;
  float out[6] = { cov.c0, cov.c1, cov.c2, cov.c3, cov.c4, cov.c5 };
#pragma unroll
  for (int k = 0; k < 3; ++k)
  {
    const int ind1 = (k * (k + 3)) >> 1;
    const int indk = ind1 - k;
    const double ak = sqrt(packedDot(A, k, k));
    out[ind1] = float(ak);
    if (ak > 0)
    {
      const double aki = double(1) / ak;
#pragma unroll
      for (int j = k + 1; j < 3; ++j)
      {
        const int indj = (j * (j + 1)) >> 1;
        const int indkj = indj + k;
        double c = packedDot(A, j, k) * aki;
        out[indkj] = float(c);
        c *= aki;
        A[j + 6] -= c * A[k + 6];
#pragma unroll
        for (int l = 0; l <= k; ++l)
        {
          A[indj + l] -= c * A[indk + l];
        }
      }
    }
  }
  cov.c0 = out[0];
  cov.c1 = out[1];
  cov.c2 = out[2];
  cov.c3 = out[3];
  cov.c4 = out[4];
  cov.c5 = out[5];
  return initialised;
}

/// ohm/VoxelOccupancyCompute.h:144-153 (null_update == false)
__device__ inline float occupancyAdjustDown(const MapConst &mc, float initial_value, float adjusted_value)
{
  const float inf = __int_as_float(0x7f800000);
  const bool uninitialised = initial_value == inf;
  adjusted_value = (uninitialised || (mc.sat_min < initial_value && initial_value < mc.sat_max)) ? adjusted_value :
                                                                                                 initial_value;
  return (adjusted_value != inf) ? fmaxf(mc.min_value, adjusted_value) : adjusted_value;
}

/// ohm/VoxelOccupancyCompute.h:78-87 (null_update == false)
__device__ inline float occupancyAdjustUp(const MapConst &mc, float initial_value, float adjusted_value)
{
  const float inf = __int_as_float(0x7f800000);
  const bool uninitialised = initial_value == inf;
  adjusted_value = (uninitialised || (mc.sat_min < initial_value && initial_value < mc.sat_max)) ? adjusted_value :
                                                                                                 initial_value;
  return (adjusted_value != inf) ? fminf(mc.max_value, adjusted_value) : adjusted_value;
}

/// ohm/CovarianceVoxelCompute.h:391-411
__device__ inline void calculateIntensityUpdateOnHit(const MapConst &mc, float &intensity_mean, float &intensity_cov,
                                                     float voxel_value, float intensity_sample, uint32_t point_count)
{
  const bool needs_reset = point_count == 0 || (voxel_value < mc.reinit_threshold && point_count >= mc.reinit_count);
  const float delta = intensity_mean - intensity_sample;
  const float point_count_float = float(point_count);
  const float inv = 1.0f / (point_count_float + 1.0f);
  const float new_mean = (!needs_reset) ? inv * (point_count_float * intensity_mean + intensity_sample) :
                                          intensity_sample;
  const float new_cov = (!needs_reset) ? inv * (point_count_float * intensity_cov + inv * delta * delta) :
                                         mc.initial_intensity_cov;
  intensity_mean = new_mean;
  intensity_cov = new_cov;
}

/// ohm/CovarianceVoxelCompute.h:447-505
__device__ inline void calculateHitMissUpdateOnHit(const MapConst &mc, const Cov6 &cov, float voxel_value,
                                                   uint32_t &hit_count, uint32_t &miss_count, const D3 &sensor,
                                                   const D3 &sample, const D3 &voxel_mean, uint32_t point_count)
{
  const float inf = __int_as_float(0x7f800000);
  const bool needs_reset =
    voxel_value == inf ||
    (point_count == 0 || (voxel_value < mc.reinit_threshold && point_count >= mc.reinit_count));
  const uint32_t initial_hit = (!needs_reset) ? hit_count : 0;
  const uint32_t initial_miss = (!needs_reset) ? miss_count : 0;
  double p_voxel, p_sample;
  calculateSampleLikelihoods(cov, sensor, sample, voxel_mean, mc.sensor_noise, p_voxel, p_sample);
  const double prod = p_voxel * p_sample;
  const double eta = 0.5 * mc.adaptation_rate;
  const bool inc_hit =
    needs_reset || point_count < mc.sample_threshold || (point_count >= mc.sample_threshold && prod >= eta);
  const bool inc_miss = !needs_reset && point_count >= mc.sample_threshold && prod < eta && p_voxel >= eta;
  hit_count = initial_hit + (inc_hit ? 1 : 0);
  miss_count = initial_miss + (inc_miss ? 1 : 0);
}

/// ohm/VoxelTsdfCompute.h:57-68
__device__ inline float tsdfComputeDistance(const D3 &sensor, const D3 &sample, const D3 &voxel_centre)
{
  const D3 sensor_to_voxel = voxel_centre - sensor;
  const D3 sensor_to_sample = sample - sensor;
  const float distance_g = float(sqrt(dot(sensor_to_sample, sensor_to_sample)));
  const float distance_g_v = float(dot(sensor_to_voxel, sensor_to_sample)) / distance_g;
  return distance_g - distance_g_v;
}

/// ohm/VoxelTsdfCompute.h:87-136 given the precomputed sdf.
__device__ inline void tsdfUpdate(const MapConst &mc, float sdf, float &voxel_weight, float &voxel_distance)
{
  const float trunc = mc.tsdf_trunc;
  const float initial_weight = voxel_weight;
  float updated_weight = 1.0f;
  updated_weight *= (mc.tsdf_dropoff > 0) ? ((trunc + sdf) / (trunc - mc.tsdf_dropoff)) : 1.0f;
  updated_weight = (updated_weight < 0.0f) ? 0.0f : updated_weight;  // std::max(updated_weight, 0.0f)
  updated_weight *= (mc.tsdf_sparsity > 0 && fabsf(sdf) < trunc) ? mc.tsdf_sparsity : 1.0f;
  const float new_weight = initial_weight + updated_weight;
  const bool near_zero_weight = fabsf(new_weight) < 0.00001f;
  const float new_sdf = (!near_zero_weight) ? (sdf * updated_weight + voxel_distance * initial_weight) / new_weight :
                                              0.0f;
  // std::min(a, b) = (b < a) ? b : a; std::max(a, b) = (a < b) ? b : a
  const float clamped =
    (new_sdf > 0.0f) ? ((new_sdf < trunc) ? new_sdf : trunc) : ((-trunc < new_sdf) ? new_sdf : -trunc);
  voxel_distance = (!near_zero_weight) ? clamped : voxel_distance;
  voxel_weight = (!near_zero_weight) ? ((mc.tsdf_max_weight < new_weight) ? mc.tsdf_max_weight : new_weight) :
                                       initial_weight;
}
}  // namespace ohmhip

#endif  // OHMHIP_NDT_TSDF_DEVICE_H
