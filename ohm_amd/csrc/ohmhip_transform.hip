// ohmhip_transform.hip -- GpuTransformSamples equivalent: sensor-frame samples + a timestamped trajectory in, world-frame
// ray pairs (sensor origin, sample) out, resident on the device and ready for ohmhip_map_integrate_rays_device().
//
// Semantics follow ohmgpu/GpuTransformSamples.cpp:97-210 (host side: sample filter, compaction) and
// ohmgpu/gpu/TransformSamples.cl:13-228 (kernel: bracketing search, lerp of the translation, the reference's rotation
// rule rot[from] * slerp(rot[from], rot[to], f)), in fp64 throughout -- the reference kernel is fp32 only because it had
// to run on devices without doubles (it rebases the time stamps to keep them representable).
#include "ohmhip_internal.h"

#include <hip/hip_runtime.h>

#include <string.h>

#include <rocprim/rocprim.hpp>

#include <cmath>

namespace ohmhip
{
struct Quat
{
  double x, y, z, w;
};

/// TransformSamples.cl:13-54
__device__ inline Quat slerp(const Quat &from, const Quat &to, double f)
{
  if (from.x == to.x && from.y == to.y && from.z == to.z && from.w == to.w)
  {
    return from;
  }
  double cos_angle = ((from.x * to.x + from.y * to.y) + from.z * to.z) + from.w * to.w;
  Quat temp = to;
  if (!(cos_angle >= 0))
  {
    temp.x = -1.0 * to.x;
    temp.y = -1.0 * to.y;
    temp.z = -1.0 * to.z;
    temp.w = -1.0 * to.w;
    cos_angle = -1.0 * cos_angle;
  }
  double coeff0, coeff1;
  if (1.0 - cos_angle > 1e-12)
  {
    const double angle = acos(cos_angle);
    const double inv_sin = 1.0 / sin(angle);
    coeff0 = sin((1.0 - f) * angle) * inv_sin;
    coeff1 = sin(f * angle) * inv_sin;
  }
  else
  {
    coeff0 = 1.0 - f;
    coeff1 = f;
  }
  Quat r;
  r.x = coeff0 * from.x + coeff1 * temp.x;
  r.y = coeff0 * from.y + coeff1 * temp.y;
  r.z = coeff0 * from.z + coeff1 * temp.z;
  r.w = coeff0 * from.w + coeff1 * temp.w;
  return r;
}

/// TransformSamples.cl:57-65
__device__ inline Quat quatMul(const Quat &a, const Quat &b)
{
  Quat q;
  q.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  q.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  q.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  q.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return q;
}

/// ohmgpu/GpuTransformSamples.cpp:47-60 (goodSample).  As in the reference the SQUARED length is compared with
/// max_range itself.
__global__ void __launch_bounds__(256)
  k_sample_flags(const double *__restrict__ local, uint32_t n, double max_range, uint32_t *__restrict__ good)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n)
  {
    return;
  }
  const double x = local[3 * size_t(i)], y = local[3 * size_t(i) + 1], z = local[3 * size_t(i) + 2];
  const bool nan = (x != x) || (y != y) || (z != z);
  const bool far = ((x * x + y * y) + z * z) > max_range;
  good[i] = (!nan && !far) ? 1u : 0u;
}

__global__ void __launch_bounds__(256)
  k_transform_samples(const double *__restrict__ times, const double *__restrict__ positions,
                      const double *__restrict__ rotations, uint32_t transform_count,
                      const double *__restrict__ sample_times, const double *__restrict__ local, uint32_t n,
                      const uint32_t *__restrict__ good, const uint32_t *__restrict__ slot, double *__restrict__ out)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !good[i])
  {
    return;
  }
  double sample_time = sample_times[i];
  // TransformSamples.cl:128-196: the pair of transforms bracketing the sample time.
  uint32_t from = 0;
  uint32_t to = transform_count - 1;
  if (transform_count > 2)
  {
    if (times[0] <= sample_time && sample_time <= times[transform_count - 1])
    {
      uint32_t iterations = 0;
      while (from <= to && iterations < 100000u)
      {
        ++iterations;
        const uint32_t mid_low = (from + to) / 2;
        const uint32_t mid_high = min(mid_low + 1, transform_count - 1);
        if (sample_time >= times[mid_low] && sample_time <= times[mid_high])
        {
          from = mid_low;
          to = mid_high;
          break;
        }
        else if (sample_time <= times[mid_low])
        {
          to = mid_low - 1;
        }
        else
        {
          from = mid_low + 1;
        }
      }
    }
    else if (sample_time < times[0])
    {
      sample_time = times[0];
      from = to = 0;
    }
    else
    {
      sample_time = times[transform_count - 1];
      from = to = transform_count - 1;
    }
  }
  // A sample outside the trajectory takes the end pose (the reference divides 0 / 0 there and emits NaN rays).
  const double span = times[to] - times[from];
  const double f = (span != 0) ? (sample_time - times[from]) / span : 0.0;
  double position[3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
  {
    position[a] = positions[3 * size_t(from) + a] + f * (positions[3 * size_t(to) + a] - positions[3 * size_t(from) + a]);
  }
  Quat qf, qt;
  qf.x = rotations[4 * size_t(from) + 0];
  qf.y = rotations[4 * size_t(from) + 1];
  qf.z = rotations[4 * size_t(from) + 2];
  qf.w = rotations[4 * size_t(from) + 3];
  qt.x = rotations[4 * size_t(to) + 0];
  qt.y = rotations[4 * size_t(to) + 1];
  qt.z = rotations[4 * size_t(to) + 2];
  qt.w = rotations[4 * size_t(to) + 3];
  const Quat q = quatMul(qf, slerp(qf, qt, f));
  // TransformSamples.cl:68-91
  const double vx = local[3 * size_t(i)], vy = local[3 * size_t(i) + 1], vz = local[3 * size_t(i) + 2];
  const double xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
  const double yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w;
  const double zz = q.z * q.z, zw = q.z * q.w;
  const double rx = (1 - 2 * (yy + zz)) * vx + (2 * (xy - zw)) * vy + (2 * (xz + yw)) * vz;
  const double ry = (2 * (xy + zw)) * vx + (1 - 2 * (xx + zz)) * vy + (2 * (yz - xw)) * vz;
  const double rz = (2 * (xz - yw)) * vx + (2 * (yz + xw)) * vy + (1 - 2 * (xx + yy)) * vz;
  double *o = out + 6 * size_t(slot[i]);
  o[0] = position[0];
  o[1] = position[1];
  o[2] = position[2];
  o[3] = position[0] + rx;
  o[4] = position[1] + ry;
  o[5] = position[2] + rz;
}
}  // namespace ohmhip

using namespace ohmhip;

extern "C" int ohmhip_transform_samples(const double *transform_times, const double *transform_translations,
                                        const double *transform_rotations_xyzw, uint32_t transform_count,
                                        const double *sample_times, const double *local_samples, uint32_t point_count,
                                        double max_range, ohmhip_stream_t stream, ohmhip_buffer_t output,
                                        uint32_t *ray_elements)
try
{
  if (ray_elements)
  {
    *ray_elements = 0;
  }
  if (!output || (point_count && (!sample_times || !local_samples)) ||
      (transform_count && (!transform_times || !transform_translations || !transform_rotations_xyzw)))
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (point_count == 0 || transform_count == 0)
  {
    return OHMHIP_OK;  // GpuTransformSamples.cpp:103-106
  }
  hipStream_t s = stream ? stream->stream : nullptr;
  size_t actual = 0;
  int status = ohmhip_buffer_resize(output, sizeof(double) * 6 * size_t(point_count), &actual);
  if (status)
  {
    return status;
  }
  const size_t tn = transform_count, pn = point_count;
  const size_t bytes_d = sizeof(double) * (8 * tn + 4 * pn);
  double *d_d = nullptr;
  uint32_t *d_u = nullptr;
  void *d_temp = nullptr;
  auto cleanup = [&]() {
    (void)hipFree(d_d);
    (void)hipFree(d_u);
    (void)hipFree(d_temp);
  };
  if ((status = hipMalloc(reinterpret_cast<void **>(&d_d), bytes_d)) != 0 ||
      (status = hipMalloc(reinterpret_cast<void **>(&d_u), sizeof(uint32_t) * (2 * pn + 1))) != 0)
  {
    cleanup();
    return status;
  }
  double *d_times = d_d, *d_pos = d_times + tn, *d_rot = d_pos + 3 * tn, *d_stimes = d_rot + 4 * tn;
  double *d_local = d_stimes + pn;
  uint32_t *d_good = d_u, *d_slot = d_u + pn;
  auto up = [&](double *dst, const double *src, size_t count) {
    return int(hipMemcpyAsync(dst, src, sizeof(double) * count, hipMemcpyHostToDevice, s));
  };
  status = up(d_times, transform_times, tn);
  status = status ? status : up(d_pos, transform_translations, 3 * tn);
  status = status ? status : up(d_rot, transform_rotations_xyzw, 4 * tn);
  status = status ? status : up(d_stimes, sample_times, pn);
  status = status ? status : up(d_local, local_samples, 3 * pn);
  if (status)
  {
    cleanup();
    return status;
  }
  const dim3 grid(uint32_t((pn + 255) / 256)), block(256);
  hipLaunchKernelGGL(k_sample_flags, grid, block, 0, s, d_local, point_count, max_range, d_good);
  // Stable compaction of the rejected samples (the reference skips them while staging, GpuTransformSamples.cpp:131-142).
  size_t temp_bytes = 0;
  status = int(rocprim::exclusive_scan(nullptr, temp_bytes, d_good, d_slot, 0u, pn, rocprim::plus<uint32_t>(), s));
  if (!status)
  {
    status = int(hipMalloc(&d_temp, temp_bytes));
  }
  if (!status)
  {
    status = int(rocprim::exclusive_scan(d_temp, temp_bytes, d_good, d_slot, 0u, pn, rocprim::plus<uint32_t>(), s));
  }
  if (!status)
  {
    void *out_ptr = nullptr;
    (void)ohmhip_buffer_ptr(output, &out_ptr);
    hipLaunchKernelGGL(k_transform_samples, grid, block, 0, s, d_times, d_pos, d_rot, transform_count, d_stimes, d_local,
                       point_count, d_good, d_slot, static_cast<double *>(out_ptr));
    uint32_t tail[2] = { 0, 0 };
    status = int(hipMemcpyAsync(&tail[0], d_good + (pn - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    status = status ? status : int(hipMemcpyAsync(&tail[1], d_slot + (pn - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    status = status ? status : int(hipStreamSynchronize(s));
    if (!status && ray_elements)
    {
      *ray_elements = 2u * (tail[0] + tail[1]);
    }
  }
  cleanup();
  return status;
}
OHMHIP_ABI_CATCH
