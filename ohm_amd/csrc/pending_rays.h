// pending_rays.h -- the two staging slots of host / deferred device batches: growth, settling an asynchronous launch,
// flushing what is pending (batch coalescing, ohmhip_map_set_batch_coalescing / _set_async_launch).
//
// Part of ohmhip_map.hip's translation unit (included there, in order): not a stand-alone header.
#ifndef OHMHIP_PENDING_RAYS_H
#define OHMHIP_PENDING_RAYS_H

namespace
{
inline char *slotRays(ohmhip_map_s::RaySlot &sl) { return sl.h; }
inline char *slotTimes(ohmhip_map_s::RaySlot &sl) { return sl.h + sl.capacity * 48; }
inline char *slotIntens(ohmhip_map_s::RaySlot &sl) { return sl.h + sl.capacity * 56; }
inline char *slotFilterFlags(ohmhip_map_s::RaySlot &sl) { return sl.h + sl.capacity * 60; }

/// Make room for `rays` rays in the filling slot, keeping what is pending in it.
int growRaySlot(ohmhip_map_t m, ohmhip_map_s::RaySlot &sl, size_t rays)
{
  if (rays <= sl.capacity)
  {
    return OHMHIP_OK;
  }
  ohmhip_map_s::RaySlot grown;
  grown.capacity = std::max<size_t>(rays + rays / 4, 4096);
  void *block = nullptr;
  OHMHIP_CHECK(hipHostMalloc(&block, grown.capacity * 61, hipHostMallocDefault));
  grown.h = static_cast<char *>(block);
  if (m->pending_rays)
  {
    std::memcpy(slotRays(grown), slotRays(sl), m->pending_rays * 48);
    std::memcpy(slotTimes(grown), slotTimes(sl), m->pending_rays * 8);
    std::memcpy(slotIntens(grown), slotIntens(sl), m->pending_rays * 4);
    std::memcpy(slotFilterFlags(grown), slotFilterFlags(sl), m->pending_rays);
  }
  if (sl.h)
  {
    OHMHIP_CHECK(hipHostFree(sl.h));
  }
  sl.h = grown.h;
  sl.capacity = grown.capacity;
  return OHMHIP_OK;
}

int integrateRaysDevice(ohmhip_map_t m, const double *d_rays, size_t element_count, const float *d_intensities,
                        const double *d_timestamps, unsigned ray_flags, size_t *integrated,
                        const unsigned char *d_filter_flags = nullptr);
int validateBatchRequest(ohmhip_map_t m, unsigned ray_flags);

/// Wait for the launch thread to finish the batch handed to it (ohmhip_map_set_async_launch) and collect its status.
int settleLaunch(ohmhip_map_t m)
{
  if (!m->launch_busy)
  {
    return OHMHIP_OK;
  }
  m->launch_thread->wait();
  m->launch_busy = false;
  const int err = m->launch_result;
  m->launch_result = OHMHIP_OK;
  return err;
}

/// Launch what the filling slot holds: H2D on the copy stream, the batch on the compute stream behind it.
/// `may_hand_over`: the caller is the host-pointer integrate call itself and needs nothing from the batch -- with
/// ohmhip_map_set_async_launch the launch sequence then runs on the map's thread.  Everybody else (the observers,
/// OHMHIP_SETTLE) gets the batch fully launched before this returns.
int flushPendingRays(ohmhip_map_t m, size_t *integrated = nullptr, bool may_hand_over = false)
{
  // One batch at a time is being launched; its error surfaces here.  The rays waiting in the filling slot are NOT that
  // batch's: they stay queued and run with the next flush -- which must then send the whole block again, because calls
  // appended from now on only reach the pinned block (a stale "uploaded" flag made that flush skip both the
  // resize of the device copy and the transfer).
  {
    const int settle_err = settleLaunch(m);
    if (settle_err != OHMHIP_OK)
    {
      m->ray_slots[m->fill_slot].rays_uploaded = false;
      return settle_err;
    }
  }
  const size_t n = m->pending_rays;
  if (n == 0)
  {
    return OHMHIP_OK;
  }
  ohmhip_map_s::RaySlot &sl = m->ray_slots[m->fill_slot];
  m->pending_rays = 0;
  m->pending_calls = 0;
  const bool on_device = m->pending_on_device;
  m->pending_on_device = false;
  const double *d_ts = nullptr;
  const float *d_int = nullptr;
  const unsigned char *d_ff = nullptr;
  if (on_device)
  {
    // (device-pointer calls: the copy stream has the device-to-device copies queued already)
    d_ts = m->pending_times ? static_cast<const double *>(sl.d_times.ptr) : nullptr;
    d_int = m->pending_intens ? static_cast<const float *>(sl.d_intens.ptr) : nullptr;
  }
  else
  {
    if (!sl.rays_uploaded)
    {
      OHMHIP_CHECK(sl.d_rays.ensure(n * 48, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_rays.ptr, slotRays(sl), n * 48, hipMemcpyHostToDevice, m->copy_stream));
    }
    sl.rays_uploaded = false;
    if (m->pending_times)
    {
      OHMHIP_CHECK(sl.d_times.ensure(n * 8, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_times.ptr, slotTimes(sl), n * 8, hipMemcpyHostToDevice, m->copy_stream));
      d_ts = static_cast<const double *>(sl.d_times.ptr);
    }
    if (m->pending_intens)
    {
      OHMHIP_CHECK(sl.d_intens.ensure(n * 4, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_intens.ptr, slotIntens(sl), n * 4, hipMemcpyHostToDevice, m->copy_stream));
      d_int = static_cast<const float *>(sl.d_intens.ptr);
    }
    if (m->pending_fflags)
    {
      OHMHIP_CHECK(sl.d_fflags.ensure(n, false, m->stream));
      OHMHIP_CHECK(hipMemcpyAsync(sl.d_fflags.ptr, slotFilterFlags(sl), n, hipMemcpyHostToDevice, m->copy_stream));
      d_ff = static_cast<const unsigned char *>(sl.d_fflags.ptr);
    }
  }
  OHMHIP_CHECK(hipEventRecord(sl.uploaded, m->copy_stream));
  OHMHIP_CHECK(hipStreamWaitEvent(m->stream, sl.uploaded, 0));
  OHMHIP_CHECK(hipStreamWaitEvent(m->front_stream, sl.uploaded, 0));  // (the set-up pass reads the rays first)
  if (may_hand_over && m->async_launch && !on_device && !integrated)
  {
    // The launch sequence blocks on the batch's plan summary in its middle: it runs on the launch thread, the caller
    // goes on (typically to stage its next block into the other slot, whose upload then runs beside this wait).
    if (!m->launch_thread)
    {
      m->launch_thread.reset(new StagePool(1));
    }
    ohmhip_map_s::RaySlot *slot = &sl;
    const double *d_r = static_cast<const double *>(sl.d_rays.ptr);
    const unsigned flags = m->pending_flags;
    sl.in_flight = true;
    m->fill_slot ^= 1;
    m->launch_busy = true;
    m->launch_thread->start(1, [m, slot, d_r, n, d_int, d_ts, flags, d_ff](unsigned) {
      int err = int(hipSetDevice(m->device));
      if (err == 0)
      {
        err = integrateRaysDevice(m, d_r, n * 2, d_int, d_ts, flags, nullptr, d_ff);
      }
      const int rec = int(hipEventRecord(slot->done, m->stream));
      m->launch_result = err ? err : rec;
    });
    return OHMHIP_OK;
  }
  const int err = integrateRaysDevice(m, static_cast<const double *>(sl.d_rays.ptr), n * 2, d_int, d_ts,
                                      m->pending_flags, integrated, d_ff);
  OHMHIP_CHECK(hipEventRecord(sl.done, m->stream));
  sl.in_flight = true;
  m->fill_slot ^= 1;
  return err;
}

#define OHMHIP_SETTLE(m)                          \
  if (m)                                          \
  {                                               \
    const int settle_err_ = flushPendingRays(m);  \
    if (settle_err_ != OHMHIP_OK)                 \
    {                                             \
      return settle_err_;                         \
    }                                             \
  }
}  // namespace


#endif  // OHMHIP_PENDING_RAYS_H
