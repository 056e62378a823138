// gputilHipDetail.h -- private parts of the HIP backend of gputil::Device / Queue / Event (the reference's gputil/*.h
// declare the classes; gputil/cuda/*Detail.h and gputil/cl/*Detail.h are the reference's counterparts of this file).
#ifndef GPUTIL_HIP_DETAIL_H
#define GPUTIL_HIP_DETAIL_H

#include <gputil/gpuDeviceInfo.h>

#include <ohmhip.h>

#include <atomic>
#include <string>

namespace gputil
{
class Event;
struct EventDetail;
/// Hand a freshly created detail (reference count 1) to an invalid Event (gputilHip.cpp; used by Queue::mark()).
void adoptEventDetail(Event &event, EventDetail *detail);

struct DeviceDetail
{
  int device = -1;  ///< HIP device index; -1: invalid
  DeviceInfo info;
  std::string name;
  std::string description;
  std::string search_paths;
  uint64_t device_memory = 0;
  uint64_t max_allocation = 0;
  bool unified_memory = false;
  int debug_level = 0;
};

/// A stream.  The default queue (stream == nullptr) is the legacy default stream as far as this backend is concerned:
/// the map integrates on its own streams inside libohmhip.so, so the queue objects handed out here only serve code
/// that marks / waits (ohm::GpuCache::gpuQueue(), tests).
struct QueueDetail
{
  ohmhip_stream_t stream = nullptr;
  bool owned = false;
  bool synchronous = false;
  ~QueueDetail()
  {
    if (stream && owned)
    {
      ohmhip_stream_destroy(stream);
    }
  }
};

/// gputil::Buffer: one C-ABI buffer (device memory, or -- kBfHostAccess -- pinned host memory the device maps) plus
/// the size the caller asked for (the allocation itself is padded).
struct BufferDetail
{
  ohmhip_buffer_t buffer = nullptr;
  size_t requested = 0;
  unsigned flags = 0;
  int device = -1;
};

/// Reference counted event (gputil::Event is copyable; copies share the underlying event).
struct EventDetail
{
  ohmhip_event_t event = nullptr;
  std::atomic_int references{ 1 };
};
}  // namespace gputil

#endif  // GPUTIL_HIP_DETAIL_H
