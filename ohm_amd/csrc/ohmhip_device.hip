// ohmhip_device.hip -- device / stream / event / buffer plumbing of the C ABI (replaces the subset of gputil the
// ray-integration path uses; see include/ohmhip.h for the reference citations).  gfx950 / ROCm only.
#include "ohmhip_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <new>

extern "C" {

#ifndef OHMHIP_BUILD_ID
#define OHMHIP_BUILD_ID "unversioned"
#endif
const char *ohmhip_build_id(void)
{
  return OHMHIP_BUILD_ID;
}

const char *ohmhip_error_string(int status)
{
  switch (status)
  {
  case OHMHIP_OK:
    return "ok";
  case OHMHIP_ERR_INVALID_ARG:
    return "ohmhip: invalid argument";
  case OHMHIP_ERR_NO_DEVICE:
    return "ohmhip: no HIP device available";
  case OHMHIP_ERR_CAPACITY:
    return "ohmhip: region pool capacity exhausted";
  case OHMHIP_ERR_UNSUPPORTED:
    return "ohmhip: unsupported flag or layout";
  case OHMHIP_ERR_NOT_FOUND:
    return "ohmhip: region not found";
  case OHMHIP_ERR_INTERNAL:
    return "ohmhip: internal error";
  case OHMHIP_ERR_PEER:
    return "ohmhip: collective abandoned, another rank failed";
  default:
    break;
  }
  if (status > 0)
  {
    return hipGetErrorString(static_cast<hipError_t>(status));
  }
  return "ohmhip: unknown error";
}

int ohmhip_device_count(int *count)
try
{
  if (!count)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  *count = 0;
  const hipError_t err = hipGetDeviceCount(count);
  if (err != hipSuccess)
  {
    *count = 0;
    return (err == hipErrorNoDevice) ? OHMHIP_ERR_NO_DEVICE : static_cast<int>(err);
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_device_select(int device)
try
{
  OHMHIP_CHECK(hipSetDevice(device));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_device_get_info(int device, ohmhip_device_info *info)
try
{
  if (!info)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  hipDeviceProp_t prop;
  OHMHIP_CHECK(hipGetDeviceProperties(&prop, device));
  std::memset(info, 0, sizeof(*info));
  std::snprintf(info->arch, sizeof(info->arch), "%s", prop.gcnArchName);
  if (prop.name[0])
  {
    std::snprintf(info->name, sizeof(info->name), "%s", prop.name);
  }
  else
  {
    // (some driver / runtime combinations leave the marketing name empty: fall back to the architecture)
    std::snprintf(info->name, sizeof(info->name), "AMD GPU (%s, %d CUs)", prop.gcnArchName, prop.multiProcessorCount);
  }
  info->total_memory = prop.totalGlobalMem;
  info->max_allocation = prop.totalGlobalMem;
  info->compute_units = prop.multiProcessorCount;
  info->lds_bytes_per_block = static_cast<int>(prop.sharedMemPerBlock);
  info->unified_memory = prop.integrated;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_stream_create(ohmhip_stream_t *stream)
try
{
  if (!stream)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  hipStream_t s = nullptr;
  OHMHIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  *stream = new (std::nothrow) ohmhip_stream_s{ s };
  return *stream ? OHMHIP_OK : OHMHIP_ERR_INTERNAL;
}
OHMHIP_ABI_CATCH

int ohmhip_stream_destroy(ohmhip_stream_t stream)
try
{
  if (!stream)
  {
    return OHMHIP_OK;
  }
  const hipError_t err = hipStreamDestroy(stream->stream);
  delete stream;
  return static_cast<int>(err);
}
OHMHIP_ABI_CATCH

int ohmhip_stream_finish(ohmhip_stream_t stream)
try
{
  OHMHIP_CHECK(hipStreamSynchronize(stream ? stream->stream : nullptr));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_stream_wait_event(ohmhip_stream_t stream, ohmhip_event_t event)
try
{
  if (!event)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!event->recorded)
  {
    return OHMHIP_OK;  // An invalid (never recorded) gputil::Event is a no-op dependency.
  }
  OHMHIP_CHECK(hipStreamWaitEvent(stream ? stream->stream : nullptr, event->event, 0));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_event_create(ohmhip_event_t *event)
try
{
  if (!event)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  hipEvent_t e = nullptr;
  OHMHIP_CHECK(hipEventCreate(&e));
  *event = new (std::nothrow) ohmhip_event_s{ e, false };
  return *event ? OHMHIP_OK : OHMHIP_ERR_INTERNAL;
}
OHMHIP_ABI_CATCH

int ohmhip_event_destroy(ohmhip_event_t event)
try
{
  if (!event)
  {
    return OHMHIP_OK;
  }
  const hipError_t err = hipEventDestroy(event->event);
  delete event;
  return static_cast<int>(err);
}
OHMHIP_ABI_CATCH

int ohmhip_event_record(ohmhip_event_t event, ohmhip_stream_t stream)
try
{
  if (!event)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(hipEventRecord(event->event, stream ? stream->stream : nullptr));
  event->recorded = true;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_event_wait(ohmhip_event_t event)
try
{
  if (!event)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!event->recorded)
  {
    return OHMHIP_OK;
  }
  OHMHIP_CHECK(hipEventSynchronize(event->event));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_event_is_complete(ohmhip_event_t event, int *complete)
try
{
  if (!event || !complete)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (!event->recorded)
  {
    *complete = 1;
    return OHMHIP_OK;
  }
  const hipError_t err = hipEventQuery(event->event);
  if (err == hipSuccess)
  {
    *complete = 1;
    return OHMHIP_OK;
  }
  if (err == hipErrorNotReady)
  {
    *complete = 0;
    return OHMHIP_OK;
  }
  return static_cast<int>(err);
}
OHMHIP_ABI_CATCH

int ohmhip_event_elapsed_ms(ohmhip_event_t start, ohmhip_event_t stop, float *ms)
try
{
  if (!start || !stop || !ms)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(hipEventElapsedTime(ms, start->event, stop->event));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_device_synchronize(void)
try
{
  OHMHIP_CHECK(hipDeviceSynchronize());
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_create(ohmhip_buffer_t *buffer, size_t bytes, unsigned flags)
try
{
  if (!buffer)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  void *ptr = nullptr;
  if (bytes)
  {
    if (flags & OHMHIP_BF_HOST_ACCESS)
    {
      OHMHIP_CHECK(hipHostMalloc(&ptr, bytes, hipHostMallocDefault));
    }
    else
    {
      OHMHIP_CHECK(hipMalloc(&ptr, bytes));
    }
  }
  *buffer = new (std::nothrow) ohmhip_buffer_s{ ptr, bytes, flags };
  return *buffer ? OHMHIP_OK : OHMHIP_ERR_INTERNAL;
}
OHMHIP_ABI_CATCH

static int freeBufferMemory(ohmhip_buffer_t buffer)
{
  hipError_t err = hipSuccess;
  if (buffer->ptr)
  {
    err = (buffer->flags & OHMHIP_BF_HOST_ACCESS) ? hipHostFree(buffer->ptr) : hipFree(buffer->ptr);
    buffer->ptr = nullptr;
    buffer->bytes = 0;
  }
  return static_cast<int>(err);
}

int ohmhip_buffer_destroy(ohmhip_buffer_t buffer)
try
{
  if (!buffer)
  {
    return OHMHIP_OK;
  }
  const int err = freeBufferMemory(buffer);
  delete buffer;
  return err;
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_resize(ohmhip_buffer_t buffer, size_t bytes, size_t *actual)
try
{
  if (!buffer)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (bytes > buffer->bytes)
  {
    // Grow-only; contents are not preserved, as with gputil::Buffer::resize (gputil/gpuBuffer.h:161).
    const unsigned flags = buffer->flags;
    const int err = freeBufferMemory(buffer);
    if (err)
    {
      return err;
    }
    void *ptr = nullptr;
    if (flags & OHMHIP_BF_HOST_ACCESS)
    {
      OHMHIP_CHECK(hipHostMalloc(&ptr, bytes, hipHostMallocDefault));
    }
    else
    {
      OHMHIP_CHECK(hipMalloc(&ptr, bytes));
    }
    buffer->ptr = ptr;
    buffer->bytes = bytes;
  }
  if (actual)
  {
    *actual = buffer->bytes;
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_size(ohmhip_buffer_t buffer, size_t *bytes)
try
{
  if (!buffer || !bytes)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  *bytes = buffer->bytes;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_ptr(ohmhip_buffer_t buffer, void **device_ptr)
try
{
  if (!buffer || !device_ptr)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  *device_ptr = buffer->ptr;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

static int copyCommon(void *dst, const void *src, size_t bytes, hipMemcpyKind kind, ohmhip_stream_t stream,
                      ohmhip_event_t block_on, ohmhip_event_t completion)
{
  if (!stream)
  {
    if (block_on && block_on->recorded)
    {
      OHMHIP_CHECK(hipEventSynchronize(block_on->event));
    }
    OHMHIP_CHECK(hipMemcpy(dst, src, bytes, kind));
    if (completion)
    {
      OHMHIP_CHECK(hipEventRecord(completion->event, nullptr));
      completion->recorded = true;
    }
    return OHMHIP_OK;
  }
  if (block_on && block_on->recorded)
  {
    OHMHIP_CHECK(hipStreamWaitEvent(stream->stream, block_on->event, 0));
  }
  OHMHIP_CHECK(hipMemcpyAsync(dst, src, bytes, kind, stream->stream));
  if (completion)
  {
    OHMHIP_CHECK(hipEventRecord(completion->event, stream->stream));
    completion->recorded = true;
  }
  return OHMHIP_OK;
}

int ohmhip_buffer_write(ohmhip_buffer_t buffer, const void *src, size_t bytes, size_t dst_offset,
                        ohmhip_stream_t stream, ohmhip_event_t block_on, ohmhip_event_t completion)
try
{
  if (!buffer || (!src && bytes) || dst_offset + bytes > buffer->bytes)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  return copyCommon(static_cast<char *>(buffer->ptr) + dst_offset, src, bytes, hipMemcpyHostToDevice, stream, block_on,
                    completion);
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_read(ohmhip_buffer_t buffer, void *dst, size_t bytes, size_t src_offset, ohmhip_stream_t stream,
                       ohmhip_event_t block_on, ohmhip_event_t completion)
try
{
  if (!buffer || (!dst && bytes) || src_offset + bytes > buffer->bytes)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  return copyCommon(dst, static_cast<const char *>(buffer->ptr) + src_offset, bytes, hipMemcpyDeviceToHost, stream,
                    block_on, completion);
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_fill(ohmhip_buffer_t buffer, int byte_value, size_t bytes, size_t offset, ohmhip_stream_t stream)
try
{
  if (!buffer || offset + bytes > buffer->bytes)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (buffer->flags & OHMHIP_BF_HOST_ACCESS)
  {
    std::memset(static_cast<char *>(buffer->ptr) + offset, byte_value, bytes);
    return OHMHIP_OK;
  }
  if (stream)
  {
    OHMHIP_CHECK(hipMemsetAsync(static_cast<char *>(buffer->ptr) + offset, byte_value, bytes, stream->stream));
  }
  else
  {
    OHMHIP_CHECK(hipMemset(static_cast<char *>(buffer->ptr) + offset, byte_value, bytes));
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

namespace
{
struct FillPattern
{
  unsigned char bytes[64];
};

__global__ void k_fill_pattern(unsigned char *dst, FillPattern pattern, uint32_t pattern_size, size_t count)
{
  const size_t stride = size_t(gridDim.x) * blockDim.x;
  for (size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x; i < count; i += stride)
  {
    dst[i] = pattern.bytes[i % pattern_size];
  }
}
}  // namespace

int ohmhip_buffer_fill_pattern(ohmhip_buffer_t buffer, const void *pattern, size_t pattern_size, size_t bytes,
                               size_t offset, ohmhip_stream_t stream, ohmhip_event_t block_on,
                               ohmhip_event_t completion)
try
{
  if (!buffer || !pattern || pattern_size == 0 || offset + bytes > buffer->bytes)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  if (pattern_size > sizeof(FillPattern))
  {
    return OHMHIP_ERR_UNSUPPORTED;
  }
  hipStream_t s = stream ? stream->stream : nullptr;
  if (block_on && block_on->recorded)
  {
    if (stream)
    {
      OHMHIP_CHECK(hipStreamWaitEvent(s, block_on->event, 0));
    }
    else
    {
      OHMHIP_CHECK(hipEventSynchronize(block_on->event));
    }
  }
  unsigned char *dst = static_cast<unsigned char *>(buffer->ptr) + offset;
  if (bytes)
  {
    if (buffer->flags & OHMHIP_BF_HOST_ACCESS)
    {
      if (stream)
      {
        OHMHIP_CHECK(hipStreamSynchronize(s));  // host-side fill: ordered behind what the stream already holds
      }
      const unsigned char *p = static_cast<const unsigned char *>(pattern);
      for (size_t i = 0; i < bytes; ++i)
      {
        dst[i] = p[i % pattern_size];
      }
    }
    else
    {
      FillPattern fp;
      std::memcpy(fp.bytes, pattern, pattern_size);
      const unsigned blocks = unsigned(std::min<size_t>((bytes + 255) / 256, 4096));
      hipLaunchKernelGGL(k_fill_pattern, dim3(blocks), dim3(256), 0, s, dst, fp, uint32_t(pattern_size), bytes);
      OHMHIP_CHECK(hipGetLastError());
    }
  }
  if (completion)
  {
    OHMHIP_CHECK(hipEventRecord(completion->event, s));
    completion->recorded = true;
  }
  if (!stream)
  {
    OHMHIP_CHECK(hipStreamSynchronize(nullptr));
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_copy(ohmhip_buffer_t dst, size_t dst_offset, ohmhip_buffer_t src, size_t src_offset, size_t bytes,
                       ohmhip_stream_t stream, ohmhip_event_t block_on, ohmhip_event_t completion)
try
{
  if (!dst || !src || dst_offset + bytes > dst->bytes || src_offset + bytes > src->bytes)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  return copyCommon(static_cast<char *>(dst->ptr) + dst_offset, static_cast<const char *>(src->ptr) + src_offset, bytes,
                    hipMemcpyDefault, stream, block_on, completion);
}
OHMHIP_ABI_CATCH

int ohmhip_buffer_flags(ohmhip_buffer_t buffer, unsigned *flags)
try
{
  if (!buffer || !flags)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  *flags = buffer->flags;
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_host_alloc(void **ptr, size_t bytes)
try
{
  if (!ptr)
  {
    return OHMHIP_ERR_INVALID_ARG;
  }
  OHMHIP_CHECK(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

int ohmhip_host_free(void *ptr)
try
{
  if (ptr)
  {
    OHMHIP_CHECK(hipHostFree(ptr));
  }
  return OHMHIP_OK;
}
OHMHIP_ABI_CATCH

}  // extern "C"
