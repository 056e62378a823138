// gputil_hip.h -- C++14 adaptor with the shape of the reference's `gputil` classes over the ohmhip C ABI, so that
// reference-style host code (ohmgpu/GpuMap.cpp, tests/gputiltest/*) keeps its vocabulary: gputil::Device, Queue,
// Event, Buffer, PinnedBuffer, ApiException.  Header only; links libohmhip.so; no HIP headers needed by callers.
//
// Reference interfaces mirrored (file:line in the reference checkout):
//   gputil::Device        gputil/gpuDevice.h:23      gputil::Queue   gputil/gpuQueue.h:39
//   gputil::Event         gputil/gpuEvent.h:23       gputil::Buffer  gputil/gpuBuffer.h:73
//   gputil::PinnedBuffer  gputil/gpuPinnedBuffer.h:28   gputil::ApiException  gputil/gpuApiException.h
#ifndef OHMHIP_GPUTIL_HIP_H
#define OHMHIP_GPUTIL_HIP_H

#include <ohmhip.h>

#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace gputil
{
/// gputil::ApiException (gputil/gpuApiException.h): thrown by the adaptor where the reference throws; nothing is
/// thrown across the C ABI itself.
class ApiException : public std::runtime_error
{
public:
  ApiException(int error_code, const char *what_call)
    : std::runtime_error(std::string(what_call) + ": " + ohmhip_error_string(error_code))
    , error_code_(error_code)
  {}
  int errorCode() const { return error_code_; }

private:
  int error_code_;
};

#define OHMHIP_GPUAPICHECK(call)                    \
  do                                                \
  {                                                 \
    const int gpu_api_status__ = (call);            \
    if (gpu_api_status__ != OHMHIP_OK)              \
    {                                               \
      throw gputil::ApiException(gpu_api_status__, #call); \
    }                                               \
  } while (0)

class Event
{
public:
  Event() = default;  // default constructed == invalid, as in the reference (gpuEvent.h:23-80)
  bool isValid() const { return bool(handle_); }
  bool isComplete() const
  {
    if (!handle_)
    {
      return true;
    }
    int complete = 1;
    OHMHIP_GPUAPICHECK(ohmhip_event_is_complete(handle_.get(), &complete));
    return complete != 0;
  }
  void wait() const
  {
    if (handle_)
    {
      OHMHIP_GPUAPICHECK(ohmhip_event_wait(handle_.get()));
    }
  }
  void release() { handle_.reset(); }
  /// Internal: create on demand.
  ohmhip_event_t ensure()
  {
    if (!handle_)
    {
      ohmhip_event_t e = nullptr;
      OHMHIP_GPUAPICHECK(ohmhip_event_create(&e));
      handle_ = std::shared_ptr<ohmhip_event_s>(e, [](ohmhip_event_t p) { ohmhip_event_destroy(p); });
    }
    return handle_.get();
  }
  ohmhip_event_t handle() const { return handle_.get(); }

private:
  std::shared_ptr<ohmhip_event_s> handle_;  // copyable, reference counted like gputil::Event
};

class Queue
{
public:
  Queue() = default;  // null queue == default stream
  static Queue create()
  {
    Queue q;
    ohmhip_stream_t s = nullptr;
    OHMHIP_GPUAPICHECK(ohmhip_stream_create(&s));
    q.handle_ = std::shared_ptr<ohmhip_stream_s>(s, [](ohmhip_stream_t p) { ohmhip_stream_destroy(p); });
    return q;
  }
  void finish() { OHMHIP_GPUAPICHECK(ohmhip_stream_finish(handle_.get())); }
  void flush() {}
  void insertBarrier() {}
  Event mark()
  {
    Event e;
    OHMHIP_GPUAPICHECK(ohmhip_event_record(e.ensure(), handle_.get()));
    return e;
  }
  ohmhip_stream_t handle() const { return handle_.get(); }

private:
  std::shared_ptr<ohmhip_stream_s> handle_;
};

class Device
{
public:
  explicit Device(int index = 0)
    : index_(index)
  {
    int count = 0;
    valid_ = ohmhip_device_count(&count) == OHMHIP_OK && index < count;
    if (valid_)
    {
      valid_ = ohmhip_device_select(index) == OHMHIP_OK && ohmhip_device_get_info(index, &info_) == OHMHIP_OK;
    }
  }
  static unsigned enumerateDevices()
  {
    int count = 0;
    return (ohmhip_device_count(&count) == OHMHIP_OK) ? unsigned(count) : 0u;
  }
  bool isValid() const { return valid_; }
  const char *name() const { return info_.name; }
  uint64_t deviceMemory() const { return info_.total_memory; }
  uint64_t maxAllocationSize() const { return info_.max_allocation; }
  bool unifiedMemory() const { return info_.unified_memory != 0; }
  Queue defaultQueue() const { return Queue(); }
  Queue createQueue() const { return Queue::create(); }

private:
  int index_ = 0;
  bool valid_ = false;
  ohmhip_device_info info_{};
};

enum BufferFlag : unsigned
{
  kBfRead = OHMHIP_BF_READ,
  kBfWrite = OHMHIP_BF_WRITE,
  kBfHostAccess = OHMHIP_BF_HOST_ACCESS,
  kBfReadWrite = kBfRead | kBfWrite
};

class Buffer
{
public:
  Buffer() = default;
  Buffer(const Device &, size_t byte_size, unsigned flags = kBfReadWrite) { create(byte_size, flags); }
  Buffer(const Buffer &) = delete;
  Buffer &operator=(const Buffer &) = delete;
  Buffer(Buffer &&other) noexcept { std::swap(handle_, other.handle_); }
  ~Buffer() { release(); }
  void create(size_t byte_size, unsigned flags = kBfReadWrite)
  {
    release();
    OHMHIP_GPUAPICHECK(ohmhip_buffer_create(&handle_, byte_size, flags));
  }
  void release()
  {
    if (handle_)
    {
      ohmhip_buffer_destroy(handle_);
      handle_ = nullptr;
    }
  }
  bool isValid() const { return handle_ != nullptr; }
  size_t size() const
  {
    size_t bytes = 0;
    if (handle_)
    {
      ohmhip_buffer_size(handle_, &bytes);
    }
    return bytes;
  }
  size_t actualSize() const { return size(); }
  /// Grow-only resize; returns the actual size (gputil/gpuBuffer.h:161).
  size_t resize(size_t new_size)
  {
    size_t actual = 0;
    OHMHIP_GPUAPICHECK(ohmhip_buffer_resize(handle_, new_size, &actual));
    return actual;
  }
  template <typename T>
  size_t elementsResize(size_t element_count)
  {
    return resize(sizeof(T) * element_count) / sizeof(T);
  }
  size_t write(const void *src, size_t byte_count, size_t dst_offset = 0, Queue *queue = nullptr,
               Event *block_on = nullptr, Event *completion = nullptr)
  {
    OHMHIP_GPUAPICHECK(ohmhip_buffer_write(handle_, src, byte_count, dst_offset, queue ? queue->handle() : nullptr,
                                           block_on ? block_on->handle() : nullptr,
                                           completion ? completion->ensure() : nullptr));
    return byte_count;
  }
  size_t read(void *dst, size_t byte_count, size_t src_offset = 0, Queue *queue = nullptr, Event *block_on = nullptr,
              Event *completion = nullptr)
  {
    OHMHIP_GPUAPICHECK(ohmhip_buffer_read(handle_, dst, byte_count, src_offset, queue ? queue->handle() : nullptr,
                                          block_on ? block_on->handle() : nullptr,
                                          completion ? completion->ensure() : nullptr));
    return byte_count;
  }
  void fill(int byte_value, size_t byte_count, size_t offset = 0, Queue *queue = nullptr)
  {
    OHMHIP_GPUAPICHECK(ohmhip_buffer_fill(handle_, byte_value, byte_count, offset, queue ? queue->handle() : nullptr));
  }
  void *argPtr() const
  {
    void *ptr = nullptr;
    if (handle_)
    {
      ohmhip_buffer_ptr(handle_, &ptr);
    }
    return ptr;
  }
  ohmhip_buffer_t handle() const { return handle_; }

private:
  ohmhip_buffer_t handle_ = nullptr;
};

enum PinMode
{
  kPinNone = 0,
  kPinRead,
  kPinWrite,
  kPinReadWrite
};

/// gputil::PinnedBuffer as implemented by the reference's CUDA backend: a pinned host staging allocation flushed with
/// an async copy on unpin() (gputil/cuda/gpuPinnedBuffer.cpp:66-131).
class PinnedBuffer
{
public:
  PinnedBuffer(Buffer &buffer, PinMode mode)
    : buffer_(&buffer)
    , mode_(mode)
  {
    pin();
  }
  ~PinnedBuffer()
  {
    try
    {
      unpin();
    }
    catch (...)
    {
    }
    if (staging_)
    {
      ohmhip_host_free(staging_);
    }
  }
  void pin()
  {
    if (!staging_ || staging_size_ < buffer_->size())
    {
      if (staging_)
      {
        ohmhip_host_free(staging_);
      }
      staging_size_ = buffer_->size();
      OHMHIP_GPUAPICHECK(ohmhip_host_alloc(&staging_, staging_size_));
    }
    dirty_begin_ = staging_size_;
    dirty_end_ = 0;
    pinned_ = true;
    if (mode_ == kPinRead || mode_ == kPinReadWrite)
    {
      buffer_->read(staging_, staging_size_);
    }
  }
  size_t write(const void *src, size_t byte_count, size_t dst_offset = 0)
  {
    std::memcpy(static_cast<char *>(staging_) + dst_offset, src, byte_count);
    dirty_begin_ = dst_offset < dirty_begin_ ? dst_offset : dirty_begin_;
    dirty_end_ = dst_offset + byte_count > dirty_end_ ? dst_offset + byte_count : dirty_end_;
    return byte_count;
  }
  size_t read(void *dst, size_t byte_count, size_t src_offset = 0) const
  {
    std::memcpy(dst, static_cast<const char *>(staging_) + src_offset, byte_count);
    return byte_count;
  }
  void unpin(Queue *queue = nullptr, Event *block_on = nullptr, Event *completion = nullptr)
  {
    if (!pinned_)
    {
      return;
    }
    pinned_ = false;
    if ((mode_ == kPinWrite || mode_ == kPinReadWrite) && dirty_end_ > dirty_begin_)
    {
      buffer_->write(static_cast<char *>(staging_) + dirty_begin_, dirty_end_ - dirty_begin_, dirty_begin_, queue,
                     block_on, completion);
    }
  }

private:
  Buffer *buffer_;
  PinMode mode_;
  void *staging_ = nullptr;
  size_t staging_size_ = 0;
  size_t dirty_begin_ = 0;
  size_t dirty_end_ = 0;
  bool pinned_ = false;
};
}  // namespace gputil

#endif  // OHMHIP_GPUTIL_HIP_H
