// gputilHipBuffer.cpp -- the HIP backend of gputil::Buffer, gputil::copyBuffer and gputil::PinnedBuffer, declared in
// the reference's gputil/gpuBuffer.h:73-510 and gputil/gpuPinnedBuffer.h:28-205, defined here over the buffer group of
// include/ohmhip.h.  Stands where gputil/cuda/gpuBuffer.cpp and gpuPinnedBuffer.cpp stand in a CUDA build.
// (gputil::EventList is backend independent in the reference -- gputil/gpuEventList.cpp -- and is compiled from there.)
//
// Semantics kept from the class documentation (gpuBuffer.h:56-71):
//   queue == nullptr            -> blocking call (after block_on, if given)
//   queue != nullptr            -> asynchronous on that queue's stream, after block_on, `completion` set to an event
//                                  recorded behind the transfer
// Design of this backend (not the CUDA one's):
//   * a kBfHostAccess buffer IS pinned host memory mapped into the device's address space (hipHostMalloc), so
//     PinnedBuffer pins by handing out that very address -- zero copies, nothing to flush on unpin but ordering;
//   * a plain device buffer is never "pinned": PinnedBuffer falls back to Buffer::read / write, as the class
//     documentation allows (gpuPinnedBuffer.h:24: "Falls back to unpinned memory transfers when required");
//   * allocations are padded to 256 bytes (actualSize() >= size()); resize() only ever grows the allocation.
#include <gputil/gpuBuffer.h>
#include <gputil/gpuDevice.h>
#include <gputil/gpuEvent.h>
#include <gputil/gpuPinnedBuffer.h>
#include <gputil/gpuQueue.h>

#include "gputilHipDetail.h"

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

namespace gputil
{
namespace
{
size_t paddedSize(size_t bytes)
{
  return (bytes + 255u) & ~size_t(255u);
}

ohmhip_stream_t streamOf(Queue *queue)
{
  return (queue && queue->isValid()) ? queue->internal()->stream : nullptr;
}

ohmhip_event_t eventOf(Event *event)
{
  return (event && event->isValid()) ? event->detail()->event : nullptr;
}

/// A fresh event for `completion` (the caller's object may hold an older event: it is released first).
ohmhip_event_t prepareCompletion(Event *completion)
{
  if (!completion)
  {
    return nullptr;
  }
  EventDetail *detail = new EventDetail;
  if (ohmhip_event_create(&detail->event) != OHMHIP_OK)
  {
    delete detail;
    completion->release();
    return nullptr;
  }
  adoptEventDetail(*completion, detail);
  return detail->event;
}

/// The reference's queues may be flagged synchronous (Queue::setSynchronous): then "asynchronous" calls finish first.
void settle(Queue *queue)
{
  if (queue && queue->isValid() && queue->synchronous())
  {
    queue->finish();
  }
}

bool allocate(BufferDetail &d, size_t byte_size, unsigned flags)
{
  d.buffer = nullptr;
  d.requested = 0;
  d.flags = flags;
  unsigned abi_flags = 0;
  abi_flags |= (flags & kBfRead) ? OHMHIP_BF_READ : 0u;
  abi_flags |= (flags & kBfWrite) ? OHMHIP_BF_WRITE : 0u;
  abi_flags |= (flags & kBfHostAccess) ? OHMHIP_BF_HOST_ACCESS : 0u;
  if (ohmhip_buffer_create(&d.buffer, paddedSize(std::max<size_t>(byte_size, 1)), abi_flags) != OHMHIP_OK)
  {
    d.buffer = nullptr;
    return false;
  }
  d.requested = byte_size;
  return true;
}

void freeDetail(BufferDetail *d)
{
  if (d)
  {
    if (d->buffer)
    {
      ohmhip_buffer_destroy(d->buffer);
    }
    delete d;
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Buffer
// ---------------------------------------------------------------------------------------------------------------------
Buffer::Buffer()
  : imp_(new BufferDetail)
{}

Buffer::Buffer(const Device &device, size_t byte_size, unsigned flags)
  : imp_(new BufferDetail)
{
  create(device, byte_size, flags);
}

Buffer::Buffer(Buffer &&other) noexcept
  : imp_(other.imp_)
{
  other.imp_ = nullptr;
}

Buffer::~Buffer()
{
  freeDetail(imp_);
  imp_ = nullptr;
}

Buffer &Buffer::operator=(Buffer &&other) noexcept
{
  if (this != &other)
  {
    freeDetail(imp_);
    imp_ = other.imp_;
    other.imp_ = nullptr;
  }
  return *this;
}

void Buffer::create(const Device &device, size_t byte_size, unsigned flags)
{
  if (!imp_)
  {
    imp_ = new BufferDetail;
  }
  release();
  if (device.isValid())
  {
    imp_->device = device.detail()->device;
    allocate(*imp_, byte_size, flags);
  }
}

void Buffer::release()
{
  if (imp_ && imp_->buffer)
  {
    ohmhip_buffer_destroy(imp_->buffer);
    imp_->buffer = nullptr;
  }
  if (imp_)
  {
    imp_->requested = 0;
    imp_->device = -1;
  }
}

void Buffer::swap(Buffer &other) noexcept
{
  std::swap(imp_, other.imp_);
}

bool Buffer::isValid() const
{
  return imp_ && imp_->buffer;
}

unsigned Buffer::flags() const
{
  return imp_ ? imp_->flags : 0u;
}

size_t Buffer::size() const
{
  return imp_ ? imp_->requested : 0u;
}

size_t Buffer::actualSize() const
{
  size_t bytes = 0;
  if (isValid())
  {
    ohmhip_buffer_size(imp_->buffer, &bytes);
  }
  return bytes;
}

size_t Buffer::resize(size_t new_size)
{
  if (!isValid())
  {
    return 0;
  }
  if (new_size > actualSize())
  {
    size_t actual = 0;
    if (ohmhip_buffer_resize(imp_->buffer, paddedSize(new_size), &actual) != OHMHIP_OK)
    {
      return actualSize();
    }
  }
  imp_->requested = new_size;
  return actualSize();
}

size_t Buffer::forceResize(size_t new_size)
{
  if (!isValid())
  {
    return 0;
  }
  if (paddedSize(std::max<size_t>(new_size, 1)) != actualSize())
  {
    // the C ABI's resize only grows: a smaller best size needs a new allocation
    BufferDetail fresh;
    fresh.device = imp_->device;
    if (allocate(fresh, new_size, imp_->flags))
    {
      ohmhip_buffer_destroy(imp_->buffer);
      imp_->buffer = fresh.buffer;
    }
  }
  imp_->requested = new_size;
  return actualSize();
}

void Buffer::fill(const void *pattern, size_t pattern_size, Queue *queue, Event *block_on, Event *completion)
{
  if (isValid())
  {
    ohmhip_buffer_fill_pattern(imp_->buffer, pattern, pattern_size, actualSize(), 0, streamOf(queue), eventOf(block_on),
                               queue ? prepareCompletion(completion) : nullptr);
    settle(queue);
  }
}

void Buffer::fillPartial(const void *pattern, size_t pattern_size, size_t fill_bytes, size_t offset, Queue *queue)
{
  if (isValid() && offset < size())
  {
    fill_bytes = std::min(fill_bytes, size() - offset);
    ohmhip_buffer_fill_pattern(imp_->buffer, pattern, pattern_size, fill_bytes, offset, streamOf(queue), nullptr, nullptr);
    settle(queue);
  }
}

size_t Buffer::read(void *dst, size_t read_byte_count, size_t src_offset, Queue *queue, Event *block_on,
                    Event *completion)
{
  if (!isValid() || src_offset >= size())
  {
    return 0;
  }
  const size_t bytes = std::min(read_byte_count, size() - src_offset);
  const int err = ohmhip_buffer_read(imp_->buffer, dst, bytes, src_offset, streamOf(queue), eventOf(block_on),
                                     queue ? prepareCompletion(completion) : nullptr);
  settle(queue);
  return err == OHMHIP_OK ? bytes : 0u;
}

size_t Buffer::write(const void *src, size_t byte_count, size_t dst_offset, Queue *queue, Event *block_on,
                     Event *completion)
{
  if (!isValid() || dst_offset >= size())
  {
    return 0;
  }
  const size_t bytes = std::min(byte_count, size() - dst_offset);
  const int err = ohmhip_buffer_write(imp_->buffer, src, bytes, dst_offset, streamOf(queue), eventOf(block_on),
                                      queue ? prepareCompletion(completion) : nullptr);
  settle(queue);
  return err == OHMHIP_OK ? bytes : 0u;
}

size_t Buffer::readElements(void *dst, size_t element_size, size_t element_count, size_t offset_elements,
                            size_t buffer_element_size, Queue *queue, Event *block_on, Event *completion)
{
  if (buffer_element_size == 0 || buffer_element_size == element_size)
  {
    return read(dst, element_size * element_count, offset_elements * element_size, queue, block_on, completion) /
           std::max<size_t>(element_size, 1);
  }
  // Strides differ (a 3-float host vector against a 4-float device vector, gpuBuffer.h:291-300): the device span is
  // fetched in one blocking transfer and re-strided on the host.
  if (!isValid() || offset_elements * buffer_element_size >= size())
  {
    return 0;
  }
  const size_t first = offset_elements * buffer_element_size;
  const size_t count = std::min(element_count, (size() - first) / buffer_element_size);
  std::vector<uint8_t> staged(count * buffer_element_size);
  if (ohmhip_buffer_read(imp_->buffer, staged.data(), staged.size(), first, nullptr, eventOf(block_on), nullptr) != OHMHIP_OK)
  {
    return 0;
  }
  const size_t copy = std::min(element_size, buffer_element_size);
  for (size_t i = 0; i < count; ++i)
  {
    std::memcpy(static_cast<uint8_t *>(dst) + i * element_size, staged.data() + i * buffer_element_size, copy);
  }
  if (queue && completion)
  {
    *completion = queue->mark();
  }
  return count;
}

size_t Buffer::writeElements(const void *src, size_t element_size, size_t element_count, size_t offset_elements,
                             size_t buffer_element_size, Queue *queue, Event *block_on, Event *completion)
{
  if (buffer_element_size == 0 || buffer_element_size == element_size)
  {
    return write(src, element_size * element_count, offset_elements * element_size, queue, block_on, completion) /
           std::max<size_t>(element_size, 1);
  }
  if (!isValid() || offset_elements * buffer_element_size >= size())
  {
    return 0;
  }
  const size_t first = offset_elements * buffer_element_size;
  const size_t count = std::min(element_count, (size() - first) / buffer_element_size);
  // Re-stride on the host (the padding bytes of every device element are written as zero), one blocking transfer.
  std::vector<uint8_t> staged(count * buffer_element_size, 0);
  const size_t copy = std::min(element_size, buffer_element_size);
  for (size_t i = 0; i < count; ++i)
  {
    std::memcpy(staged.data() + i * buffer_element_size, static_cast<const uint8_t *>(src) + i * element_size, copy);
  }
  if (ohmhip_buffer_write(imp_->buffer, staged.data(), staged.size(), first, nullptr, eventOf(block_on), nullptr) != OHMHIP_OK)
  {
    return 0;
  }
  if (queue && completion)
  {
    *completion = queue->mark();
  }
  return count;
}

void *Buffer::argPtr() const
{
  return address();
}

void *Buffer::address() const
{
  void *ptr = nullptr;
  if (isValid())
  {
    ohmhip_buffer_ptr(imp_->buffer, &ptr);
  }
  return ptr;
}

size_t copyBuffer(Buffer &dst, const Buffer &src, Queue *queue, Event *block_on, Event *completion)
{
  return copyBuffer(dst, 0, src, 0, src.size(), queue, block_on, completion);
}

size_t copyBuffer(Buffer &dst, const Buffer &src, size_t byte_count, Queue *queue, Event *block_on, Event *completion)
{
  return copyBuffer(dst, 0, src, 0, byte_count, queue, block_on, completion);
}

size_t copyBuffer(Buffer &dst, size_t dst_offset, const Buffer &src, size_t src_offset, size_t byte_count, Queue *queue,
                  Event *block_on, Event *completion)
{
  if (!dst.isValid() || !src.isValid() || dst_offset >= dst.size() || src_offset >= src.size())
  {
    return 0;
  }
  byte_count = std::min(byte_count, std::min(dst.size() - dst_offset, src.size() - src_offset));
  const int err = ohmhip_buffer_copy(dst.detail()->buffer, dst_offset, src.detail()->buffer, src_offset, byte_count,
                                     streamOf(queue), eventOf(block_on), queue ? prepareCompletion(completion) : nullptr);
  settle(queue);
  return err == OHMHIP_OK ? byte_count : 0u;
}

// ---------------------------------------------------------------------------------------------------------------------
// PinnedBuffer
// ---------------------------------------------------------------------------------------------------------------------
PinnedBuffer::PinnedBuffer() = default;

PinnedBuffer::PinnedBuffer(Buffer &buffer, PinMode mode)
  : buffer_(&buffer)
  , mode_(mode)
{
  pin();
}

PinnedBuffer::PinnedBuffer(PinnedBuffer &&other) noexcept
  : buffer_(other.buffer_)
  , pinned_(other.pinned_)
  , mode_(other.mode_)
{
  other.buffer_ = nullptr;
  other.pinned_ = nullptr;
  other.mode_ = kPinNone;
}

PinnedBuffer::~PinnedBuffer()
{
  unpin();
}

bool PinnedBuffer::isPinned() const
{
  return pinned_ != nullptr;
}

void PinnedBuffer::pin()
{
  if (buffer_ && !pinned_ && buffer_->isValid() && (buffer_->flags() & kBfHostAccess))
  {
    // The allocation is host memory the device reads and writes in place.  What the host is about to read (or
    // overwrite) may still be in flight on the device: pinning is the fence.
    ohmhip_device_synchronize();
    pinned_ = buffer_->address();
  }
}

void PinnedBuffer::unpin(Queue *queue, Event *block_on, Event *completion)
{
  if (buffer_ && pinned_)
  {
    // Nothing to transfer: host writes landed in the allocation itself.  The event contract still holds: `completion`
    // is recorded on the queue, behind block_on, so work enqueued after it sees the writes.
    if (queue && queue->isValid())
    {
      if (ohmhip_event_t wait_for = eventOf(block_on))
      {
        ohmhip_stream_wait_event(queue->internal()->stream, wait_for);
      }
      if (completion)
      {
        *completion = queue->mark();
      }
    }
    else if (block_on)
    {
      block_on->wait();
    }
    pinned_ = nullptr;
  }
}

size_t PinnedBuffer::read(void *dst, size_t byte_count, size_t src_offset) const
{
  if (!buffer_)
  {
    return 0;
  }
  if (!pinned_)
  {
    return buffer_->read(dst, byte_count, src_offset);
  }
  if (src_offset >= buffer_->size())
  {
    return 0;
  }
  byte_count = std::min(byte_count, buffer_->size() - src_offset);
  std::memcpy(dst, static_cast<const uint8_t *>(pinned_) + src_offset, byte_count);
  return byte_count;
}

size_t PinnedBuffer::write(const void *src, size_t byte_count, size_t dst_offset)
{
  if (!buffer_)
  {
    return 0;
  }
  if (!pinned_)
  {
    return buffer_->write(src, byte_count, dst_offset);
  }
  if (dst_offset >= buffer_->size())
  {
    return 0;
  }
  byte_count = std::min(byte_count, buffer_->size() - dst_offset);
  std::memcpy(static_cast<uint8_t *>(pinned_) + dst_offset, src, byte_count);
  return byte_count;
}

size_t PinnedBuffer::readElements(void *dst, size_t element_size, size_t element_count, size_t offset_elements,
                                  size_t buffer_element_size)
{
  if (!buffer_)
  {
    return 0;
  }
  if (!pinned_)
  {
    return buffer_->readElements(dst, element_size, element_count, offset_elements, buffer_element_size);
  }
  const size_t stride = buffer_element_size ? buffer_element_size : element_size;
  const size_t first = offset_elements * stride;
  if (stride == 0 || first >= buffer_->size())
  {
    return 0;
  }
  const size_t count = std::min(element_count, (buffer_->size() - first) / stride);
  const size_t copy = std::min(element_size, stride);
  const uint8_t *from = static_cast<const uint8_t *>(pinned_) + first;
  for (size_t i = 0; i < count; ++i)
  {
    std::memcpy(static_cast<uint8_t *>(dst) + i * element_size, from + i * stride, copy);
  }
  return count;
}

size_t PinnedBuffer::writeElements(const void *src, size_t element_size, size_t element_count, size_t offset_elements,
                                   size_t buffer_element_size)
{
  if (!buffer_)
  {
    return 0;
  }
  if (!pinned_)
  {
    return buffer_->writeElements(src, element_size, element_count, offset_elements, buffer_element_size);
  }
  const size_t stride = buffer_element_size ? buffer_element_size : element_size;
  const size_t first = offset_elements * stride;
  if (stride == 0 || first >= buffer_->size())
  {
    return 0;
  }
  const size_t count = std::min(element_count, (buffer_->size() - first) / stride);
  const size_t copy = std::min(element_size, stride);
  uint8_t *to = static_cast<uint8_t *>(pinned_) + first;
  for (size_t i = 0; i < count; ++i)
  {
    std::memcpy(to + i * stride, static_cast<const uint8_t *>(src) + i * element_size, copy);
  }
  return count;
}

PinnedBuffer &PinnedBuffer::operator=(PinnedBuffer &&other) noexcept
{
  if (this != &other)
  {
    unpin();
    buffer_ = other.buffer_;
    pinned_ = other.pinned_;
    mode_ = other.mode_;
    other.buffer_ = nullptr;
    other.pinned_ = nullptr;
    other.mode_ = kPinNone;
  }
  return *this;
}
}  // namespace gputil
