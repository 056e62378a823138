// gpuQueue.cpp -- gputil::Queue (declared in the reference's gputil/gpuQueue.h) for HIP.  Replaces
// gputil/cuda/gpuQueue.cpp.
#include <gputil/gpuQueue.h>

#include "gputilHipDetail.h"

namespace gputil
{
Queue::Queue()
  : queue_(nullptr)
{}

Queue::Queue(const Queue &other) = default;

Queue::Queue(Queue &&other) noexcept
  : queue_(std::move(other.queue_))
{}

Queue::Queue(void *platform_queue)
  : queue_(new QueueDetail)
{
  queue_->stream = static_cast<ohmhip_stream_t>(platform_queue);  // null: the default stream
}

Queue::~Queue() = default;

bool Queue::isValid() const
{
  return queue_ != nullptr;
}

void Queue::insertBarrier()
{
  // In-order streams: every operation is a barrier for the next one.
}

Event Queue::mark()
{
  Event event;
  if (!queue_)
  {
    return event;
  }
  EventDetail *detail = new EventDetail;
  if (ohmhip_event_create(&detail->event) != OHMHIP_OK || ohmhip_event_record(detail->event, queue_->stream) != OHMHIP_OK)
  {
    if (detail->event)
    {
      ohmhip_event_destroy(detail->event);
    }
    delete detail;
    return event;
  }
  adoptEventDetail(event, detail);  // (reference count 1)
  if (queue_->synchronous)
  {
    event.wait();
  }
  return event;
}

void Queue::setSynchronous(bool synchronous)
{
  if (queue_)
  {
    queue_->synchronous = synchronous;
  }
}

bool Queue::synchronous() const
{
  return queue_ && queue_->synchronous;
}

void Queue::flush()
{
  // HIP submits eagerly.
}

void Queue::finish()
{
  if (queue_)
  {
    ohmhip_stream_finish(queue_->stream);
  }
}

void Queue::queueCallback(const std::function<void(void)> &callback)
{
  // The C ABI has no stream callbacks: run it once the work queued so far has finished.
  finish();
  if (callback)
  {
    callback();
  }
}

QueueDetail *Queue::internal() const
{
  return queue_.get();
}

Queue &Queue::operator=(const Queue &other) = default;

Queue &Queue::operator=(Queue &&other) noexcept
{
  queue_ = std::move(other.queue_);
  return *this;
}
}  // namespace gputil
