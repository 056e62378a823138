// gpuEvent.cpp -- gputil::Event (declared in the reference's gputil/gpuEvent.h) for HIP.  Replaces
// gputil/cuda/gpuEvent.cpp.  Copies of an Event share one reference-counted hipEvent.
#include <gputil/gpuEvent.h>

#include "gputilHipDetail.h"

#include <cstring>

namespace gputil
{
namespace
{
void releaseDetail(EventDetail *detail)
{
  if (detail && detail->references.fetch_sub(1) == 1)
  {
    if (detail->event)
    {
      ohmhip_event_destroy(detail->event);
    }
    delete detail;
  }
}

EventDetail *retainDetail(EventDetail *detail)
{
  if (detail)
  {
    detail->references.fetch_add(1);
  }
  return detail;
}
}  // namespace

/// Used by Queue::mark(): hand a freshly created detail (reference count 1) to an invalid Event.  Event keeps its
/// pointer private and the reference's header offers no setter; Event is a single pointer (gputil/gpuEvent.h:74-80),
/// which the static_assert pins.
void adoptEventDetail(Event &event, EventDetail *detail)
{
  static_assert(sizeof(Event) == sizeof(EventDetail *), "gputil::Event is expected to hold exactly one pointer");
  event.release();
  std::memcpy(static_cast<void *>(&event), &detail, sizeof(detail));
}

Event::Event() = default;

Event::Event(const Event &other)
  : imp_(retainDetail(other.imp_))
{}

Event::Event(Event &&other) noexcept
  : imp_(other.imp_)
{
  other.imp_ = nullptr;
}

Event::~Event()
{
  release();
}

bool Event::isValid() const
{
  return imp_ && imp_->event;
}

void Event::release()
{
  releaseDetail(imp_);
  imp_ = nullptr;
}

bool Event::isComplete() const
{
  if (!isValid())
  {
    return true;
  }
  int complete = 1;
  ohmhip_event_is_complete(imp_->event, &complete);
  return complete != 0;
}

void Event::wait() const
{
  if (isValid())
  {
    ohmhip_event_wait(imp_->event);
  }
}

void Event::wait(const Event *events, size_t event_count)
{
  for (size_t i = 0; i < event_count; ++i)
  {
    events[i].wait();
  }
}

void Event::wait(const Event **events, size_t event_count)
{
  for (size_t i = 0; i < event_count; ++i)
  {
    if (events[i])
    {
      events[i]->wait();
    }
  }
}

Event &Event::operator=(const Event &other)
{
  if (this != &other)
  {
    EventDetail *retained = retainDetail(other.imp_);
    release();
    imp_ = retained;
  }
  return *this;
}

Event &Event::operator=(Event &&other) noexcept
{
  if (this != &other)
  {
    release();
    imp_ = other.imp_;
    other.imp_ = nullptr;
  }
  return *this;
}

EventDetail *Event::detail()
{
  return imp_;
}

EventDetail *Event::detail() const
{
  return imp_;
}
}  // namespace gputil
