// gputilHip.cpp -- the HIP backend of the part of gputil the ray-integration path exposes: gputil::Event, Queue and
// Device, declared in the reference's gputil/gpuEvent.h, gpuQueue.h and gpuDevice.h, defined here over the device group
// of include/ohmhip.h (one translation unit: the three classes share EventDetail / the stream handles of
// gputilHipDetail.h).  Stands where gputil/cuda/gpuEvent.cpp, gpuQueue.cpp and gpuDevice.cpp stand in a CUDA build.
#include <gputil/gpuDevice.h>
#include <gputil/gpuEvent.h>
#include <gputil/gpuQueue.h>

#include "gputilHipDetail.h"

#include <cstdlib>
#include <cstring>
#include <sstream>

namespace gputil
{
// ---------------------------------------------------------------------------------------------------------------------
// Event: copies share one reference-counted hipEvent.
// ---------------------------------------------------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------------------------------------------------
// Queue
// ---------------------------------------------------------------------------------------------------------------------
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

// ---------------------------------------------------------------------------------------------------------------------
// Device
// ---------------------------------------------------------------------------------------------------------------------
namespace
{
bool fill(DeviceDetail &d, int device)
{
  ohmhip_device_info info;
  if (device < 0 || ohmhip_device_get_info(device, &info) != OHMHIP_OK)
  {
    d = DeviceDetail{};
    return false;
  }
  d.device = device;
  d.name = info.name;
  d.info.name = info.name;
  d.info.platform = "HIP";
  d.info.type = kDeviceGpu;
  d.info.version = Version{};
  std::ostringstream text;
  text << info.name << " (" << info.arch << ", " << info.compute_units << " CUs, " << (info.total_memory >> 30)
       << " GiB)";
  d.description = text.str();
  d.device_memory = info.total_memory;
  d.max_allocation = info.max_allocation;
  d.unified_memory = info.unified_memory != 0;
  return true;
}

int deviceFromArgs(int argc, const char **argv, const char *default_device)
{
  const char *hint = default_device;
  for (int i = 1; i < argc; ++i)
  {
    if (std::strncmp(argv[i], "--device=", 9) == 0)
    {
      hint = argv[i] + 9;
    }
  }
  int count = 0;
  if (ohmhip_device_count(&count) != OHMHIP_OK || count <= 0)
  {
    return -1;
  }
  if (hint && *hint)
  {
    char *end = nullptr;
    const long index = std::strtol(hint, &end, 10);
    if (end && *end == '\0' && index >= 0 && index < count)
    {
      return int(index);
    }
  }
  return 0;
}
}  // namespace

Device::Device(bool default_device)
  : imp_(new DeviceDetail)
{
  if (default_device)
  {
    int count = 0;
    if (ohmhip_device_count(&count) == OHMHIP_OK && count > 0)
    {
      fill(*imp_, 0);
    }
  }
}

Device::Device(const DeviceInfo &device_info)
  : imp_(new DeviceDetail)
{
  select(device_info);
}

Device::Device(int argc, const char **argv, const char *default_device, unsigned device_type_flags)
  : imp_(new DeviceDetail)
{
  select(argc, argv, default_device, device_type_flags);
}

Device::Device(const Device &other)
  : imp_(new DeviceDetail(*other.imp_))
{}

Device::Device(Device &&other) noexcept
  : imp_(std::move(other.imp_))
{}

Device::~Device() = default;

unsigned Device::enumerateDevices(std::vector<DeviceInfo> &devices)
{
  int count = 0;
  if (ohmhip_device_count(&count) != OHMHIP_OK)
  {
    return 0;
  }
  unsigned added = 0;
  for (int i = 0; i < count; ++i)
  {
    DeviceDetail d;
    if (fill(d, i))
    {
      devices.push_back(d.info);
      ++added;
    }
  }
  return added;
}

const char *Device::name() const
{
  return imp_->name.c_str();
}

const char *Device::description() const
{
  return imp_->description.c_str();
}

const DeviceInfo &Device::info() const
{
  return imp_->info;
}

Queue Device::defaultQueue() const
{
  return Queue(nullptr);
}

Queue Device::createQueue(unsigned flags) const
{
  (void)flags;
  ohmhip_stream_t stream = nullptr;
  if (!isValid() || ohmhip_stream_create(&stream) != OHMHIP_OK)
  {
    return Queue();
  }
  Queue queue(stream);
  queue.internal()->owned = true;
  return queue;
}

bool Device::select(int argc, const char **argv, const char *default_device, unsigned device_type_flags)
{
  if (!(device_type_flags & kGpu))
  {
    *imp_ = DeviceDetail{};
    return false;
  }
  const int device = deviceFromArgs(argc, argv, default_device);
  return device >= 0 && ohmhip_device_select(device) == OHMHIP_OK && fill(*imp_, device);
}

bool Device::select(const DeviceInfo &device_info)
{
  int count = 0;
  if (ohmhip_device_count(&count) != OHMHIP_OK)
  {
    return false;
  }
  for (int i = 0; i < count; ++i)
  {
    DeviceDetail d;
    if (fill(d, i) && d.info == device_info)
    {
      *imp_ = d;
      return ohmhip_device_select(i) == OHMHIP_OK;
    }
  }
  return false;
}

void Device::setDebugGpu(DebugLevel debug_level)
{
  imp_->debug_level = int(debug_level);
}

Device::DebugLevel Device::debugGpu() const
{
  return DebugLevel(imp_->debug_level);
}

bool Device::supportsFeature(const char *feature_id) const
{
  (void)feature_id;  // OpenCL extension strings in the reference
  return false;
}

void Device::addSearchPath(const char *path)
{
  // (kernel source search path in the OpenCL backend; kept for the accessor)
  if (!imp_->search_paths.empty())
  {
    imp_->search_paths += ",";
  }
  imp_->search_paths += path ? path : "";
}

const char *Device::searchPaths() const
{
  return imp_->search_paths.c_str();
}

bool Device::isValid() const
{
  return imp_ && imp_->device >= 0;
}

uint64_t Device::deviceMemory() const
{
  return imp_->device_memory;
}

uint64_t Device::maxAllocationSize() const
{
  return imp_->max_allocation;
}

bool Device::unifiedMemory() const
{
  return imp_->unified_memory;
}

Device &Device::operator=(const Device &other)
{
  if (this != &other)
  {
    imp_.reset(new DeviceDetail(*other.imp_));
  }
  return *this;
}

Device &Device::operator=(Device &&other) noexcept
{
  imp_ = std::move(other.imp_);
  return *this;
}
}  // namespace gputil
