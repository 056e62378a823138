// gpuDevice.cpp -- gputil::Device (declared in the reference's gputil/gpuDevice.h) for HIP, over the device group of
// include/ohmhip.h.  Replaces gputil/cuda/gpuDevice.cpp.
#include <gputil/gpuDevice.h>

#include "gputilHipDetail.h"

#include <cstdlib>
#include <cstring>
#include <sstream>

namespace gputil
{
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
