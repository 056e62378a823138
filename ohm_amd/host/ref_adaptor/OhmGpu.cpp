// OhmGpu.cpp -- device selection functions declared in the reference's ohmgpu/OhmGpu.h, for the HIP backend.  Replaces
// ohmgpu/OhmGpu.cpp:27-298 (OpenCL platform / CUDA device matching): one process uses one HIP device.
#include <ohmgpu/OhmGpu.h>

#include <gputil/gpuDevice.h>
#include <gputil/gpuProgram.h>

#include <ohmhip.h>

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <mutex>
#include <string>

namespace ohm
{
namespace
{
std::mutex g_device_mutex;
std::unique_ptr<gputil::Device> g_device;

int selectDevice(const char *device_hint, bool show_device)
{
  std::lock_guard<std::mutex> guard(g_device_mutex);
  int count = 0;
  if (ohmhip_device_count(&count) != OHMHIP_OK || count <= 0)
  {
    return 1;
  }
  int chosen = 0;
  if (device_hint && *device_hint)
  {
    // "--device" is a partial, lower-case name match in the reference; an index is accepted as well.
    char *end = nullptr;
    const long index = std::strtol(device_hint, &end, 10);
    if (end && *end == '\0' && index >= 0 && index < count)
    {
      chosen = int(index);
    }
    else
    {
      for (int i = 0; i < count; ++i)
      {
        ohmhip_device_info info;
        if (ohmhip_device_get_info(i, &info) == OHMHIP_OK)
        {
          std::string name(info.name);
          for (char &c : name)
          {
            c = char(std::tolower(static_cast<unsigned char>(c)));
          }
          if (name.find(device_hint) != std::string::npos)
          {
            chosen = i;
            break;
          }
        }
      }
    }
  }
  if (ohmhip_device_select(chosen) != OHMHIP_OK)
  {
    return 1;
  }
  g_device.reset(new gputil::Device(true));
  if (show_device && g_device->isValid())
  {
    std::cout << g_device->description() << std::endl;
  }
  return g_device->isValid() ? 0 : 1;
}
}  // namespace

int configureGpuFromArgs(int argc, const char **argv, bool show_device)
{
  const char *device = nullptr;
  for (int i = 1; i < argc; ++i)
  {
    if (std::strncmp(argv[i], "--device=", 9) == 0)
    {
      device = argv[i] + 9;
    }
    else if (std::strcmp(argv[i], "--device") == 0 && i + 1 < argc)
    {
      device = argv[++i];
    }
    // --accel, --platform, --clver and --gpu-debug select among OpenCL platforms in the reference: no meaning here.
  }
  return selectDevice(device, show_device);
}

int configureGpu(unsigned accel, const char *device_name, bool show_device)
{
  if (!(accel & kGpuAccel))
  {
    return 1;  // this backend has no CPU accelerator
  }
  return selectDevice(device_name, show_device);
}

gputil::Device &gpuDevice()
{
  {
    std::lock_guard<std::mutex> guard(g_device_mutex);
    if (g_device)
    {
      return *g_device;
    }
  }
  selectDevice(nullptr, true);
  std::lock_guard<std::mutex> guard(g_device_mutex);
  if (!g_device)
  {
    g_device.reset(new gputil::Device(false));  // invalid device: GpuMap::gpuOk() will be false
  }
  return *g_device;
}

unsigned gpuArgsInfo(const char **args_info, int *arg_type, unsigned max_pairs)
{
  static const char *const kPairs[] = { "device", "HIP device: index, or part of its name (lower case)." };
  const unsigned pair_count = 1;
  if (args_info)
  {
    for (unsigned i = 0; i < pair_count && i < max_pairs; ++i)
    {
      args_info[2 * i] = kPairs[2 * i];
      args_info[2 * i + 1] = kPairs[2 * i + 1];
      if (arg_type)
      {
        arg_type[i] = 1;  // string
      }
    }
  }
  return pair_count;
}

const char *gpuBuildStdArg()
{
  return "";  // no run-time kernel compilation: the kernels are in libohmhip.so
}

void setGpuBuildVersion(gputil::BuildArgs &build_args)
{
  (void)build_args;
}
}  // namespace ohm
