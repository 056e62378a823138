// gputil_hip_check.cpp -- exercises the HIP backend of gputil::Device / Queue / Event (this directory) through the
// reference's own class declarations.  Built by __graft_entry__.build() when the reference checkout is present (the
// headers are compiled against where they lie), run on the GPU by tests/test_gpu_cpp_host.py.
#include <gputil/gpuDevice.h>
#include <gputil/gpuEvent.h>
#include <gputil/gpuQueue.h>

#include <cstdio>
#include <vector>

#define CHECK(cond)                                                   \
  if (!(cond))                                                        \
  {                                                                   \
    std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
    return 1;                                                         \
  }

int main()
{
  std::vector<gputil::DeviceInfo> devices;
  const unsigned count = gputil::Device::enumerateDevices(devices);
  CHECK(count >= 1 && devices.size() == count);
  gputil::Device invalid;
  CHECK(!invalid.isValid());
  gputil::Device gpu(true);
  CHECK(gpu.isValid());
  CHECK(gpu.deviceMemory() > (uint64_t(1) << 30) && gpu.maxAllocationSize() > 0);
  CHECK(gpu.info() == devices[0] && gpu.info().type == gputil::kDeviceGpu);
  gputil::Device copy = gpu;
  CHECK(copy.isValid() && copy.info() == gpu.info());
  gputil::Device selected(devices[0]);
  CHECK(selected.isValid());
  const char *args[] = { "prog", "--device=0" };
  gputil::Device from_args(2, args);
  CHECK(from_args.isValid());

  gputil::Queue none;
  CHECK(!none.isValid());
  gputil::Queue queue = gpu.createQueue();
  CHECK(queue.isValid());
  gputil::Queue same = queue;  // copies share the stream
  CHECK(same.internal() == queue.internal());
  gputil::Event unset;
  CHECK(!unset.isValid() && unset.isComplete());
  gputil::Event mark = queue.mark();
  CHECK(mark.isValid());
  mark.wait();
  CHECK(mark.isComplete());
  gputil::Event shared = mark;  // reference counted
  mark.release();
  CHECK(!mark.isValid() && shared.isValid() && shared.isComplete());
  gputil::Event moved(std::move(shared));
  CHECK(moved.isValid() && !shared.isValid());
  const gputil::Event list[2] = { moved, gpu.defaultQueue().mark() };
  gputil::Event::wait(list, 2);
  bool called = false;
  queue.queueCallback([&called]() { called = true; });
  CHECK(called);
  queue.setSynchronous(true);
  CHECK(queue.synchronous() && queue.mark().isComplete());
  queue.finish();
  std::printf("GPUTIL_HIP_OK %s | %s\n", gpu.name(), gpu.description());
  return 0;
}
