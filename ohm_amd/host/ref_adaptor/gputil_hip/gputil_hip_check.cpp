// gputil_hip_check.cpp -- exercises the HIP backend of gputil (this directory) through the reference's own class
// declarations: Device / Queue / Event (gputilHip.cpp), Buffer / copyBuffer / PinnedBuffer (gputilHipBuffer.cpp) and the
// reference's backend-independent EventList compiled on top of them.  The buffer cases follow
// /root/reference/tests/gputiltest/GpuBufferTest.cpp:431-640 (Ref, ReadWriteCopy, Pinned, Allocation) with the
// asynchronous forms of the class documentation (gputil/gpuBuffer.h:56-71) added.  Built by __graft_entry__.build()
// when the reference checkout is present (headers compiled against where they lie), run on the GPU by
// tests/test_gpu_cpp_host.py.
#include <gputil/gpuBuffer.h>
#include <gputil/gpuDevice.h>
#include <gputil/gpuEvent.h>
#include <gputil/gpuEventList.h>
#include <gputil/gpuPinnedBuffer.h>
#include <gputil/gpuQueue.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(cond)                                                   \
  if (!(cond))                                                        \
  {                                                                   \
    std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);     \
    return 1;                                                         \
  }

namespace
{
int checkDeviceQueueEvent(gputil::Device &gpu)
{
  std::vector<gputil::DeviceInfo> devices;
  const unsigned count = gputil::Device::enumerateDevices(devices);
  CHECK(count >= 1 && devices.size() == count);
  gputil::Device invalid;
  CHECK(!invalid.isValid());
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
  return 0;
}

// tests/gputiltest/GpuBufferTest.cpp:431 (Ref): an event handed out as `completion` of a queued write is shared by
// reference count -- copies keep it alive, releasing the original does not end the others.
int checkRef(gputil::Device &gpu)
{
  gputil::Event event;
  gputil::Buffer buffer(gpu, 64 * 1024u, gputil::kBfReadWriteHost);
  gputil::Queue queue = gpu.createQueue();
  std::vector<uint8_t> host(buffer.size());
  for (size_t i = 0; i < host.size(); ++i)
  {
    host[i] = uint8_t(i % 256);
  }
  CHECK(buffer.write(host.data(), host.size(), 0, &queue, nullptr, &event) == host.size());
  CHECK(event.isValid());
  gputil::Event second = event;  // +ref
  queue.finish();
  CHECK(event.isComplete() && second.isComplete());
  event.release();
  CHECK(!event.isValid() && second.isValid() && second.isComplete());
  std::vector<uint8_t> back(host.size(), 0);
  CHECK(buffer.read(back.data(), back.size()) == back.size() && back == host);
  return 0;
}

// tests/gputiltest/GpuBufferTest.cpp:474 (ReadWriteCopy), then the same through the asynchronous forms.
int checkReadWriteCopy(gputil::Device &gpu)
{
  const std::string ref = "The quick brown fox jumps over the lazy dog.";
  std::string got;
  got.resize(ref.size());
  gputil::Buffer buffer(gpu, ref.size() * 2);
  CHECK(buffer.isValid() && buffer.size() == ref.size() * 2 && buffer.actualSize() >= buffer.size());
  CHECK(buffer.flags() == gputil::kBfReadWrite);
  CHECK(buffer.write(ref.data(), ref.size()) == ref.size());
  CHECK(buffer.read(&got.front(), got.size()) == got.size());
  CHECK(got == ref);
  const unsigned offset = 4;
  buffer.write(ref.data(), ref.size(), offset);
  buffer.read(&got.front(), got.size(), offset);
  CHECK(got == ref);
  const std::string offset_ref = "The The quick brown fox jumps over the lazy dog.";
  got.resize(ref.size() + offset);
  buffer.read(&got.front(), got.size());
  CHECK(got == offset_ref);
  buffer.write(ref.data(), ref.size());
  gputil::Buffer buffer2(gpu, ref.size() * 2);
  CHECK(gputil::copyBuffer(buffer2, buffer) == buffer.size());
  got.assign(ref.size(), '\0');
  buffer2.read(&got.front(), got.size());
  CHECK(got == ref);
  // reads and writes are clipped to size()
  CHECK(buffer.read(&got.front(), got.size(), buffer.size()) == 0);
  CHECK(buffer.write(ref.data(), ref.size(), buffer.size() - 3) == 3);

  // asynchronous: write on a queue with a completion event, copy blocked on it, read blocked on the copy
  gputil::Queue queue = gpu.createQueue();
  std::vector<uint32_t> big(1u << 20);
  for (size_t i = 0; i < big.size(); ++i)
  {
    big[i] = uint32_t(i * 2654435761u);
  }
  gputil::Buffer a(gpu, big.size() * sizeof(uint32_t));
  gputil::Buffer b(gpu, big.size() * sizeof(uint32_t));
  gputil::Event written, copied, read_done;
  CHECK(a.writeElements(big.data(), big.size(), 0, &queue, nullptr, &written) == big.size());
  CHECK(written.isValid());
  gputil::Queue queue2 = gpu.createQueue();
  CHECK(gputil::copyBuffer(b, a, &queue2, &written, &copied) == a.size());
  std::vector<uint32_t> back(big.size(), 0);
  CHECK(b.readElements(back.data(), back.size(), 0, &queue, &copied, &read_done) == back.size());
  CHECK(read_done.isValid());
  read_done.wait();
  CHECK(written.isComplete() && copied.isComplete() && back == big);
  // blocking call that waits on an event first (queue == nullptr, block_on given)
  gputil::Event filled;
  const uint32_t pattern = 0xA5C31E0Fu;
  b.fill(&pattern, sizeof(pattern), &queue, nullptr, &filled);
  CHECK(b.readElements(back.data(), back.size(), 0, nullptr, &filled) == back.size());
  CHECK(std::all_of(back.begin(), back.end(), [pattern](uint32_t v) { return v == pattern; }));
  // fillPartial + clear
  const uint8_t three[3] = { 1, 2, 3 };
  b.clear(0);
  b.fillPartial(three, sizeof(three), 10, 6);
  uint8_t head[20];
  b.read(head, sizeof(head));
  const uint8_t expect[20] = { 0, 0, 0, 0, 0, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 1, 0, 0, 0, 0 };
  CHECK(std::memcmp(head, expect, sizeof(expect)) == 0);
  // strided elements: 3-float host vectors against 4-float device vectors (gpuBuffer.h:291-300)
  struct F3
  {
    float x, y, z;
  };
  struct F4
  {
    float x, y, z, w;
  };
  std::vector<F3> pts(100);
  for (size_t i = 0; i < pts.size(); ++i)
  {
    pts[i] = F3{ float(i), float(2 * i), float(3 * i) };
  }
  gputil::Buffer vec(gpu, pts.size() * sizeof(F4));
  CHECK((vec.writeElements<F4>(pts.data(), pts.size()) == pts.size()));
  std::vector<F4> raw(pts.size());
  vec.read(raw.data(), raw.size() * sizeof(F4));
  CHECK(raw[7].x == 7.0f && raw[7].z == 21.0f && raw[7].w == 0.0f);
  std::vector<F3> pts_back(pts.size());
  CHECK((vec.readElements<F4>(pts_back.data(), pts_back.size(), 0) == pts.size()));
  CHECK(std::memcmp(pts.data(), pts_back.data(), pts.size() * sizeof(F3)) == 0);
  CHECK((vec.readElements<F4>(pts_back.data(), 10, 95) == 5));  // clipped to what the buffer holds
  CHECK(pts_back[0].x == 95.0f);
  // resize: grow only; forceResize may shrink; swap; move
  const size_t before = vec.actualSize();
  CHECK(vec.resize(16) == before && vec.size() == 16);
  CHECK(vec.resize(before * 2) >= before * 2 && vec.size() == before * 2);
  CHECK(vec.forceResize(1000) < before * 2 && vec.size() == 1000 && vec.actualSize() >= 1000);
  CHECK(vec.elementsResize<double>(500) >= 500 && vec.elementCount<double>() == 500);
  gputil::Buffer other(gpu, 64);
  void *vec_address = vec.address();
  other.swap(vec);
  CHECK(other.address() == vec_address && vec.size() == 64);
  gputil::Buffer taken(std::move(other));
  CHECK(taken.address() == vec_address && taken.arg<char *>() == static_cast<char *>(vec_address));
  taken.release();
  CHECK(!taken.isValid() && taken.size() == 0);
  gputil::Buffer unset;
  CHECK(!unset.isValid() && unset.read(head, 4) == 0);
  unset.create(gpu, 128, gputil::kBfReadWriteHost);
  CHECK(unset.isValid() && unset.size() == 128 && (unset.flags() & gputil::kBfHostAccess));
  return 0;
}

// tests/gputiltest/GpuBufferTest.cpp:522 (Pinned), plus the asynchronous unpin and the unpinned fallback.
int checkPinned(gputil::Device &gpu)
{
  const std::string ref = "The quick brown fox jumps over the lazy dog.";
  std::string got;
  got.resize(ref.size());
  gputil::Buffer buffer(gpu, ref.size() * 2, gputil::kBfReadWriteHost);
  {
    gputil::PinnedBuffer pin;
    CHECK(!pin.isPinned());
    pin = gputil::PinnedBuffer(buffer, gputil::kPinWrite);
    CHECK(pin.isPinned() && pin.mode() == gputil::kPinWrite && pin.buffer() == &buffer);
    CHECK(pin.write(ref.data(), ref.size()) == ref.size());
    pin.unpin();
    CHECK(!pin.isPinned());
    pin = gputil::PinnedBuffer(buffer, gputil::kPinRead);
    CHECK(pin.isPinned());
    CHECK(pin.read(&got.front(), got.size()) == got.size());
    pin.unpin();
    CHECK(!pin.isPinned());
    CHECK(got == ref);
  }
  const unsigned offset = 4;
  gputil::PinnedBuffer write_pin(buffer, gputil::kPinWrite);
  CHECK(write_pin.isPinned());
  write_pin.write(ref.data(), ref.size(), offset);
  write_pin.unpin();
  CHECK(!write_pin.isPinned());
  gputil::PinnedBuffer read_pin(buffer, gputil::kPinRead);
  CHECK(read_pin.isPinned());
  read_pin.read(&got.front(), got.size(), offset);
  read_pin.unpin();
  CHECK(!read_pin.isPinned());
  CHECK(got == ref);
  const std::string offset_ref = "The The quick brown fox jumps over the lazy dog.";
  read_pin.pin();
  CHECK(read_pin.isPinned());
  got.resize(ref.size() + offset);
  read_pin.read(&got.front(), got.size());
  read_pin.unpin();
  CHECK(!read_pin.isPinned());
  CHECK(got == offset_ref);

  // pinned write, asynchronous unpin with a completion event, device-side copy blocked on it (gpuPinnedBuffer.h:53-67)
  gputil::Queue queue = gpu.createQueue();
  std::vector<uint32_t> data(1u << 18);
  for (size_t i = 0; i < data.size(); ++i)
  {
    data[i] = uint32_t(i ^ 0x5bd1e995u);
  }
  gputil::Buffer staging(gpu, data.size() * sizeof(uint32_t), gputil::kBfReadWriteHost);
  gputil::Buffer device(gpu, data.size() * sizeof(uint32_t));
  gputil::PinnedBuffer wp(staging, gputil::kPinWrite);
  CHECK(wp.isPinned());
  CHECK(wp.writeElements(data.data(), data.size()) == data.size());
  gputil::Event unpinned, copied;
  wp.unpin(&queue, nullptr, &unpinned);
  CHECK(!wp.isPinned() && unpinned.isValid());
  CHECK(gputil::copyBuffer(device, staging, &queue, &unpinned, &copied) == staging.size());
  std::vector<uint32_t> back(data.size(), 0);
  CHECK(device.readElements(back.data(), back.size(), 0, nullptr, &copied) == back.size());
  CHECK(back == data);
  // a plain device buffer is not pinned: the same calls fall back to transfers
  gputil::PinnedBuffer fallback(device, gputil::kPinReadWrite);
  CHECK(!fallback.isPinned());
  uint32_t word = 0;
  CHECK(fallback.read(&word, sizeof(word), 5 * sizeof(uint32_t)) == sizeof(word) && word == data[5]);
  word = 77;
  CHECK(fallback.write(&word, sizeof(word), 0) == sizeof(word));
  CHECK(fallback.readElements(back.data(), 2, 0) == 2 && back[0] == 77 && back[1] == data[1]);
  return 0;
}

// tests/gputiltest/GpuBufferTest.cpp:590 (Allocation): memory is really released -- 20 x (max allocation, <= 2 GiB) / 4
int checkAllocation(gputil::Device &gpu)
{
  uint64_t alloc_size = std::min<uint64_t>(gpu.maxAllocationSize(), uint64_t(2) << 30) / 4;
  const int fill_value = 42;
  for (int i = 0; i < 20; ++i)
  {
    gputil::Buffer mem(gpu, size_t(alloc_size));
    CHECK(mem.isValid());
    mem.fill(&fill_value, sizeof(fill_value));
    if (i == 19)
    {
      int tail[4] = { 0, 0, 0, 0 };
      CHECK(mem.read(tail, sizeof(tail), size_t(alloc_size) - sizeof(tail)) == sizeof(tail));
      CHECK(tail[0] == 42 && tail[3] == 42);
    }
  }
  return 0;
}

// gputil/gpuEventList.h: the reference's own (backend independent) implementation on this backend's Event.
int checkEventList(gputil::Device &gpu)
{
  gputil::Queue queue = gpu.createQueue();
  gputil::EventList empty;
  CHECK(empty.count() == 0 && empty.capacity() == gputil::EventList::kShortCount);
  gputil::Event first = queue.mark();
  gputil::EventList one(first);
  CHECK(one.count() == 1 && one.events()[0].isValid());
  gputil::EventList many;
  for (int i = 0; i < 20; ++i)
  {
    many.add(queue.mark());
  }
  CHECK(many.size() == 20 && many.capacity() >= 20);
  gputil::Event::wait(many.events(), many.count());
  for (size_t i = 0; i < many.count(); ++i)
  {
    CHECK(many.events()[i].isComplete());
  }
  many.clear();
  CHECK(many.count() == 0 && first.isValid());
  const gputil::Event second = queue.mark();
  gputil::EventList pair({ &first, &second });
  CHECK(pair.count() == 2);
  return 0;
}
}  // namespace

int main()
{
  gputil::Device gpu(true);
  if (checkDeviceQueueEvent(gpu) || checkRef(gpu) || checkReadWriteCopy(gpu) || checkPinned(gpu) || checkAllocation(gpu) ||
      checkEventList(gpu))
  {
    return 1;
  }
  std::printf("GPUTIL_HIP_OK %s | %s\n", gpu.name(), gpu.description());
  return 0;
}
