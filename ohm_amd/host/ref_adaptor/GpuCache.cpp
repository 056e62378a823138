// GpuCache.cpp -- ohm::GpuCache (declared in the reference's ohmgpu/GpuCache.h) over libohmhip.so.
//
// The reference's GpuCache owns one fixed-slot GpuLayerCache per voxel layer and moves regions between host and device
// on demand (ohmgpu/GpuCache.cpp:44-186, ohmgpu/GpuLayerCache.cpp:429-633).  Here the whole map is resident in one
// device map (include/ohmhip.h); this class is the MapRegionCache face the core map calls through
// OccupancyMapDetail::gpu_cache (ohm/OccupancyMap.cpp:1202-1234 cullRegions -> remove; layout changes -> reinitialise;
// the map's destructor deletes it, ohm/private/OccupancyMapDetail.cpp:21-24) and the owner of the device map.
#include <ohmgpu/GpuCache.h>

#include "private/HipMapBinding.h"

#include <ohmgpu/OhmGpu.h>

#include <ohm/OccupancyMap.h>

#include <gputil/gpuDevice.h>
#include <gputil/gpuQueue.h>

namespace ohm
{
struct GpuCacheDetail
{
  HipMapBinding binding;
  size_t target_gpu_alloc_size = 0;
  unsigned flags = 0;
  gputil::Device gpu;
  gputil::Queue queue;
};

GpuCache::GpuCache(OccupancyMap &map, size_t target_gpu_alloc_size, unsigned flags)
  : imp_(new GpuCacheDetail)
{
  imp_->binding.map = &map;
  imp_->binding.gpu_mem_size = target_gpu_alloc_size;
  imp_->target_gpu_alloc_size = target_gpu_alloc_size;
  imp_->flags = flags;
  imp_->gpu = ohm::gpuDevice();
  imp_->queue = imp_->gpu.defaultQueue();
  registerHipBinding(map, &imp_->binding);
}

GpuCache::~GpuCache()
{
  if (imp_)
  {
    unregisterHipBinding(*imp_->binding.map);
    delete imp_;
  }
}

void GpuCache::reinitialise()
{
  // The host layout changed (OccupancyMap::updateLayout): bring the host up to date, then rebuild the device map for
  // the new layer set from the host copy.
  HipMapBinding &binding = imp_->binding;
  if (binding.core.valid())
  {
    binding.download({}, true);
    binding.create(binding.core.kind(), nullptr, nullptr);  // the next batch pushes the NDT / TSDF parameters again
  }
}

void GpuCache::flush()
{
  imp_->binding.download({}, true);
}

void GpuCache::clear()
{
  // Drop residency without download (used after CPU-side edits, e.g. tests/ohmtestgpu/GpuNdtTests.cpp:151-152): the
  // host copy is authoritative afterwards, so everything the host holds is uploaded again before the next batch.
  imp_->binding.core.clearResidency();
}

void GpuCache::removeLayers()
{
  imp_->binding.core.destroy();
}

void GpuCache::remove(const glm::i16vec3 &region_key)
{
  const int16_t key[3] = { region_key.x, region_key.y, region_key.z };
  imp_->binding.core.removeRegion(key);
}

bool GpuCache::syncLayerTo(MapChunk &dst_chunk, unsigned dst_layer, const MapChunk &src_chunk, unsigned src_layer)
{
  // The reference copies a layer of a cached region straight into another map's chunk (ohmgpu/GpuLayerCache.cpp:
  // 636-667).  With the whole map resident the simplest correct answer is "not handled here": the caller
  // (ohm/CopyUtil.cpp) then syncs the source map and copies on the host.
  (void)dst_chunk;
  (void)dst_layer;
  (void)src_chunk;
  (void)src_layer;
  return false;
}

MapRegionCache *GpuCache::findLayerCache(unsigned layer)
{
  (void)layer;
  return this;
}

size_t GpuCache::targetGpuAllocSize() const
{
  return imp_->target_gpu_alloc_size;
}

unsigned GpuCache::layerCount() const
{
  return imp_->binding.core.layerCount();
}

GpuLayerCache *GpuCache::createCache(unsigned id, const GpuLayerCacheParams &params)
{
  // Per-layer caches do not exist in this backend.
  (void)id;
  (void)params;
  return nullptr;
}

GpuLayerCache *GpuCache::layerCache(unsigned id)
{
  (void)id;
  return nullptr;
}

gputil::Device &GpuCache::gpu()
{
  return imp_->gpu;
}

const gputil::Device &GpuCache::gpu() const
{
  return imp_->gpu;
}

gputil::Queue &GpuCache::gpuQueue()
{
  return imp_->queue;
}

const gputil::Queue &GpuCache::gpuQueue() const
{
  return imp_->queue;
}
}  // namespace ohm
