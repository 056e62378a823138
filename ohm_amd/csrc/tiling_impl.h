// tiling_impl.h -- regions larger than one LDS tile (include/ohmhip.h, "LARGE REGIONS"; ohm/OccupancyMap.h:287 takes any
// glm::u8vec3 region size, up to 255 voxels per axis).  Included at the end of ohmhip_map.hip.
//
// Inside the library such a region is a set of equal TILES of at most 32768 voxels (MapConst::dim); the whole pipeline --
// hash table, pool slots, LDS count tile, 15-bit voxel indices in segment, sample and event keys -- works in tiles and
// never knows.  Tiles are cut so that each is a contiguous piece of the region's MapChunk block (index x + y*dx +
// z*dx*dy): full x rows always, then either whole x-y layers in z slabs, or -- when one layer alone exceeds a tile --
// y strips of single layers.  What is left is this file: the C ABI speaks the caller's region keys, so the entry points
// that name or list regions translate between a region and its tiles.
#ifndef OHMHIP_TILING_IMPL_H
#define OHMHIP_TILING_IMPL_H

namespace
{
/// The tile edge per axis for a region of dims[]: {dx, ty, tz} with ty | dy, tz | dz and dx * ty * tz <= limit.
void chooseTileDims(const int dims[3], int limit, int tile[3])
{
  tile[0] = dims[0];
  tile[1] = dims[1];
  tile[2] = dims[2];
  if (int64_t(dims[0]) * dims[1] * dims[2] <= limit)
  {
    return;
  }
  auto largestDivisor = [](int n, int64_t unit, int64_t budget) {
    int best = 1;
    for (int d = 1; d <= n; ++d)
    {
      if (n % d == 0 && unit * d <= budget)
      {
        best = d;
      }
    }
    return best;
  };
  if (int64_t(dims[0]) * dims[1] <= limit)
  {
    tile[2] = largestDivisor(dims[2], int64_t(dims[0]) * dims[1], limit);  // z slabs of whole layers
  }
  else
  {
    tile[2] = 1;  // single layers, cut into y strips
    tile[1] = largestDivisor(dims[1], dims[0], limit);
  }
}

/// While > 0 the entry points treat keys as tile keys (the translation layer calling back into them).
struct TilePassthrough
{
  ohmhip_map_t m;
  explicit TilePassthrough(ohmhip_map_t map) : m(map) { ++m->tile_passthrough; }
  ~TilePassthrough() { --m->tile_passthrough; }
};

struct TileRef
{
  int16_t key[3];
  size_t voxel_offset;  ///< of the tile's first voxel in the region's block
};

/// The tiles of region `key`, in block order.
void tilesOfRegion(const MapConst &mc, const int16_t *key, std::vector<TileRef> &out)
{
  out.clear();
  for (int jz = 0; jz < mc.tile_split[2]; ++jz)
  {
    for (int jy = 0; jy < mc.tile_split[1]; ++jy)
    {
      TileRef t;
      t.key[0] = key[0];
      t.key[1] = int16_t(int(key[1]) * mc.tile_split[1] + jy);
      t.key[2] = int16_t(int(key[2]) * mc.tile_split[2] + jz);
      t.voxel_offset = (size_t(jz) * size_t(mc.dim[2]) * size_t(mc.kdim[1]) + size_t(jy) * size_t(mc.dim[1])) * size_t(mc.kdim[0]);
      out.push_back(t);
    }
  }
}

bool regionFitsTileKeys(const MapConst &mc, const int16_t *key)
{
  for (int a = 1; a < 3; ++a)
  {
    const int lo = int(key[a]) * mc.tile_split[a];
    if (lo < -32768 || lo + mc.tile_split[a] - 1 > 32767)
    {
      return false;
    }
  }
  return true;
}

void regionOfTile(const MapConst &mc, const int16_t *tile, int16_t *region)
{
  region[0] = tile[0];
  region[1] = int16_t(floorDiv(tile[1], mc.tile_split[1]));
  region[2] = int16_t(floorDiv(tile[2], mc.tile_split[2]));
}

/// Tile keys -> the distinct regions they belong to, in order of first appearance.
void regionsOfTiles(const MapConst &mc, const std::vector<int16_t> &tiles, std::vector<int16_t> &regions)
{
  regions.clear();
  std::unordered_map<uint64_t, char> seen;
  for (size_t i = 0; i + 2 < tiles.size(); i += 3)
  {
    int16_t r[3];
    regionOfTile(mc, &tiles[i], r);
    if (seen.emplace(packRegionKey(r[0], r[1], r[2]), 1).second)
    {
      regions.insert(regions.end(), r, r + 3);
    }
  }
}

void fillLayerClear(int layer_id, void *dst, size_t voxels)
{
  // ohm/DefaultLayer.cpp:87-91: occupancy clears to +inf, every other layer to zero bytes
  if (layer_id == OHMHIP_LID_OCCUPANCY)
  {
    uint32_t *p = static_cast<uint32_t *>(dst);
    for (size_t i = 0; i < voxels; ++i)
    {
      p[i] = 0x7f800000u;
    }
  }
  else
  {
    std::memset(dst, 0, voxels * kLayerBytes[layer_id]);
  }
}

int tiledListRegions(ohmhip_map_t m, bool dirty_only, int16_t *keys_xyz, size_t capacity, size_t *count)
{
  std::vector<int16_t> tiles;
  {
    TilePassthrough pass(m);
    size_t n = 0;
    OHMHIP_CHECK(dirty_only ? ohmhip_map_dirty_regions(m, nullptr, 0, &n) : ohmhip_map_regions(m, nullptr, 0, &n));
    tiles.resize(3 * n);
    if (n)
    {
      OHMHIP_CHECK(dirty_only ? ohmhip_map_dirty_regions(m, tiles.data(), n, &n) : ohmhip_map_regions(m, tiles.data(), n, &n));
      tiles.resize(3 * n);
    }
  }
  std::vector<int16_t> regions;
  regionsOfTiles(m->mc, tiles, regions);
  *count = regions.size() / 3;
  if (keys_xyz)
  {
    std::memcpy(keys_xyz, regions.data(), sizeof(int16_t) * 3 * std::min(capacity, regions.size() / 3));
  }
  return OHMHIP_OK;
}

bool tileExists(ohmhip_map_t m, const int16_t *tile)
{
  const uint64_t packed = packRegionKey(tile[0], tile[1], tile[2]);
  return m->region_slots.find(packed) != m->region_slots.end() || m->spilled.find(packed) != m->spilled.end();
}

int tiledReadRegions(ohmhip_map_t m, int layer_id, const int16_t *keys_xyz, size_t count, void *const *dsts)
{
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  OHMHIP_CHECK(refreshHostRegionTable(m));
  const size_t vb = kLayerBytes[layer_id];
  std::vector<int16_t> tile_keys;
  std::vector<void *> tile_dsts;
  std::vector<TileRef> tiles;
  for (size_t i = 0; i < count; ++i)
  {
    const int16_t *key = keys_xyz + 3 * i;
    if (!regionFitsTileKeys(m->mc, key))
    {
      return OHMHIP_ERR_NOT_FOUND;
    }
    tilesOfRegion(m->mc, key, tiles);
    size_t present = 0;
    for (const TileRef &t : tiles)
    {
      char *dst = static_cast<char *>(dsts[i]) + t.voxel_offset * vb;
      if (tileExists(m, t.key))
      {
        tile_keys.insert(tile_keys.end(), t.key, t.key + 3);
        tile_dsts.push_back(dst);
        ++present;
      }
      else
      {
        fillLayerClear(layer_id, dst, size_t(m->mc.region_voxels));  // a tile no ray has reached yet
      }
    }
    if (present == 0)
    {
      return OHMHIP_ERR_NOT_FOUND;
    }
  }
  TilePassthrough pass(m);
  return ohmhip_map_read_regions(m, layer_id, tile_keys.data(), tile_dsts.size(), tile_dsts.data());
}

int tiledWriteRegions(ohmhip_map_t m, int layer_id, const int16_t *keys_xyz, size_t count, const void *const *srcs)
{
  const size_t vb = kLayerBytes[layer_id];
  std::vector<int16_t> tile_keys;
  std::vector<const void *> tile_srcs;
  std::vector<TileRef> tiles;
  for (size_t i = 0; i < count; ++i)
  {
    if (!regionFitsTileKeys(m->mc, keys_xyz + 3 * i))
    {
      return OHMHIP_ERR_INVALID_ARG;
    }
    tilesOfRegion(m->mc, keys_xyz + 3 * i, tiles);
    for (const TileRef &t : tiles)
    {
      tile_keys.insert(tile_keys.end(), t.key, t.key + 3);
      tile_srcs.push_back(static_cast<const char *>(srcs[i]) + t.voxel_offset * vb);
    }
  }
  TilePassthrough pass(m);
  return ohmhip_map_write_regions(m, layer_id, tile_keys.data(), tile_srcs.size(), tile_srcs.data());
}

int tiledRemoveRegions(ohmhip_map_t m, const int16_t *keys_xyz, size_t count, size_t *removed)
{
  OHMHIP_CHECK(hipStreamSynchronize(m->stream));
  OHMHIP_CHECK(refreshHostRegionTable(m));
  std::vector<int16_t> tile_keys;
  std::vector<TileRef> tiles;
  size_t regions_hit = 0;
  for (size_t i = 0; i < count; ++i)
  {
    if (!regionFitsTileKeys(m->mc, keys_xyz + 3 * i))
    {
      continue;
    }
    tilesOfRegion(m->mc, keys_xyz + 3 * i, tiles);
    bool any = false;
    for (const TileRef &t : tiles)
    {
      if (tileExists(m, t.key))
      {
        tile_keys.insert(tile_keys.end(), t.key, t.key + 3);
        any = true;
      }
    }
    regions_hit += any ? 1u : 0u;
  }
  size_t tiles_removed = 0;
  TilePassthrough pass(m);
  const int err = ohmhip_map_remove_regions(m, tile_keys.data(), tile_keys.size() / 3, &tiles_removed);
  if (removed)
  {
    *removed = (err == OHMHIP_OK) ? regions_hit : 0;
  }
  return err;
}
}  // namespace

#endif  // OHMHIP_TILING_IMPL_H
