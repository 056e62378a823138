/*
 * ohm_oracle.c -- CPU restatement (plain C99) of the reference ohm CPU ray mappers.
 *
 * TEST INFRASTRUCTURE ONLY (see ohm_oracle.h).  Restates, it does not copy: every function names the
 * reference file:line whose arithmetic and evaluation order it follows.  Build with
 *   gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math -shared -fPIC
 * so that double/float results are bit-identical to an x86-64 (SSE2) build of the reference.
 *
 * glm semantics relied on (glm is header-only vector plumbing, it holds none of the path's arithmetic):
 *   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   length(v)     = sqrt(dot(v,v));  length2(v) = dot(v,v)
 *   normalize(v)  = v * (1 / sqrt(dot(v,v)))
 */
#include "ohm_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------- */
/* Small vector helpers (double)                                                                                  */
/* ------------------------------------------------------------------------------------------------------------- */
typedef struct
{
  double x, y, z;
} dv3;

static dv3 dv3_make(double x, double y, double z)
{
  dv3 v;
  v.x = x;
  v.y = y;
  v.z = z;
  return v;
}
static dv3 dv3_sub(dv3 a, dv3 b) { return dv3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static dv3 dv3_add(dv3 a, dv3 b) { return dv3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static dv3 dv3_scale(dv3 a, double s) { return dv3_make(a.x * s, a.y * s, a.z * s); }
static double dv3_dot(dv3 a, dv3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static dv3 dv3_normalize(dv3 v) { return dv3_scale(v, 1.0 / sqrt(dv3_dot(v, v))); }

/* ------------------------------------------------------------------------------------------------------------- */
/* Map + chunk storage                                                                                            */
/* ------------------------------------------------------------------------------------------------------------- */
static const size_t k_layer_bytes[ORACLE_LID_COUNT] = { 4, 8, 24, 4, 4, 4, 8, 8, 8 };

typedef struct OracleChunk
{
  int16_t region[3];
  int used;
  void *layers[ORACLE_LID_COUNT];
} OracleChunk;

struct OracleMap
{
  double resolution;
  int dim[3];
  double region_dim[3]; /* spatial, ohm/OccupancyMap.cpp:200-202 */
  double origin[3];
  unsigned layers;
  float hit_value, miss_value, threshold_value;
  float min_value, max_value;
  int saturate_min, saturate_max;
  int filter_mode;
  /* RayFilterFlag bits per ray of the NEXT integrate call, when the caller ran the RayFilterFunction itself (an
   * arbitrary host function in the reference, ohm/RayFilter.h:45); NULL: the built-in filter below applies. */
  const unsigned char *batch_filter_flags;
  double filter_range;
  double first_ray_time;
  /* NDT */
  float sensor_noise;
  unsigned sample_threshold;
  float adaptation_rate;
  float reinit_threshold;
  unsigned reinit_count;
  float initial_intensity_cov;
  int ndt_tm;
  /* TSDF */
  float tsdf_max_weight, tsdf_trunc, tsdf_dropoff, tsdf_sparsity;
  /* chunk hash table (open addressing) */
  OracleChunk *table;
  size_t table_cap; /* power of two */
  size_t table_count;
  uint64_t visits;
};

size_t oracle_layer_voxel_bytes(int layer_id)
{
  return (layer_id >= 0 && layer_id < ORACLE_LID_COUNT) ? k_layer_bytes[layer_id] : 0;
}

/* ohm/MapProbability.h:20-36 (float instantiation) */
float oracle_probability_to_value(float p)
{
  return logf(p / (1.0f - p));
}
float oracle_value_to_probability(float v)
{
  return (v == -INFINITY) ? 0.0f : 1.0f - (1.0f / (1.0f + expf(v)));
}

/* ohm/NdtMap.h:146-149 */
static float ndt_rate_from_miss_probability(float miss_probability, float scale)
{
  float r = scale * (1.0f - 2.0f * miss_probability);
  r = (r < 1.0f) ? r : 1.0f;
  return (0.0f < r) ? r : 0.0f;
}

OracleMap *oracle_map_create(double resolution, int dimx, int dimy, int dimz, unsigned layers)
{
  OracleMap *m = (OracleMap *)calloc(1, sizeof(OracleMap));
  m->resolution = resolution;
  m->dim[0] = dimx > 0 ? dimx : 32; /* ohm/OccupancyMap.h:24-26 default 32 */
  m->dim[1] = dimy > 0 ? dimy : 32;
  m->dim[2] = dimz > 0 ? dimz : 32;
  for (int i = 0; i < 3; ++i)
  {
    m->region_dim[i] = m->dim[i] * resolution; /* ohm/OccupancyMap.cpp:200-202 */
  }
  m->layers = layers;
  /* ohm/OccupancyMap.cpp:205-213 */
  m->min_value = -2.0f;
  m->max_value = 3.511f;
  m->hit_value = oracle_probability_to_value(0.9f);
  m->miss_value = oracle_probability_to_value(0.45f);
  m->threshold_value = oracle_probability_to_value(0.5f);
  m->filter_mode = ORACLE_FILTER_GOOD; /* ohm/OccupancyMap.cpp:215-218 */
  m->filter_range = 1e10;
  m->first_ray_time = -1.0;
  /* ohm/private/NdtMapDetail.h:20-45 */
  m->sensor_noise = 0.05f;
  m->sample_threshold = 3;
  m->adaptation_rate = ndt_rate_from_miss_probability(oracle_value_to_probability(m->miss_value), 2.0f);
  m->reinit_threshold = oracle_probability_to_value(0.2f);
  m->reinit_count = 100;
  m->initial_intensity_cov = 1.0f;
  /* ohm/VoxelTsdf.h:27-37 */
  m->tsdf_max_weight = 1e4f;
  m->tsdf_trunc = 0.1f;
  m->tsdf_dropoff = 0.0f;
  m->tsdf_sparsity = 1.0f;
  m->table_cap = 1024;
  m->table = (OracleChunk *)calloc(m->table_cap, sizeof(OracleChunk));
  return m;
}

static void chunk_free(OracleChunk *c)
{
  for (int l = 0; l < ORACLE_LID_COUNT; ++l)
  {
    free(c->layers[l]);
    c->layers[l] = NULL;
  }
}

void oracle_map_destroy(OracleMap *m)
{
  if (!m)
  {
    return;
  }
  for (size_t i = 0; i < m->table_cap; ++i)
  {
    if (m->table[i].used)
    {
      chunk_free(&m->table[i]);
    }
  }
  free(m->table);
  free(m);
}

void oracle_map_set_origin(OracleMap *m, double x, double y, double z)
{
  m->origin[0] = x;
  m->origin[1] = y;
  m->origin[2] = z;
}
void oracle_map_set_hit_probability(OracleMap *m, float p) { m->hit_value = oracle_probability_to_value(p); }
void oracle_map_set_miss_probability(OracleMap *m, float p)
{
  m->miss_value = oracle_probability_to_value(p);
}
void oracle_map_set_threshold_probability(OracleMap *m, float p)
{
  m->threshold_value = oracle_probability_to_value(p);
}
void oracle_map_set_hit_value(OracleMap *m, float v) { m->hit_value = v; }
void oracle_map_set_miss_value(OracleMap *m, float v) { m->miss_value = v; }
void oracle_map_set_min_max(OracleMap *m, float mn, float mx)
{
  m->min_value = mn;
  m->max_value = mx;
}
void oracle_map_set_saturation(OracleMap *m, int at_min, int at_max)
{
  m->saturate_min = at_min;
  m->saturate_max = at_max;
}
void oracle_map_set_batch_filter_flags(OracleMap *m, const unsigned char *flags)
{
  m->batch_filter_flags = flags;
}

void oracle_map_set_ray_filter(OracleMap *m, int mode, double range)
{
  m->filter_mode = mode;
  m->filter_range = range;
}
float oracle_map_hit_value(const OracleMap *m) { return m->hit_value; }
float oracle_map_miss_value(const OracleMap *m) { return m->miss_value; }

void oracle_map_set_ndt(OracleMap *m, float sensor_noise, unsigned sample_threshold, float adaptation_rate,
                        float reinit_threshold, unsigned reinit_count, float initial_intensity_cov, int ndt_tm)
{
  m->sensor_noise = sensor_noise;
  m->sample_threshold = sample_threshold;
  m->adaptation_rate = (adaptation_rate > 0) ?
                         adaptation_rate :
                         ndt_rate_from_miss_probability(oracle_value_to_probability(m->miss_value), 2.0f);
  m->reinit_threshold = reinit_threshold;
  m->reinit_count = reinit_count;
  m->initial_intensity_cov = initial_intensity_cov;
  m->ndt_tm = ndt_tm;
}
float oracle_map_ndt_adaptation_rate(const OracleMap *m) { return m->adaptation_rate; }

void oracle_map_set_tsdf(OracleMap *m, float max_weight, float trunc, float dropoff, float sparsity)
{
  m->tsdf_max_weight = max_weight;
  m->tsdf_trunc = trunc;
  m->tsdf_dropoff = dropoff;
  m->tsdf_sparsity = sparsity;
}

uint64_t oracle_map_visit_count(const OracleMap *m) { return m->visits; }

static size_t region_hash(int rx, int ry, int rz)
{
  uint64_t h = (uint64_t)(uint16_t)rx | ((uint64_t)(uint16_t)ry << 16) | ((uint64_t)(uint16_t)rz << 32);
  h ^= h >> 33;
  h *= 0xff51afd7ed558ccdULL;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ULL;
  h ^= h >> 33;
  return (size_t)h;
}

static void chunk_init(const OracleMap *m, OracleChunk *c, int rx, int ry, int rz)
{
  const size_t n = (size_t)m->dim[0] * m->dim[1] * m->dim[2];
  c->region[0] = (int16_t)rx;
  c->region[1] = (int16_t)ry;
  c->region[2] = (int16_t)rz;
  c->used = 1;
  for (int l = 0; l < ORACLE_LID_COUNT; ++l)
  {
    c->layers[l] = NULL;
    if (m->layers & (1u << l))
    {
      c->layers[l] = calloc(n, k_layer_bytes[l]); /* all layers clear to 0 ... */
    }
  }
  if (c->layers[ORACLE_LID_OCCUPANCY])
  {
    /* ... except occupancy, cleared to +inf == unobserved. ohm/DefaultLayer.cpp:87-91, ohm/VoxelOccupancy.h:42-45 */
    float *occ = (float *)c->layers[ORACLE_LID_OCCUPANCY];
    for (size_t i = 0; i < n; ++i)
    {
      occ[i] = INFINITY;
    }
  }
}

static OracleChunk *table_find(const OracleMap *m, int rx, int ry, int rz)
{
  size_t i = region_hash(rx, ry, rz) & (m->table_cap - 1);
  while (m->table[i].used)
  {
    OracleChunk *c = &m->table[i];
    if (c->region[0] == rx && c->region[1] == ry && c->region[2] == rz)
    {
      return c;
    }
    i = (i + 1) & (m->table_cap - 1);
  }
  return NULL;
}

static void table_grow(OracleMap *m)
{
  OracleChunk *old = m->table;
  const size_t old_cap = m->table_cap;
  m->table_cap *= 2;
  m->table = (OracleChunk *)calloc(m->table_cap, sizeof(OracleChunk));
  for (size_t k = 0; k < old_cap; ++k)
  {
    if (old[k].used)
    {
      size_t i = region_hash(old[k].region[0], old[k].region[1], old[k].region[2]) & (m->table_cap - 1);
      while (m->table[i].used)
      {
        i = (i + 1) & (m->table_cap - 1);
      }
      m->table[i] = old[k];
    }
  }
  free(old);
}

/* OccupancyMap::region(key, allow_create=true) */
static OracleChunk *map_region(OracleMap *m, int rx, int ry, int rz)
{
  OracleChunk *c = table_find(m, rx, ry, rz);
  if (c)
  {
    return c;
  }
  if ((m->table_count + 1) * 2 > m->table_cap)
  {
    table_grow(m);
  }
  size_t i = region_hash(rx, ry, rz) & (m->table_cap - 1);
  while (m->table[i].used)
  {
    i = (i + 1) & (m->table_cap - 1);
  }
  chunk_init(m, &m->table[i], rx, ry, rz);
  ++m->table_count;
  return &m->table[i];
}

size_t oracle_region_count(const OracleMap *m) { return m->table_count; }

size_t oracle_region_keys(const OracleMap *m, int16_t *keys_xyz, size_t cap)
{
  size_t n = 0;
  for (size_t i = 0; i < m->table_cap; ++i)
  {
    if (m->table[i].used)
    {
      if (n < cap)
      {
        keys_xyz[3 * n + 0] = m->table[i].region[0];
        keys_xyz[3 * n + 1] = m->table[i].region[1];
        keys_xyz[3 * n + 2] = m->table[i].region[2];
      }
      ++n;
    }
  }
  return n;
}

/* OccupancyMap::setFirstRayTime (ohm/OccupancyMap.h:346): the touch-time base, set regardless of the current value. */
void oracle_map_set_first_ray_time(OracleMap *m, double time)
{
  m->first_ray_time = time;
}

void *oracle_region_layer(OracleMap *m, int rx, int ry, int rz, int layer_id)
{
  OracleChunk *c = table_find(m, rx, ry, rz);
  if (!c || layer_id < 0 || layer_id >= ORACLE_LID_COUNT)
  {
    return NULL;
  }
  return c->layers[layer_id];
}

/* ------------------------------------------------------------------------------------------------------------- */
/* Key maths                                                                                                      */
/* ------------------------------------------------------------------------------------------------------------- */

/* ohm/MapCoord.h:85-93 (double instantiation): floor(coord / resolution + 0.5) */
int oracle_point_to_region_coord(double coord, double resolution)
{
  return (int)floor(coord / resolution + 0.5);
}

/* ohm/MapCoord.h:45-80 (double instantiation) */
int oracle_point_to_region_voxel(double coord, double voxel_resolution, double region_resolution)
{
  const double epsilon = (double)1e-6f; /* note: float literal widened, as the reference writes it */
  if (-epsilon <= coord && coord < 0)
  {
    coord = 0;
  }
  else if (coord >= region_resolution && coord - epsilon < region_resolution)
  {
    coord -= epsilon;
  }
  return (int)floor(coord / voxel_resolution);
}

/* ohm/OccupancyMap.cpp:859-886 -> ohm/MapRegion.cpp:32-43 (ctor) and :46-69 (voxelKey) */
int oracle_voxel_key(const OracleMap *m, const double p[3], OracleKey *key)
{
  int q[3];
  int16_t coord[3];
  for (int a = 0; a < 3; ++a)
  {
    /* MapRegion ctor: quantise (stored as int16), centre = coord * region_dim (MapCoord.h:32-37) */
    coord[a] = (int16_t)oracle_point_to_region_coord((double)(p[a] - m->origin[a]), m->region_dim[a]);
    const double centre = coord[a] * m->region_dim[a];
    /* MapRegion::voxelKey: region_min = centre - 0.5 * region_dim; p_local = point - origin - region_min */
    const double region_min = centre - 0.5 * m->region_dim[a];
    const double pl = p[a] - m->origin[a] - region_min;
    q[a] = oracle_point_to_region_voxel(pl, m->resolution, m->region_dim[a]);
  }
  if (0 <= q[0] && q[0] < m->dim[0] && 0 <= q[1] && q[1] < m->dim[1] && 0 <= q[2] && q[2] < m->dim[2])
  {
    for (int a = 0; a < 3; ++a)
    {
      key->region[a] = coord[a];
      key->local[a] = (uint8_t)q[a];
    }
    key->pad = 0;
    return 1;
  }
  /* Key::kNull: region all int16 min (ohm/Key.cpp) */
  key->region[0] = key->region[1] = key->region[2] = INT16_MIN;
  key->local[0] = key->local[1] = key->local[2] = 0;
  key->pad = 0;
  return 0;
}

static int key_is_null(const OracleKey *k)
{
  return k->region[0] == INT16_MIN && k->region[1] == INT16_MIN && k->region[2] == INT16_MIN;
}

/* ohm/OccupancyMap.h:757-778 */
static dv3 voxel_centre(const OracleMap *m, const OracleKey *key)
{
  double c[3];
  for (int a = 0; a < 3; ++a)
  {
    double v = (double)(float)key->region[a]; /* centre = glm::vec3(regionKey()) -> float, exact for int16 */
    v *= m->region_dim[a];
    v -= 0.5 * m->region_dim[a];
    v += m->origin[a];
    v += (double)key->local[a] * m->resolution;
    v += 0.5 * m->resolution;
    c[a] = v;
  }
  return dv3_make(c[0], c[1], c[2]);
}

void oracle_voxel_centre(const OracleMap *m, const OracleKey *key, double centre[3])
{
  const dv3 c = voxel_centre(m, key);
  centre[0] = c.x;
  centre[1] = c.y;
  centre[2] = c.z;
}

/* ohm/MapChunk.h:47-50 */
static unsigned voxel_index(const OracleMap *m, const OracleKey *k)
{
  return (unsigned)k->local[0] + (unsigned)k->local[1] * m->dim[0] + (unsigned)k->local[2] * m->dim[0] * m->dim[1];
}

/* ohm/OccupancyMap.h:827-845 */
static void step_key(const OracleMap *m, OracleKey *key, int axis, int dir)
{
  int local_key = key->local[axis] + dir;
  int region_key = key->region[axis];
  if (local_key < 0)
  {
    --region_key;
    local_key = m->dim[axis] - 1;
  }
  else if (local_key >= m->dim[axis])
  {
    ++region_key;
    local_key = 0;
  }
  key->local[axis] = (uint8_t)local_key;
  key->region[axis] = (int16_t)(uint16_t)region_key;
}

/* ohm/OccupancyMap.h:887-901: to - from */
static void range_between(const OracleMap *m, const OracleKey *from, const OracleKey *to, int diff[3])
{
  for (int i = 0; i < 3; ++i)
  {
    const int region_diff = (int)to->region[i] - (int)from->region[i];
    diff[i] = (int)to->local[i] - (int)from->local[i] + region_diff * m->dim[i];
  }
}

static int keys_equal(const OracleKey *a, const OracleKey *b)
{
  return a->region[0] == b->region[0] && a->region[1] == b->region[1] && a->region[2] == b->region[2] &&
         a->local[0] == b->local[0] && a->local[1] == b->local[1] && a->local[2] == b->local[2];
}

/* ------------------------------------------------------------------------------------------------------------- */
/* Line walk: ohm/LineWalkCompute.h (CPU instantiation: WalkReal = double, WalkKey = Key)                         */
/* ------------------------------------------------------------------------------------------------------------- */
typedef int (*VisitFn)(void *ctx, const OracleKey *key, double enter_range, double exit_range);

typedef struct
{
  double time_next[3];
  double initial_delta[3];
  double step_delta[3];
  int sign[3];
  double length;
} WalkSteps;

/* ohm/LineWalkCompute.h:164-171 */
static int walk_step_dir(int sign) { return -2 * sign + 1; }

/* ohm/LineWalkCompute.h:188-248 (walkInitRay) + :260-280 (walkCalculateSteps) */
static void walk_calculate_steps(WalkSteps *ws, const double start[3], const double end[3],
                                 const double start_voxel_centre[3], double res, double length_epsilon)
{
  double dir[3], dir_inv[3], vmin[3], vmax[3], exit0[3], exit1[3];
  for (int a = 0; a < 3; ++a)
  {
    dir[a] = end[a] - start[a];
  }
  double length = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
  length = (length > length_epsilon) ? sqrt(length) : 0;
  for (int a = 0; a < 3; ++a)
  {
    ws->sign[a] = dir[a] < 0;
  }
  for (int a = 0; a < 3; ++a)
  {
    dir[a] /= length;
  }
  for (int a = 0; a < 3; ++a)
  {
    dir_inv[a] = (length > 0) ? 1 / dir[a] : 0;
  }
  for (int a = 0; a < 3; ++a)
  {
    vmin[a] = start_voxel_centre[a] - 0.5 * res;
    vmax[a] = start_voxel_centre[a] + 0.5 * res;
  }
  /* walkCalculateVoxelWallExit: bounds[1 - sign] */
  for (int a = 0; a < 3; ++a)
  {
    const double bound = ws->sign[a] ? vmin[a] : vmax[a];
    exit0[a] = (bound - start[a]) * dir_inv[a];
  }
  for (int a = 0; a < 3; ++a)
  {
    const double shift = walk_step_dir(ws->sign[a]) * res;
    vmin[a] += shift;
    vmax[a] += shift;
  }
  for (int a = 0; a < 3; ++a)
  {
    const double bound = ws->sign[a] ? vmin[a] : vmax[a];
    exit1[a] = (bound - start[a]) * dir_inv[a];
    if (exit1[a] != INFINITY)
    {
      exit1[a] -= exit0[a];
    }
  }
  for (int a = 0; a < 3; ++a)
  {
    ws->initial_delta[a] = ws->time_next[a] = exit0[a];
    ws->step_delta[a] = exit1[a];
  }
  ws->length = length;
}

/* ohm/LineWalkCompute.h:282-289 */
static int walk_select_next_axis(const double *time_next)
{
  int axis = 0;
  axis = (time_next[axis] < time_next[1]) ? axis : 1;
  axis = (time_next[axis] < time_next[2]) ? axis : 2;
  return axis;
}

/* ohm/LineWalkCompute.h:291-307 */
static unsigned walk_step_next(const OracleMap *m, WalkSteps *steps, OracleKey *current, unsigned *axis,
                               int *steps_remaining, int *stepped)
{
  const int step_dir = walk_step_dir(steps->sign[*axis]);
  step_key(m, current, (int)*axis, step_dir);
  steps_remaining[*axis] -= step_dir;
  stepped[*axis] += step_dir;
  steps->time_next[*axis] = (steps_remaining[*axis]) ?
                              steps->initial_delta[*axis] + steps->step_delta[*axis] * abs(stepped[*axis]) :
                              INFINITY;
  const unsigned limit_flags_change = (unsigned)(steps_remaining[*axis] == 0) * (1u << *axis);
  *axis = (unsigned)walk_select_next_axis(steps->time_next);
  return limit_flags_change;
}

/* ohm/LineWalkCompute.h:345-413 */
static unsigned walk_line_voxels(const OracleMap *m, VisitFn visit, void *ctx, const double start[3],
                                 const double end[3], const OracleKey *start_key, const OracleKey *end_key,
                                 const double start_voxel_centre[3], unsigned flags, double length_epsilon)
{
  WalkSteps steps;
  walk_calculate_steps(&steps, start, end, start_voxel_centre, m->resolution, length_epsilon);

  int steps_remaining[3] = { 0, 0, 0 };
  int stepped[3] = { 0, 0, 0 };
  range_between(m, start_key, end_key, steps_remaining); /* walkKeyDiff: end - start (ohm/LineWalk.h:63-70) */

  OracleKey current = *start_key;
  double last_time = 0;
  unsigned axis = 0;
  unsigned voxel_count = 0;
  unsigned limit_flags = 0;
  int continue_traversal = 1;

  limit_flags |= (unsigned)(steps_remaining[0] == 0) * (1u << 0u);
  limit_flags |= (unsigned)(steps_remaining[1] == 0) * (1u << 1u);
  limit_flags |= (unsigned)(steps_remaining[2] == 0) * (1u << 2u);

  for (int i = 0; i < 3; ++i)
  {
    steps.time_next[i] = (steps_remaining[i]) ? steps.initial_delta[i] : INFINITY;
  }
  axis = (unsigned)walk_select_next_axis(steps.time_next);

  if (flags & ORACLE_WALK_EXCLUDE_START)
  {
    last_time = steps.time_next[axis];
    ++voxel_count;
    limit_flags |= walk_step_next(m, &steps, &current, &axis, steps_remaining, stepped);
  }

  while (continue_traversal && limit_flags < 7u && !keys_equal(&current, end_key))
  {
    continue_traversal = visit(ctx, &current, last_time, steps.time_next[axis]);
    last_time = steps.time_next[axis];
    ++voxel_count;
    limit_flags |= walk_step_next(m, &steps, &current, &axis, steps_remaining, stepped);
  }

  if (continue_traversal && (flags & ORACLE_WALK_EXCLUDE_END) == 0u)
  {
    visit(ctx, end_key, last_time, steps.length);
    ++voxel_count;
  }
  return voxel_count;
}

/* ohm/LineWalk.h:112-129 */
static unsigned walk_segment_keys(const OracleMap *m, VisitFn visit, void *ctx, const double start[3],
                                  const double end[3], unsigned flags)
{
  OracleKey start_key, end_key;
  oracle_voxel_key(m, start, &start_key);
  oracle_voxel_key(m, end, &end_key);
  if (key_is_null(&start_key) || key_is_null(&end_key))
  {
    return 0;
  }
  double centre[3];
  oracle_voxel_centre(m, &start_key, centre);
  return walk_line_voxels(m, visit, ctx, start, end, &start_key, &end_key, centre, flags, 1e-6);
}

typedef struct
{
  OracleKey *keys;
  double *enter;
  double *exit;
  size_t cap;
  size_t count;
} CollectCtx;

static int collect_visit(void *vctx, const OracleKey *key, double enter_range, double exit_range)
{
  CollectCtx *c = (CollectCtx *)vctx;
  if (c->count < c->cap)
  {
    if (c->keys)
    {
      c->keys[c->count] = *key;
    }
    if (c->enter)
    {
      c->enter[c->count] = enter_range;
    }
    if (c->exit)
    {
      c->exit[c->count] = exit_range;
    }
  }
  ++c->count;
  return 1;
}

size_t oracle_walk_segment_keys(const OracleMap *m, const double start[3], const double end[3], unsigned walk_flags,
                                OracleKey *keys, double *enter, double *exit, size_t cap)
{
  CollectCtx c;
  c.keys = keys;
  c.enter = enter;
  c.exit = exit;
  c.cap = cap;
  c.count = 0;
  walk_segment_keys(m, collect_visit, &c, start, end, walk_flags);
  return c.count;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* Occupancy arithmetic: ohm/VoxelOccupancyCompute.h                                                              */
/* ------------------------------------------------------------------------------------------------------------- */

/* :44-54 */
void oracle_occupancy_adjust_hit(float *occupancy, float initial_value, float hit_adjustment, float uninit,
                                 float max_value, float sat_min, float sat_max, int null_update)
{
  const int uninitialised = initial_value == uninit;
  const float base_value = (null_update || !uninitialised) ? initial_value : 0.0f;
  hit_adjustment =
    (!null_update && (uninitialised || (sat_min < initial_value && initial_value < sat_max))) ? hit_adjustment : 0.0f;
  *occupancy = (base_value != uninit) ? (float)fmin(base_value + hit_adjustment, max_value) : base_value;
}

/* :78-87 */
void oracle_occupancy_adjust_up(float *occupancy, float initial_value, float adjusted_value, float uninit,
                                float max_value, float sat_min, float sat_max, int null_update)
{
  const int uninitialised = initial_value == uninit;
  adjusted_value = (!null_update && (uninitialised || (sat_min < initial_value && initial_value < sat_max))) ?
                     adjusted_value :
                     initial_value;
  *occupancy = (adjusted_value != uninit) ? (float)fmin(max_value, adjusted_value) : adjusted_value;
}

/* :110-120 */
void oracle_occupancy_adjust_miss(float *occupancy, float initial_value, float miss_adjustment, float uninit,
                                  float min_value, float sat_min, float sat_max, int null_update)
{
  const int uninitialised = initial_value == uninit;
  const float base_value = (null_update || !uninitialised) ? initial_value : 0.0f;
  miss_adjustment =
    (!null_update && (uninitialised || (sat_min < initial_value && initial_value < sat_max))) ? miss_adjustment : 0.0f;
  *occupancy = (base_value != uninit) ? (float)fmax(min_value, base_value + miss_adjustment) : base_value;
}

/* :144-153 */
void oracle_occupancy_adjust_down(float *occupancy, float initial_value, float adjusted_value, float uninit,
                                  float min_value, float sat_min, float sat_max, int null_update)
{
  const int uninitialised = initial_value == uninit;
  adjusted_value = (!null_update && (uninitialised || (sat_min < initial_value && initial_value < sat_max))) ?
                     adjusted_value :
                     initial_value;
  *occupancy = (adjusted_value != uninit) ? (float)fmax(min_value, adjusted_value) : adjusted_value;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* Voxel mean: ohm/VoxelMeanCompute.h (Vec3 = dvec3, coord_real = double as instantiated by the CPU mappers)      */
/* ------------------------------------------------------------------------------------------------------------- */

/* :69-92 */
unsigned oracle_sub_voxel_coord(const double v[3], double resolution)
{
  const int mean_positions = (1 << 10) - 1;
  const unsigned used_bit = (1u << 31u);
  const double mean_resolution = resolution / (double)mean_positions;
  const double offset = (double)0.5f * resolution;
  int pos[3];
  for (int a = 0; a < 3; ++a)
  {
    pos[a] = oracle_point_to_region_coord(v[a] + offset, mean_resolution);
    pos[a] = (pos[a] >= 0 ? (pos[a] < (1 << 10) ? pos[a] : mean_positions) : 0);
  }
  unsigned pattern = 0;
  pattern |= (unsigned)pos[0];
  pattern |= ((unsigned)pos[1] << 10);
  pattern |= ((unsigned)pos[2] << 20);
  pattern |= used_bit;
  return pattern;
}

/* :102-122. Note the reference tests the constant `used_bit`, not `pattern & used_bit`: always decodes. */
void oracle_sub_voxel_to_local(unsigned pattern, double resolution, double out[3])
{
  const int mean_positions = (1 << 10) - 1;
  const double mean_resolution = resolution / (double)mean_positions;
  const double offset = (double)0.5f * resolution;
  out[0] = (int)(pattern & mean_positions) * mean_resolution - offset;
  out[1] = (int)((pattern >> 10) & mean_positions) * mean_resolution - offset;
  out[2] = (int)((pattern >> 20) & mean_positions) * mean_resolution - offset;
}

/* :134-152 */
unsigned oracle_sub_voxel_update(unsigned coord, unsigned point_count, const double v[3], double resolution)
{
  double mean[3];
  oracle_sub_voxel_to_local(coord, resolution, mean);
  const double one_on_count_plus_one = (double)1 / (double)(point_count + 1);
  mean[0] += (v[0] - mean[0]) * one_on_count_plus_one;
  mean[1] += (v[1] - mean[1]) * one_on_count_plus_one;
  mean[2] += (v[2] - mean[2]) * one_on_count_plus_one;
  return oracle_sub_voxel_coord(mean, resolution);
}

/* ------------------------------------------------------------------------------------------------------------- */
/* Incident normal: ohm/VoxelIncidentCompute.h (float maths)                                                      */
/* ------------------------------------------------------------------------------------------------------------- */
static float f_max(float a, float b) { return (a < b) ? b : a; } /* std::max */
static float f_min(float a, float b) { return (b < a) ? b : a; } /* std::min */

/* :35-55 */
static void decode_normal(unsigned packed, float n[3])
{
  n[0] = (2.0f * (((packed >> 0) & 0x3FFF) / 16383.0f)) - 1.0f;
  n[1] = (2.0f * (((packed >> 15) & 0x3FFF) / 16383.0f)) - 1.0f;
  n[0] = f_max(-1.0f, f_min(n[0], 1.0f));
  n[1] = f_max(-1.0f, f_min(n[1], 1.0f));
  n[2] = f_max(-1.0f, f_min(1.0f - (n[0] * n[0] + n[1] * n[1]), 1.0f));
  n[0] = (packed & (1u << 30)) ? n[0] : 0.0f;
  n[1] = (packed & (1u << 30)) ? n[1] : 0.0f;
  n[2] = (packed & (1u << 30)) ? sqrtf(n[2]) : 0.0f;
  n[2] *= (packed & (1u << 31)) ? -1.0f : 1.0f;
}

/* :57-80 */
static unsigned encode_normal(const float normal_in[3])
{
  float normal[3] = { normal_in[0], normal_in[1], normal_in[2] };
  unsigned n = 0;
  normal[0] = 0.5f * (f_max(-1.0f, f_min(normal[0], 1.0f)) + 1.0f);
  normal[1] = 0.5f * (f_max(-1.0f, f_min(normal[1], 1.0f)) + 1.0f);
  unsigned i = (unsigned)(normal[0] * 16383.0f);
  n |= (i & 0x3FFF) << 0;
  i = (unsigned)(normal[1] * 16383.0f);
  n |= (i & 0x3FFF) << 15;
  n &= ~((1u << 30) | (1u << 31));
  n |= (normal[2] < 0) ? (1u << 31) : 0;
  n |= (normal[0] || normal[1] || normal[2]) ? (1u << 30) : 0;
  return n;
}

/* :82-112 */
static unsigned update_incident_normal(unsigned packed, const float ray_in[3], unsigned point_count)
{
  float normal[3];
  float ray[3] = { ray_in[0], ray_in[1], ray_in[2] };
  decode_normal(packed, normal);
  point_count = ((normal[0] != 0 || normal[1] != 0 || normal[2] != 0) && point_count) ? point_count : 0;
  const float one_on_count_plus_one = 1.0f / (float)(point_count + 1);
  float len2 = ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2];
  float s = (len2 > 1e-6f) ? 1.0f / sqrtf(len2) : 0.0f;
  ray[0] *= s;
  ray[1] *= s;
  ray[2] *= s;
  normal[0] += (ray[0] - normal[0]) * one_on_count_plus_one;
  normal[1] += (ray[1] - normal[1]) * one_on_count_plus_one;
  normal[2] += (ray[2] - normal[2]) * one_on_count_plus_one;
  len2 = normal[0] * normal[0] + normal[1] * normal[1] + normal[2] * normal[2];
  s = (len2 > 1e-6f) ? 1.0f / sqrtf(len2) : 0.0f;
  normal[0] *= s;
  normal[1] *= s;
  normal[2] *= s;
  return encode_normal(normal);
}

/* ohm/VoxelTouchTimeCompute.h:24-27 */
static unsigned encode_touch_time(double timebase, double timestamp)
{
  return (unsigned)((timestamp - timebase) / 0.001);
}

/* ------------------------------------------------------------------------------------------------------------- */
/* Ray filters: ohm/RayFilter.cpp:12-58                                                                           */
/* ------------------------------------------------------------------------------------------------------------- */
enum
{
  RFF_INVALID = 1u << 0,
  RFF_CLIPPED_START = 1u << 1,
  RFF_CLIPPED_END = 1u << 2
};

static int all_finite(const double v[3]) { return isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2]); }

static int apply_filter(const OracleMap *m, double start[3], double end[3], unsigned *filter_flags, size_t ray_index)
{
  if (m->batch_filter_flags)
  {
    /* the caller's filter already accepted (and possibly moved) the ray; its flags are consumed as the mappers do:
     * ohm/RayMapperOccupancy.cpp:209-223 */
    *filter_flags |= m->batch_filter_flags[ray_index];
    return 1;
  }
  if (m->filter_mode == ORACLE_FILTER_NONE)
  {
    return 1;
  }
  int good = all_finite(start) && all_finite(end);
  const dv3 ray = dv3_make(end[0] - start[0], end[1] - start[1], end[2] - start[2]);
  const double len2 = dv3_dot(ray, ray);
  if (m->filter_mode == ORACLE_FILTER_GOOD)
  {
    good = good && (m->filter_range <= 0 || len2 <= m->filter_range * m->filter_range);
    if (!good)
    {
      *filter_flags |= RFF_INVALID;
    }
    return good;
  }
  /* clipRayFilter */
  if (good && m->filter_range > 0 && len2 > m->filter_range * m->filter_range)
  {
    const double len = sqrt(len2);
    const dv3 dir = dv3_make(ray.x / len, ray.y / len, ray.z / len);
    end[0] = start[0] + dir.x * m->filter_range;
    end[1] = start[1] + dir.y * m->filter_range;
    end[2] = start[2] + dir.z * m->filter_range;
    *filter_flags |= RFF_CLIPPED_END;
  }
  *filter_flags |= (unsigned)(!good) * RFF_INVALID;
  return good;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* RayMapperOccupancy::integrateRays -- ohm/RayMapperOccupancy.cpp:68-339                                         */
/* ------------------------------------------------------------------------------------------------------------- */
typedef struct
{
  OracleMap *map;
  unsigned ray_flags;
  OracleChunk *last_chunk;
  double last_exit_range;
  int stop_adjustments;
  float sat_min, sat_max;
} OccCtx;

/* :105-193 */
static int occ_visit(void *vctx, const OracleKey *key, double enter_range, double exit_range)
{
  OccCtx *c = (OccCtx *)vctx;
  OracleMap *m = c->map;
  OracleChunk *chunk = (c->last_chunk && key->region[0] == c->last_chunk->region[0] &&
                        key->region[1] == c->last_chunk->region[1] && key->region[2] == c->last_chunk->region[2]) ?
                         c->last_chunk :
                         map_region(m, key->region[0], key->region[1], key->region[2]);
  c->last_chunk = chunk;
  const unsigned vi = voxel_index(m, key);
  float *occ = (float *)chunk->layers[ORACLE_LID_OCCUPANCY];
  float occupancy_value = occ[vi];
  const float initial_value = occupancy_value;
  const int initially_unobserved = initial_value == INFINITY;
  const int initially_free = !initially_unobserved && initial_value < m->threshold_value;
  const int initially_occupied = !initially_unobserved && initial_value >= m->threshold_value;

  float miss_adjustment = m->miss_value;
  miss_adjustment = (initially_unobserved && (c->ray_flags & ORACLE_RF_EXCLUDE_UNOBSERVED)) ? INFINITY : miss_adjustment;
  miss_adjustment = (initially_free && (c->ray_flags & ORACLE_RF_EXCLUDE_FREE)) ? 0.0f : miss_adjustment;
  miss_adjustment = (initially_occupied && (c->ray_flags & ORACLE_RF_EXCLUDE_OCCUPIED)) ? 0.0f : miss_adjustment;

  oracle_occupancy_adjust_miss(&occupancy_value, initial_value, miss_adjustment, INFINITY, m->min_value, c->sat_min,
                               c->sat_max, c->stop_adjustments);
  occ[vi] = occupancy_value;

  if (chunk->layers[ORACLE_LID_TRAVERSAL])
  {
    float *trav = (float *)chunk->layers[ORACLE_LID_TRAVERSAL];
    trav[vi] += (float)(exit_range - enter_range);
  }

  c->stop_adjustments =
    c->stop_adjustments || ((c->ray_flags & ORACLE_RF_STOP_ON_FIRST_OCCUPIED) && initially_occupied);
  c->last_exit_range = exit_range;
  ++m->visits;
  return 1;
}

size_t oracle_integrate_occupancy(OracleMap *m, const double *rays, size_t element_count, const double *timestamps,
                                  unsigned ray_flags)
{
  OccCtx ctx;
  memset(&ctx, 0, sizeof(ctx));
  ctx.map = m;
  ctx.ray_flags = ray_flags;
  ctx.sat_min = m->saturate_min ? m->min_value : -3.402823466e+38f; /* numeric_limits<float>::lowest() */
  ctx.sat_max = m->saturate_max ? m->max_value : 3.402823466e+38f;  /* numeric_limits<float>::max() */

  if (timestamps)
  {
    m->first_ray_time = (m->first_ray_time < 0) ? *timestamps : m->first_ray_time; /* OccupancyMap.cpp:343-347 */
  }
  const double time_base = m->first_ray_time;

  for (size_t i = 0; i < element_count; i += 2)
  {
    unsigned filter_flags = 0;
    double start[3] = { rays[3 * i + 0], rays[3 * i + 1], rays[3 * i + 2] };
    double end[3] = { rays[3 * i + 3], rays[3 * i + 4], rays[3 * i + 5] };
    if (!apply_filter(m, start, end, &filter_flags, i >> 1))
    {
      continue;
    }

    const int include_sample_in_ray = (filter_flags & RFF_CLIPPED_END) || (ray_flags & ORACLE_RF_END_POINT_AS_FREE);
    unsigned walk_flags = (!include_sample_in_ray) ? ORACLE_WALK_EXCLUDE_END : 0u;
    walk_flags |= (ray_flags & ORACLE_RF_EXCLUDE_ORIGIN) ? ORACLE_WALK_EXCLUDE_START : 0u;

    if (!(ray_flags & ORACLE_RF_EXCLUDE_RAY))
    {
      ctx.stop_adjustments = 0;
      walk_segment_keys(m, occ_visit, &ctx, start, end, walk_flags);
    }

    if (!ctx.stop_adjustments && !include_sample_in_ray && !(ray_flags & ORACLE_RF_EXCLUDE_SAMPLE))
    {
      OracleKey key;
      oracle_voxel_key(m, end, &key);
      OracleChunk *chunk =
        (ctx.last_chunk && key.region[0] == ctx.last_chunk->region[0] && key.region[1] == ctx.last_chunk->region[1] &&
         key.region[2] == ctx.last_chunk->region[2]) ?
          ctx.last_chunk :
          map_region(m, key.region[0], key.region[1], key.region[2]);
      ctx.last_chunk = chunk;
      const unsigned vi = voxel_index(m, &key);
      float *occ = (float *)chunk->layers[ORACLE_LID_OCCUPANCY];
      float occupancy_value = occ[vi];
      const float initial_value = occupancy_value;
      const int initially_unobserved = initial_value == INFINITY;
      const int initially_free = !initially_unobserved && initial_value < m->threshold_value;
      const int initially_occupied = !initially_unobserved && initial_value >= m->threshold_value;

      float hit_adjustment = m->hit_value;
      hit_adjustment = (initially_unobserved && (ray_flags & ORACLE_RF_EXCLUDE_UNOBSERVED)) ? INFINITY : hit_adjustment;
      hit_adjustment = (initially_free && (ray_flags & ORACLE_RF_EXCLUDE_FREE)) ? 0.0f : hit_adjustment;
      hit_adjustment = (initially_occupied && (ray_flags & ORACLE_RF_EXCLUDE_OCCUPIED)) ? 0.0f : hit_adjustment;

      oracle_occupancy_adjust_hit(&occupancy_value, initial_value, hit_adjustment, INFINITY, m->max_value,
                                  ctx.sat_min, ctx.sat_max, ctx.stop_adjustments);

      unsigned sample_count = 0;
      if (chunk->layers[ORACLE_LID_MEAN])
      {
        uint32_t *mean = (uint32_t *)chunk->layers[ORACLE_LID_MEAN] + 2 * (size_t)vi;
        const dv3 centre = voxel_centre(m, &key);
        const double local[3] = { end[0] - centre.x, end[1] - centre.y, end[2] - centre.z };
        mean[0] = oracle_sub_voxel_update(mean[0], mean[1], local, m->resolution);
        sample_count = mean[1];
        ++mean[1];
      }
      occ[vi] = occupancy_value;

      if (chunk->layers[ORACLE_LID_TRAVERSAL])
      {
        float *trav = (float *)chunk->layers[ORACLE_LID_TRAVERSAL];
        const dv3 d = dv3_make(end[0] - start[0], end[1] - start[1], end[2] - start[2]);
        trav[vi] += (float)(sqrt(dv3_dot(d, d)) - ctx.last_exit_range);
      }
      if (chunk->layers[ORACLE_LID_TOUCH_TIME] && timestamps)
      {
        ((uint32_t *)chunk->layers[ORACLE_LID_TOUCH_TIME])[vi] = encode_touch_time(time_base, timestamps[i >> 1]);
      }
      if (chunk->layers[ORACLE_LID_INCIDENT])
      {
        uint32_t *inc = (uint32_t *)chunk->layers[ORACLE_LID_INCIDENT];
        const float ray[3] = { (float)(start[0] - end[0]), (float)(start[1] - end[1]), (float)(start[2] - end[2]) };
        inc[vi] = update_incident_normal(inc[vi], ray, sample_count);
      }
      ++m->visits;
    }
  }
  return element_count / 2;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* NDT: ohm/CovarianceVoxelCompute.h (CovReal = double, CovVec3 = dvec3)                                          */
/* ------------------------------------------------------------------------------------------------------------- */

/* :90-98 */
static void initialise_covariance(float cov[6], float voxel_resolution)
{
  const float covariance_scale_factor = 0.1f;
  cov[0] = cov[2] = cov[5] = covariance_scale_factor * voxel_resolution;
  cov[1] = cov[3] = cov[4] = 0;
}

/* :107-120 */
static double packed_dot(const double A[9], int j, int k)
{
  const int col_first_el[] = { 0, 1, 3 };
  const int indj = col_first_el[j];
  const int indk = col_first_el[k];
  const int mm = (j <= k) ? j : k;
  double d = A[6 + k] * A[6 + j];
  for (int i = 0; i <= mm; ++i)
  {
    d += A[indj + i] * A[indk + i];
  }
  return d;
}

/* :152-170 */
static void unpack_covariance(const float cov[6], unsigned point_count, dv3 sample_to_mean, double *matrix)
{
  const double one_on_num_pt_plus_one = (double)1 / (point_count + (double)1);
  const double sc_1 = point_count ? sqrt(point_count * one_on_num_pt_plus_one) : (double)1;
  const double sc_2 = one_on_num_pt_plus_one * sqrt((double)point_count);
  for (int i = 0; i < 6; ++i)
  {
    matrix[i] = sc_1 * cov[i];
  }
  matrix[0 + 6] = sc_2 * sample_to_mean.x;
  matrix[1 + 6] = sc_2 * sample_to_mean.y;
  matrix[2 + 6] = sc_2 * sample_to_mean.z;
}

/* :183-204 */
static dv3 solve_triangular(const float cov[6], dv3 y)
{
  dv3 x;
  double d;
  d = y.x;
  x.x = d / cov[0];
  d = y.y;
  d -= cov[1 + 0] * x.x;
  x.y = d / cov[1 + 1];
  d = y.z;
  d -= cov[3 + 0] * x.x;
  d -= cov[3 + 1] * x.y;
  x.z = d / cov[3 + 2];
  return x;
}

/* :227-267 */
static dv3 calculate_sample_likelihoods(const float cov[6], dv3 sensor, dv3 sample, dv3 voxel_mean, float sensor_noise,
                                        double *p_x_ml_given_voxel, double *p_x_ml_given_sample)
{
  const double k_half = 0.5;
  const dv3 sensor_to_sample = dv3_sub(sample, sensor);
  const dv3 sensor_ray = dv3_normalize(sensor_to_sample);
  const dv3 mean_to_sensor = dv3_sub(sensor, voxel_mean);
  const dv3 a = solve_triangular(cov, sensor_ray);
  const dv3 b_norm = solve_triangular(cov, mean_to_sensor);
  const double t = -dv3_dot(a, b_norm) / dv3_dot(a, a);
  const dv3 voxel_ml = dv3_add(dv3_scale(sensor_ray, t), sensor);
  const dv3 s1 = solve_triangular(cov, dv3_sub(voxel_ml, voxel_mean));
  *p_x_ml_given_voxel = exp(-k_half * dv3_dot(s1, s1));
  const double sensor_noise_variance = sensor_noise * sensor_noise; /* float*float then widened */
  const dv3 d2 = dv3_sub(voxel_ml, sample);
  *p_x_ml_given_sample = exp(-k_half * dv3_dot(d2, d2) / sensor_noise_variance);
  return voxel_ml;
}

/* :301-375 */
int oracle_calculate_hit_with_covariance(float cov[6], float *voxel_value, const double sample_a[3],
                                         const double mean_a[3], unsigned point_count, float hit_value, float uninit,
                                         float voxel_resolution, float reinit_threshold, unsigned reinit_count)
{
  const dv3 sample = dv3_make(sample_a[0], sample_a[1], sample_a[2]);
  const dv3 voxel_mean = dv3_make(mean_a[0], mean_a[1], mean_a[2]);
  const float initial_value = *voxel_value;
  const int was_uncertain = initial_value == uninit;
  int initialised_covariance = 0;

  if (point_count == 0 || (initial_value < reinit_threshold && point_count >= reinit_count))
  {
    initialise_covariance(cov, voxel_resolution);
    initialised_covariance = 1;
    point_count = 0;
  }

  *voxel_value = (!was_uncertain) ? hit_value + initial_value : hit_value;

  const dv3 sample_to_mean = (!initialised_covariance) ? dv3_sub(sample, voxel_mean) : dv3_make(0, 0, 0);
  double A[9];
  unpack_covariance(cov, point_count, sample_to_mean, A);

  for (int k = 0; k < 3; ++k)
  {
    const int ind1 = (k * (k + 3)) >> 1;
    const int indk = ind1 - k;
    const double ak = sqrt(packed_dot(A, k, k));
    cov[ind1] = (float)ak;
    if (ak > 0)
    {
      const double aki = (double)1 / ak;
      for (int j = k + 1; j < 3; ++j)
      {
        const int indj = (j * (j + 1)) >> 1;
        const int indkj = indj + k;
        double c = packed_dot(A, j, k) * aki;
        cov[indkj] = (float)c;
        c *= aki;
        A[j + 6] -= c * A[k + 6];
        for (int l = 0; l <= k; ++l)
        {
          A[indj + l] -= c * A[indk + l];
        }
      }
    }
  }
  return initialised_covariance;
}

/* :391-411 */
static void calculate_intensity_update_on_hit(float intensity[2], float voxel_value, float intensity_sample,
                                              float initial_intensity_covariance, unsigned point_count,
                                              float reinit_threshold, unsigned reinit_count)
{
  const float initial_value = voxel_value;
  const int needs_reset = point_count == 0 || (initial_value < reinit_threshold && point_count >= reinit_count);
  const float delta = intensity[0] - intensity_sample;
  const float point_count_float = (float)point_count;
  const float inv = 1.0f / (point_count_float + 1.0f);
  const float new_mean = (!needs_reset) ? inv * (point_count_float * intensity[0] + intensity_sample) : intensity_sample;
  const float new_cov =
    (!needs_reset) ? inv * (point_count_float * intensity[1] + inv * delta * delta) : initial_intensity_covariance;
  intensity[0] = new_mean;
  intensity[1] = new_cov;
}

/* :447-505 */
static void calculate_hit_miss_update_on_hit(const float cov[6], float voxel_value, uint32_t hit_miss[2], dv3 sensor,
                                             dv3 sample, dv3 voxel_mean, unsigned point_count, float uninit,
                                             int reinit_with_cov, float adaptation_rate, float sensor_noise,
                                             float reinit_threshold, unsigned reinit_count, unsigned sample_threshold)
{
  const double k_half = 0.5;
  const int needs_reset =
    voxel_value == uninit ||
    (reinit_with_cov && (point_count == 0 || (voxel_value < reinit_threshold && point_count >= reinit_count)));
  const unsigned initial_hit = (!needs_reset) ? hit_miss[0] : 0;
  const unsigned initial_miss = (!needs_reset) ? hit_miss[1] : 0;
  double p_voxel, p_sample;
  calculate_sample_likelihoods(cov, sensor, sample, voxel_mean, sensor_noise, &p_voxel, &p_sample);
  const double prod = p_voxel * p_sample;
  const double eta = k_half * adaptation_rate;
  const int inc_hit = needs_reset || point_count < sample_threshold || (point_count >= sample_threshold && prod >= eta);
  const int inc_miss = !needs_reset && point_count >= sample_threshold && prod < eta && p_voxel >= eta;
  hit_miss[0] = initial_hit + (inc_hit ? 1 : 0);
  hit_miss[1] = initial_miss + (inc_miss ? 1 : 0);
}

/* :542-635 */
void oracle_calculate_miss_ndt(const float cov[6], float *voxel_value, int *is_miss, const double sensor_a[3],
                               const double sample_a[3], const double mean_a[3], unsigned point_count, float uninit,
                               float miss_value, float adaptation_rate, float sensor_noise, unsigned sample_threshold)
{
  const double k_one = 1.0;
  const double k_half = 0.5;
  if (*voxel_value == uninit)
  {
    *voxel_value = miss_value;
    *is_miss = 1;
    return;
  }
  if (point_count < sample_threshold)
  {
    *voxel_value += miss_value;
    *is_miss = 1;
    return;
  }
  const dv3 sensor = dv3_make(sensor_a[0], sensor_a[1], sensor_a[2]);
  const dv3 sample = dv3_make(sample_a[0], sample_a[1], sample_a[2]);
  const dv3 voxel_mean = dv3_make(mean_a[0], mean_a[1], mean_a[2]);
  double p_voxel, p_sample;
  calculate_sample_likelihoods(cov, sensor, sample, voxel_mean, sensor_noise, &p_voxel, &p_sample);
  const double scaling_factor = k_half * adaptation_rate;
  const double prod = p_voxel * (k_one - p_sample);
  const double probability_update = k_half - scaling_factor * prod;
  *is_miss = prod < scaling_factor;
  if (probability_update == probability_update)
  {
    *voxel_value += (float)log(probability_update / (k_one - probability_update));
  }
}

/* RayMapperNdt::integrateRays -- ohm/RayMapperNdt.cpp:84-407 */
typedef struct
{
  OracleMap *map;
  OracleChunk *last_chunk;
  double start[3];
  double sample[3];
  double last_exit_range;
  int stop_adjustments;
  float sat_min, sat_max;
} NdtCtx;

static OracleChunk *ctx_chunk(OracleMap *m, OracleChunk **last, const OracleKey *key)
{
  OracleChunk *chunk = (*last && key->region[0] == (*last)->region[0] && key->region[1] == (*last)->region[1] &&
                        key->region[2] == (*last)->region[2]) ?
                         *last :
                         map_region(m, key->region[0], key->region[1], key->region[2]);
  *last = chunk;
  return chunk;
}

/* :135-230 */
static int ndt_visit(void *vctx, const OracleKey *key, double enter_range, double exit_range)
{
  NdtCtx *c = (NdtCtx *)vctx;
  OracleMap *m = c->map;
  OracleChunk *chunk = ctx_chunk(m, &c->last_chunk, key);
  const unsigned vi = voxel_index(m, key);
  float *occ = (float *)chunk->layers[ORACLE_LID_OCCUPANCY];
  const float *cov = (const float *)chunk->layers[ORACLE_LID_COVARIANCE] + 6 * (size_t)vi;
  const uint32_t *vmean = (const uint32_t *)chunk->layers[ORACLE_LID_MEAN] + 2 * (size_t)vi;
  float occupancy_value = occ[vi];
  double local[3];
  oracle_sub_voxel_to_local(vmean[0], m->resolution, local);
  const dv3 centre = voxel_centre(m, key);
  const double mean[3] = { local[0] + centre.x, local[1] + centre.y, local[2] + centre.z };
  const float initial_value = occupancy_value;
  float adjusted_value = initial_value;
  int is_miss = 0;
  oracle_calculate_miss_ndt(cov, &adjusted_value, &is_miss, c->start, c->sample, mean, vmean[1], INFINITY,
                            m->miss_value, m->adaptation_rate, m->sensor_noise, m->sample_threshold);
  if (m->ndt_tm && chunk->layers[ORACLE_LID_HIT_MISS])
  {
    uint32_t *hm = (uint32_t *)chunk->layers[ORACLE_LID_HIT_MISS] + 2 * (size_t)vi;
    hm[1] += is_miss ? 1u : 0u;
  }
  oracle_occupancy_adjust_down(&occupancy_value, initial_value, adjusted_value, INFINITY, m->min_value, c->sat_min,
                               c->sat_max, c->stop_adjustments);
  occ[vi] = occupancy_value;
  if (chunk->layers[ORACLE_LID_TRAVERSAL])
  {
    float *trav = (float *)chunk->layers[ORACLE_LID_TRAVERSAL];
    trav[vi] += (float)(exit_range - enter_range);
  }
  c->last_exit_range = exit_range;
  ++m->visits;
  return 1;
}

size_t oracle_integrate_ndt(OracleMap *m, const double *rays, size_t element_count, const float *intensities,
                            const double *timestamps, unsigned ray_flags)
{
  NdtCtx ctx;
  memset(&ctx, 0, sizeof(ctx));
  ctx.map = m;
  ctx.sat_min = m->saturate_min ? m->min_value : -3.402823466e+38f;
  ctx.sat_max = m->saturate_max ? m->max_value : 3.402823466e+38f;
  float intensity = 0.0f;
  if (timestamps)
  {
    m->first_ray_time = (m->first_ray_time < 0) ? *timestamps : m->first_ray_time;
  }
  const double time_base = m->first_ray_time;

  for (size_t i = 0; i < element_count; i += 2)
  {
    unsigned filter_flags = 0;
    for (int a = 0; a < 3; ++a)
    {
      ctx.start[a] = rays[3 * i + a];
      ctx.sample[a] = rays[3 * i + 3 + a];
    }
    if (intensities)
    {
      intensity = intensities[i >> 1];
    }
    if (!apply_filter(m, ctx.start, ctx.sample, &filter_flags, i >> 1))
    {
      continue;
    }
    const int include_sample_in_ray = (filter_flags & RFF_CLIPPED_END) || (ray_flags & ORACLE_RF_END_POINT_AS_FREE);
    unsigned walk_flags = (!include_sample_in_ray) ? ORACLE_WALK_EXCLUDE_END : 0u;
    walk_flags |= (ray_flags & ORACLE_RF_EXCLUDE_ORIGIN) ? ORACLE_WALK_EXCLUDE_START : 0u;

    if (!(ray_flags & ORACLE_RF_EXCLUDE_RAY))
    {
      ctx.stop_adjustments = 0;
      walk_segment_keys(m, ndt_visit, &ctx, ctx.start, ctx.sample, walk_flags);
    }

    if (!ctx.stop_adjustments && !include_sample_in_ray)
    {
      OracleKey key;
      oracle_voxel_key(m, ctx.sample, &key);
      OracleChunk *chunk = ctx_chunk(m, &ctx.last_chunk, &key);
      const unsigned vi = voxel_index(m, &key);
      const dv3 centre = voxel_centre(m, &key);
      float *occ = (float *)chunk->layers[ORACLE_LID_OCCUPANCY];
      float *cov = (float *)chunk->layers[ORACLE_LID_COVARIANCE] + 6 * (size_t)vi;
      uint32_t *vmean = (uint32_t *)chunk->layers[ORACLE_LID_MEAN] + 2 * (size_t)vi;
      float occupancy_value = occ[vi];
      double local[3];
      oracle_sub_voxel_to_local(vmean[0], m->resolution, local);
      const double mean[3] = { local[0] + centre.x, local[1] + centre.y, local[2] + centre.z };
      const float initial_value = occupancy_value;
      float adjusted_value = initial_value;

      if (m->ndt_tm)
      {
        float *iv = (float *)chunk->layers[ORACLE_LID_INTENSITY] + 2 * (size_t)vi;
        uint32_t *hm = (uint32_t *)chunk->layers[ORACLE_LID_HIT_MISS] + 2 * (size_t)vi;
        calculate_hit_miss_update_on_hit(cov, adjusted_value, hm, dv3_make(ctx.start[0], ctx.start[1], ctx.start[2]),
                                         dv3_make(ctx.sample[0], ctx.sample[1], ctx.sample[2]),
                                         dv3_make(mean[0], mean[1], mean[2]), vmean[1], INFINITY, 1,
                                         m->adaptation_rate, m->sensor_noise, m->reinit_threshold, m->reinit_count,
                                         m->sample_threshold);
        calculate_intensity_update_on_hit(iv, adjusted_value, intensity, m->initial_intensity_cov, vmean[1],
                                          m->reinit_threshold, m->reinit_count);
      }

      const int reset_mean = oracle_calculate_hit_with_covariance(cov, &adjusted_value, ctx.sample, mean, vmean[1],
                                                                  m->hit_value, INFINITY, (float)m->resolution,
                                                                  m->reinit_threshold, m->reinit_count);
      oracle_occupancy_adjust_up(&occupancy_value, initial_value, adjusted_value, INFINITY, m->max_value, ctx.sat_min,
                                 ctx.sat_max, ctx.stop_adjustments);
      vmean[1] = (!reset_mean) ? vmean[1] : 0;
      const double sample_local[3] = { ctx.sample[0] - centre.x, ctx.sample[1] - centre.y, ctx.sample[2] - centre.z };
      vmean[0] = oracle_sub_voxel_update(vmean[0], vmean[1], sample_local, m->resolution);
      ++vmean[1];
      occ[vi] = occupancy_value;

      if (chunk->layers[ORACLE_LID_TRAVERSAL])
      {
        float *trav = (float *)chunk->layers[ORACLE_LID_TRAVERSAL];
        const dv3 d =
          dv3_make(ctx.sample[0] - ctx.start[0], ctx.sample[1] - ctx.start[1], ctx.sample[2] - ctx.start[2]);
        trav[vi] += (float)(sqrt(dv3_dot(d, d)) - ctx.last_exit_range);
      }
      if (chunk->layers[ORACLE_LID_TOUCH_TIME] && timestamps)
      {
        ((uint32_t *)chunk->layers[ORACLE_LID_TOUCH_TIME])[vi] = encode_touch_time(time_base, timestamps[i >> 1]);
      }
      if (chunk->layers[ORACLE_LID_INCIDENT])
      {
        uint32_t *inc = (uint32_t *)chunk->layers[ORACLE_LID_INCIDENT];
        const float ray[3] = { (float)(ctx.start[0] - ctx.sample[0]), (float)(ctx.start[1] - ctx.sample[1]),
                               (float)(ctx.start[2] - ctx.sample[2]) };
        inc[vi] = update_incident_normal(inc[vi], ray, vmean[1] - 1);
      }
      ++m->visits;
    }
  }
  return element_count / 2;
}

/* ------------------------------------------------------------------------------------------------------------- */
/* TSDF: ohm/VoxelTsdfCompute.h (Vec3 = dvec3), ohm/RayMapperTsdf.cpp:87-182                                      */
/* ------------------------------------------------------------------------------------------------------------- */

/* :57-68 */
static float compute_distance(dv3 sensor, dv3 sample, dv3 voxel_centre_v)
{
  const dv3 sensor_to_voxel = dv3_sub(voxel_centre_v, sensor);
  const dv3 sensor_to_sample = dv3_sub(sample, sensor);
  const float distance_g = (float)sqrt(dv3_dot(sensor_to_sample, sensor_to_sample));
  const float distance_g_v = (float)dv3_dot(sensor_to_voxel, sensor_to_sample) / distance_g;
  const float sdf = distance_g - distance_g_v;
  return sdf;
}

/* :87-136 */
int oracle_calculate_tsdf(const double sensor_a[3], const double sample_a[3], const double centre_a[3],
                          float default_truncation_distance, float max_weight, float dropoff_epsilon,
                          float sparsity_compensation_factor, float *voxel_weight, float *voxel_distance)
{
  const float sdf = compute_distance(dv3_make(sensor_a[0], sensor_a[1], sensor_a[2]),
                                     dv3_make(sample_a[0], sample_a[1], sample_a[2]),
                                     dv3_make(centre_a[0], centre_a[1], centre_a[2]));
  const float initial_weight = *voxel_weight;
  float updated_weight = 1.0f;
  updated_weight *= (dropoff_epsilon > 0) ?
                      ((default_truncation_distance + sdf) / (default_truncation_distance - dropoff_epsilon)) :
                      1.0f;
  updated_weight = f_max(updated_weight, 0.0f);
  /* fabs(float) resolves to the float overload under `using namespace std` */
  updated_weight *=
    (sparsity_compensation_factor > 0 && fabsf(sdf) < default_truncation_distance) ? sparsity_compensation_factor : 1.0f;
  const float new_weight = initial_weight + updated_weight;
  const float abs_new_weight = fabsf(new_weight);
  const int near_zero_weight = abs_new_weight < 0.00001f;
  const float new_sdf =
    (!near_zero_weight) ? (sdf * updated_weight + *voxel_distance * initial_weight) / new_weight : 0.0f;
  *voxel_distance = (!near_zero_weight) ? ((new_sdf > 0.0f) ? f_min(default_truncation_distance, new_sdf) :
                                                              f_max(-default_truncation_distance, new_sdf)) :
                                          *voxel_distance;
  *voxel_weight = (!near_zero_weight) ? f_min(new_weight, max_weight) : initial_weight;
  return !near_zero_weight;
}

typedef struct
{
  OracleMap *map;
  OracleChunk *last_chunk;
  double sensor[3];
  double sample[3];
} TsdfCtx;

static int tsdf_visit(void *vctx, const OracleKey *key, double enter_range, double exit_range)
{
  (void)enter_range;
  (void)exit_range;
  TsdfCtx *c = (TsdfCtx *)vctx;
  OracleMap *m = c->map;
  OracleChunk *chunk = ctx_chunk(m, &c->last_chunk, key);
  const unsigned vi = voxel_index(m, key);
  float *tsdf = (float *)chunk->layers[ORACLE_LID_TSDF] + 2 * (size_t)vi;
  double centre[3];
  oracle_voxel_centre(m, key, centre);
  oracle_calculate_tsdf(c->sensor, c->sample, centre, m->tsdf_trunc, m->tsdf_max_weight, m->tsdf_dropoff,
                        m->tsdf_sparsity, &tsdf[0], &tsdf[1]);
  ++m->visits;
  return 1;
}

size_t oracle_integrate_tsdf(OracleMap *m, const double *rays, size_t element_count)
{
  TsdfCtx ctx;
  memset(&ctx, 0, sizeof(ctx));
  ctx.map = m;
  for (size_t i = 0; i < element_count; i += 2)
  {
    unsigned filter_flags = 0;
    double ray_start[3], ray_end[3];
    for (int a = 0; a < 3; ++a)
    {
      ray_start[a] = ctx.sensor[a] = rays[3 * i + a];
      ray_end[a] = ctx.sample[a] = rays[3 * i + 3 + a];
    }
    if (!apply_filter(m, ray_start, ray_end, &filter_flags, i >> 1))
    {
      continue;
    }
    walk_segment_keys(m, tsdf_visit, &ctx, ray_start, ray_end, 0u);
  }
  return element_count / 2;
}

/* ------------------------------------------------------------------------------------------------------------------ */
/* GpuTransformSamples (sensor frame -> world frame), fp64 restatement of the reference kernel's arithmetic.            */
/* ------------------------------------------------------------------------------------------------------------------ */
typedef struct
{
  double x, y, z, w;
} OQuat;

/* ohmgpu/gpu/TransformSamples.cl:13-54 */
static OQuat o_slerp(OQuat from, OQuat to, double f)
{
  if (from.x == to.x && from.y == to.y && from.z == to.z && from.w == to.w)
  {
    return from;
  }
  double cos_angle = ((from.x * to.x + from.y * to.y) + from.z * to.z) + from.w * to.w;
  OQuat temp = to;
  if (!(cos_angle >= 0))
  {
    temp.x = -1.0 * to.x;
    temp.y = -1.0 * to.y;
    temp.z = -1.0 * to.z;
    temp.w = -1.0 * to.w;
    cos_angle = -1.0 * cos_angle;
  }
  double coeff0, coeff1;
  if (1.0 - cos_angle > 1e-12)
  {
    const double angle = acos(cos_angle);
    const double inv_sin = 1.0 / sin(angle);
    coeff0 = sin((1.0 - f) * angle) * inv_sin;
    coeff1 = sin(f * angle) * inv_sin;
  }
  else
  {
    coeff0 = 1.0 - f;
    coeff1 = f;
  }
  OQuat r;
  r.x = coeff0 * from.x + coeff1 * temp.x;
  r.y = coeff0 * from.y + coeff1 * temp.y;
  r.z = coeff0 * from.z + coeff1 * temp.z;
  r.w = coeff0 * from.w + coeff1 * temp.w;
  return r;
}

/* ohmgpu/gpu/TransformSamples.cl:57-65 */
static OQuat o_quat_mul(OQuat a, OQuat b)
{
  OQuat q;
  q.x = a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y;
  q.y = a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x;
  q.z = a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w;
  q.w = a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z;
  return q;
}

unsigned oracle_transform_samples(const double *times, const double *translations, const double *rotations_xyzw,
                                  unsigned transform_count, const double *sample_times, const double *local_samples,
                                  unsigned point_count, double max_range, double *out)
{
  unsigned valid = 0;
  if (point_count == 0 || transform_count == 0)
  {
    return 0;
  }
  for (unsigned i = 0; i < point_count; ++i)
  {
    const double vx = local_samples[3 * (size_t)i], vy = local_samples[3 * (size_t)i + 1];
    const double vz = local_samples[3 * (size_t)i + 2];
    /* goodSample, GpuTransformSamples.cpp:47-60: squared length against max_range itself */
    if (vx != vx || vy != vy || vz != vz || ((vx * vx + vy * vy) + vz * vz) > max_range)
    {
      continue;
    }
    double sample_time = sample_times[i];
    unsigned from = 0, to = transform_count - 1;
    if (transform_count > 2)
    {
      if (times[0] <= sample_time && sample_time <= times[transform_count - 1])
      {
        unsigned iterations = 0;
        while (from <= to && iterations < 100000u)
        {
          ++iterations;
          const unsigned mid_low = (from + to) / 2;
          const unsigned mid_high = (mid_low + 1 < transform_count - 1) ? mid_low + 1 : transform_count - 1;
          if (sample_time >= times[mid_low] && sample_time <= times[mid_high])
          {
            from = mid_low;
            to = mid_high;
            break;
          }
          else if (sample_time <= times[mid_low])
          {
            to = mid_low - 1;
          }
          else
          {
            from = mid_low + 1;
          }
        }
      }
      else if (sample_time < times[0])
      {
        sample_time = times[0];
        from = to = 0;
      }
      else
      {
        sample_time = times[transform_count - 1];
        from = to = transform_count - 1;
      }
    }
    const double span = times[to] - times[from];
    const double f = (span != 0) ? (sample_time - times[from]) / span : 0.0;
    double position[3];
    for (int a = 0; a < 3; ++a)
    {
      position[a] = translations[3 * (size_t)from + a] + f * (translations[3 * (size_t)to + a] - translations[3 * (size_t)from + a]);
    }
    OQuat qf = { rotations_xyzw[4 * (size_t)from], rotations_xyzw[4 * (size_t)from + 1], rotations_xyzw[4 * (size_t)from + 2],
                 rotations_xyzw[4 * (size_t)from + 3] };
    OQuat qt = { rotations_xyzw[4 * (size_t)to], rotations_xyzw[4 * (size_t)to + 1], rotations_xyzw[4 * (size_t)to + 2],
                 rotations_xyzw[4 * (size_t)to + 3] };
    const OQuat q = o_quat_mul(qf, o_slerp(qf, qt, f));
    /* ohmgpu/gpu/TransformSamples.cl:68-91 */
    const double xx = q.x * q.x, xy = q.x * q.y, xz = q.x * q.z, xw = q.x * q.w;
    const double yy = q.y * q.y, yz = q.y * q.z, yw = q.y * q.w;
    const double zz = q.z * q.z, zw = q.z * q.w;
    const double rx = (1 - 2 * (yy + zz)) * vx + (2 * (xy - zw)) * vy + (2 * (xz + yw)) * vz;
    const double ry = (2 * (xy + zw)) * vx + (1 - 2 * (xx + zz)) * vy + (2 * (yz - xw)) * vz;
    const double rz = (2 * (xz - yw)) * vx + (2 * (yz + xw)) * vy + (1 - 2 * (xx + yy)) * vz;
    double *o = out + 6 * (size_t)valid;
    o[0] = position[0];
    o[1] = position[1];
    o[2] = position[2];
    o[3] = position[0] + rx;
    o[4] = position[1] + ry;
    o[5] = position[2] + rz;
    ++valid;
  }
  return 2 * valid;
}
