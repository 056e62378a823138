/*
 * ohm_oracle.h -- CPU restatement (plain C) of the reference ohm CPU ray mappers.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped product path (ohm_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / timed CPU baseline.
 *
 * Every function cites the reference file:line (relative to the reference checkout) whose
 * arithmetic and operation ORDER it restates.  All maths is IEEE double/float with FP
 * contraction off (build with -ffp-contract=off), matching an x86-64 build of the reference.
 *
 * Parity pinning: see oracle/README.md -- pinned against the reference's own known-answer tests
 * (tests/test_oracle_pins.py) and against the glm-free reference headers compiled in place
 * (oracle/_ref, tests/test_oracle_vs_ref.py).
 */
#ifndef OHM_ORACLE_H
#define OHM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ohm/Key.h:100-101: i16vec3 region + u8vec3 local. */
typedef struct OracleKey
{
  int16_t region[3];
  uint8_t local[3];
  uint8_t pad;
} OracleKey;

/* Layer selection bits for oracle_map_create(). Layout per ohm/DefaultLayer.cpp:76-311. */
enum OracleLayer
{
  ORACLE_LAYER_OCCUPANCY = 1u << 0, /* float, clear = +inf */
  ORACLE_LAYER_MEAN = 1u << 1,      /* {u32 coord, u32 count} */
  ORACLE_LAYER_COVARIANCE = 1u << 2, /* float[6] */
  ORACLE_LAYER_TRAVERSAL = 1u << 3, /* float */
  ORACLE_LAYER_TOUCH_TIME = 1u << 4, /* u32 */
  ORACLE_LAYER_INCIDENT = 1u << 5,  /* u32 */
  ORACLE_LAYER_INTENSITY = 1u << 6, /* {float mean, float cov} */
  ORACLE_LAYER_HIT_MISS = 1u << 7,  /* {u32 hit, u32 miss} */
  ORACLE_LAYER_TSDF = 1u << 8       /* {float weight, float distance} */
};

/* Layer ids for oracle_region_layer(). */
enum OracleLayerId
{
  ORACLE_LID_OCCUPANCY = 0,
  ORACLE_LID_MEAN,
  ORACLE_LID_COVARIANCE,
  ORACLE_LID_TRAVERSAL,
  ORACLE_LID_TOUCH_TIME,
  ORACLE_LID_INCIDENT,
  ORACLE_LID_INTENSITY,
  ORACLE_LID_HIT_MISS,
  ORACLE_LID_TSDF,
  ORACLE_LID_COUNT
};

/* ohm/RayFlag.h:16-60 */
enum OracleRayFlag
{
  ORACLE_RF_DEFAULT = 0,
  ORACLE_RF_END_POINT_AS_FREE = 1u << 0,
  ORACLE_RF_STOP_ON_FIRST_OCCUPIED = 1u << 1,
  ORACLE_RF_EXCLUDE_ORIGIN = 1u << 2,
  ORACLE_RF_EXCLUDE_SAMPLE = 1u << 3,
  ORACLE_RF_EXCLUDE_RAY = 1u << 4,
  ORACLE_RF_EXCLUDE_UNOBSERVED = 1u << 5,
  ORACLE_RF_EXCLUDE_FREE = 1u << 6,
  ORACLE_RF_EXCLUDE_OCCUPIED = 1u << 7
};

/* Ray filter selection (ohm/RayFilter.cpp:12-58). */
enum OracleRayFilter
{
  ORACLE_FILTER_NONE = 0,
  ORACLE_FILTER_GOOD = 1, /* goodRayFilter(max_range) -- map default with 1e10 */
  ORACLE_FILTER_CLIP = 2  /* clipRayFilter(max_length) */
};

/* Walk flags: ohm/LineWalk.h:51-57 */
enum OracleWalkFlag
{
  ORACLE_WALK_EXCLUDE_START = 1u << 0,
  ORACLE_WALK_EXCLUDE_END = 1u << 1
};

typedef struct OracleMap OracleMap;

OracleMap *oracle_map_create(double resolution, int dimx, int dimy, int dimz, unsigned layers);
void oracle_map_destroy(OracleMap *map);
void oracle_map_set_origin(OracleMap *map, double x, double y, double z);
/* ohm/OccupancyMap.cpp:762-800 */
void oracle_map_set_hit_probability(OracleMap *map, float p);
void oracle_map_set_miss_probability(OracleMap *map, float p);
void oracle_map_set_threshold_probability(OracleMap *map, float p);
void oracle_map_set_hit_value(OracleMap *map, float v);
void oracle_map_set_miss_value(OracleMap *map, float v);
void oracle_map_set_min_max(OracleMap *map, float min_value, float max_value);
void oracle_map_set_saturation(OracleMap *map, int at_min, int at_max);
void oracle_map_set_ray_filter(OracleMap *map, int mode, double range);
/* Per-ray RayFilterFlag bits (ohm/RayFilter.h:21-29: 1 invalid, 2 clipped start, 4 clipped end) for integrate calls whose rays the caller already passed through a
 * RayFilterFunction; NULL restores the built-in filter.  The array must outlive the calls it covers. */
void oracle_map_set_batch_filter_flags(OracleMap *map, const unsigned char *flags);
float oracle_map_hit_value(const OracleMap *map);
float oracle_map_miss_value(const OracleMap *map);
/* NDT parameters: ohm/private/NdtMapDetail.h:20-45. adaptation_rate < 0 => derive from miss probability
 * (ohm/NdtMap.cpp:31-36, ohm/NdtMap.h:146-149). */
void oracle_map_set_ndt(OracleMap *map, float sensor_noise, unsigned sample_threshold, float adaptation_rate,
                        float reinit_threshold, unsigned reinit_count, float initial_intensity_cov, int ndt_tm);
float oracle_map_ndt_adaptation_rate(const OracleMap *map);
/* TSDF options: ohm/VoxelTsdf.h:27-37 */
void oracle_map_set_tsdf(OracleMap *map, float max_weight, float trunc, float dropoff, float sparsity);

/* Key maths. ohm/OccupancyMap.cpp:859-886, ohm/MapRegion.cpp:32-69, ohm/MapCoord.h:32-93 */
int oracle_voxel_key(const OracleMap *map, const double p[3], OracleKey *key);
/* ohm/OccupancyMap.h:757-778 */
void oracle_voxel_centre(const OracleMap *map, const OracleKey *key, double centre[3]);

/* ohm/LineWalk.h:112-129 + ohm/LineWalkCompute.h:345-413. Writes up to cap keys/enter/exit (any may be NULL).
 * Returns the number of voxels visited (even if > cap). */
size_t oracle_walk_segment_keys(const OracleMap *map, const double start[3], const double end[3], unsigned walk_flags,
                                OracleKey *keys, double *enter, double *exit, size_t cap);

/* The three CPU mappers. rays = 2*n_rays dvec3 (origin, sample pairs); element_count = number of POINTS.
 * ohm/RayMapperOccupancy.cpp:68-339, ohm/RayMapperNdt.cpp:84-407, ohm/RayMapperTsdf.cpp:87-182.
 * Return element_count/2 like the reference. */
size_t oracle_integrate_occupancy(OracleMap *map, const double *rays, size_t element_count, const double *timestamps,
                                  unsigned ray_flags);
size_t oracle_integrate_ndt(OracleMap *map, const double *rays, size_t element_count, const float *intensities,
                            const double *timestamps, unsigned ray_flags);
size_t oracle_integrate_tsdf(OracleMap *map, const double *rays, size_t element_count);

/* Total voxel visits (miss visits + sample updates) performed so far. */
uint64_t oracle_map_visit_count(const OracleMap *map);

/* Region enumeration / raw layer access (MapChunk layout: index = x + y*dx + z*dx*dy, ohm/MapChunk.h:33-50). */
size_t oracle_region_count(const OracleMap *map);
size_t oracle_region_keys(const OracleMap *map, int16_t *keys_xyz, size_t cap);
void oracle_map_set_first_ray_time(OracleMap *m, double time);
void *oracle_region_layer(OracleMap *map, int rx, int ry, int rz, int layer_id);
size_t oracle_layer_voxel_bytes(int layer_id);

/* Stand-alone arithmetic entry points (for pinning against the reference headers / tests). */
void oracle_occupancy_adjust_hit(float *occ, float initial, float adj, float uninit, float max_value, float sat_min,
                                 float sat_max, int null_update);
void oracle_occupancy_adjust_miss(float *occ, float initial, float adj, float uninit, float min_value, float sat_min,
                                  float sat_max, int null_update);
void oracle_occupancy_adjust_up(float *occ, float initial, float adjusted, float uninit, float max_value, float sat_min,
                                float sat_max, int null_update);
void oracle_occupancy_adjust_down(float *occ, float initial, float adjusted, float uninit, float min_value,
                                  float sat_min, float sat_max, int null_update);
int oracle_point_to_region_coord(double coord, double resolution);
int oracle_point_to_region_voxel(double coord, double voxel_resolution, double region_resolution);
unsigned oracle_sub_voxel_coord(const double local[3], double resolution);
void oracle_sub_voxel_to_local(unsigned pattern, double resolution, double local[3]);
unsigned oracle_sub_voxel_update(unsigned coord, unsigned count, const double local[3], double resolution);
int oracle_calculate_tsdf(const double sensor[3], const double sample[3], const double centre[3], float trunc,
                          float max_weight, float dropoff, float sparsity, float *weight, float *distance);
int oracle_calculate_hit_with_covariance(float cov[6], float *value, const double sample[3], const double mean[3],
                                         unsigned point_count, float hit_value, float uninit, float resolution,
                                         float reinit_threshold, unsigned reinit_count);
void oracle_calculate_miss_ndt(const float cov[6], float *value, int *is_miss, const double sensor[3],
                               const double sample[3], const double mean[3], unsigned point_count, float uninit,
                               float miss_value, float adaptation_rate, float sensor_noise, unsigned sample_threshold);
/* GpuTransformSamples semantics in fp64 (ohmgpu/GpuTransformSamples.cpp:47-60, 131-142 sample filter + compaction;
 * ohmgpu/gpu/TransformSamples.cl:13-228 bracketing search, lerp, rot[from] * slerp(rot[from], rot[to], f), rotate +
 * translate).  out: 6 doubles per valid sample (sensor origin, sample).  Returns 2 x valid samples. */
unsigned oracle_transform_samples(const double *transform_times, const double *translations, const double *rotations_xyzw,
                                  unsigned transform_count, const double *sample_times, const double *local_samples,
                                  unsigned point_count, double max_range, double *out);
float oracle_probability_to_value(float p);
float oracle_value_to_probability(float v);

#ifdef __cplusplus
}
#endif

#endif /* OHM_ORACLE_H */
