/*
 * chisel_hip_selftest.h -- the remaining exports of libchisel_hip.so: device-side known-answer checks and debug read-outs used
 * by tests/ and tools/.  Not part of the drop-in boundary (include/chisel_hip.h is); declared here so that every symbol the
 * library exports has a declaration.
 */
#ifndef CHISEL_HIP_SELFTEST_H_
#define CHISEL_HIP_SELFTEST_H_
#include "chisel_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* device arithmetic against the reference-built golden vectors (tests/golden/ref_kat.json): truncation distance and
 * ConstantWeighter weight per reading; DistVoxel::Integrate on (sdf, w, update, weight) 4-tuples; ColorVoxel::Integrate */
int chisel_hip_kat_truncation(int kind, float param, const float *depths, int n, float *trunc, float *weight1);
int chisel_hip_kat_dist(const float *ops, int n, float *out);
int chisel_hip_kat_color(const uint8_t *ops, int n, uint8_t *out);
/* exhaustive device checks of the kernels' equivalent rewrites (each returns the number of mismatches, 0 expected):
 * colour running average for weights < 8 / any weight; reciprocal_in_range() against IEEE 1.0f / z on [2^-40, 2^40];
 * the one-instruction floor-to-int against v_floor_f32 + v_cvt_i32_f32 on all 2^32 bit patterns */
int chisel_hip_kat_color_fresh(unsigned *mismatches);
int chisel_hip_kat_color_any(unsigned *mismatches);
int chisel_hip_kat_reciprocal(unsigned long long *mismatches, unsigned *example_bits);
int chisel_hip_kat_floor(unsigned long long *mismatches, unsigned *example_bits);
/* Raycast (src/geometry/Raycast.cpp:34-128) of n rays (6 floats each) clipped to [lo, hi]: the cells met, in order */
int chisel_hip_kat_raycast(const float *rays, int n, const int lo[3], const int hi[3], int *cells, int cap, int *count);
/* point-cloud path: listed chunks, (unit, point) pairs, busiest unit, units of the last cloud */
int chisel_hip_debug_cloud_stats(chisel_hip_map *map, int64_t out[4]);
/* host evaluation of the candidate enumeration: the ids cull_kernel hands to one shard; the reference's id range + planes + corners */
int chisel_hip_debug_cull_space(const int range_min[3], const int range_dim[3], int n_shards, int shard_rank, int shard_block, int *ids,
                                int capacity, int *count);
int chisel_hip_debug_frustum_range(const float *pose, float near_plane, float far_plane, float fy, float cy, int W, int H, int chunk_n,
                                   float res, int *range_min3, int *range_dim3, float *planes24, float *corners24);

#ifdef __cplusplus
}
#endif
#endif /* CHISEL_HIP_SELFTEST_H_ */
