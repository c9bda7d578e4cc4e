/*
 * chisel_oracle.h -- C interface of the CPU ORACLE for the OpenChisel dense-TSDF hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The shipped path (cvids_amd/csrc -> libchisel_hip.so) never links, loads or calls it.
 *
 * It is a plain C++ restatement (Eigen-free, g++ -O3 -ffp-contract=off, no -march, i.e. the
 * reference's own flags open_chisel/catkin.cmake:10-12) of the reference algorithm under
 * /root/reference/OpenChisel/open_chisel; every function cites the file:line it follows.
 *
 * PARITY PINNING: the reference has no tests, fixtures or golden vectors and
 * the path as a whole is unbuildable here (needs Eigen, absent).  The Eigen-free rows
 * (DistVoxel, ColorVoxel, the three truncators, ConstantWeighter, ColorImage::At) ARE pinned:
 * oracle/ref_kat builds a generator from the reference's own unmodified headers and its outputs
 * are committed as tests/golden/ref_kat.json, which this oracle must reproduce bit for bit.
 * Everything that touches Eigen types (projection, frustum, chunk enumeration, marching cubes,
 * the point-cloud fusion mode with its ray walk) is "parity unpinned": a restatement of the cited
 * lines with the fp32 operation order of Eigen >= 3.3 fixed-size expressions (3-term sums reduce
 * as a0 + (a1 + a2); Transform * Vec3 as the homogeneous 4-vector product; Transform::inverse()
 * by cofactors).
 */
#ifndef CHISEL_ORACLE_H_
#define CHISEL_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* truncator kinds (truncation/{Constant,Inverse,Quadratic}Truncator.h) */
enum { OC_TRUNC_CONSTANT = 0, OC_TRUNC_INVERSE = 1, OC_TRUNC_QUADRATIC = 2 };

/* counters, indices into the array filled by oc_get_counters (SURVEY.md 8d definitions) */
enum {
    OC_CNT_SDF = 0,      /* voxels that took the in-band branch (DistVoxel::Integrate)          */
    OC_CNT_COL = 1,      /* of those: ColorVoxel::Integrate executed (colour weight < 8)       */
    OC_CNT_COL_SAT = 2,  /* of those: on the colour image but colour weight saturated (read)    */
    OC_CNT_PROBE = 3,    /* voxels of RESIDENT chunks that took the carve test branch           */
    OC_CNT_CARVED = 4,   /* of those: Carve() or weight decay fired                             */
    OC_CNT_VISITED = 5,  /* voxels visited by the reference loop (all candidates x N^3)         */
    OC_CNT_CANDIDATES = 6, /* candidate chunks enumerated (GetChunkIDsIntersecting)              */
    OC_CNT_CREATED = 7,  /* chunks allocated this frame                                         */
    OC_CNT_COLLECTED = 8,/* chunks garbage-collected this frame                                 */
    OC_CNT_UPDATED_CHUNKS = 9, /* chunks whose Integrate returned true                          */
    OC_NUM_COUNTERS = 10
};

typedef struct oc_map oc_map;

oc_map *oc_create(int csx, int csy, int csz, float resolution, int use_color);
void oc_destroy(oc_map *m);
void oc_reset(oc_map *m);
/* ChiselServer::SetupProjectionIntegrator (chisel_ros/src/ChiselServer.cpp:480-487) */
void oc_set_integrator(oc_map *m, int trunc_kind, float trunc_param, float weight,
                       int carving_enabled, float carving_dist);
/* threads used by IntegrateDepthScanColor / RecomputeMeshes (reference: fixed 16) */
void oc_set_threads(oc_map *m, int n_threads);

/* pose = camera->world rigid transform, row-major 3x4 [R|t] */
void oc_integrate_depth(oc_map *m, const float *depth, int W, int H, const float *pose,
                        float fx, float fy, float cx, float cy, float near_plane, float far_plane);
void oc_integrate_depth_color(oc_map *m, const float *depth, int W, int H, const float *pose,
                              float fx, float fy, float cx, float cy, float near_plane, float far_plane,
                              const uint8_t *color, int CW, int CH, int channels, const float *color_pose,
                              float cfx, float cfy, float ccx, float ccy);

/* point-cloud fusion mode (Chisel.cpp:107-157): points in the sensor frame (xyz per point), optional colours in [0, 1]
 * (rgb per point).  Counters afterwards: SDF, CARVED, VISITED (= ray cells inside listed chunks), CANDIDATES (= listed chunks),
 * CREATED, COLLECTED, UPDATED_CHUNKS. */
void oc_integrate_pointcloud(oc_map *m, const float *points_xyz, int n_points, const float *colors_rgb /* or NULL */,
                             const float *pose /* row-major 3x4 */, float truncation, float max_dist);
/* geometry/Raycast.cpp:35-128; returns the number of cells, writes at most `capacity` of them */
int oc_raycast(const float *start3, const float *end3, const int *min3, const int *max3, int *cells_xyz, int capacity);
/* Eigen::Affine3f::inverse() of a row-major 3x4 pose */
void oc_invert_pose(const float *pose, float *inverse);
void oc_get_counters(const oc_map *m, uint64_t *out /* OC_NUM_COUNTERS, last frame */);
void oc_get_phase_ms(const oc_map *m, double *out4 /* intersect, allocation, integration, garbage */);

int oc_num_chunks(const oc_map *m);
void oc_list_chunks(const oc_map *m, int *ids_xyz /* 3*num */);
int oc_has_chunk(const oc_map *m, int x, int y, int z);
/* returns 0 when absent; arrays are N^3 long, rgbw may be NULL */
int oc_get_chunk(const oc_map *m, int x, int y, int z, float *sdf, float *weight, uint8_t *rgbw);
int oc_remove_chunk(oc_map *m, int x, int y, int z);

int oc_num_meshes_to_update(const oc_map *m);
void oc_list_meshes_to_update(const oc_map *m, int *ids_xyz);
/* Chisel::UpdateMeshes: recompute on every 10th call unless force != 0 */
void oc_update_meshes(oc_map *m, int force);
int oc_num_meshes(const oc_map *m);
void oc_list_meshes(const oc_map *m, int *ids_xyz);
int oc_mesh_size(const oc_map *m, int x, int y, int z, int *n_vertices, int *n_grids);
int oc_get_mesh(const oc_map *m, int x, int y, int z, float *vertices, float *normals, float *colors,
                float *grids);
int oc_save_ply(const oc_map *m, const char *path);
int oc_get_sdf(const oc_map *m, float x, float y, float z, double *dist);
int oc_get_sdf_and_gradient(const oc_map *m, float x, float y, float z, double *dist, float *grad3);

/* ---- scalar known-answer entry points ---- */
float oc_truncation(int kind, float param, float depth);
float oc_weight(float weight, float surface_dist, float truncation);
void oc_dist_integrate(float *sdf, float *weight, float dist_update, float weight_update);
void oc_color_integrate(uint8_t *rgbw, uint8_t r, uint8_t g, uint8_t b, uint8_t weight_update);
void oc_color_at(const uint8_t *data, int width, int channels, int row, int col, uint8_t *rgba_out);
uint64_t oc_chunk_hash(int x, int y, int z);
void oc_project_point(float fx, float fy, float cx, float cy, const float *p3, float *out3);
/* out: corners[8*3], planes[6*4] in order far,near,top,bottom,left,right (normal xyz, distance), aabb[6] */
void oc_frustum(const float *pose, float near_plane, float far_plane, float fy, float cy, int W, int H,
                float *corners, float *planes, float *aabb);
/* candidate ids for a frustum, reference enumeration order; returns count, writes up to max */
int oc_candidates(const oc_map *m, const float *pose, float near_plane, float far_plane, float fy, float cy,
                  int W, int H, int *ids_xyz, int max_ids);
/* marching cubes on one cube: corner sdf[8], coords origin + res; returns #vertices, writes <= 15*3 floats */
int oc_mesh_cube(const float *corner_sdf8, const float *origin3, float res, float *verts, float *normals);
/* triangle table row (16 ints) */
void oc_triangle_table_row(int index, int *row16);

#ifdef __cplusplus
}
#endif
#endif
