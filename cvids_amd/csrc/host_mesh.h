// host_mesh.h -- mesh-side entry points of the C ABI (included by chisel_hip.hip).
extern "C" {
int chisel_hip_update_meshes(chisel_hip_map *m, int force) { (void)m; (void)force; return fail(CHISEL_HIP_ERR_UNSUPPORTED, "mesh extraction not built yet"); }
int chisel_hip_num_meshes(chisel_hip_map *m, int64_t *out) { if (!m || !out) return fail(CHISEL_HIP_ERR_INVALID, "null"); *out = (int64_t)m->meshes.size(); return CHISEL_HIP_OK; }
int chisel_hip_list_meshes(chisel_hip_map *m, int *ids, int64_t max_ids, int64_t *count) { (void)ids; (void)max_ids; if (!m || !count) return fail(CHISEL_HIP_ERR_INVALID, "null"); *count = 0; return CHISEL_HIP_OK; }
int chisel_hip_mesh_size(chisel_hip_map *m, const int id[3], int64_t *nv, int64_t *ng) { (void)m; (void)id; (void)nv; (void)ng; return fail(CHISEL_HIP_ERR_NOT_FOUND, "no mesh"); }
int chisel_hip_download_mesh(chisel_hip_map *m, const int id[3], float *v, float *n, float *c, float *g) { (void)m; (void)id; (void)v; (void)n; (void)c; (void)g; return fail(CHISEL_HIP_ERR_NOT_FOUND, "no mesh"); }
int chisel_hip_get_sdf(chisel_hip_map *m, const float pos[3], double *dist, int *found) { (void)m; (void)pos; (void)dist; (void)found; return fail(CHISEL_HIP_ERR_UNSUPPORTED, "not built yet"); }
int chisel_hip_get_sdf_and_gradient(chisel_hip_map *m, const float pos[3], double *dist, float grad[3], int *found) { (void)m; (void)pos; (void)dist; (void)grad; (void)found; return fail(CHISEL_HIP_ERR_UNSUPPORTED, "not built yet"); }
int chisel_hip_save_ply(chisel_hip_map *m, const char *path) { (void)m; (void)path; return fail(CHISEL_HIP_ERR_UNSUPPORTED, "not built yet"); }
}
