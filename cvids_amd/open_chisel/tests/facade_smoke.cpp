// facade_smoke.cpp -- drives the chisel::* facade the way chisel_ros::ChiselServer does (ChiselServer.cpp:480-516):
// setup integrator -> IntegrateDepthScanColor per frame -> IntegratePointCloud -> UpdateMeshes -> queries, and dumps the voxel fields so that
// tests/test_gpu_facade.py can compare them with the Python host path on the same frames.
#include <open_chisel/Chisel.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

using namespace chisel;

int main(int argc, char **argv) {
    const char *out_path = argc > 1 ? argv[1] : nullptr;
    const int W = 64, H = 48, N = 8;
    const float res = 0.05f;
    try {
        Chisel map(Eigen::Vector3i(N, N, N), res, true);
        TruncatorPtr trunc(new InverseTruncator(2.0f));
        WeighterPtr weigh(new ConstantWeighter(1.0f));
        ProjectionIntegrator integ(trunc, weigh, 0.05f, true, map.GetChunkManager().GetCentroids());
        PinholeCamera cam;
        Intrinsics K;
        K.SetFx(52.5f); K.SetFy(52.5f); K.SetCx(31.5f); K.SetCy(23.5f);
        cam.SetIntrinsics(K);
        cam.SetWidth(W); cam.SetHeight(H);
        cam.SetNearPlane(0.05f); cam.SetFarPlane(5.0f);
        std::shared_ptr<DepthImage<float>> depth(new DepthImage<float>(W, H));
        std::shared_ptr<ColorImage<uint8_t>> color(new ColorImage<uint8_t>(W, H, 3));
        for (int v = 0; v < H; v++)
            for (int u = 0; u < W; u++) {
                uint8_t *p = color->GetMutableData() + color->Index(v, u, 0);
                p[0] = (uint8_t)(u % 256); p[1] = (uint8_t)(v % 256); p[2] = (uint8_t)((u + v) % 256);
            }
        for (int k = 0; k < 3; k++) {
            // a wall at z = 1.5 + 0.1 k metres, camera at the origin looking along +z
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++) depth->SetDataAt(v, u, 1.5f + 0.1f * k);
            Transform T;  // identity
            map.IntegrateDepthScanColor<float, uint8_t>(integ, depth, T, cam, color, T, cam);
        }
        {
            // PointCloud fusion mode (ChiselServer.cpp:517-524): a coloured wall at 1.2 m seen from a sensor at (0.05, 0, 0.02)
            PointCloud cloud;
            for (int v = 0; v < H; v++)
                for (int u = 0; u < W; u++) {
                    const float z = 1.2f;
                    cloud.AddPointAndColor(Vec3((((float)u - 31.5f) / 52.5f) * z, (((float)v - 23.5f) / 52.5f) * z, z),
                                           Vec3((float)(u % 256) / 255.0f, (float)(v % 256) / 255.0f, (float)((u + v) % 256) / 255.0f));
                }
            Transform T;
            T.translation() = Vec3(0.05f, 0.0f, 0.02f);
            map.IntegratePointCloud(integ, cloud, T, 0.1f, 5.0f);
        }
        for (int k = 0; k < 10; k++) map.UpdateMeshes();  // the first call recomputes (Chisel.cpp:53)
        const ChunkMap &chunks = map.GetChunkManager().GetChunks();
        const MeshMap &meshes = map.GetChunkManager().GetAllMeshes();
        size_t nverts = 0;
        for (const auto &kv : meshes) nverts += kv.second->vertices.size();
        printf("chunks %zu meshes %zu vertices %zu to_update %zu\n", chunks.size(), meshes.size(), nverts, map.GetMeshesToUpdate().size());
        bool threw = false;
        try {
            map.GetChunkManager().GetChunk(ChunkID(1000, 1000, 1000));
        } catch (const std::out_of_range &) {
            threw = true;
        }
        if (!threw || chunks.empty() || meshes.empty()) return 2;
        double dist = 0;
        Vec3 grad;
        const bool has = map.GetChunkManager().GetSDFAndGradient(Vec3(0.01f, 0.01f, 1.62f), &dist, &grad);
        printf("sdf at (0.01, 0.01, 1.62): found %d dist %.9g grad %.9g %.9g %.9g\n", (int)has, dist, grad(0), grad(1), grad(2));
        if (out_path) {
            FILE *f = fopen(out_path, "wb");
            if (!f) return 3;
            for (const auto &kv : chunks) {
                ChunkPtr c = map.GetChunkManager().GetChunk(kv.first);
                const int id[3] = {kv.first(0), kv.first(1), kv.first(2)};
                fwrite(id, sizeof(int), 3, f);
                for (const DistVoxel &v : c->GetVoxels()) {
                    fwrite(&v.sdf, 4, 1, f);
                    fwrite(&v.weight, 4, 1, f);
                }
                for (const ColorVoxel &v : c->GetColorVoxels()) {
                    const uint8_t b[4] = {v.red, v.green, v.blue, v.weight};
                    fwrite(b, 1, 4, f);
                }
            }
            fclose(f);
        }
        if (!map.SaveAllMeshesToPLY("/tmp/chisel_hip_facade_smoke.ply")) return 4;
    } catch (const std::exception &e) {
        fprintf(stderr, "facade_smoke: %s\n", e.what());
        return 1;
    }
    return 0;
}
