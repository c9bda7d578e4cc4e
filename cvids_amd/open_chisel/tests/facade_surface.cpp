// facade_surface.cpp -- the parts of OpenChisel's public surface that chisel_ros does not call (a third-party caller might):
// ChunkManager(const Vector3i &, float, bool), GetDistanceVoxel / GetColorVoxel(pos), GetChunkIDsIntersecting(AABB / Frustum),
// GenerateMesh / ColorizeMesh / ComputeNormalsFromGradients / InterpolateColor / RecomputeMesh / GetMutableMesh / CacheCentroids,
// ProjectionIntegrator::Integrate / IntegrateColor per chunk, Frustum::Intersects / Contains, Plane's constructors
// (ChunkManager.h:61-212, ProjectionIntegrator.h:51-52 / 101-102, Frustum.cpp:41-99, Plane.cpp:32-63).  Every check is a consistency the
// reference's own code guarantees: RecomputeMesh == GenerateMesh -> ColorizeMesh -> ComputeNormalsFromGradients (ChunkManager.cpp:91-128),
// per-chunk Integrate over the candidates of a frame == IntegrateDepthScanColor of that frame (Chisel.h:114-213), and so on.
#include <open_chisel/Chisel.h>
#include <open_chisel/FixedPointFloat.h>
#include <open_chisel/geometry/Interpolate.h>
#include <open_chisel/geometry/Raycast.h>
#include <open_chisel/io/PLY.h>
#include <open_chisel/marching_cubes/MarchingCubes.h>
#include <open_chisel/threading/Threading.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>

using namespace chisel;

#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            fprintf(stderr, "facade_surface: line %d: %s\n", __LINE__, #cond); \
            return 10;                                                     \
        }                                                                  \
    } while (0)

static bool same_bits(const Vec3 &a, const Vec3 &b) { return std::memcmp(&a, &b, 3 * sizeof(float)) == 0; }

int main() {
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
                p[0] = (uint8_t)(u * 3); p[1] = (uint8_t)(v * 5); p[2] = (uint8_t)(u + v);
                depth->SetDataAt(v, u, 1.5f + 0.004f * u - 0.003f * v);  // a tilted wall
            }
        // ---- value types: camera projection, image accessors, truncators (PinholeCamera.cpp:38-64, DepthImage.h:59-89, ColorImage.h:66-118)
        {
            const Vec3 q = cam.ProjectPoint(Vec3(0.2f, -0.1f, 2.0f));
            CHECK(q(0) == 52.5f * 0.2f * (1.0f / 2.0f) + 31.5f && q(1) == 52.5f * -0.1f * (1.0f / 2.0f) + 23.5f && q(2) == 2.0f);
            const Vec3 back = cam.UnprojectPoint(q);
            CHECK(std::fabs(back(0) - 0.2f) < 1e-6f && std::fabs(back(1) + 0.1f) < 1e-6f && back(2) == 2.0f);
            CHECK(cam.IsPointOnImage(q) && !cam.IsPointOnImage(Vec3(-0.5f, 3.0f, 1.0f)) && !cam.IsPointOnImage(Vec3(64.0f, 3.0f, 1.0f)));
            CHECK(cam.GetIntrinsics().GetMatrix()(0, 2) == 31.5f && cam.GetIntrinsics().GetMatrix()(1, 1) == 52.5f);
            CHECK(depth->At(3, 5) == depth->DepthAt(3, 5) && depth->IsInside(3, 5) && !depth->IsInside(-1, 0));
            depth->AtMutable(3, 5) = depth->At(3, 5);
            const float mid = depth->BilinearInterpolateDepth(5.5f, 3.5f);
            CHECK(std::fabs(mid - 0.25f * (depth->At(3, 5) + depth->At(3, 6) + depth->At(4, 5) + depth->At(4, 6))) < 1e-6f);
            CHECK(BilinearInterpolate(1.0f, 3.0f, 5.0f, 7.0f, 0.5f, 0.5f) == 4.0f && LinearInterpolate(2.0f, 4.0f, 0.25f) == 2.5f);
            Color<uint8_t> px = {0, 0, 0, 0};
            color->At(7, 9, &px);  // BGR: red is byte 2
            CHECK(px.red == color->At(7, 9, 2) && px.green == color->At(7, 9, 1) && px.blue == color->At(7, 9, 0) && px.alpha == px.red);
            CHECK(ColorVoxel::Saturate(300.0f) == 255.0f && ColorVoxel::Saturate(-2.0f) == 0.0f && ColorVoxel::Saturate(17.5f) == 17.5f);
            QuadraticTruncator qt(2.0f);
            CHECK(qt.GetScalingFactor() == 2.0f && qt.GetTruncationDistance(1.0f) == (float)(std::abs(qt.GetQuadraticTerm() * std::pow(1.0f, 2) + qt.GetLinearTerm() * 1.0f + qt.GetConstantTerm()) * 2.0f));
            ConstantTruncator ct(0.1f);
            ct.SetTruncationDistance(0.3f);
            CHECK(ct.GetTruncationDistance(5.0f) == 0.3f);
        }
        Transform T;
        for (int k = 0; k < 2; k++) map.IntegrateDepthScanColor<float, uint8_t>(integ, depth, T, cam, color, T, cam);
        ChunkManager &cm = map.GetMutableChunkManager();

        // ---- GetChunks keeps its mirrors while the chunk set stands still
        const ChunkMap &chunks = cm.GetChunks();
        CHECK(chunks.size() > 10);
        const ChunkID some = chunks.begin()->first;
        const Chunk *mirror = chunks.begin()->second.get();
        CHECK(cm.GetChunks().find(some)->second.get() == mirror);
        map.IntegrateDepthScanColor<float, uint8_t>(integ, depth, T, cam, color, T, cam);  // same surface: no new chunk
        CHECK(cm.GetChunks().find(some)->second.get() == mirror);
        CHECK(cm.RemoveChunk(some));
        CHECK(cm.GetChunks().count(some) == 0 && !cm.HasChunk(some));

        // ---- GetDistanceVoxel / GetColorVoxel (ChunkManager.cpp:575-607) against the chunk's own voxel array
        ChunkID full(0, 0, 0);
        for (const auto &kv : cm.GetChunks()) {
            size_t known = 0;
            for (const DistVoxel &v : kv.second->GetVoxels()) known += v.GetWeight() > 0;
            if (known > 100) { full = kv.first; break; }
        }
        ChunkPtr c = cm.GetChunk(full);
        const Vec3 p = c->GetOrigin() + Vec3(3.4f * res, 2.6f * res, 5.5f * res);
        const DistVoxel *dv = cm.GetDistanceVoxel(p);
        const ColorVoxel *cv = cm.GetColorVoxel(p);
        CHECK(dv && cv);
        const DistVoxel &want = c->GetDistVoxel(3, 2, 5);
        CHECK(dv->GetSDF() == want.GetSDF() && dv->GetWeight() == want.GetWeight());
        CHECK(cv->GetRed() == c->GetColorVoxel(3, 2, 5).GetRed() && cv->GetWeight() == c->GetColorVoxel(3, 2, 5).GetWeight());
        CHECK(cm.GetDistanceVoxel(Vec3(100.0f, 100.0f, 100.0f)) == nullptr);
        {   // Chunk::IsCoordValid / GetColorAt / ComputeStatistics (Chunk.h:106-109, Chunk.cpp:89-136) on the mirror
            CHECK(c->IsCoordValid(0, 0, N - 1) && !c->IsCoordValid(N, 0, 0) && !c->IsCoordValid(0, -1, 0));
            const Vec3 col = c->GetColorAt(p);
            CHECK(col(0) == (float)c->GetColorVoxel(3, 2, 5).GetRed() / 255.0f && col(2) == (float)c->GetColorVoxel(3, 2, 5).GetBlue() / 255.0f);
            CHECK(same_bits(c->GetColorAt(c->GetOrigin() - Vec3(1.0f, 0.0f, 0.0f)), Vec3(0.0f, 0.0f, 0.0f)));
            ChunkStatistics st = {0, 0, 0, 0.0f};
            c->ComputeStatistics(&st);
            CHECK(st.numKnownInside + st.numKnownOutside + st.numUnknown == c->GetTotalNumVoxels() && st.numKnownInside + st.numKnownOutside > 100 && st.totalWeight > 0.0f);
            Chunk blank(ChunkID(9, 9, 9), Eigen::Vector3i(N, N, N), res, false);
            blank.AllocateColorVoxels();
            CHECK(blank.HasColors() && blank.GetColorVoxels().size() == blank.GetTotalNumVoxels() && blank.GetDistVoxel(0).GetWeight() == 0.0f);
        }

        // ---- frustum: SetupFrustum -> Intersects / Contains, GetChunkIDsIntersecting (ChunkManager.cpp:72-89, 182-212)
        Frustum fr;
        cam.SetupFrustum(T, &fr);
        CHECK(fr.Contains(Vec3(0.0f, 0.0f, 1.0f)) == fr.Contains(Vec3(0.0f, 0.0f, 1.0f)));
        {
            // Frustum::SetFromVectors on the vectors SetFromParams derives (Frustum.cpp:143-153) is the same frustum; and
            // SetFromOpenGLViewProjection (:124-141) of a view / projection pair that encodes them
            const Mat3x3 R = T.linear();
            const float fy = cam.GetIntrinsics().GetFy(), cyp = cam.GetIntrinsics().GetCy(), W = (float)cam.GetWidth(), H = (float)cam.GetHeight();
            const float aspect = (fy * W) / (fy * H), fov = (float)(std::atan2((double)cyp, (double)fy) + std::atan2((double)(H - cyp), (double)fy));
            Frustum fv;
            fv.SetFromVectors(Vec3(R(0, 2), R(1, 2), R(2, 2)), T.translation(), Vec3(R(0, 0), R(1, 0), R(2, 0)), Vec3(-R(0, 1), -R(1, 1), -R(2, 1)),
                              cam.GetNearPlane(), cam.GetFarPlane(), fov, aspect);
            for (int i = 0; i < 8; i++) CHECK(fv.GetCorners()[i] == fr.GetCorners()[i]);
            for (int i = 0; i < 24; i++) CHECK(fv.GetLines()[i] == fr.GetLines()[i]);
            CHECK(fv.GetFarPlane().distance == fr.GetFarPlane().distance && fv.GetLeftPlane().normal == fr.GetLeftPlane().normal);
            Mat4x4 view = Mat4x4::Zero(), proj = Mat4x4::Zero();
            for (int k = 0; k < 3; k++) {
                view(0, k) = R(k, 0);
                view(1, k) = -R(k, 1);
                view(2, k) = -R(k, 2);
                view(k, 3) = T.translation()(k);
            }
            view(3, 3) = 1.0f;
            const float nearD = 0.5f, farD = 4.0f, kk = farD / nearD;
            proj(1, 1) = 1.0f / std::tan(0.5f * fov);
            proj(0, 0) = proj(1, 1) / aspect;
            proj(2, 2) = (1.0f + kk) / (1.0f - kk);     // (cc - 1) / (cc + 1) = kk
            proj(2, 3) = 2.0f * kk * nearD / (1.0f - kk);  // dd (1 - kk) / (2 kk) = n
            Frustum fo;
            fo.SetFromOpenGLViewProjection(view, proj);
            // (what :124-141 derive from the two matrices, handed to SetFromVectors directly: the same frustum bit for bit; and its far
            // plane's corners lie farD in front of the camera, up to the rounding of the matrix entries)
            const float bb = proj(1, 1), cc = proj(2, 2), dd = proj(2, 3), k2 = (cc - 1.0f) / (cc + 1.0f), n2 = (dd * (1.0f - k2)) / (2.0f * k2);
            Frustum fe;
            fe.SetFromVectors(Vec3(R(0, 2), R(1, 2), R(2, 2)), T.translation(), Vec3(R(0, 0), R(1, 0), R(2, 0)), Vec3(-R(0, 1), -R(1, 1), -R(2, 1)), n2, k2 * n2,
                              (float)(2.0f * std::atan((double)(1.0f / bb))), bb / proj(0, 0));
            for (int i = 0; i < 8; i++) CHECK(fo.GetCorners()[i] == fe.GetCorners()[i]);
            CHECK(fo.GetNearPlane().distance == fe.GetNearPlane().distance && fo.GetTopPlane().normal == fe.GetTopPlane().normal);
            const Vec3 far_centre = (fo.GetCorners()[0] + fo.GetCorners()[1] + fo.GetCorners()[2] + fo.GetCorners()[3]) * 0.25f;
            CHECK((far_centre - T * Vec3(0.0f, 0.0f, farD)).norm() < 1e-3f);
        }
        ChunkIDList ids;
        cm.GetChunkIDsIntersecting(fr, &ids);
        AABB box;
        fr.ComputeBoundingBox(&box);
        const ChunkID lo = cm.GetIDAt(box.min), hi = cm.GetIDAt(box.max) + Eigen::Vector3i(1, 1, 1);
        size_t expect = 0;
        for (int x = lo(0) - 1; x <= hi(0) + 1; x++)
            for (int y = lo(1) - 1; y <= hi(1) + 1; y++)
                for (int z = lo(2) - 1; z <= hi(2) + 1; z++) {
                    const Vec3 mn = Vec3((float)(x * N), (float)(y * N), (float)(z * N)) * res;
                    expect += fr.Intersects(AABB(mn, mn + Vec3((float)N, (float)N, (float)N) * res));
                }
        CHECK(ids.size() == expect && expect > 100);
        CHECK(ids.front() == ChunkID(lo(0) - 1, lo(1) - 1, lo(2) - 1) || !fr.Intersects(AABB(Vec3(), Vec3())));
        for (const auto &kv : cm.GetChunks()) {  // every chunk the frame created is one of the reference's candidates
            bool found = false;
            for (const ChunkID &i : ids) found = found || i == kv.first;
            CHECK(found);
        }
        ChunkIDList in_box;
        cm.GetChunkIDsIntersecting(AABB(Vec3(0.0f, 0.0f, 0.0f), Vec3(0.5f, 0.3f, 0.1f)), &in_box);
        CHECK(in_box.size() == 2 * 1 * 1 && in_box[0] == ChunkID(0, 0, 0) && in_box[1] == ChunkID(1, 0, 0));
        const Plane pq(Vec3(0, 0, 1), 5.0f);  // Plane.cpp:38-42: the distance argument is dropped (sic)
        CHECK(pq.distance == 0.0f && pq.ClassifyPoint(Vec3(0, 0, -1)) == Plane::IntersectionType::Inside);
        const Plane p3(Vec3(0, 0, 1), Vec3(2, 0, 1), Vec3(0, 3, 1));  // cross = (0, 0, 6): normal (0, 0, 1), offset -6 (un-normalised)
        CHECK(p3.normal(2) == 1.0f && p3.distance == -6.0f);

        // ---- RecomputeMesh == GenerateMesh -> ColorizeMesh -> ComputeNormalsFromGradients (ChunkManager.cpp:91-128)
        std::mutex mu;
        ChunkID meshed(0, 0, 0);
        bool have = false;
        for (const auto &kv : cm.GetChunks()) {
            cm.RecomputeMesh(kv.first, mu);
            if (cm.HasMesh(kv.first) && cm.GetMesh(kv.first)->vertices.size() >= 30) { meshed = kv.first; have = true; break; }
        }
        CHECK(have);
        const MeshPtr stored = cm.GetMutableMesh(meshed);
        Mesh mine;
        cm.GenerateMesh(cm.GetChunk(meshed), &mine);
        CHECK(mine.vertices.size() == stored->vertices.size() && mine.grids.size() == stored->grids.size() && mine.colors.empty());
        size_t face_normals_kept = 0;
        for (size_t i = 0; i < mine.vertices.size(); i++) {
            CHECK(same_bits(mine.vertices[i], stored->vertices[i]));
            CHECK(mine.indices[i] == i);
            face_normals_kept += same_bits(mine.normals[i], stored->normals[i]);
        }
        for (size_t t = 0; t + 2 < mine.normals.size(); t += 3)  // MeshCube: one face normal per triangle (MarchingCubes.h:95-101)
            CHECK(same_bits(mine.normals[t], mine.normals[t + 1]) && same_bits(mine.normals[t], mine.normals[t + 2]));
        CHECK(face_normals_kept < mine.vertices.size());  // gradient normals differ from face normals on a tilted wall
        {
            // GenerateMesh is this loop in the reference (ChunkManager.cpp:395-441): interior cubes, then the max-x, max-y and max-z planes
            Mesh cubes;
            VertIndex next = 0;
            ChunkPtr ch = cm.GetChunk(meshed);
            const Vec3 org = ch->GetOrigin();
            auto centroid = [&](int x, int y, int z) { return Vec3((float)x * res + res * 0.5f, (float)y * res + res * 0.5f, (float)z * res + res * 0.5f) + org; };
            for (int z = 0; z < N - 1; z++)
                for (int y = 0; y < N - 1; y++)
                    for (int x = 0; x < N - 1; x++) cm.ExtractInsideVoxelMesh(ch, Eigen::Vector3i(x, y, z), centroid(x, y, z), &next, &cubes);
            for (int z = 0; z < N - 1; z++)
                for (int y = 0; y < N; y++) cm.ExtractBorderVoxelMesh(ch, Eigen::Vector3i(N - 1, y, z), centroid(N - 1, y, z), &next, &cubes);
            for (int z = 0; z < N - 1; z++)
                for (int x = 0; x < N - 1; x++) cm.ExtractBorderVoxelMesh(ch, Eigen::Vector3i(x, N - 1, z), centroid(x, N - 1, z), &next, &cubes);
            for (int y = 0; y < N; y++)
                for (int x = 0; x < N; x++) cm.ExtractBorderVoxelMesh(ch, Eigen::Vector3i(x, y, N - 1), centroid(x, y, N - 1), &next, &cubes);
            CHECK(cubes.vertices.size() == mine.vertices.size() && cubes.grids.size() == mine.grids.size() && next == mine.vertices.size());
            for (size_t i = 0; i < cubes.vertices.size(); i++)
                CHECK(same_bits(cubes.vertices[i], mine.vertices[i]) && same_bits(cubes.normals[i], mine.normals[i]) && cubes.indices[i] == i);
            for (size_t i = 0; i < cubes.grids.size(); i++) CHECK(same_bits(cubes.grids[i], mine.grids[i]));
        }
        cm.ColorizeMesh(&mine);
        cm.ComputeNormalsFromGradients(&mine);
        for (size_t i = 0; i < mine.vertices.size(); i++) {
            CHECK(same_bits(mine.normals[i], stored->normals[i]));
            CHECK(same_bits(mine.colors[i], stored->colors[i]));
        }
        CHECK(same_bits(cm.InterpolateColor(mine.vertices[0]), stored->colors[0]));
        {   // SaveMeshPLYASCII of one mesh (io/PLY.cpp:29-88): header, one line per vertex with its colour bytes, one per face
            const char *path = "/tmp/facade_surface_mesh.ply";
            CHECK(SaveMeshPLYASCII(path, MeshConstPtr(stored)));
            CHECK(!SaveMeshPLYASCII("/nonexistent-dir/x.ply", MeshConstPtr(stored)));
            FILE *f = fopen(path, "r");
            CHECK(f != nullptr);
            char line[256];
            size_t lines = 0, nv = 0, nf = 0;
            while (fgets(line, sizeof(line), f)) {
                lines++;
                if (sscanf(line, "element vertex %zu", &nv) == 1 || sscanf(line, "element face %zu", &nf) == 1) continue;
            }
            fclose(f);
            CHECK(nv == stored->vertices.size() && nf == nv / 3 && lines == 12 + nv + nf);
        }
        CHECK(map.GetMeshesToUpdate().size() > 0);  // GenerateMesh left meshesToUpdate alone (RecomputeMesh above cleared only its own chunks)

        // ---- per-chunk ProjectionIntegrator::IntegrateColor over the candidates of a frame == Chisel::IntegrateDepthScanColor of it
        ChunkManager own(Eigen::Vector3i(N, N, N), res, true);  // ChunkManager.h:62
        own.CacheCentroids();
        CHECK(own.GetCentroids().size() == (size_t)N * N * N);
        Chisel ref_map(Eigen::Vector3i(N, N, N), res, true);
        ref_map.IntegrateDepthScanColor<float, uint8_t>(integ, depth, T, cam, color, T, cam);
        size_t created = 0, updated = 0;
        for (const ChunkID &id : ids) {  // Chisel.h:133-143: create every candidate, integrate, erase the untouched ones (:202-207)
            own.CreateChunk(id);
            created++;
            ChunkPtr ch = own.GetChunk(id);
            if (integ.IntegrateColor<float, uint8_t>(depth, cam, T, color, cam, T, ch.get())) updated++;
            else own.RemoveChunk(id);
            if (created >= 400 && updated >= 12) break;  // (a few hundred one-chunk launches are enough for the check)
        }
        CHECK(updated >= 12);
        for (const auto &kv : own.GetChunks()) {
            CHECK(ref_map.GetChunkManager().HasChunk(kv.first));
            ChunkPtr a = own.GetChunk(kv.first), b = ref_map.GetChunkManager().GetChunk(kv.first);
            for (size_t i = 0; i < a->GetTotalNumVoxels(); i++) {
                CHECK(a->GetDistVoxel(i).GetSDF() == b->GetDistVoxel(i).GetSDF() && a->GetDistVoxel(i).GetWeight() == b->GetDistVoxel(i).GetWeight());
                CHECK(a->GetColorVoxel(i).GetRed() == b->GetColorVoxel(i).GetRed() && a->GetColorVoxel(i).GetWeight() == b->GetColorVoxel(i).GetWeight());
            }
        }
        // a free-standing chunk (not part of any manager) goes through a map of its own
        Chunk loose(meshed, Eigen::Vector3i(N, N, N), res, true);
        const bool touched = integ.IntegrateColor<float, uint8_t>(depth, cam, T, color, cam, T, &loose);
        CHECK(touched == ref_map.GetChunkManager().HasChunk(meshed));
        if (touched) {
            ChunkPtr b = ref_map.GetChunkManager().GetChunk(meshed);
            for (size_t i = 0; i < loose.GetTotalNumVoxels(); i++) CHECK(loose.GetDistVoxel(i).GetSDF() == b->GetDistVoxel(i).GetSDF());
        }
        // ---- GetChunkIDsIntersecting(PointCloud, ...) (ChunkManager.cpp:214-257): the chunks Chisel::IntegratePointCloud lists; every chunk
        // that call creates is one of them, and a cloud on a plane well inside the far limit lists at least the chunks of its own points
        {
            PointCloud cloud;
            for (int v = 0; v < H; v += 2)
                for (int u = 0; u < W; u += 2) {
                    const float d = 1.2f + 0.002f * u;
                    cloud.AddPoint(Vec3(((float)u - 31.5f) / 52.5f * d, ((float)v - 23.5f) / 52.5f * d, d));
                }
            ChunkIDList listed;
            Chisel cloud_map(Eigen::Vector3i(N, N, N), res, false);
            cloud_map.GetMutableChunkManager().GetChunkIDsIntersecting(cloud, T, 0.1f, 5.0f, &listed);
            CHECK(listed.size() > 20);
            for (size_t i = 1; i < listed.size(); i++) CHECK(!(listed[i] == listed[i - 1]));
            cloud_map.IntegratePointCloud(integ, cloud, T, 0.1f, 5.0f);
            size_t made = 0;
            for (const auto &kv : cloud_map.GetChunkManager().GetChunks()) {
                bool found = false;
                for (const ChunkID &i : listed) found = found || i == kv.first;
                CHECK(found);
                made++;
            }
            CHECK(made > 5);
        }
        // ---- marching_cubes/MarchingCubes.h:41-146: the public statics for a cube of the caller's own.  A unit cube cut by the plane
        // x = 0.25 (corners 0, 3, 4, 7 at x = 0 inside, cubeIndexOffsets order): case 0x99, two triangles, every vertex on the plane, the
        // normals along x; MeshCube(.., TriangleVector*) holds the same triangles with the vertex order reversed.
        {
            MarchingCubes::CornerCoords cc;
            MarchingCubes::CornerSDF sd;
            const int ox[8] = {0, 1, 1, 0, 0, 1, 1, 0}, oy[8] = {0, 0, 1, 1, 0, 0, 1, 1}, oz[8] = {0, 0, 0, 0, 1, 1, 1, 1};
            for (int i = 0; i < 8; i++) {
                cc.col(i) = Vec3((float)ox[i], (float)oy[i], (float)oz[i]);
                sd(i) = (float)ox[i] - 0.25f;
            }
            CHECK(MarchingCubes::CalculateVertexConfiguration(sd) == 0x99 && MarchingCubes::IsOccupied(sd));
            CHECK(MarchingCubes::triangleTable()[0][0] == -1 && MarchingCubes::triangleTable()[0x99][6] == -1 && MarchingCubes::edgeIndexPairs()[0][1] == 1);
            Mesh one;
            VertIndex next = 0;
            MarchingCubes::MeshCube(cc, sd, &next, &one);
            CHECK(next == 6 && one.vertices.size() == 6 && one.normals.size() == 6 && one.indices.size() == 6 && one.indices[5] == 5);
            for (size_t i = 0; i < 6; i++) CHECK(one.vertices[i](0) == 0.25f && std::fabs(std::fabs(one.normals[i](0)) - 1.0f) < 1e-6f);
            TriangleVector tris;
            MarchingCubes::MeshCube(cc, sd, &tris);
            CHECK(tris.size() == 2);
            for (int t = 0; t < 2; t++)
                for (int c = 0; c < 3; c++)
                    for (int r = 0; r < 3; r++) CHECK(tris[t](r, c) == one.vertices[3 * t + 2 - c](r));
            MarchingCubes::EdgeCoords ec;
            MarchingCubes::InterpolateEdgeVertices(cc, sd, &ec);
            const Vec3 e0 = ec.col(0);  // edge 0 joins corners 0 and 1
            CHECK(e0(0) == 0.25f && e0(1) == 0.0f && e0(2) == 0.0f);
            const Vec3 iv = MarchingCubes::InterpolateVertex(Vec3(0, 0, 0), Vec3(1, 0, 0), -0.25f, 0.75f);
            CHECK(iv(0) == 0.25f);
            const Vec3 flat = MarchingCubes::InterpolateVertex(Vec3(1, 2, 3), Vec3(2, 2, 2), 0.5f, 0.5f);  // "vertex1 + 0.5 * vertex2" (MarchingCubes.h:141)
            CHECK(flat(0) == 2.0f && flat(1) == 3.0f && flat(2) == 4.0f);
            sd(0) = 1.0f; sd(3) = 1.0f; sd(4) = 1.0f; sd(7) = 1.0f;
            CHECK(!MarchingCubes::IsOccupied(sd));
        }
        // ---- geometry/Raycast.h: a diagonal through a 4 x 4 x 4 window, against the chunks Chisel lists for a one-point cloud along the same segment
        {
            Point3List cells;
            Raycast(Vec3(0.5f, 0.5f, 0.5f), Vec3(3.5f, 0.5f, 0.5f), Point3(0, 0, 0), Point3(4, 4, 4), &cells);
            CHECK(cells.size() == 4 && cells[0] == Point3(0, 0, 0) && cells[3] == Point3(3, 0, 0));
            cells.clear();
            Raycast(Vec3(-2.5f, 1.2f, 0.3f), Vec3(5.5f, 1.2f, 0.3f), Point3(0, 0, 0), Point3(4, 4, 4), &cells);  // clipped to the window
            CHECK(cells.size() == 4 && cells[0] == Point3(0, 1, 0));
            cells.clear();
            Raycast(Vec3(0.2f, 0.2f, 0.2f), Vec3(0.8f, 0.7f, 0.1f), Point3(0, 0, 0), Point3(4, 4, 4), &cells);  // inside one cell: nothing (Raycast.cpp:79-80)
            CHECK(cells.empty());
            CHECK(signum(-3) == -1.0f && signum(0) == 0.0f && mod(-0.25f, 1.0f) == 0.75f && intbound(0.25f, 1) == 0.75f && intbound(0.25f, -1) == 0.25f);
        }
        // ---- geometry/Interpolate.h, FixedPointFloat.h, threading/Threading.h
        {
            CHECK(LinearInterpolate(1.0f, 3.0f, 0.25f) == 1.5f && BilinearInterpolate(0.0f, 1.0f, 2.0f, 3.0f, 0.5f, 0.5f) == 1.5f);
            CHECK(FloatToFixedFloat16(-1000.0f) == 0 && FloatToFixedFloat16(5000.0f) == 65535 && FloatToUFixedFloat16(-3.0f) == 0);
            CHECK(std::fabs(FixedFloat16ToFloat(FloatToFixedFloat16(12.5f)) - 12.5f) < 0.04f && std::fabs(UFixedFloat16ToFloat(FloatToUFixedFloat16(12.5f)) - 12.5f) < 0.02f);
            std::vector<int> v(5000, 1);
            std::mutex mu;
            long sum = 0;
            parallel_for(v.begin(), v.end(), [&](int &x) { x += 1; std::lock_guard<std::mutex> g(mu); sum += x; }, 4, 1000);
            CHECK(sum == 10000);
            sum = 0;
            parallel_for(v.begin(), v.begin() + 3, [&](int &x) { sum += x; });  // fewer elements than one group: the calling thread alone
            CHECK(sum == 6);
        }
        // ---- AABB::Intersects(Plane) (AABB.cpp:37-70)
        {
            const AABB unit(Vec3(0, 0, 0), Vec3(1, 1, 1));
            CHECK(unit.Intersects(Plane(1.0f, 0.0f, 0.0f, -0.25f)) == Plane::IntersectionType::Intersects);
            CHECK(unit.Intersects(Plane(1.0f, 0.0f, 0.0f, 1.0f)) == Plane::IntersectionType::Outside);
            CHECK(unit.Intersects(Plane(1.0f, 0.0f, 0.0f, -2.0f)) == Plane::IntersectionType::Inside);
            CHECK(unit.Intersects(Plane(0.0f, 1.0f, 0.0f, -0.75f)) == Plane::IntersectionType::Intersects);  // between min + half extent and max
        }
        cm.PrintMemoryStatistics();
        printf("facade_surface ok: %zu candidates, %zu per-chunk integrations, %zu vertices\n", ids.size(), created, mine.vertices.size());
    } catch (const std::exception &e) {
        fprintf(stderr, "facade_surface: %s\n", e.what());
        return 1;
    }
    return 0;
}
