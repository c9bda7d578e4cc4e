// open_chisel/Chisel.h -- the reference's class surface (chisel::Chisel, ProjectionIntegrator, ChunkManager, Chunk,
// DistVoxel, ColorVoxel, Mesh) as thin host facades over the C ABI of libchisel_hip.so (include/chisel_hip.h).
//
// Same names, signatures, argument meaning and error behaviour as OpenChisel/open_chisel/include/open_chisel/*.h so
// that chisel_ros::ChiselServer (chisel_ros/src/ChiselServer.cpp) compiles against these headers unchanged:
//   - voxels and meshes live in HBM; ChunkManager::GetChunks() / GetChunk() / GetAllMeshes() hand out host mirrors: a Chunk
//     knows its id and geometry at once (ComputeBoundingBox, GetID, GetNumVoxels need no transfer -- chisel_ros walks all chunks
//     for their box centres every frame, ChiselServer.cpp:594-603) and fetches its voxels on the first GetVoxels() /
//     GetDistVoxel(); a mirror shows the map as it was when its voxels were fetched, callers hold ChunkPtrs only inside one
//     callback;
//   - GetChunk / GetMesh of an absent id throw std::out_of_range like the reference's unordered_map::at
//     (ChunkManager.h:84-87, 171-178); bool results stay bool; C-ABI failures other than NOT_FOUND become
//     std::runtime_error carrying chisel_hip_last_error().
// Nothing is computed here: every call forwards to the C ABI.
#ifndef CHISEL_HIP_FACADE_CHISEL_H_
#define CHISEL_HIP_FACADE_CHISEL_H_
#include <chisel_hip.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iterator>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "camera/PinholeCamera.h"
#include "geometry/AABB.h"
#include "geometry/Frustum.h"
#include "geometry/Geometry.h"
#include "mesh/Mesh.h"
#include "pointcloud/PointCloud.h"
#include "truncation/Truncator.h"
#include "weighting/Weighter.h"

namespace chisel {

typedef Eigen::Vector3i ChunkID;
#ifdef CHISEL_HIP_HAVE_EIGEN
typedef std::vector<ChunkID, Eigen::aligned_allocator<ChunkID>> ChunkIDList;
#else
typedef std::vector<ChunkID> ChunkIDList;
#endif
struct ChunkHasher {  // ChunkManager.h:40-52
    static constexpr size_t p1 = 73856093, p2 = 19349663, p3 = 8349279;
    std::size_t operator()(const ChunkID &key) const { return (key(0) * p1 ^ key(1) * p2 ^ key(2) * p3); }
};

class DistVoxel {  // DistVoxel.h:33-77.  The map's voxels are updated by integrate_kernel; these methods act on a host mirror.
  public:
    float GetSDF() const { return sdf; }
    float GetWeight() const { return weight; }
    void SetSDF(float v) { sdf = v; }
    void SetWeight(float v) { weight = v; }
    void Integrate(const float &distUpdate, const float &weightUpdate) {  // DistVoxel.h:52-60
        const float oldSDF = GetSDF(), oldWeight = GetWeight();
        const float newDist = (oldWeight * oldSDF + weightUpdate * distUpdate) / (weightUpdate + oldWeight);
        SetSDF(newDist);
        SetWeight(oldWeight + weightUpdate);
    }
    void Carve() { Reset(); }  // DistVoxel.h:62-66
    void Reset() {             // DistVoxel.h:68-72
        sdf = 99999.0f;
        weight = 0.0f;
    }
    float sdf = 99999.0f, weight = 0.0f;
};
class ColorVoxel {  // ColorVoxel.h:33-100
  public:
    uint8_t GetRed() const { return red; }
    uint8_t GetGreen() const { return green; }
    uint8_t GetBlue() const { return blue; }
    uint8_t GetWeight() const { return weight; }
    void SetRed(uint8_t v) { red = v; }
    void SetGreen(uint8_t v) { green = v; }
    void SetBlue(uint8_t v) { blue = v; }
    void SetWeight(uint8_t v) { weight = v; }
    void Reset() { red = green = blue = weight = 0; }
    static float Saturate(float value) { return std::min(std::max(value, 0.0f), 255.0f); }  // ColorVoxel.h:39-42
    uint8_t red = 0, green = 0, blue = 0, weight = 0;
};

inline void hip_check(int rc) {
    if (rc != CHISEL_HIP_OK) throw std::runtime_error(std::string("chisel_hip: ") + chisel_hip_last_error());
}
// the library on the loader's path must be the one these headers were written against (array lengths and signatures are the version's)
inline void hip_check_abi() {
    if (chisel_hip_abi_version() != CHISEL_HIP_ABI_VERSION)
        throw std::runtime_error("chisel_hip: libchisel_hip.so has ABI version " + std::to_string(chisel_hip_abi_version()) + ", these headers " +
                                 std::to_string(CHISEL_HIP_ABI_VERSION));
}

namespace hipfacade {  // the C structs of the boundary from the facade's value types
inline void Pose12(const Transform &T, float out[12]) {
    for (int r = 0; r < 3; r++) {
        for (int k = 0; k < 3; k++) out[4 * r + k] = T.linear()(r, k);
        out[4 * r + 3] = T.translation()(r);
    }
}
template <class DataType>
inline chisel_hip_depth_frame DepthFrame(const DepthImage<DataType> &img, const Transform &T, const PinholeCamera &cam) {
    static_assert(sizeof(DataType) == sizeof(float), "DepthImage<float>: convert 16UC1 millimetres on the caller's side as Conversions.h:140-150 does");
    chisel_hip_depth_frame f;
    std::memset(&f, 0, sizeof(f));
    f.depth = reinterpret_cast<const float *>(img.GetData());
    f.width = img.GetWidth();
    f.height = img.GetHeight();
    f.on_device = 0;
    Pose12(T, f.pose);
    f.fx = cam.GetIntrinsics().GetFx();
    f.fy = cam.GetIntrinsics().GetFy();
    f.cx = cam.GetIntrinsics().GetCx();
    f.cy = cam.GetIntrinsics().GetCy();
    f.near_plane = cam.GetNearPlane();
    f.far_plane = cam.GetFarPlane();
    return f;
}
template <class ColorType>
inline chisel_hip_color_frame ColorFrame(const ColorImage<ColorType> &img, const Transform &T, const PinholeCamera &cam) {
    static_assert(sizeof(ColorType) == 1, "ColorImage<uint8_t>");
    chisel_hip_color_frame c;
    std::memset(&c, 0, sizeof(c));
    c.color = reinterpret_cast<const uint8_t *>(img.GetData());
    c.width = img.GetWidth();
    c.height = img.GetHeight();
    c.channels = img.GetNumChannels();
    Pose12(T, c.pose);
    c.fx = cam.GetIntrinsics().GetFx();
    c.fy = cam.GetIntrinsics().GetFy();
    c.cx = cam.GetIntrinsics().GetCx();
    c.cy = cam.GetIntrinsics().GetCy();
    return c;
}
}  // namespace hipfacade

struct ChunkStatistics {  // Chunk.h:39-45
    size_t numKnownInside, numKnownOutside, numUnknown;
    float totalWeight;
};
class Chunk {  // Chunk.h:47-140: a host object; either free-standing (to be added to a map) or the mirror of a device-resident chunk
  public:
    EIGEN_MAKE_ALIGNED_OPERATOR_NEW
    // free-standing chunk with default voxels (Chunk.cpp:33-44); ChunkManager::AddChunk uploads it
    Chunk(const ChunkID &id, const Eigen::Vector3i &nv, float res, bool color)
        : ID(id), numVoxels(nv), voxelResolutionMeters(res), hasColor(color), map(nullptr), loaded(true), voxels((size_t)nv(0) * nv(1) * nv(2)) {
        if (color) colors.resize(voxels.size());
        origin = Vec3(nv(0) * id(0) * res, nv(1) * id(1) * res, nv(2) * id(2) * res);  // Chunk.cpp:43
    }
    // mirror of the chunk `id` of `m`: the voxels are fetched when they are first asked for
    Chunk(chisel_hip_map *m, const ChunkID &id, const Eigen::Vector3i &nv, float res, bool color)
        : ID(id), numVoxels(nv), voxelResolutionMeters(res), hasColor(color), map(m), loaded(false) {
        origin = Vec3(nv(0) * id(0) * res, nv(1) * id(1) * res, nv(2) * id(2) * res);
    }
    const ChunkID &GetID() const { return ID; }
    ChunkID &GetIDMutable() { return ID; }
    void SetID(const ChunkID &id) { ID = id; }
    const Eigen::Vector3i &GetNumVoxels() const { return numVoxels; }
    float GetVoxelResolutionMeters() const { return voxelResolutionMeters; }
    size_t GetTotalNumVoxels() const { return (size_t)numVoxels(0) * numVoxels(1) * numVoxels(2); }
    bool HasColors() const { return hasColor; }
    bool HasVoxels() const { return GetTotalNumVoxels() != 0; }
    const std::vector<DistVoxel> &GetVoxels() const { return Load().voxels; }
    const std::vector<ColorVoxel> &GetColorVoxels() const { return Load().colors; }
    size_t GetVoxelID(int x, int y, int z) const { return (z * numVoxels(2) + y) * numVoxels(0) + x; }  // Chunk.h:81-84 (sic)
    const DistVoxel &GetDistVoxel(size_t i) const { return Load().voxels.at(i); }
    DistVoxel &GetDistVoxelMutable(size_t i) { return const_cast<std::vector<DistVoxel> &>(Load().voxels).at(i); }
    const ColorVoxel &GetColorVoxel(size_t i) const { return Load().colors.at(i); }
    ColorVoxel &GetColorVoxelMutable(size_t i) { return const_cast<std::vector<ColorVoxel> &>(Load().colors).at(i); }
    const DistVoxel &GetDistVoxel(int x, int y, int z) const { return GetDistVoxel(GetVoxelID(x, y, z)); }
    const ColorVoxel &GetColorVoxel(int x, int y, int z) const { return GetColorVoxel(GetVoxelID(x, y, z)); }
    const Vec3 &GetOrigin() const { return origin; }
    bool IsCoordValid(int x, int y, int z) const {  // Chunk.h:106-109
        return x >= 0 && x < numVoxels(0) && y >= 0 && y < numVoxels(1) && z >= 0 && z < numVoxels(2);
    }
    // Chunk.cpp:46-63: (re)create the voxel arrays with default voxels -- of this host object; the map's voxels change through
    // ChunkManager::AddChunk, which uploads them
    void AllocateDistVoxels() {
        loaded = true;
        voxels.assign(GetTotalNumVoxels(), DistVoxel());
    }
    void AllocateColorVoxels() {
        loaded = true;
        hasColor = true;
        if (voxels.size() != GetTotalNumVoxels()) voxels.assign(GetTotalNumVoxels(), DistVoxel());
        colors.assign(GetTotalNumVoxels(), ColorVoxel());
    }
    void ComputeStatistics(ChunkStatistics *stats) const {  // Chunk.cpp:89-116 over this mirror (the whole map: ChunkManager::PrintMemoryStatistics)
        for (const DistVoxel &vox : Load().voxels) {
            const float weight = vox.GetWeight();
            if (weight > 0) {
                if (vox.GetSDF() < 0) stats->numKnownInside++;
                else stats->numKnownOutside++;
            } else {
                stats->numUnknown++;
            }
            stats->totalWeight += weight;
        }
    }
    Vec3 GetColorAt(const Vec3 &pos) const {  // Chunk.cpp:118-136: nearest voxel's colour in [0, 1], zero outside the chunk
        if (hasColor && ComputeBoundingBox().Contains(pos)) {
            const int x = static_cast<int>((pos(0) - origin(0)) / voxelResolutionMeters), y = static_cast<int>((pos(1) - origin(1)) / voxelResolutionMeters),
                      z = static_cast<int>((pos(2) - origin(2)) / voxelResolutionMeters);
            if (IsCoordValid(x, y, z)) {
                const ColorVoxel &c = GetColorVoxel(x, y, z);
                return Vec3(static_cast<float>(c.GetRed()) / 255.0f, static_cast<float>(c.GetGreen()) / 255.0f, static_cast<float>(c.GetBlue()) / 255.0f);
            }
        }
        return Vec3(0.0f, 0.0f, 0.0f);
    }
    AABB ComputeBoundingBox() const {  // Chunk.cpp:65-70
        const Vec3 pos = origin;
        const Vec3 size = numVoxels.cast<float>() * voxelResolutionMeters;
        return AABB(pos, pos + size);
    }
    Point3 GetVoxelCoords(const Vec3 &worldCoords) const {  // Chunk.cpp:72-81
        const float roundingFactor = 1.0f / voxelResolutionMeters;
        const Vec3 rel = worldCoords - origin;
        return Point3((int)std::floor(rel(0) * roundingFactor), (int)std::floor(rel(1) * roundingFactor), (int)std::floor(rel(2) * roundingFactor));
    }
  protected:
    friend class ChunkManager;
    friend class ProjectionIntegrator;
    const Chunk &Load() const {
        if (loaded) return *this;
        loaded = true;
        const size_t V = GetTotalNumVoxels();
        voxels.assign(V, DistVoxel());
        if (hasColor) colors.assign(V, ColorVoxel());
        std::vector<float> sdf(V), w(V);
        std::vector<uint8_t> rgbw(hasColor ? V * 4 : 0);
        const int v[3] = {ID(0), ID(1), ID(2)};
        const int rc = chisel_hip_download_chunk(map, v, sdf.data(), w.data(), hasColor ? rgbw.data() : nullptr);
        if (rc == CHISEL_HIP_ERR_NOT_FOUND) return *this;  // removed since the mirror was made: default voxels
        hip_check(rc);
        for (size_t i = 0; i < V; i++) {
            voxels[i].sdf = sdf[i];
            voxels[i].weight = w[i];
            if (hasColor) {
                colors[i].red = rgbw[4 * i];
                colors[i].green = rgbw[4 * i + 1];
                colors[i].blue = rgbw[4 * i + 2];
                colors[i].weight = rgbw[4 * i + 3];
            }
        }
        return *this;
    }
    ChunkID ID;
    Eigen::Vector3i numVoxels;
    float voxelResolutionMeters;
    bool hasColor;
    chisel_hip_map *map;
    mutable bool loaded;
    mutable std::vector<DistVoxel> voxels;
    mutable std::vector<ColorVoxel> colors;
    Vec3 origin;
};
typedef std::shared_ptr<Chunk> ChunkPtr;
typedef std::shared_ptr<const Chunk> ChunkConstPtr;
typedef std::unordered_map<ChunkID, ChunkPtr, ChunkHasher> ChunkMap;
typedef std::unordered_map<ChunkID, bool, ChunkHasher> ChunkSet;
typedef std::unordered_map<ChunkID, MeshPtr, ChunkHasher> MeshMap;

class ProjectionIntegrator {  // ProjectionIntegrator.h:36-232 (Integrate / IntegrateColor per chunk run on the GPU)
  public:
    ProjectionIntegrator() : carvingDist(0), enableVoxelCarving(false) {}
    ProjectionIntegrator(const TruncatorPtr &t, const WeighterPtr &w, float carvingDist_, bool enableCarving, const Vec3List &centroids_)
        : truncator(t), weighter(w), carvingDist(carvingDist_), enableVoxelCarving(enableCarving), centroids(centroids_) {}
    const TruncatorPtr &GetTruncator() const { return truncator; }
    void SetTruncator(const TruncatorPtr &v) { truncator = v; }
    const WeighterPtr &GetWeighter() const { return weighter; }
    void SetWeighter(const WeighterPtr &v) { weighter = v; }
    float GetCarvingDist() const { return carvingDist; }
    bool IsCarvingEnabled() const { return enableVoxelCarving; }
    void SetCarvingDist(float d) { carvingDist = d; }
    void SetCarvingEnabled(bool e) { enableVoxelCarving = e; }
    void SetCentroids(const Vec3List &c) { centroids = c; }  // kept for source compatibility: the kernels recompute centroids
    // ProjectionIntegrator.h:51-99 / :101-183: one frame into ONE chunk; true when a voxel changed.  The chunk is the mirror of a chunk
    // of a map (ChunkManager::GetChunk): the update runs on that map's GPU (chisel_hip_integrate_chunk) and the mirror's voxels are
    // fetched again when next asked for.  A free-standing chunk is carried through a map of its own for the call.
    template <class DataType>
    bool Integrate(const std::shared_ptr<const DepthImage<DataType>> &depthImage, const PinholeCamera &camera, const Transform &cameraPose, Chunk *chunk) const {
        const chisel_hip_depth_frame f = hipfacade::DepthFrame(*depthImage, cameraPose, camera);
        return IntegrateChunk(&f, nullptr, chunk);
    }
    template <class DataType, class ColorType>
    bool IntegrateColor(const std::shared_ptr<const DepthImage<DataType>> &depthImage, const PinholeCamera &depthCamera, const Transform &depthCameraPose,
                        const std::shared_ptr<const ColorImage<ColorType>> &colorImage, const PinholeCamera &colorCamera, const Transform &colorCameraPose,
                        Chunk *chunk) const {
        const chisel_hip_depth_frame f = hipfacade::DepthFrame(*depthImage, depthCameraPose, depthCamera);
        const chisel_hip_color_frame c = hipfacade::ColorFrame(*colorImage, colorCameraPose, colorCamera);
        return IntegrateChunk(&f, &c, chunk);
    }
    chisel_hip_integrator HipStruct() const {
        chisel_hip_integrator s;
        s.truncator_kind = truncator ? truncator->HipKind() : CHISEL_HIP_TRUNC_INVERSE;
        s.truncator_param = truncator ? truncator->HipParam() : 8.0f;
        s.weight = weighter ? weighter->HipWeight() : 1.0f;
        s.carving_enabled = enableVoxelCarving ? 1 : 0;
        s.carving_dist = carvingDist;
        return s;
    }
  protected:
    bool IntegrateChunk(const chisel_hip_depth_frame *f, const chisel_hip_color_frame *c, Chunk *chunk) const {
        if (!chunk) throw std::invalid_argument("ProjectionIntegrator::Integrate: null chunk");  // assert(chunk != nullptr), ProjectionIntegrator.h:54
        const chisel_hip_integrator in = HipStruct();
        const int v[3] = {chunk->ID(0), chunk->ID(1), chunk->ID(2)};
        int updated = 0;
        if (chunk->map) {
            hip_check(chisel_hip_set_integrator(chunk->map, &in));
            hip_check(chisel_hip_integrate_chunk(chunk->map, v, f, c, &updated));
            chunk->loaded = false;
            return updated != 0;
        }
        // free-standing: a map of its own for the length of the call
        chisel_hip_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        for (int k = 0; k < 3; k++) cfg.chunk_size[k] = chunk->numVoxels(k);
        cfg.voxel_resolution = chunk->voxelResolutionMeters;
        cfg.use_color = chunk->hasColor ? 1 : 0;
        cfg.device_id = -1;
        cfg.max_chunks = 8;
        cfg.n_shards = 1;
        chisel_hip_map *tmp = nullptr;
        hip_check_abi();
        hip_check(chisel_hip_create(&cfg, &tmp));
        const size_t V = chunk->GetTotalNumVoxels();
        std::vector<float> sdf(V), w(V);
        std::vector<uint8_t> rgbw(chunk->hasColor ? 4 * V : 0);
        for (size_t i = 0; i < V; i++) {
            sdf[i] = chunk->voxels[i].sdf;
            w[i] = chunk->voxels[i].weight;
            if (chunk->hasColor) {
                rgbw[4 * i] = chunk->colors[i].red; rgbw[4 * i + 1] = chunk->colors[i].green; rgbw[4 * i + 2] = chunk->colors[i].blue; rgbw[4 * i + 3] = chunk->colors[i].weight;
            }
        }
        int rc = chisel_hip_upload_chunk(tmp, v, sdf.data(), w.data(), chunk->hasColor ? rgbw.data() : nullptr);
        if (!rc) rc = chisel_hip_set_integrator(tmp, &in);
        if (!rc) rc = chisel_hip_integrate_chunk(tmp, v, f, c, &updated);
        if (!rc) rc = chisel_hip_download_chunk(tmp, v, sdf.data(), w.data(), chunk->hasColor ? rgbw.data() : nullptr);
        std::string err = rc ? chisel_hip_last_error() : "";
        chisel_hip_destroy(tmp);
        if (rc) throw std::runtime_error("chisel_hip: " + err);
        for (size_t i = 0; i < V; i++) {
            chunk->voxels[i].sdf = sdf[i];
            chunk->voxels[i].weight = w[i];
            if (chunk->hasColor) {
                chunk->colors[i].red = rgbw[4 * i]; chunk->colors[i].green = rgbw[4 * i + 1]; chunk->colors[i].blue = rgbw[4 * i + 2]; chunk->colors[i].weight = rgbw[4 * i + 3];
            }
        }
        return updated != 0;
    }
    TruncatorPtr truncator;
    WeighterPtr weighter;
    float carvingDist;
    bool enableVoxelCarving;
    Vec3List centroids;
};

class ChunkManager {  // ChunkManager.h:57-216 over one chisel_hip_map
  public:
    ChunkManager() {}
    // ChunkManager.h:62: a manager of its own (chisel::Chisel builds its own the same way and hands it the map it owns)
    ChunkManager(const Eigen::Vector3i &cs, float res, bool color) : chunkSize(cs), voxelResolutionMeters(res), useColor(color) {
        chisel_hip_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        for (int k = 0; k < 3; k++) cfg.chunk_size[k] = cs(k);
        cfg.voxel_resolution = res;
        cfg.use_color = color ? 1 : 0;
        cfg.device_id = -1;
        cfg.n_shards = 1;
        hip_check_abi();
        hip_check(chisel_hip_create(&cfg, &map));
        owned = std::shared_ptr<chisel_hip_map>(map, [](chisel_hip_map *m) { chisel_hip_destroy(m); });  // copies of the manager share the map
        CacheCentroids();
    }
    ChunkManager(chisel_hip_map *m, const Eigen::Vector3i &cs, float res, bool color) : map(m), chunkSize(cs), voxelResolutionMeters(res), useColor(color) {
        CacheCentroids();
    }
    virtual ~ChunkManager() {}
    void CacheCentroids() {  // ChunkManager.cpp:50-70 (the kernels recompute the centroids with the same two roundings; the table is for callers)
        const float res = voxelResolutionMeters, half = res * 0.5f;
        centroids.clear();
        for (int z = 0; z < chunkSize(2); z++)
            for (int y = 0; y < chunkSize(1); y++)
                for (int x = 0; x < chunkSize(0); x++) centroids.push_back(Vec3(x * res + half, y * res + half, z * res + half));
    }
    const Eigen::Vector3i &GetChunkSize() const { return chunkSize; }
    float GetResolution() const { return voxelResolutionMeters; }
    const Vec3List &GetCentroids() const { return centroids; }
    bool HasChunk(const ChunkID &id) const {
        const int v[3] = {id(0), id(1), id(2)};
        int out = 0;
        hip_check(chisel_hip_has_chunk(map, v, &out));
        return out != 0;
    }
    bool HasChunk(int x, int y, int z) const { return HasChunk(ChunkID(x, y, z)); }
    ChunkPtr GetChunk(const ChunkID &id) const {  // throws std::out_of_range like chunks.at(id) (ChunkManager.h:84-87)
        if (!HasChunk(id)) throw std::out_of_range("ChunkManager::GetChunk");
        return std::make_shared<Chunk>(map, id, chunkSize, voxelResolutionMeters, useColor);
    }
    ChunkPtr GetChunk(int x, int y, int z) const { return GetChunk(ChunkID(x, y, z)); }
    // Every chunk of the map, as mirrors that know their id and box and fetch voxels on demand (ChunkManager.h:65-73).  The mirrors are
    // kept between calls: chisel_ros walks this map after every frame (ChiselServer.cpp:594-603), and nothing is done at all while the
    // chunk set stands still (same number of chunks -- integration only adds -- and same topology epoch); when it has moved, ids that
    // disappeared are dropped and new ids get a mirror, the others keep theirs (with their voxels marked stale).
    const ChunkMap &GetChunks() const {
        int64_t n = 0;
        uint64_t epoch = 0;
        hip_check(chisel_hip_num_chunks(map, &n));
        hip_check(chisel_hip_topology_epoch(map, &epoch));
        if (chunksValid && n == chunksCount && epoch == chunksEpoch) {
            for (auto &c : chunks) c.second->loaded = false;  // voxels may have changed: fetched again on demand
            return chunks;
        }
        std::vector<int> ids((size_t)n * 3);
        if (n) hip_check(chisel_hip_list_chunks(map, ids.data(), n, &n));
        ChunkSet present;
        for (int64_t i = 0; i < n; i++) {
            const ChunkID id(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
            present[id] = true;
            auto it = chunks.find(id);
            if (it == chunks.end()) chunks[id] = std::make_shared<Chunk>(map, id, chunkSize, voxelResolutionMeters, useColor);
            else it->second->loaded = false;
        }
        for (auto it = chunks.begin(); it != chunks.end();)
            it = present.count(it->first) ? std::next(it) : chunks.erase(it);
        chunksValid = true;
        chunksCount = n;
        chunksEpoch = epoch;
        return chunks;
    }
    ChunkMap &GetMutableChunks() {
        GetChunks();
        return chunks;
    }
    int GetBucketSize() const { return (int)chunks.bucket_count(); }
    // ChunkManager.h:89-92: the chunk's voxels (a free-standing chunk's, or a mirror's as last fetched) become the map's
    ChunkMap::iterator AddChunk(const ChunkPtr &chunk) {
        const size_t V = chunk->GetTotalNumVoxels();
        std::vector<float> sdf(V), w(V);
        std::vector<uint8_t> rgbw(useColor ? V * 4 : 0);
        const std::vector<DistVoxel> &dv = chunk->GetVoxels();
        for (size_t i = 0; i < V; i++) {
            sdf[i] = dv[i].sdf;
            w[i] = dv[i].weight;
        }
        if (useColor && chunk->HasColors()) {
            const std::vector<ColorVoxel> &cv = chunk->GetColorVoxels();
            for (size_t i = 0; i < V; i++) {
                rgbw[4 * i] = cv[i].red; rgbw[4 * i + 1] = cv[i].green; rgbw[4 * i + 2] = cv[i].blue; rgbw[4 * i + 3] = cv[i].weight;
            }
        }
        const ChunkID id = chunk->GetID();
        const int v[3] = {id(0), id(1), id(2)};
        hip_check(chisel_hip_upload_chunk(map, v, sdf.data(), w.data(), useColor ? rgbw.data() : nullptr));
        return chunks.insert(std::make_pair(id, std::make_shared<Chunk>(map, id, chunkSize, voxelResolutionMeters, useColor))).first;
    }
    ChunkMap::iterator CreateChunk(const ChunkID &id) {  // ChunkManager.cpp:171-174
        return AddChunk(std::make_shared<Chunk>(id, chunkSize, voxelResolutionMeters, useColor));
    }
    ChunkID GetIDAt(const Vec3 &pos) const {  // ChunkManager.h:136-145
        const float rx = 1.0f / (chunkSize(0) * voxelResolutionMeters), ry = 1.0f / (chunkSize(1) * voxelResolutionMeters),
                    rz = 1.0f / (chunkSize(2) * voxelResolutionMeters);
        return ChunkID((int)std::floor(pos(0) * rx), (int)std::floor(pos(1) * ry), (int)std::floor(pos(2) * rz));
    }
    ChunkPtr GetChunkAt(const Vec3 &pos) const {  // ChunkManager.h:124-134
        const ChunkID id = GetIDAt(pos);
        if (HasChunk(id)) return GetChunk(id);
        return ChunkPtr();
    }
    // ChunkManager.cpp:575-607: the voxel that contains `pos` (nullptr when its chunk is not resident or the linear id is out of range);
    // the pointer stays valid until the next call of the same member (it points into a mirror kept here, one per member: asking
    // for the colour voxel of a position does not take the distance voxel away)
    const DistVoxel *GetDistanceVoxel(const Vec3 &pos) {
        const size_t i = VoxelAt(pos, distVoxelChunk);
        return i == (size_t)-1 ? nullptr : &distVoxelChunk->GetDistVoxel(i);
    }
    const ColorVoxel *GetColorVoxel(const Vec3 &pos) {
        const size_t i = VoxelAt(pos, colorVoxelChunk);
        return (i == (size_t)-1 || !colorVoxelChunk->HasColors()) ? nullptr : &colorVoxelChunk->GetColorVoxel(i);
    }
    // ChunkManager.cpp:72-89
    void GetChunkIDsIntersecting(const AABB &box, ChunkIDList *chunkList) {
        const ChunkID minID = GetIDAt(box.min);
        const ChunkID maxID = GetIDAt(box.max) + Eigen::Vector3i(1, 1, 1);
        for (int x = minID(0); x < maxID(0); x++)
            for (int y = minID(1); y < maxID(1); y++)
                for (int z = minID(2); z < maxID(2); z++) chunkList->push_back(ChunkID(x, y, z));
    }
    // ChunkManager.cpp:214-257 (chisel_hip_cloud_candidates: the ray walk over the chunk grid, on the GPU; ascending ids where the
    // reference has the order of an unordered_map)
    void GetChunkIDsIntersecting(const PointCloud &cloud, const Transform &cameraTransform, float truncation, float maxDist, ChunkIDList *chunkList) {
        chunkList->clear();
        chisel_hip_pointcloud pc;
        std::memset(&pc, 0, sizeof(pc));
        pc.n_points = (int64_t)cloud.GetPoints().size();
        pc.points = pc.n_points ? reinterpret_cast<const float *>(cloud.GetPoints().data()) : nullptr;
        hipfacade::Pose12(cameraTransform, pc.pose);
        pc.truncation = truncation;
        pc.max_dist = maxDist;
        int64_t n = 0;
        hip_check(chisel_hip_cloud_candidates(map, &pc, nullptr, 0, &n));
        std::vector<int> ids((size_t)std::max<int64_t>(n, 1) * 3);
        hip_check(chisel_hip_cloud_candidates(map, &pc, ids.data(), n, &n));
        for (int64_t i = 0; i < n; i++) chunkList->push_back(ChunkID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
    }
    // ChunkManager.cpp:182-212 (chisel_hip_candidates: the reference's range and plane test, its order)
    void GetChunkIDsIntersecting(const Frustum &frustum, ChunkIDList *chunkList) {
        float c[24], p[24];
        for (int i = 0; i < 8; i++)
            for (int k = 0; k < 3; k++) c[3 * i + k] = frustum.GetCorners()[i](k);
        const Plane *planes[6] = {&frustum.GetFarPlane(), &frustum.GetNearPlane(), &frustum.GetTopPlane(), &frustum.GetBottomPlane(),
                                  &frustum.GetLeftPlane(), &frustum.GetRightPlane()};
        for (int i = 0; i < 6; i++) {
            for (int k = 0; k < 3; k++) p[4 * i + k] = planes[i]->normal(k);
            p[4 * i + 3] = planes[i]->distance;
        }
        const int cs[3] = {chunkSize(0), chunkSize(1), chunkSize(2)};
        int64_t n = 0;
        hip_check(chisel_hip_candidates(c, p, cs, voxelResolutionMeters, nullptr, 0, &n));
        std::vector<int> ids((size_t)n * 3);
        if (n) hip_check(chisel_hip_candidates(c, p, cs, voxelResolutionMeters, ids.data(), n, &n));
        for (int64_t i = 0; i < n; i++) chunkList->push_back(ChunkID(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]));
    }
    // ChunkManager.cpp:381-447: marching cubes of one chunk into the caller's Mesh -- vertices, face normals, sequential indices, grids
    // (no colours, no gradient normals: those are ColorizeMesh / ComputeNormalsFromGradients below, as RecomputeMesh chains them)
    void GenerateMesh(const ChunkPtr &chunk, Mesh *mesh) { MeshChunk(chunk->GetID(), 0, mesh); }
    // ChunkManager.cpp:259-379: the triangles of ONE cube (voxel `index` of `chunk`, cube origin `coordinates`) appended to the caller's
    // Mesh -- vertices, face normals, indices from *nextMeshIndex on, a grid entry when the cube is occupied.  The voxels are the map's
    // (chisel_hip_mesh_cube), not those of the host mirror `chunk` points to; both members read neighbour chunks where the cube leaves
    // the chunk, the reference's "inside" form merely assumes it does not.
    void ExtractBorderVoxelMesh(const ChunkPtr &chunk, const Eigen::Vector3i &index, const Vec3 &coordinates, VertIndex *nextMeshIndex, Mesh *mesh) {
        const ChunkID id = chunk->GetID();
        const int c[3] = {id(0), id(1), id(2)}, v[3] = {index(0), index(1), index(2)};
        const float xyz[3] = {coordinates(0), coordinates(1), coordinates(2)};
        float ve[45], no[45];
        int nv = 0, occupied = 0;
        hip_check(chisel_hip_mesh_cube(map, c, v, xyz, ve, no, &nv, &occupied));
        for (int i = 0; i < nv; i++) {
            mesh->vertices.push_back(Vec3(ve[3 * i], ve[3 * i + 1], ve[3 * i + 2]));
            mesh->normals.push_back(Vec3(no[3 * i], no[3 * i + 1], no[3 * i + 2]));
            mesh->indices.push_back((*nextMeshIndex)++);
        }
        if (occupied) mesh->grids.push_back(coordinates);
    }
    void ExtractInsideVoxelMesh(const ChunkPtr &chunk, const Eigen::Vector3i &index, const Vec3 &coords, VertIndex *nextMeshIndex, Mesh *mesh) {
        ExtractBorderVoxelMesh(chunk, index, coords, nextMeshIndex, mesh);
    }
    void ColorizeMesh(Mesh *mesh) {  // ChunkManager.cpp:628-639
        const size_t n = mesh->vertices.size();
        mesh->colors.clear();
        mesh->colors.resize(n);
        if (!n) return;
        std::vector<float> v(3 * n), c(3 * n, 0.0f);
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) v[3 * i + k] = mesh->vertices[i](k);
        hip_check(chisel_hip_shade_vertices(map, v.data(), (int64_t)n, nullptr, c.data(), 2));
        for (size_t i = 0; i < n; i++) mesh->colors[i] = Vec3(c[3 * i], c[3 * i + 1], c[3 * i + 2]);
    }
    Vec3 InterpolateColor(const Vec3 &colorPos) {  // ChunkManager.cpp:501-573
        const float v[3] = {colorPos(0), colorPos(1), colorPos(2)};
        float c[3] = {0, 0, 0};
        hip_check(chisel_hip_shade_vertices(map, v, 1, nullptr, c, 2));
        return Vec3(c[0], c[1], c[2]);
    }
    void ComputeNormalsFromGradients(Mesh *mesh) {  // ChunkManager.cpp:609-626: a normal is replaced where the gradient lookup succeeds
        const size_t n = mesh->vertices.size();
        if (!n) return;
        std::vector<float> v(3 * n), nr(3 * n);
        for (size_t i = 0; i < n; i++)
            for (int k = 0; k < 3; k++) {
                v[3 * i + k] = mesh->vertices[i](k);
                nr[3 * i + k] = mesh->normals[i](k);
            }
        hip_check(chisel_hip_shade_vertices(map, v.data(), (int64_t)n, nr.data(), nullptr, 1));
        for (size_t i = 0; i < n; i++) mesh->normals[i] = Vec3(nr[3 * i], nr[3 * i + 1], nr[3 * i + 2]);
    }
    // ChunkManager.cpp:91-128 (the mutex of the reference's 16 worker threads has nothing to guard here)
    template <class Mutex>
    void RecomputeMesh(const ChunkID &chunkID, Mutex &) {
        const int v[3] = {chunkID(0), chunkID(1), chunkID(2)};
        hip_check(chisel_hip_recompute_mesh(map, v));
    }
    MeshPtr &GetMutableMesh(const ChunkID &chunkID) {  // ChunkManager.h:175-178
        allMeshes[chunkID] = GetMesh(chunkID);         // throws std::out_of_range like allMeshes.at(chunkID)
        return allMeshes[chunkID];
    }
    const MeshMap &GetAllMeshes() const {
        int64_t n = 0;
        hip_check(chisel_hip_list_meshes(map, nullptr, 0, &n));
        std::vector<int> ids((size_t)n * 3);
        if (n) hip_check(chisel_hip_list_meshes(map, ids.data(), n, &n));
        allMeshes.clear();
        for (int64_t i = 0; i < n; i++) {
            const ChunkID id(ids[3 * i], ids[3 * i + 1], ids[3 * i + 2]);
            allMeshes[id] = GetMesh(id);
        }
        return allMeshes;
    }
    bool HasMesh(const ChunkID &id) const {
        const int v[3] = {id(0), id(1), id(2)};
        int64_t nv = 0, ng = 0;
        return chisel_hip_mesh_size(map, v, &nv, &ng) == CHISEL_HIP_OK;
    }
    MeshPtr GetMesh(const ChunkID &id) const {  // throws std::out_of_range like allMeshes.at(id) (ChunkManager.h:171-178)
        const int v[3] = {id(0), id(1), id(2)};
        int64_t nv = 0, ng = 0;
        const int rc = chisel_hip_mesh_size(map, v, &nv, &ng);
        if (rc == CHISEL_HIP_ERR_NOT_FOUND) throw std::out_of_range("ChunkManager::GetMesh");
        hip_check(rc);
        std::vector<float> ve((size_t)nv * 3), no((size_t)nv * 3), co(useColor ? (size_t)nv * 3 : 0), gr((size_t)ng * 3);
        hip_check(chisel_hip_download_mesh(map, v, ve.data(), no.data(), useColor ? co.data() : nullptr, gr.data()));
        MeshPtr mesh = std::make_shared<Mesh>();
        for (int64_t i = 0; i < nv; i++) {
            mesh->vertices.push_back(Vec3(ve[3 * i], ve[3 * i + 1], ve[3 * i + 2]));
            mesh->normals.push_back(Vec3(no[3 * i], no[3 * i + 1], no[3 * i + 2]));
            if (useColor) mesh->colors.push_back(Vec3(co[3 * i], co[3 * i + 1], co[3 * i + 2]));
            mesh->indices.push_back((size_t)i);  // MarchingCubes.h:92-94: sequential
        }
        for (int64_t i = 0; i < ng; i++) mesh->grids.push_back(Vec3(gr[3 * i], gr[3 * i + 1], gr[3 * i + 2]));
        return mesh;
    }
    bool RemoveChunk(const ChunkID &id) {  // ChunkManager.h:99-108
        if (!HasChunk(id)) return false;
        const int v[3] = {id(0), id(1), id(2)};
        hip_check(chisel_hip_garbage_collect(map, v, 1));
        chunks.erase(id);
        return true;
    }
    bool RemoveChunk(const ChunkPtr &chunk) { return RemoveChunk(chunk->GetID()); }  // ChunkManager.h:110-113
    void RemoveChunk(const ChunkMap::iterator &it) {                                // ChunkManager.h:94-97
        const ChunkID id = it->first;
        RemoveChunk(id);
    }
    bool GetUseColor() const { return useColor; }
    MeshMap &GetAllMutableMeshes() {
        GetAllMeshes();
        return allMeshes;
    }
    // ChunkManager.cpp:130-169: marching cubes of the given chunks now (Chisel::UpdateMeshes passes meshesToUpdate)
    void RecomputeMeshes(const ChunkSet &which) {
        std::vector<int> ids;
        for (const std::pair<const ChunkID, bool> &c : which) {
            ids.push_back(c.first(0)); ids.push_back(c.first(1)); ids.push_back(c.first(2));
        }
        hip_check(chisel_hip_update_meshes_of(map, ids.data(), (int)which.size()));
    }
    void PrintMemoryStatistics() const {  // ChunkManager.cpp:641-678: the same three lines, the census from one reduction kernel
        chisel_hip_statistics st;
        hip_check(chisel_hip_memory_statistics(map, &st));
        const float big = std::numeric_limits<float>::max();
        float lo[3] = {big, big, big}, hi[3] = {-big, -big, -big};
        for (int a = 0; a < 3 && st.n_chunks; a++) {
            lo[a] = (float)(chunkSize(a) * st.id_min[a]) * voxelResolutionMeters;                                                  // Chunk.cpp:43
            hi[a] = (float)(chunkSize(a) * st.id_max[a]) * voxelResolutionMeters + (float)chunkSize(a) * voxelResolutionMeters;   // Chunk.cpp:65-70
        }
        float numVoxels[3];
        for (int a = 0; a < 3; a++) numVoxels[a] = ((hi[a] - lo[a]) * 0.5f) * 2 / voxelResolutionMeters;                         // AABB::GetExtents
        const float totalNum = numVoxels[0] * numVoxels[1] * numVoxels[2];
        const float maxMemory = totalNum * 16 / 1000000.0f;  // sizeof(DistVoxel) of the reference build (vptr + two floats, padded)
        const size_t currentNum = (size_t)st.n_chunks * (size_t)(chunkSize(0) * chunkSize(1) * chunkSize(2));
        const float currentMemory = currentNum * 16 / 1000000.0f;
        std::printf("Num Unknown: %lu, Num KnownIn: %lu, Num KnownOut: %lu Weight: %f\n", (unsigned long)st.n_unknown, (unsigned long)st.n_known_inside,
                    (unsigned long)st.n_known_outside, (float)st.total_weight);
        std::printf("Bounds: %f %f %f %f %f %f\n", lo[0], lo[1], lo[2], hi[0], hi[1], hi[2]);
        std::printf("Theoretical max (MB): %f, Current (MB): %f\n", maxMemory, currentMemory);
    }
    bool GetSDF(const Vec3 &pos, double *dist) const {  // ChunkManager.cpp:476-499
        const float p[3] = {pos(0), pos(1), pos(2)};
        int found = 0;
        hip_check(chisel_hip_get_sdf(map, p, dist, &found));
        return found != 0;
    }
    bool GetSDFAndGradient(const Vec3 &pos, double *dist, Vec3 *grad) const {  // ChunkManager.cpp:449-474
        const float p[3] = {pos(0), pos(1), pos(2)};
        float g[3] = {0, 0, 0};
        int found = 0;
        hip_check(chisel_hip_get_sdf_and_gradient(map, p, dist, g, &found));
        if (found && grad) *grad = Vec3(g[0], g[1], g[2]);
        return found != 0;
    }
    void Reset() {  // ChunkManager.cpp:176-180
        hip_check(chisel_hip_reset(map));
        chunks.clear();
        allMeshes.clear();
    }
    chisel_hip_map *HipMap() const { return map; }
  protected:
    size_t VoxelAt(const Vec3 &pos, ChunkPtr &voxelChunk) {  // GetChunkAt + GetVoxelCoords + GetVoxelID (ChunkManager.cpp:575-607); -1: none
        const ChunkID id = GetIDAt(pos);
        if (!HasChunk(id)) return (size_t)-1;
        voxelChunk = std::make_shared<Chunk>(map, id, chunkSize, voxelResolutionMeters, useColor);
        const Point3 c = voxelChunk->GetVoxelCoords(pos);
        const long i = ((long)c(2) * chunkSize(2) + c(1)) * chunkSize(0) + c(0);  // Chunk::GetVoxelID, Chunk.h:81-84 (sic)
        if (i < 0 || i >= (long)voxelChunk->GetTotalNumVoxels()) return (size_t)-1;
        return (size_t)i;
    }
    void MeshChunk(const ChunkID &id, int stages, Mesh *mesh) {
        mesh->Clear();
        const int v[3] = {id(0), id(1), id(2)};
        const int64_t V = (int64_t)chunkSize(0) * chunkSize(1) * chunkSize(2), cap = 15 * V;
        std::vector<float> ve((size_t)cap * 3), no((size_t)cap * 3), co(useColor ? (size_t)cap * 3 : 0), gr((size_t)V * 3);
        int64_t nv = 0, ng = 0;
        hip_check(chisel_hip_generate_mesh(map, v, stages, cap, V, ve.data(), no.data(), useColor ? co.data() : nullptr, gr.data(), &nv, &ng));
        for (int64_t i = 0; i < nv; i++) {
            mesh->vertices.push_back(Vec3(ve[3 * i], ve[3 * i + 1], ve[3 * i + 2]));
            mesh->normals.push_back(Vec3(no[3 * i], no[3 * i + 1], no[3 * i + 2]));
            if ((stages & 2) && useColor) mesh->colors.push_back(Vec3(co[3 * i], co[3 * i + 1], co[3 * i + 2]));
            mesh->indices.push_back((size_t)i);
        }
        for (int64_t i = 0; i < ng; i++) mesh->grids.push_back(Vec3(gr[3 * i], gr[3 * i + 1], gr[3 * i + 2]));
    }
    std::shared_ptr<chisel_hip_map> owned;  // set when this manager created the map itself (the three-argument constructor)
    ChunkPtr distVoxelChunk, colorVoxelChunk;  // the mirrors GetDistanceVoxel / GetColorVoxel last pointed into
    mutable bool chunksValid = false;
    mutable int64_t chunksCount = 0;
    mutable uint64_t chunksEpoch = 0;
    chisel_hip_map *map = nullptr;
    Eigen::Vector3i chunkSize;
    float voxelResolutionMeters = 0.0f;
    bool useColor = false;
    Vec3List centroids;
    mutable ChunkMap chunks;    // host mirrors, rebuilt on request
    mutable MeshMap allMeshes;
};

class Chisel {  // Chisel.h:38-230
  public:
    Chisel() : map(nullptr) {}
    Chisel(const Eigen::Vector3i &chunkSize, float voxelResolution, bool useColor) : map(nullptr) {
        chisel_hip_config cfg;
        std::memset(&cfg, 0, sizeof(cfg));
        for (int k = 0; k < 3; k++) cfg.chunk_size[k] = chunkSize(k);
        cfg.voxel_resolution = voxelResolution;
        cfg.use_color = useColor ? 1 : 0;
        cfg.device_id = -1;
        cfg.n_shards = 1;
        // CHISEL_HIP_DEVICES=0,1,2,3 (environment of the node, e.g. in the launch file): the map is spread over these GPUs inside
        // this process (chisel_hip_create_group); every call below stays the same
        std::vector<int> devices;
        if (const char *env = std::getenv("CHISEL_HIP_DEVICES")) {
            for (const char *p = env; *p;) {
                char *end = nullptr;
                const long d = std::strtol(p, &end, 10);
                if (end == p) break;
                devices.push_back((int)d);
                p = (*end == ',') ? end + 1 : end;
            }
        }
        hip_check_abi();
        if (devices.size() > 1) hip_check(chisel_hip_create_group(&cfg, devices.data(), (int)devices.size(), &map));
        else {
            if (devices.size() == 1) cfg.device_id = devices[0];
            hip_check(chisel_hip_create(&cfg, &map));
        }
        chunkManager = ChunkManager(map, chunkSize, voxelResolution, useColor);
    }
    virtual ~Chisel() {
        if (map) chisel_hip_destroy(map);
    }
    Chisel(const Chisel &) = delete;
    Chisel &operator=(const Chisel &) = delete;

    const ChunkManager &GetChunkManager() const { return chunkManager; }
    ChunkManager &GetMutableChunkManager() { return chunkManager; }
    // Chisel.h:52-55.  A ChunkManager is a view of one device map: a manager taken from another Chisel would make this object
    // integrate into one map and answer from the other, so only views of this map are accepted.
    void SetChunkManager(const ChunkManager &manager) {
        if (manager.HipMap() != map) throw std::invalid_argument("chisel-hip: SetChunkManager needs a ChunkManager of this Chisel's map");
        chunkManager = manager;
    }

    template <class DataType>
    void IntegrateDepthScan(const ProjectionIntegrator &integrator, const std::shared_ptr<const DepthImage<DataType>> &depthImage,
                            const Transform &extrinsic, const PinholeCamera &camera) {  // Chisel.h:59-112
        static_assert(sizeof(DataType) == sizeof(float), "DepthImage<float>: convert 16UC1 millimetres on the caller's side as Conversions.h:140-150 does");
        const chisel_hip_integrator in = integrator.HipStruct();
        hip_check(chisel_hip_set_integrator(map, &in));
        chisel_hip_depth_frame f = hipfacade::DepthFrame(*depthImage, extrinsic, camera);
        hip_check(chisel_hip_integrate_depth(map, &f));
        hip_check(chisel_hip_meshes_to_update_prefetch(map, updateCursor));  // (GetMeshesToUpdate is what chisel_ros asks next: its listing rides on the wait below)
        hip_check(chisel_hip_synchronize(map));  // the reference returns with every voxel update visible
    }
    template <class DataType, class ColorType>
    void IntegrateDepthScanColor(const ProjectionIntegrator &integrator, const std::shared_ptr<const DepthImage<DataType>> &depthImage,
                                 const Transform &depthExtrinsic, const PinholeCamera &depthCamera,
                                 const std::shared_ptr<const ColorImage<ColorType>> &colorImage, const Transform &colorExtrinsic,
                                 const PinholeCamera &colorCamera) {  // Chisel.h:114-213
        static_assert(sizeof(DataType) == sizeof(float) && sizeof(ColorType) == 1, "DepthImage<float>, ColorImage<uint8_t>");
        const chisel_hip_integrator in = integrator.HipStruct();
        hip_check(chisel_hip_set_integrator(map, &in));
        chisel_hip_depth_frame f = hipfacade::DepthFrame(*depthImage, depthExtrinsic, depthCamera);
        chisel_hip_color_frame c;
        std::memset(&c, 0, sizeof(c));
        c.color = reinterpret_cast<const uint8_t *>(colorImage->GetData());
        c.width = colorImage->GetWidth();
        c.height = colorImage->GetHeight();
        c.channels = colorImage->GetNumChannels();
        Pose12(colorExtrinsic, c.pose);
        c.fx = colorCamera.GetIntrinsics().GetFx();
        c.fy = colorCamera.GetIntrinsics().GetFy();
        c.cx = colorCamera.GetIntrinsics().GetCx();
        c.cy = colorCamera.GetIntrinsics().GetCy();
        hip_check(chisel_hip_integrate_depth_color(map, &f, &c));
        hip_check(chisel_hip_meshes_to_update_prefetch(map, updateCursor));  // (GetMeshesToUpdate is what chisel_ros asks next, ChiselServer.cpp:346: its listing rides on the wait below)
        hip_check(chisel_hip_synchronize(map));
    }
    // Chisel.cpp:107-157 (fusion_mode = PointCloud, ChiselServer.cpp:523; CVIDS itself launches DepthImage mode, sample.launch:21)
    void IntegratePointCloud(const ProjectionIntegrator &integrator, const PointCloud &cloud, const Transform &extrinsic, float truncation,
                             float maxDist) {
        static_assert(sizeof(Vec3) == 3 * sizeof(float), "Vec3 is three packed floats");
        const chisel_hip_integrator in = integrator.HipStruct();
        hip_check(chisel_hip_set_integrator(map, &in));
        chisel_hip_pointcloud pc;
        std::memset(&pc, 0, sizeof(pc));
        pc.n_points = (int64_t)cloud.GetPoints().size();
        pc.points = pc.n_points ? reinterpret_cast<const float *>(cloud.GetPoints().data()) : nullptr;
        if (cloud.HasColor()) {
            // the reference reads colors[i] for the i-th point that passes the depth limit (ProjectionIntegrator.cpp:124): one colour per point
            if (cloud.GetColors().size() < cloud.GetPoints().size()) throw std::invalid_argument("chisel-hip: PointCloud with fewer colours than points");
            pc.colors = reinterpret_cast<const float *>(cloud.GetColors().data());
        }
        Pose12(extrinsic, pc.pose);
        pc.truncation = truncation;
        pc.max_dist = maxDist;
        hip_check(chisel_hip_integrate_pointcloud(map, &pc));
        hip_check(chisel_hip_synchronize(map));  // the reference returns with every voxel update visible
    }
    void UpdateMeshes() { hip_check(chisel_hip_update_meshes(map, 0)); }  // Chisel.cpp:50-59 (every 10th call recomputes)
    void GarbageCollect(const ChunkIDList &chunks) {  // Chisel.cpp:61-67
        std::vector<int> ids;
        for (const ChunkID &c : chunks) {
            ids.push_back(c(0)); ids.push_back(c(1)); ids.push_back(c(2));
        }
        hip_check(chisel_hip_garbage_collect(map, ids.data(), (int)chunks.size()));
    }
    bool SaveAllMeshesToPLY(const std::string &filename) {  // Chisel.cpp:69-105
        const int rc = chisel_hip_save_ply(map, filename.c_str());
        if (rc == CHISEL_HIP_ERR_IO) return false;
        hip_check(rc);
        return true;
    }
    void Reset() {  // Chisel.cpp:44-48
        hip_check(chisel_hip_reset(map));
        meshesToUpdate.clear();
    }
    // Chisel.h:220-223.  chisel_ros reads it after every frame (ChiselServer.cpp:346): the copy kept here is brought up to date with
    // what joined the set since the previous call (chisel_hip_meshes_to_update_since), not rebuilt.
    const ChunkSet &GetMeshesToUpdate() const {
        if (updateIds.empty()) updateIds.resize(3 * 8192);
        for (;;) {
            int64_t n = 0;
            int cleared = 0;
            hip_check(chisel_hip_meshes_to_update_since(map, updateCursor, updateIds.data(), (int64_t)(updateIds.size() / 3), &n, &cleared));
            if ((size_t)n * 3 > updateIds.size()) {
                updateIds.resize((size_t)n * 3 + 3 * 1024);
                continue;
            }
            if (cleared) meshesToUpdate.clear();
            for (int64_t i = 0; i < n; i++) meshesToUpdate[ChunkID(updateIds[3 * i], updateIds[3 * i + 1], updateIds[3 * i + 2])] = true;
            return meshesToUpdate;
        }
    }
    chisel_hip_map *HipMap() const { return map; }

  protected:
    static void Pose12(const Transform &T, float out[12]) { hipfacade::Pose12(T, out); }
    chisel_hip_map *map;
    ChunkManager chunkManager;
    mutable ChunkSet meshesToUpdate;
    mutable uint64_t updateCursor[2] = {0, 0};
    mutable std::vector<int> updateIds;
};
typedef std::shared_ptr<Chisel> ChiselPtr;
typedef std::shared_ptr<const Chisel> ChiselConstPtr;

}  // namespace chisel
#endif
