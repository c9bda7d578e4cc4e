// replay.cpp -- a ROS-free caller of the chisel::* facade that walks a recorded depth(+colour) stream through the per-frame
// sequence of chisel_ros::ChiselServer and, on the way, compiles every chisel::* call site of chisel_ros
// against the facade headers:
//
//   CallbackAll                ChiselServer.cpp:233-244   colour info -> depth info -> colour image -> depth image
//   Set*CameraInfo / Set*Image ChiselServer.cpp:246-296, 369-421; Conversions.h:107-231 (16UC1 millimetres -> metres, BGR8 copy,
//                              tf -> chisel::Transform through a quaternion and Transform::inverse())
//   DepthImageCallback         ChiselServer.cpp:297-367   integrate, chunk boxes, meshes when none are pending, pose, frustum
//   IntegrateLastDepthImage    ChiselServer.cpp:489-516   IntegrateDepthScan[Color] -> latest chunk boxes -> frustum -> UpdateMeshes
//   Publish*                   ChiselServer.cpp:97-180, 534-605, 607-716
//   SaveMesh / GetAllChunks / Reset services  ChiselServer.cpp:718-740, 426-432; Serialization.h:31-84 (FillChunkMessage)
//
// Written against the reference's call sites, not copied from them: the ROS plumbing (node handle, topics, tf listener) is
// replaced by ros_stubs.h and a file reader.  Recording format ("CVIDSRC1", little endian):
//   char magic[8]; int32 n_frames, width, height, depth_encoding (0 = 32FC1 metres, 1 = 16UC1 millimetres), color_channels (0, 1, 3, 4),
//   reserved[3]; then per frame: double P[4] (CameraInfo.P[0], P[5], P[2], P[6]), double tf_origin[3], double tf_rotation[4] (x y z w:
//   the camera-frame <- base-frame transform tf would return), the depth image, the colour image.
//
//   replay <recording> <out_prefix> [chunk_edge voxel_res use_color near far truncation_scale carving_dist]
// writes <out_prefix>.map (id + sdf/weight pairs + rgbw per chunk, ascending id), <out_prefix>.poses (12 floats per frame: the
// camera->world poses handed to the library), <out_prefix>.ply, and prints what each publisher last carried.
#include <open_chisel/Chisel.h>
#include <open_chisel/truncation/InverseTruncator.h>
#include <open_chisel/weighting/ConstantWeighter.h>

#include <algorithm>
#include <cassert>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "ros_stubs.h"

namespace replay {

typedef float DepthData;
typedef uint8_t ColorData;

// ---- Conversions.h:140-231 ----------------------------------------------------------------------------------------------
template <class DataType>
void ROSImgToDepthImg(const sensor_msgs::ImageConstPtr &image, chisel::DepthImage<DataType> *depthImage) {
    assert(depthImage->GetHeight() == (int)image->height && depthImage->GetWidth() == (int)image->width);
    DataType *out = depthImage->GetMutableData();
    const int total = (int)(image->width * image->height);
    if (image->encoding == "32FC1") {
        const DataType *in = reinterpret_cast<const DataType *>(image->data.data());
        for (int i = 0; i < total; i++) out[i] = in[i];
    } else if (image->encoding == "16UC1") {  // millimetres
        const uint16_t *in = reinterpret_cast<const uint16_t *>(image->data.data());
        for (int i = 0; i < total; i++) out[i] = (1.0f / 1000.0f) * in[i];
    } else {
        std::fprintf(stderr, "Unrecognized depth image format.\n");
    }
}
template <class DataType>
chisel::ColorImage<DataType> *ROSImgToColorImg(const sensor_msgs::ImageConstPtr &image) {
    size_t numChannels = 0;
    if (image->encoding == "mono8") numChannels = 1;
    else if (image->encoding == "bgr8" || image->encoding == "rgb8") numChannels = 3;
    else if (image->encoding == "bgra8") numChannels = 4;
    chisel::ColorImage<DataType> *out = new chisel::ColorImage<DataType>(image->width, image->height, numChannels);
    if (image->step / image->width != numChannels * sizeof(DataType)) {
        std::fprintf(stderr, "Inconsistent channel width\n");
        return out;
    }
    const DataType *in = reinterpret_cast<const DataType *>(image->data.data());
    DataType *dst = out->GetMutableData();
    const int total = (int)(image->width * image->height * numChannels);
    for (int i = 0; i < total; i++) dst[i] = in[i];
    return out;
}
inline chisel::Transform RosTfToChiselTf(const tf::StampedTransform &tf) {
    chisel::Transform transform;
    transform.translation()(0) = tf.getOrigin().x();
    transform.translation()(1) = tf.getOrigin().y();
    transform.translation()(2) = tf.getOrigin().z();
    chisel::Quaternion quat;
    quat.x() = tf.getRotation().x();
    quat.y() = tf.getRotation().y();
    quat.z() = tf.getRotation().z();
    quat.w() = tf.getRotation().w();
    transform.linear() = quat.toRotationMatrix();
    return transform.inverse();
}
inline chisel::PinholeCamera RosCameraToChiselCamera(const sensor_msgs::CameraInfoConstPtr &camera) {
    chisel::PinholeCamera cam;
    chisel::Intrinsics intrinsics;
    intrinsics.SetFx(camera->P[0]);
    intrinsics.SetFy(camera->P[5]);
    intrinsics.SetCx(camera->P[2]);
    intrinsics.SetCy(camera->P[6]);
    cam.SetIntrinsics(intrinsics);
    cam.SetWidth(camera->width);
    cam.SetHeight(camera->height);
    return cam;
}

// ---- Serialization.h:31-84 (the reference's own bit packing, kept as it is: it loses data -- the lossless counterpart is
// chisel_hip_save_map) --------------------------------------------------------------------------------------------------------
inline void FillChunkMessage(chisel::ChunkConstPtr chunk, chisel_ros::ChunkMessage *message) {
    chisel::ChunkHasher hasher;
    assert(message != nullptr);
    message->header.stamp = ros::Time::now();
    const chisel::ChunkID id = chunk->GetID();
    message->ID_x = id.x();
    message->ID_y = id.y();
    message->ID_z = id.z();
    message->spatial_hash = hasher(id);
    message->resolution_meters = chunk->GetVoxelResolutionMeters();
    const Eigen::Vector3i size = chunk->GetNumVoxels();
    message->num_voxels_x = size.x();
    message->num_voxels_y = size.y();
    message->num_voxels_z = size.z();
    message->distance_data.reserve(chunk->GetTotalNumVoxels());
    if (chunk->HasColors()) message->color_data.reserve(chunk->GetTotalNumVoxels());
    for (const chisel::DistVoxel &voxel : chunk->GetVoxels()) {
        float sdf = voxel.GetSDF(), weight = voxel.GetWeight();
        uint32_t a, b;
        std::memcpy(&a, &sdf, 4);
        std::memcpy(&b, &weight, 4);
        message->distance_data.push_back(a | b << sizeof(uint32_t));
    }
    for (const chisel::ColorVoxel &voxel : chunk->GetColorVoxels())
        message->color_data.push_back(static_cast<uint32_t>(voxel.GetRed()) | static_cast<uint32_t>(voxel.GetBlue()) << sizeof(uint8_t) |
                                      static_cast<uint32_t>(voxel.GetGreen()) << 2 * sizeof(uint8_t) |
                                      static_cast<uint32_t>(voxel.GetBlue()) << 3 * sizeof(uint8_t) |
                                      static_cast<uint32_t>(voxel.GetWeight()) << 4 * sizeof(uint8_t));
}

// ---- ChiselServer ---------------------------------------------------------------------------------------------------------------
struct RosCameraTopic {
    chisel::PinholeCamera cameraModel;
    chisel::Transform lastPose;
    ros::Time lastImageTimestamp;
    bool gotPose = false, gotInfo = false, gotImage = false;
    LastMessage<visualization_msgs::Marker> frustumPublisher;
    LastMessage<geometry_msgs::PoseStamped> lastPosePublisher;
};

class Server {
  public:
    Server(int chunkSize, float resolution, bool color, float nearPlane, float farPlane)
        : useColor(color), nearPlaneDist(nearPlane), farPlaneDist(farPlane) {
        chiselMap.reset(new chisel::Chisel(Eigen::Vector3i(chunkSize, chunkSize, chunkSize), resolution, color));  // ChiselServer.cpp:46-54
    }
    chisel::ChiselPtr GetChiselMap() { return chiselMap; }

    void SetupProjectionIntegrator(chisel::TruncatorPtr truncator, uint16_t weight, bool useCarving, float carvingDist) {  // :480-487
        projectionIntegrator.SetCentroids(GetChiselMap()->GetChunkManager().GetCentroids());
        projectionIntegrator.SetTruncator(truncator);
        projectionIntegrator.SetWeighter(chisel::WeighterPtr(new chisel::ConstantWeighter(weight)));
        projectionIntegrator.SetCarvingDist(carvingDist);
        projectionIntegrator.SetCarvingEnabled(useCarving);
    }

    void CallbackAll(sensor_msgs::ImageConstPtr depth_image, sensor_msgs::CameraInfoConstPtr depth_info, sensor_msgs::ImageConstPtr color_image,
                     sensor_msgs::CameraInfoConstPtr color_info, const tf::StampedTransform &depth_tf, const tf::StampedTransform &color_tf) {  // :233-244
        if (useColor) SetColorCameraInfo(color_info);
        SetDepthCameraInfo(depth_info);
        if (useColor) ColorImageCallback(color_image, color_tf);
        DepthImageCallback(depth_image, depth_tf);
    }

    void SetDepthCameraInfo(const sensor_msgs::CameraInfoConstPtr &info) {  // :260-269
        depthCamera.cameraModel = RosCameraToChiselCamera(info);
        depthCamera.cameraModel.SetNearPlane(nearPlaneDist);
        depthCamera.cameraModel.SetFarPlane(farPlaneDist);
        depthCamera.gotInfo = true;
    }
    void SetColorCameraInfo(const sensor_msgs::CameraInfoConstPtr &info) {  // :369-377
        colorCamera.cameraModel = RosCameraToChiselCamera(info);
        colorCamera.cameraModel.SetNearPlane(nearPlaneDist);
        colorCamera.cameraModel.SetFarPlane(farPlaneDist);
        colorCamera.gotInfo = true;
    }
    void ColorImageCallback(sensor_msgs::ImageConstPtr colorImage, const tf::StampedTransform &tf) {  // :379-421
        if (!lastColorImage.get()) lastColorImage.reset(ROSImgToColorImg<ColorData>(colorImage));
        else lastColorImage.reset(ROSImgToColorImg<ColorData>(colorImage));
        colorCamera.lastImageTimestamp = colorImage->header.stamp;
        colorCamera.gotImage = true;
        colorCamera.gotPose = true;
        colorCamera.lastPose = RosTfToChiselTf(tf);
    }
    void SetDepthImage(const sensor_msgs::ImageConstPtr &img) {  // :282-295
        if (!lastDepthImage.get()) lastDepthImage.reset(new chisel::DepthImage<DepthData>(img->width, img->height));
        ROSImgToDepthImg(img, lastDepthImage.get());
        depthCamera.lastImageTimestamp = img->header.stamp;
        depthCamera.gotImage = true;
    }
    void DepthImageCallback(sensor_msgs::ImageConstPtr depthImage, const tf::StampedTransform &tf) {  // :297-367
        SetDepthImage(depthImage);
        depthCamera.gotPose = true;
        depthCamera.lastPose = RosTfToChiselTf(tf);
        hasNewData = true;
        IntegrateLastDepthImage();
        PublishChunkBoxes();
        if (chiselMap->GetMeshesToUpdate().size() == 0) PublishMeshes();
        PublishDepthPose();
        PublishDepthFrustum();
        if (useColor) {
            PublishColorPose();
            PublishColorFrustum();
        }
    }
    void IntegrateLastDepthImage() {  // :489-516
        if (depthCamera.gotInfo && depthCamera.gotPose && lastDepthImage.get()) {
            if (useColor)
                chiselMap->IntegrateDepthScanColor<DepthData, ColorData>(projectionIntegrator, lastDepthImage, depthCamera.lastPose, depthCamera.cameraModel,
                                                                          lastColorImage, colorCamera.lastPose, colorCamera.cameraModel);
            else
                chiselMap->IntegrateDepthScan<DepthData>(projectionIntegrator, lastDepthImage, depthCamera.lastPose, depthCamera.cameraModel);
            PublishLatestChunkBoxes();
            PublishDepthFrustum();
            chiselMap->UpdateMeshes();
            hasNewData = false;
        }
    }

    // :97-134
    void PublishDepthFrustum() {
        chisel::Frustum frustum;
        depthCamera.cameraModel.SetupFrustum(depthCamera.lastPose, &frustum);
        depthCamera.frustumPublisher.publish(CreateFrustumMarker(frustum));
    }
    void PublishColorFrustum() {
        chisel::Frustum frustum;
        colorCamera.cameraModel.SetupFrustum(colorCamera.lastPose, &frustum);
        colorCamera.frustumPublisher.publish(CreateFrustumMarker(frustum));
    }
    // what the markers take from the library: the frustum's 24 line end points, the pose's translation and rotation, the chunk
    // boxes' centres (marker types, scales and colours are rviz business with no chisel:: call in them)
    visualization_msgs::Marker CreateFrustumMarker(const chisel::Frustum &frustum) {
        visualization_msgs::Marker marker;
        const chisel::Vec3 *lines = frustum.GetLines();
        for (int i = 0; i < 24; i++) marker.points.push_back(point_of(lines[i]));
        return marker;
    }
    // :136-180
    void PublishPose(RosCameraTopic &cam) {
        geometry_msgs::PoseStamped pose;
        pose.header.stamp = cam.lastImageTimestamp;
        pose.pose.position = point_of(cam.lastPose.translation());
        const chisel::Quaternion quat(cam.lastPose.rotation());
        pose.pose.orientation.x = quat.x(); pose.pose.orientation.y = quat.y();
        pose.pose.orientation.z = quat.z(); pose.pose.orientation.w = quat.w();
        cam.lastPosePublisher.publish(pose);
    }
    void PublishDepthPose() { PublishPose(depthCamera); }
    void PublishColorPose() { PublishPose(colorCamera); }

    // :534-605
    void PublishLatestChunkBoxes() {
        const chisel::ChunkManager &chunkManager = chiselMap->GetChunkManager();
        visualization_msgs::Marker marker;
        marker.scale.x = chunkManager.GetChunkSize()(0) * chunkManager.GetResolution();
        for (const std::pair<const chisel::ChunkID, bool> &id : chiselMap->GetMeshesToUpdate())
            if (chunkManager.HasChunk(id.first)) marker.points.push_back(point_of(chunkManager.GetChunk(id.first)->ComputeBoundingBox().GetCenter()));
        latestChunkPublisher.publish(marker);
    }
    void PublishChunkBoxes() {
        visualization_msgs::Marker marker;
        for (const std::pair<const chisel::ChunkID, chisel::ChunkPtr> &pair : chiselMap->GetChunkManager().GetChunks())
            marker.points.push_back(point_of(pair.second->ComputeBoundingBox().GetCenter()));
        chunkBoxPublisher.publish(marker);
    }
    // :607-716: what the mesh markers take from the library -- GetAllMeshes() and, per mesh, grids / vertices / HasColors() ? colors :
    // HasNormals() ? normals (the marker's own colouring of those values is rviz business and has no chisel:: call in it)
    void PublishMeshes() {
        visualization_msgs::Marker marker, marker2;
        const chisel::MeshMap &meshMap = chiselMap->GetChunkManager().GetAllMeshes();
        for (const std::pair<const chisel::ChunkID, chisel::MeshPtr> &meshes : meshMap) {
            const chisel::MeshPtr &mesh = meshes.second;
            for (size_t i = 0; i < mesh->grids.size(); i++) marker2.points.push_back(point_of(mesh->grids[i]));
            for (size_t i = 0; i < mesh->vertices.size(); i++) {
                marker.points.push_back(point_of(mesh->vertices[i]));
                const chisel::Vec3 shade = mesh->HasColors() ? mesh->colors[i] : (mesh->HasNormals() ? mesh->normals[i] : mesh->vertices[i]);
                std_msgs::ColorRGBA color;
                color.r = shade[0]; color.g = shade[1]; color.b = shade[2]; color.a = 1.0;
                marker.colors.push_back(color);
            }
        }
        if (!marker.points.empty()) {
            meshPublisher.publish(marker);
            normalPublisher.publish(marker2);
        }
    }
    static geometry_msgs::Point point_of(const chisel::Vec3 &v) {
        geometry_msgs::Point pt;
        pt.x = v[0]; pt.y = v[1]; pt.z = v[2];
        return pt;
    }
    // services: :426-432, :718-740
    bool Reset() {
        chiselMap->Reset();
        return true;
    }
    bool SaveMesh(const std::string &file_name) { return chiselMap->SaveAllMeshesToPLY(file_name); }
    bool GetAllChunks(std::vector<chisel_ros::ChunkMessage> *chunks) {
        const chisel::ChunkMap &chunkmap = chiselMap->GetChunkManager().GetChunks();
        chunks->resize(chunkmap.size());
        size_t i = 0;
        for (const std::pair<const chisel::ChunkID, chisel::ChunkPtr> &chunkPair : chiselMap->GetChunkManager().GetChunks()) {
            chisel_ros::ChunkMessage &msg = chunks->at(i);
            FillChunkMessage(chunkPair.second, &msg);
            i++;
        }
        return true;
    }

    RosCameraTopic depthCamera, colorCamera;
    LastMessage<visualization_msgs::Marker> chunkBoxPublisher, latestChunkPublisher, meshPublisher, normalPublisher;

  private:
    chisel::ChiselPtr chiselMap;
    chisel::ProjectionIntegrator projectionIntegrator;
    std::shared_ptr<chisel::DepthImage<DepthData>> lastDepthImage;
    std::shared_ptr<chisel::ColorImage<ColorData>> lastColorImage;
    bool useColor, hasNewData = false;
    float nearPlaneDist, farPlaneDist;
};

}  // namespace replay

namespace {
struct RecHeader {
    char magic[8];
    int32_t n_frames, width, height, depth_encoding, color_channels, reserved[3];
};
}  // namespace

int main(int argc, char **argv) {
    if (argc < 3) {
        std::fprintf(stderr, "usage: replay <recording> <out_prefix> [chunk_edge voxel_res use_color near far truncation_scale carving_dist]\n");
        return 2;
    }
    const int chunkSize = argc > 3 ? std::atoi(argv[3]) : 16;
    const float res = argc > 4 ? (float)std::atof(argv[4]) : 0.03f;
    const bool useColor = argc > 5 ? std::atoi(argv[5]) != 0 : true;
    const float nearPlane = argc > 6 ? (float)std::atof(argv[6]) : 0.05f, farPlane = argc > 7 ? (float)std::atof(argv[7]) : 5.0f;
    const float truncScale = argc > 8 ? (float)std::atof(argv[8]) : 8.0f, carvingDist = argc > 9 ? (float)std::atof(argv[9]) : 0.05f;
    std::ifstream in(argv[1], std::ios::binary);
    RecHeader h;
    if (!in.read(reinterpret_cast<char *>(&h), sizeof(h)) || std::memcmp(h.magic, "CVIDSRC1", 8) != 0) {
        std::fprintf(stderr, "replay: %s is not a CVIDSRC1 recording\n", argv[1]);
        return 1;
    }
    if (useColor && h.color_channels == 0) {
        std::fprintf(stderr, "replay: the recording has no colour images\n");
        return 1;
    }
    replay::Server server(chunkSize, res, useColor, nearPlane, farPlane);
    // ChiselNode.cpp:98: the inverse truncator; weight 1, carving on
    server.SetupProjectionIntegrator(chisel::TruncatorPtr(new chisel::InverseTruncator(truncScale)), 1, true, carvingDist);
    const std::string prefix = argv[2];
    std::ofstream poses(prefix + ".poses", std::ios::binary);
    const size_t depthBytes = (size_t)h.width * h.height * (h.depth_encoding ? 2 : 4), colorBytes = (size_t)h.width * h.height * h.color_channels;
    for (int k = 0; k < h.n_frames; k++) {
        double P[4], origin[3], rot[4];
        in.read(reinterpret_cast<char *>(P), sizeof(P));
        in.read(reinterpret_cast<char *>(origin), sizeof(origin));
        in.read(reinterpret_cast<char *>(rot), sizeof(rot));
        std::shared_ptr<sensor_msgs::Image> depth(new sensor_msgs::Image), color(new sensor_msgs::Image);
        depth->width = color->width = h.width;
        depth->height = color->height = h.height;
        depth->encoding = h.depth_encoding ? "16UC1" : "32FC1";
        depth->step = h.width * (h.depth_encoding ? 2 : 4);
        depth->data.resize(depthBytes);
        in.read(reinterpret_cast<char *>(depth->data.data()), depthBytes);
        color->encoding = h.color_channels == 1 ? "mono8" : (h.color_channels == 3 ? "bgr8" : "bgra8");
        color->step = h.width * h.color_channels;
        color->data.resize(colorBytes);
        if (colorBytes) in.read(reinterpret_cast<char *>(color->data.data()), colorBytes);
        if (!in) {
            std::fprintf(stderr, "replay: recording ends inside frame %d\n", k);
            return 1;
        }
        std::shared_ptr<sensor_msgs::CameraInfo> info(new sensor_msgs::CameraInfo);
        info->width = h.width;
        info->height = h.height;
        info->P[0] = P[0]; info->P[5] = P[1]; info->P[2] = P[2]; info->P[6] = P[3];
        tf::StampedTransform tf;
        for (int i = 0; i < 3; i++) tf.origin.v[i] = origin[i];
        for (int i = 0; i < 4; i++) tf.rotation.v[i] = rot[i];
        server.CallbackAll(depth, info, color, info, tf, tf);
        float pose12[12];
        for (int r = 0; r < 3; r++) {
            for (int c = 0; c < 3; c++) pose12[4 * r + c] = server.depthCamera.lastPose.linear()(r, c);
            pose12[4 * r + 3] = server.depthCamera.lastPose.translation()(r);
        }
        poses.write(reinterpret_cast<const char *>(pose12), sizeof(pose12));
        std::printf("frame %d: chunk boxes %zu, latest boxes %zu, frustum points %zu, mesh vertices %zu (published %zu times), pose published %zu\n", k,
                    server.chunkBoxPublisher.last.points.size(), server.latestChunkPublisher.last.points.size(),
                    server.depthCamera.frustumPublisher.last.points.size(), server.meshPublisher.last.points.size(), server.meshPublisher.published,
                    server.depthCamera.lastPosePublisher.published);
    }
    // the services
    std::vector<chisel_ros::ChunkMessage> msgs;
    server.GetAllChunks(&msgs);
    size_t words = 0;
    for (const chisel_ros::ChunkMessage &m : msgs) words += m.distance_data.size() + m.color_data.size();
    const bool saved = server.SaveMesh(prefix + ".ply");
    std::printf("services: GetAllChunks %zu messages (%zu payload words), SaveMesh %s\n", msgs.size(), words, saved ? "ok" : "failed");
    // the map, through the mirrors GetChunks() hands out (ascending id)
    const chisel::ChunkMap &chunks = server.GetChiselMap()->GetChunkManager().GetChunks();
    std::vector<chisel::ChunkID> ids;
    for (const std::pair<const chisel::ChunkID, chisel::ChunkPtr> &c : chunks) ids.push_back(c.first);
    std::sort(ids.begin(), ids.end(), [](const chisel::ChunkID &a, const chisel::ChunkID &b) {
        return a(2) != b(2) ? a(2) < b(2) : (a(1) != b(1) ? a(1) < b(1) : a(0) < b(0));
    });
    std::ofstream out(prefix + ".map", std::ios::binary);
    for (const chisel::ChunkID &id : ids) {
        const chisel::ChunkPtr &c = chunks.at(id);
        const int v[3] = {id(0), id(1), id(2)};
        out.write(reinterpret_cast<const char *>(v), sizeof(v));
        for (const chisel::DistVoxel &d : c->GetVoxels()) {
            const float sw[2] = {d.GetSDF(), d.GetWeight()};
            out.write(reinterpret_cast<const char *>(sw), sizeof(sw));
        }
        for (const chisel::ColorVoxel &cv : c->GetColorVoxels()) {
            const uint8_t px[4] = {cv.GetRed(), cv.GetGreen(), cv.GetBlue(), cv.GetWeight()};
            out.write(reinterpret_cast<const char *>(px), 4);
        }
    }
    std::printf("map: %zu chunks written\n", ids.size());
    server.Reset();
    std::printf("after Reset: %zu chunks\n", server.GetChiselMap()->GetChunkManager().GetChunks().size());
    return 0;
}
