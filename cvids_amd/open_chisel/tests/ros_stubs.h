// ros_stubs.h -- just enough of the ROS message / tf types chisel_ros touches for a ROS-free build of its chisel::* call sites
// (replay.cpp).  Field names follow the ROS message definitions (sensor_msgs/Image, sensor_msgs/CameraInfo,
// visualization_msgs/Marker, geometry_msgs/*, std_msgs/*, chisel_ros/msg/ChunkMessage.msg); nothing here talks to a ROS master.
#ifndef CHISEL_HIP_ROS_STUBS_H_
#define CHISEL_HIP_ROS_STUBS_H_
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace ros {
struct Time {
    double t = 0.0;
    static Time now() { return Time(); }
};
}  // namespace ros
namespace std_msgs {
struct Header {
    ros::Time stamp;
    std::string frame_id;
};
struct ColorRGBA {
    float r = 0, g = 0, b = 0, a = 0;
};
}  // namespace std_msgs
namespace geometry_msgs {
struct Point {
    double x = 0, y = 0, z = 0;
};
struct Quaternion {
    double x = 0, y = 0, z = 0, w = 1;
};
struct Vector3 {
    double x = 0, y = 0, z = 0;
};
struct Pose {
    Point position;
    Quaternion orientation;
};
struct PoseStamped {
    std_msgs::Header header;
    Pose pose;
};
}  // namespace geometry_msgs
namespace visualization_msgs {
struct Marker {
    enum { CUBE_LIST = 6, LINE_LIST = 5, TRIANGLE_LIST = 11 };
    std_msgs::Header header;
    std::string ns;
    int id = 0;
    int type = 0;
    geometry_msgs::Pose pose;
    geometry_msgs::Vector3 scale;
    std_msgs::ColorRGBA color;
    std::vector<geometry_msgs::Point> points;
    std::vector<std_msgs::ColorRGBA> colors;
};
}  // namespace visualization_msgs
namespace sensor_msgs {
struct Image {
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    std::string encoding;
    uint32_t step = 0;
    std::vector<uint8_t> data;
};
typedef std::shared_ptr<const Image> ImageConstPtr;
struct CameraInfo {
    std_msgs::Header header;
    uint32_t height = 0, width = 0;
    double P[12] = {0};
};
typedef std::shared_ptr<const CameraInfo> CameraInfoConstPtr;
}  // namespace sensor_msgs
namespace tf {
struct Vec {
    double v[4];
    double x() const { return v[0]; }
    double y() const { return v[1]; }
    double z() const { return v[2]; }
    double w() const { return v[3]; }
};
struct StampedTransform {  // the transform lookupTransform(camera frame, base frame) hands back
    Vec origin{{0, 0, 0, 0}}, rotation{{0, 0, 0, 1}};
    const Vec &getOrigin() const { return origin; }
    const Vec &getRotation() const { return rotation; }
};
}  // namespace tf
namespace chisel_ros {
struct ChunkMessage {  // msg/ChunkMessage.msg:1-23
    std_msgs::Header header;
    int32_t ID_x = 0, ID_y = 0, ID_z = 0;
    uint64_t spatial_hash = 0;
    float resolution_meters = 0;
    int32_t num_voxels_x = 0, num_voxels_y = 0, num_voxels_z = 0;
    std::vector<uint32_t> distance_data;
    std::vector<uint32_t> color_data;
};
}  // namespace chisel_ros
// a publisher that keeps the last message (what rviz would have received)
template <class Msg>
struct LastMessage {
    Msg last;
    size_t published = 0;
    void publish(const Msg &m) {
        last = m;
        published++;
    }
};
#endif
