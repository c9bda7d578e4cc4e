// open_chisel/pointcloud/PointCloud.h -- facade counterpart of the reference's point container
// (open_chisel/include/open_chisel/pointcloud/PointCloud.h:32-75): chisel_ros fills one from sensor_msgs::PointCloud2
// (ChiselServer.cpp:441-451) and hands it to Chisel::IntegratePointCloud in PointCloud fusion mode (ChiselServer.cpp:523).
#pragma once
#include "../geometry/Geometry.h"

namespace chisel {

class PointCloud {
   public:
    bool HasColor() const { return !colors.empty(); }
    const Vec3List &GetPoints() const { return points; }
    Vec3List &GetMutablePoints() { return points; }
    const Vec3List &GetColors() const { return colors; }
    Vec3List &GetMutableColors() { return colors; }
    void AddPoint(const Vec3 &p) { points.push_back(p); }
    void AddColor(const Vec3 &c) { colors.push_back(c); }
    void AddPointAndColor(const Vec3 &p, const Vec3 &c) {
        points.push_back(p);
        colors.push_back(c);
    }
    void Clear() {
        points.clear();
        colors.clear();
    }

   private:
    Vec3List points, colors;
};
typedef std::shared_ptr<PointCloud> PointCloudPtr;
typedef std::shared_ptr<const PointCloud> PointCloudConstPtr;

}  // namespace chisel
