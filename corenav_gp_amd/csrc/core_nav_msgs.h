// core_nav_msgs.h -- POD mirrors of the wire formats on the hot-path boundary, usable without ROS:
//   core_nav/GP_Input   (core_navigation/msg/GP_Input.msg:1-3)
//   core_nav/GP_Output  (core_navigation/msg/GP_Output.msg:1-3)
//   core_nav/SetStopping (core_navigation/srv/SetStopping.srv:1-7)
//   std_msgs/Float64, std_msgs/Int64, geometry_msgs/Point as far as the path uses them.
// With ROS present (ros_shell.cpp, built only when <ros/ros.h> is found) the generated message
// classes are converted field by field to these structs at the node boundary.
#pragma once
#include <array>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

// The mirrors live in namespace corenav_pod so they can coexist with the generated ROS classes
// (::std_msgs, ::core_nav) inside a catkin build; everything here names them explicitly.
namespace corenav_pod {
namespace std_msgs {
struct Header { uint32_t seq = 0; double stamp = 0.0; std::string frame_id; };
struct Float64 { double data = 0.0; };
struct Int64 { int64_t data = 0; typedef std::shared_ptr<const Int64> ConstPtr; };
}  // namespace std_msgs

namespace geometry_msgs {
struct Point { double x = 0.0, y = 0.0, z = 0.0; };
}  // namespace geometry_msgs

namespace core_nav {
struct GP_Input {
  std_msgs::Header header;
  std::vector<double> time_array, slip_array;
  typedef std::shared_ptr<const GP_Input> ConstPtr;
};
struct GP_Output {
  std_msgs::Header header;
  std::vector<double> mean, sigma;
  typedef std::shared_ptr<const GP_Output> ConstPtr;
};
struct SetStopping {
  struct Request { bool stopping = false; } request;
  struct Response {
    std::array<double, 225> PvecData{}, QvecData{}, STMvecData{};
    std::array<double, 60> HvecData{};
    geometry_msgs::Point PosData;
  } response;
};
}  // namespace core_nav

}  // namespace corenav_pod

namespace corenav {
// What GpPredictor needs from its middleware: the three endpoints of gp_predictor.cpp:11-13 and a
// clock.  ros_shell.cpp implements it over roscpp; tests and the replay harness implement it in
// process.  Names are the reference's absolute topic / service names.
struct NodeHandle {
  static constexpr const char *kGpResultTopic = "/core_nav/core_nav/gp_result";
  static constexpr const char *kStoppingService = "/core_nav/core_nav/stopping_service";
  static constexpr const char *kStopCmdTopic = "/core_nav/core_nav/stop_cmd";
  static constexpr const char *kGpInputTopic = "/core_nav/core_nav/gp_input";
  std::function<bool(corenav_pod::core_nav::SetStopping &)> call_set_stopping;   // service client (:12,:26)
  std::function<void(const corenav_pod::std_msgs::Float64 &)> publish_stop_cmd;  // publisher, queue 1 (:13,:118)
  std::function<double()> now;                                      // ros::Time::now().toSec()
  std::function<bool(const std::string &, double &)> get_param;     // ros::param::get (:135-140)
};
}  // namespace corenav
