// gp_predictor_node.cpp -- the roscpp shell around the ROS-free GpPredictor class.
// Same node name, topics, service, queue sizes as the reference's gp_predictor/src/gp_predictor.cpp:
//   subscribes  /core_nav/core_nav/gp_result        core_nav/GP_Output   queue 1   (:11)
//   calls       /core_nav/core_nav/stopping_service core_nav/SetStopping           (:12)
//   advertises  /core_nav/core_nav/stop_cmd         std_msgs/Float64     queue 1   (:13)
// Built only inside a catkin workspace that has roscpp and the core_nav messages (this image has
// neither, so this file is NOT compiled by csrc/Makefile; the class it wraps is, and is tested
// through cgp_gppredictor_callback).
#include <ros/ros.h>
#include <std_msgs/Float64.h>
#include <core_nav/GP_Output.h>
#include <core_nav/SetStopping.h>

#include "../csrc/gp_predictor.h"

namespace ros_msgs = ::core_nav;  // generated ROS types; the POD mirrors are corenav_pod::core_nav

int main(int argc, char **argv) {
  ros::init(argc, argv, "gp_predictor");                                              // :182
  ros::NodeHandle nh("");                                                             // :183
  ros::ServiceClient clt = nh.serviceClient<ros_msgs::SetStopping>(corenav::NodeHandle::kStoppingService);
  ros::Publisher pub = nh.advertise<std_msgs::Float64>(corenav::NodeHandle::kStopCmdTopic, 1);

  corenav::NodeHandle bridge;
  bridge.now = [] { return ros::Time::now().toSec(); };
  bridge.call_set_stopping = [&clt](corenav_pod::core_nav::SetStopping &srv) {
    ros_msgs::SetStopping r;
    r.request.stopping = srv.request.stopping;
    if (!clt.call(r)) return false;                                                   // :26,:53-56
    std::copy(r.response.PvecData.begin(), r.response.PvecData.end(), srv.response.PvecData.begin());
    std::copy(r.response.QvecData.begin(), r.response.QvecData.end(), srv.response.QvecData.begin());
    std::copy(r.response.STMvecData.begin(), r.response.STMvecData.end(), srv.response.STMvecData.begin());
    std::copy(r.response.HvecData.begin(), r.response.HvecData.end(), srv.response.HvecData.begin());
    srv.response.PosData.x = r.response.PosData.x;
    srv.response.PosData.y = r.response.PosData.y;
    srv.response.PosData.z = r.response.PosData.z;
    return true;
  };
  bridge.publish_stop_cmd = [&pub](const corenav_pod::std_msgs::Float64 &m) {
    std_msgs::Float64 out;
    out.data = m.data;
    pub.publish(out);                                                                 // :118
  };
  bridge.get_param = [](const std::string &name, double &v) { return ros::param::get(name, v); };  // :135-140

  GpPredictor node(bridge);
  node.LoadParameters(bridge);  // the reference defines but never calls it (:134-142); defaults = init_params.yaml
  ros::Subscriber sub = nh.subscribe<ros_msgs::GP_Output>(
      corenav::NodeHandle::kGpResultTopic, 1, [&node](const ros_msgs::GP_Output::ConstPtr &in) {
        auto msg = std::make_shared<corenav_pod::core_nav::GP_Output>();
        msg->mean = in->mean;
        msg->sigma = in->sigma;
        node.GPCallBack(msg);
      });
  ros::spin();                                                                        // :187
  return 0;
}
