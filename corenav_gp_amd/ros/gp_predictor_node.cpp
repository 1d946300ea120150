// gp_predictor_node.cpp -- the node executable: the reference's main (gp_predictor/src/gp_predictor.cpp:180-190)
// verbatim in structure, because inside a catkin workspace csrc/gp_predictor.h IS the reference's class
// (GpPredictor(ros::NodeHandle &), gp_sub_ / stop_cmd_pub_ / clt_setStopping_ wired in the constructor).
// Same node name "gp_predictor" (:182), empty namespace handle (:183), single-threaded ros::spin() (:187).
// Built only where roscpp, Eigen and the core_nav messages exist (ros/CMakeLists.txt); this image has none
// of them, so this file and the ROS branch of csrc/gp_predictor.{h,cpp} are NOT compiled here.
#include <ros/ros.h>

#include "../csrc/gp_predictor.h"

#ifndef CORENAV_HAVE_ROS
#error "gp_predictor_node.cpp needs roscpp: build it with catkin (ros/CMakeLists.txt)"
#endif

int main(int argc, char **argv) {
  ros::init(argc, argv, "gp_predictor");
  ros::NodeHandle nh("");

  GpPredictor gp_predictor(nh);

  ros::spin();

  return 0;
}
