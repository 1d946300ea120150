// gp_predictor.h -- the reference's GpPredictor class (gp_predictor/include/gp_predictor/gp_predictor.h:18-62),
// name for name, in two configurations of ONE class chosen by what the build finds:
//
//   <ros/ros.h> present (a catkin workspace; CORENAV_NO_ROS not defined):
//       the reference's own signature -- GpPredictor(ros::NodeHandle &) (:22), the generated core_nav / std_msgs
//       message classes in the callbacks and members (:28-47), and the private gp_sub_ / stop_cmd_pub_ /
//       clt_setStopping_ / nh_ (:57-60) which the constructor wires exactly as gp_predictor.cpp:9-14 does.
//       core_navigation and any other client then compile against this header unchanged.
//   otherwise (this image has no ROS): the same members over the POD mirrors of core_nav_msgs.h and
//       corenav::NodeHandle, which is what libcorenav_gp.so and the tests build.
//
//   <Eigen/Dense> present: the matrix members are the reference's Eigen types (:36-43) and
//       `typedef Eigen::MatrixXd Matrix` (:25) exists; otherwise fixed-size corenav::Mat<R, C> with the same
//       (row, col) element access.
//
// Both configurations run the same arithmetic (gp_predictor_core.cpp).  The ROS configuration is NOT compiled
// in this repository's build (no roscpp / Eigen in the image) -- INTEGRATION.md says so.
#ifndef CORENAV_GP_PREDICTOR_H_
#define CORENAV_GP_PREDICTOR_H_

#include <array>

#if !defined(CORENAV_NO_ROS) && defined(__has_include)
#if __has_include(<ros/ros.h>)
#define CORENAV_HAVE_ROS 1
#endif
#endif
#if !defined(CORENAV_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#define CORENAV_HAVE_EIGEN 1
#endif
#endif

#include "core_nav_msgs.h"

#ifdef CORENAV_HAVE_EIGEN
#include <Eigen/Dense>
#include <Eigen/Geometry>
#endif

#ifdef CORENAV_HAVE_ROS
#include <ros/ros.h>
#include <ros/console.h>
#include "std_msgs/Int64.h"
#include <std_msgs/Bool.h>
#include <core_nav/SetStopping.h>
#include <core_nav/GP_Input.h>
#include <core_nav/GP_Output.h>
#include <std_msgs/Float64.h>
namespace corenav_types {
typedef ros::NodeHandle NodeHandle;
namespace msgs = ::core_nav;
namespace std_msgs = ::std_msgs;
}  // namespace corenav_types
#else
namespace corenav_types {
typedef corenav::NodeHandle NodeHandle;
namespace msgs = corenav_pod::core_nav;
namespace std_msgs = corenav_pod::std_msgs;
}  // namespace corenav_types
#endif

namespace corenav {
// Fixed-size row-major matrix with Eigen's element access, used when Eigen is not installed.
template <int R, int C> struct Mat {
  std::array<double, R * C> v{};
  double &operator()(int r, int c) { return v[r * C + c]; }
  double operator()(int r, int c) const { return v[r * C + c]; }
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double *data() { return v.data(); }
  const double *data() const { return v.data(); }
};
}  // namespace corenav

class GpPredictor {
 public:
  GpPredictor(corenav_types::NodeHandle &);   // ROS build: GpPredictor(ros::NodeHandle &), gp_predictor.h:22

#ifdef CORENAV_HAVE_EIGEN
  typedef Eigen::Matrix<double, 3, 1> Vector3;   // :24
  typedef Eigen::MatrixXd Matrix;                // :25
  typedef Eigen::Matrix<double, 4, 4> Mat4x4;
  typedef Eigen::Matrix<double, 15, 4> Mat15x4;
  typedef Eigen::Matrix<double, 4, 15> Mat4x15;
  typedef Eigen::Matrix<double, 15, 15> Mat15x15;
#else
  typedef corenav::Mat<3, 1> Vector3;
  typedef corenav::Mat<4, 4> Mat4x4;
  typedef corenav::Mat<15, 4> Mat15x4;
  typedef corenav::Mat<4, 15> Mat4x15;
  typedef corenav::Mat<15, 15> Mat15x15;
#endif

  // Declared but never defined in the reference (gp_predictor.h:27-28); defined here as no-ops that record
  // the flag, so a caller that references them still links.
  void mobility(bool flag);
  void mobilityCallback(const corenav_types::std_msgs::Int64::ConstPtr &msg);
  void GPCallBack(const corenav_types::msgs::GP_Output::ConstPtr &gp_data_in_);
  bool LoadParameters(const corenav_types::NodeHandle &nh_);
  GpPredictor::Vector3 llh_to_enu(const double latitude, const double longitude, const double height);

  corenav_types::msgs::GP_Input slip_msg;
  corenav_types::msgs::GP_Output gp_data_;

  Mat4x4 R_IP, R_IP_1, R_IP_2;   // :36-38
  Mat15x4 K_pred;                // :39
  Mat4x15 H_;                    // :40
  Mat15x15 P_pred, STM_, Q_;     // :41-43

  GpPredictor::Vector3 savePos, ins_enu_slip, ins_enu_slip3p, ins_enu_slip_3p;
  corenav_types::std_msgs::Float64 stop_cmd_msg_;

  bool new_gp_data_arrived_ = false;  // uninitialised in the reference (gp_predictor.h:48)
  bool gp_flag = false;
  double gp_arrived_time_ = 0.0;
  double xy_errSlip = 0.0, odomUptCount = 0.0, startRecording = 0.0, stopRecording = 0.0, saveCountOdom = 0.0;
  // Defaults = core_navigation/config/init_params.yaml:9-16.  The reference leaves these
  // uninitialised because LoadParameters is never called (gp_predictor.cpp:134-142,180-190).
  double init_ecef_x = 859153.0153, init_ecef_y = -4836303.7266, init_ecef_z = 4055378.501;
  double init_x = 0.693457963620326, init_y = -1.39498384275845, init_z = 334.993517334743;
  int slip_i = 0;
  int i = 0;

  // Build-side switches (not in the reference): H unpacking r*4+c (reference behaviour) or r*15+c,
  // and the stop threshold of gp_predictor.cpp:102.
  bool h_bug_compatible = true;
  double xy_threshold = 3.00;

 private:
#ifdef CORENAV_HAVE_ROS
  ros::Subscriber gp_sub_;              // :57
  ros::Publisher stop_cmd_pub_;         // :58
  ros::ServiceClient clt_setStopping_;  // :59
#endif
  corenav_types::NodeHandle &nh_;       // :60

  // the three middleware operations GPCallBack performs, over roscpp or over the in-process hooks
  double clock_now();
  bool call_set_stopping(corenav_pod::core_nav::SetStopping &srv);
  void publish_stop_cmd();
};

#ifdef CORENAV_HAVE_ROS
int main(int argc, char **argv);   // gp_predictor.h:63
#endif

#endif  // CORENAV_GP_PREDICTOR_H_
