// gp_predictor.h -- the reference's GpPredictor class surface (gp_predictor/include/gp_predictor/
// gp_predictor.h:18-62) kept name for name, without ROS or Eigen: fixed-size row-major arrays
// replace the Eigen members, corenav::NodeHandle replaces ros::NodeHandle.
#ifndef CORENAV_GP_PREDICTOR_H_
#define CORENAV_GP_PREDICTOR_H_

#include <array>

#include "core_nav_msgs.h"

class GpPredictor {
 public:
  explicit GpPredictor(corenav::NodeHandle &);

  typedef std::array<double, 3> Vector3;

  // Declared but never defined in the reference (gp_predictor.h:28-29); defined here as no-ops that
  // record the flag so a caller linking against them still links.
  void mobility(bool flag);
  void mobilityCallback(const corenav_pod::std_msgs::Int64::ConstPtr &msg);
  void GPCallBack(const corenav_pod::core_nav::GP_Output::ConstPtr &gp_data_in_);
  bool LoadParameters(const corenav::NodeHandle &nh_);
  GpPredictor::Vector3 llh_to_enu(const double latitude, const double longitude, const double height);

  corenav_pod::core_nav::GP_Input slip_msg;
  corenav_pod::core_nav::GP_Output gp_data_;

  std::array<double, 16> R_IP{}, R_IP_1{}, R_IP_2{};  // 4x4
  std::array<double, 60> K_pred{};                    // 15x4
  std::array<double, 60> H_{};                        // 4x15
  std::array<double, 225> P_pred{}, STM_{}, Q_{};     // 15x15

  GpPredictor::Vector3 savePos{}, ins_enu_slip{}, ins_enu_slip3p{}, ins_enu_slip_3p{};
  corenav_pod::std_msgs::Float64 stop_cmd_msg_;

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
  corenav::NodeHandle &nh_;
};

#endif  // CORENAV_GP_PREDICTOR_H_
