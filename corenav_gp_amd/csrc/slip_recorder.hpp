// slip_recorder.hpp -- the producer side of the hot path: slip computation and the
// recording-window state machine of CoreNav::Update (core_navigation/src/CoreNav.cpp:176,244-330),
// CoreNav::stopCallback (:755-759) and the gp_flag handling of CoreNav::getCmdData (:794-816), as a
// ROS-free, Eigen-free class.  Member names follow the reference.
#pragma once
#include <vector>

namespace corenav {

class SlipWindowRecorder {
 public:
  // One odometry update (10 Hz).  vel* are the wheel ground speeds as CoreNav computes them
  // (:178-181), vlin the INS forward speed in the body frame (:245), cmd_x = cmd[0].
  // Returns true when a GP_Input window is published this tick (then `time_array` / `slip_array`
  // hold it; the reference clears its message right after publishing, :307-308).
  bool Update(double velFrontLeft, double velFrontRight, double velBackLeft, double velBackRight, double vlin,
              double cmd_x);
  void stopCallback(double cmd_stop);  // :755-759
  void CmdCallBack(double cmd_x);      // getCmdData :794-816 (gp_flag / started_driving_again_flag)

  // published window (valid after Update returned true)
  std::vector<double> time_array, slip_array;
  // reference state
  double slip = 0.0;
  double odomUptCount = 0.0, saveCountOdom = 0.0, startRecording = 0.0, stopRecording = 0.0;
  double cmd_stop_ = 0.0;
  bool first_driving_flag = true, gp_flag = false, new_stop_data_arrived_ = false;
  bool started_driving_again_flag = true;
  int skipped_windows = 0;  // windows dropped for having < 15 samples (:300-302)

 private:
  std::vector<double> rec_time_, rec_slip_;  // slip_msg being filled
};

}  // namespace corenav
