#include "slip_recorder.hpp"

#include <algorithm>
#include <cmath>

namespace corenav {

bool SlipWindowRecorder::Update(double velFrontLeft_, double velFrontRight_, double velBackLeft_, double velBackRight_,
                                double vlin, double cmd_x) {
  bool published = false;
  odomUptCount = odomUptCount + 1;                                   // :176
  const double rearVel_ = (velBackLeft_ + velBackRight_) / 2.0;      // :183
  // :246  slip = max over the four wheels of (v_wheel - v_ins) / v_wheel
  slip = std::max(std::max((velFrontRight_ - vlin) / velFrontRight_, (velBackRight_ - vlin) / velBackRight_),
                  std::max((velFrontLeft_ - vlin) / velFrontLeft_, (velBackLeft_ - vlin) / velBackLeft_));
  if (std::abs(rearVel_) < 0.001) slip = 0.0;                        // :248-251
  if (slip < -1.0) slip = -1.0;                                      // :252-255
  if (slip > 1.0) slip = 1.0;                                        // :256-259

  if (slip != 0.0 && slip != -1.0 && slip != 1.0 && std::fabs(cmd_x) > 0.2) {  // :264
    if (first_driving_flag) {                                        // :266-276
      saveCountOdom = odomUptCount;
      startRecording = saveCountOdom + 10;
      stopRecording = startRecording + 150;
      first_driving_flag = false;
    }
    if (odomUptCount > startRecording && odomUptCount < stopRecording && !gp_flag) {  // :278-288
      rec_slip_.push_back(slip);
      rec_time_.push_back(odomUptCount);
    }
    if (odomUptCount >= stopRecording) {                             // :289
      if (!gp_flag) {                                                // :294-309
        gp_flag = true;
        if (rec_slip_.size() < 15) {
          ++skipped_windows;                                         // :300-302 "GP skipped"
        } else {
          time_array = rec_time_;                                    // :304 gp_pub.publish(slip_msg)
          slip_array = rec_slip_;
          published = true;
        }
        rec_slip_.clear();                                           // :307-308
        rec_time_.clear();
      }
      if (new_stop_data_arrived_) {                                  // :311-321
        new_stop_data_arrived_ = false;
        startRecording = stopRecording + std::ceil(cmd_stop_) * 10 + 10 + 50;
        stopRecording = startRecording + 150;
        gp_flag = false;
      }
    }
    // :323  integer-looking division is on doubles in the reference (odomUptCount is double)
    if (!first_driving_flag && odomUptCount / 10 - stopRecording / 10 > 10) {  // :323-329 unexpected stop
      rec_slip_.clear();
      rec_time_.clear();
      first_driving_flag = true;
      gp_flag = false;
    }
  }
  return published;
}

void SlipWindowRecorder::stopCallback(double cmd_stop) {             // :755-759
  cmd_stop_ = cmd_stop;
  new_stop_data_arrived_ = true;
}

void SlipWindowRecorder::CmdCallBack(double cmd_x) {                 // getCmdData :794-816
  if (gp_flag) {
    if (std::fabs(cmd_x) < 0.0001) started_driving_again_flag = false;
  }
  if (started_driving_again_flag == false) {
    if (std::fabs(cmd_x) > 0.0001) {
      started_driving_again_flag = true;
      gp_flag = false;
    }
  }
}

}  // namespace corenav
