// gp_predictor.cpp -- GpPredictor node logic (gp_predictor/src/gp_predictor.cpp:9-142) on the
// ROS-free NodeHandle.  The arithmetic is gp_predictor_core.cpp; this file is the message plumbing.
#include "gp_predictor.h"

#include <algorithm>

#include "gp_predictor_core.hpp"

GpPredictor::GpPredictor(corenav::NodeHandle &nh) : nh_(nh) {}

void GpPredictor::mobility(bool flag) { gp_flag = flag; }
void GpPredictor::mobilityCallback(const corenav_pod::std_msgs::Int64::ConstPtr &msg) { mobility(msg && msg->data != 0); }

void GpPredictor::GPCallBack(const corenav_pod::core_nav::GP_Output::ConstPtr &gp_data_in_) {
  gp_data_.mean = gp_data_in_->mean;     // :18-19
  gp_data_.sigma = gp_data_in_->sigma;
  gp_arrived_time_ = nh_.now ? nh_.now() : 0.0;  // :22

  corenav_pod::core_nav::SetStopping srv;
  srv.request.stopping = true;           // :25
  if (nh_.call_set_stopping && nh_.call_set_stopping(srv)) {  // :26
    std::copy(srv.response.PvecData.begin(), srv.response.PvecData.end(), P_pred.begin());   // :30-36
    std::copy(srv.response.QvecData.begin(), srv.response.QvecData.end(), Q_.begin());
    std::copy(srv.response.STMvecData.begin(), srv.response.STMvecData.end(), STM_.begin());
    corenav::unpack_H(srv.response.HvecData.data(), h_bug_compatible, H_.data());            // :38-42
    savePos = {srv.response.PosData.x, srv.response.PosData.y, srv.response.PosData.z};      // :44-46
    new_gp_data_arrived_ = true;         // :51
  }
  if (!new_gp_data_arrived_) return;     // :58

  const double init_llh[3] = {init_x, init_y, init_z}, init_ecef[3] = {init_ecef_x, init_ecef_y, init_ecef_z};
  // The core re-reads H through unpack_H; hand it the already-unpacked matrix in r*15+c layout.
  const double now = nh_.now ? nh_.now() : gp_arrived_time_;
  corenav::StopPrediction r = corenav::predict_stop(
      gp_data_.mean.data(), gp_data_.sigma.data(), (int)std::min(gp_data_.mean.size(), gp_data_.sigma.size()),
      P_pred.data(), Q_.data(), STM_.data(), H_.data(), savePos.data(), gp_arrived_time_, now, xy_threshold,
      /*h_bug_compatible=*/false, init_llh, init_ecef);
  xy_errSlip = r.xy_err;
  if (r.fired) {
    stop_cmd_msg_.data = r.stop_cmd;     // :109,:114
    if (nh_.publish_stop_cmd) nh_.publish_stop_cmd(stop_cmd_msg_);  // :118
  }
  new_gp_data_arrived_ = false;          // :126
  i = 0;                                 // :127-128
  slip_i = 0;
}

bool GpPredictor::LoadParameters(const corenav::NodeHandle &nh) {  // :134-142
  if (!nh.get_param) return false;
  if (!nh.get_param("init_llh/x", init_x)) return false;
  if (!nh.get_param("init_llh/y", init_y)) return false;
  if (!nh.get_param("init_llh/z", init_z)) return false;
  if (!nh.get_param("init_ecef/x", init_ecef_x)) return false;
  if (!nh.get_param("init_ecef/y", init_ecef_y)) return false;
  if (!nh.get_param("init_ecef/z", init_ecef_z)) return false;
  return true;
}

GpPredictor::Vector3 GpPredictor::llh_to_enu(const double latitude, const double longitude, const double height) {
  const double init_llh[3] = {init_x, init_y, init_z}, init_ecef[3] = {init_ecef_x, init_ecef_y, init_ecef_z};
  Vector3 out{};
  corenav::llh_to_enu(latitude, longitude, height, init_llh, init_ecef, out.data());
  return out;
}
