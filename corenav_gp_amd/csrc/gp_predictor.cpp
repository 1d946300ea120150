// gp_predictor.cpp -- GpPredictor node logic (gp_predictor/src/gp_predictor.cpp:9-142).  One body for both
// configurations of gp_predictor.h; the arithmetic is gp_predictor_core.cpp, this file is the message plumbing.
#include "gp_predictor.h"

#include <algorithm>

#include "gp_predictor_core.hpp"

#ifdef CORENAV_HAVE_ROS
// gp_predictor.cpp:9-14 -- same topics, service, queue sizes, same member wiring
GpPredictor::GpPredictor(ros::NodeHandle &nh) : nh_(nh) {
  gp_sub_ = nh.subscribe("/core_nav/core_nav/gp_result", 1, &GpPredictor::GPCallBack, this);
  clt_setStopping_ = nh_.serviceClient<core_nav::SetStopping>("/core_nav/core_nav/stopping_service");
  stop_cmd_pub_ = nh.advertise<std_msgs::Float64>("/core_nav/core_nav/stop_cmd", 1);
}
double GpPredictor::clock_now() { return ros::Time::now().toSec(); }   // :22, :107
bool GpPredictor::call_set_stopping(corenav_pod::core_nav::SetStopping &srv) {
  core_nav::SetStopping r;
  r.request.stopping = srv.request.stopping;      // :25
  if (!clt_setStopping_.call(r)) {                // :26
    ROS_ERROR("Failed to call Stopping Service"); // :53-56
    return false;
  }
  ROS_INFO("Called Stopping Service");
  std::copy(r.response.PvecData.begin(), r.response.PvecData.end(), srv.response.PvecData.begin());
  std::copy(r.response.QvecData.begin(), r.response.QvecData.end(), srv.response.QvecData.begin());
  std::copy(r.response.STMvecData.begin(), r.response.STMvecData.end(), srv.response.STMvecData.begin());
  std::copy(r.response.HvecData.begin(), r.response.HvecData.end(), srv.response.HvecData.begin());
  srv.response.PosData.x = r.response.PosData.x;
  srv.response.PosData.y = r.response.PosData.y;
  srv.response.PosData.z = r.response.PosData.z;
  return true;
}
void GpPredictor::publish_stop_cmd() { stop_cmd_pub_.publish(stop_cmd_msg_); }   // :118
bool GpPredictor::LoadParameters(const ros::NodeHandle &nh) {                     // :134-142
  if (!nh.getParam("init_llh/x", init_x)) return false;
  if (!nh.getParam("init_llh/y", init_y)) return false;
  if (!nh.getParam("init_llh/z", init_z)) return false;
  if (!nh.getParam("init_ecef/x", init_ecef_x)) return false;
  if (!nh.getParam("init_ecef/y", init_ecef_y)) return false;
  if (!nh.getParam("init_ecef/z", init_ecef_z)) return false;
  return true;
}
#else
GpPredictor::GpPredictor(corenav::NodeHandle &nh) : nh_(nh) {}
double GpPredictor::clock_now() { return nh_.now ? nh_.now() : 0.0; }
bool GpPredictor::call_set_stopping(corenav_pod::core_nav::SetStopping &srv) {
  return nh_.call_set_stopping && nh_.call_set_stopping(srv);
}
void GpPredictor::publish_stop_cmd() {
  if (nh_.publish_stop_cmd) nh_.publish_stop_cmd(stop_cmd_msg_);
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
#endif

void GpPredictor::mobility(bool flag) { gp_flag = flag; }
void GpPredictor::mobilityCallback(const corenav_types::std_msgs::Int64::ConstPtr &msg) { mobility(msg && msg->data != 0); }

void GpPredictor::GPCallBack(const corenav_types::msgs::GP_Output::ConstPtr &gp_data_in_) {
  gp_data_.mean = gp_data_in_->mean;     // :18-19
  gp_data_.sigma = gp_data_in_->sigma;
  gp_arrived_time_ = clock_now();        // :22

  corenav_pod::core_nav::SetStopping srv;
  srv.request.stopping = true;           // :25
  double Hrow[60];                       // H_ unpacked, row-major 4 x 15
  if (call_set_stopping(srv)) {          // :26
    for (int row = 0; row < 15; ++row)   // :30-36
      for (int col = 0; col < 15; ++col) {
        P_pred(row, col) = srv.response.PvecData[row * 15 + col];
        Q_(row, col) = srv.response.QvecData[row * 15 + col];
        STM_(row, col) = srv.response.STMvecData[row * 15 + col];
      }
    corenav::unpack_H(srv.response.HvecData.data(), h_bug_compatible, Hrow);   // :38-42
    for (int row = 0; row < 4; ++row)
      for (int col = 0; col < 15; ++col) H_(row, col) = Hrow[row * 15 + col];
    savePos[0] = srv.response.PosData.x;   // :44-46
    savePos[1] = srv.response.PosData.y;
    savePos[2] = srv.response.PosData.z;
    new_gp_data_arrived_ = true;           // :51
  }
  if (!new_gp_data_arrived_) return;       // :58

  // the core takes row-major arrays whatever the storage order of the members
  double Pr[225], Qr[225], Sr[225];
  for (int row = 0; row < 15; ++row)
    for (int col = 0; col < 15; ++col) {
      Pr[row * 15 + col] = P_pred(row, col);
      Qr[row * 15 + col] = Q_(row, col);
      Sr[row * 15 + col] = STM_(row, col);
    }
  for (int row = 0; row < 4; ++row)
    for (int col = 0; col < 15; ++col) Hrow[row * 15 + col] = H_(row, col);
  const double pos[3] = {savePos[0], savePos[1], savePos[2]};
  const double init_llh[3] = {init_x, init_y, init_z}, init_ecef[3] = {init_ecef_x, init_ecef_y, init_ecef_z};
  const double now = clock_now();          // :107
  corenav::StopPrediction r = corenav::predict_stop(
      gp_data_.mean.data(), gp_data_.sigma.data(), (int)std::min(gp_data_.mean.size(), gp_data_.sigma.size()), Pr, Qr, Sr,
      Hrow, pos, gp_arrived_time_, now, xy_threshold, /*h_bug_compatible=*/false, init_llh, init_ecef);
  xy_errSlip = r.xy_err;
  if (r.fired) {
    stop_cmd_msg_.data = r.stop_cmd;       // :109,:114
    publish_stop_cmd();                    // :118
  }
  new_gp_data_arrived_ = false;            // :126
  i = 0;                                   // :127-128
  slip_i = 0;
}

GpPredictor::Vector3 GpPredictor::llh_to_enu(const double latitude, const double longitude, const double height) {
  const double init_llh[3] = {init_x, init_y, init_z}, init_ecef[3] = {init_ecef_x, init_ecef_y, init_ecef_z};
  double out[3];
  corenav::llh_to_enu(latitude, longitude, height, init_llh, init_ecef, out);
  Vector3 v;
  v[0] = out[0];
  v[1] = out[1];
  v[2] = out[2];
  return v;
}
