// Drives OUR GpPredictor in its ROS configuration (compiled against the doubles of this directory): constructs it on
// a ros::NodeHandle, delivers one GP_Output through the subscription it registered, answers its SetStopping call
// from a canned filter snapshot and prints what it subscribed to / advertised / called / published.
//   drive_gp_predictor <input.txt>   (M, mean[M], sigma[M], P[225], Q[225], STM[225], H[60], pos[3], arrival, now)
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "gp_predictor.h"

#ifndef CORENAV_HAVE_ROS
#error "this driver must be compiled in the ROS configuration (the doubles provide <ros/ros.h>)"
#endif
#ifndef CORENAV_HAVE_EIGEN
#error "this driver must be compiled with <Eigen/Dense> found"
#endif

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  std::ifstream in(argv[1]);
  int M = 0;
  in >> M;
  auto msg = std::make_shared<core_nav::GP_Output>();
  msg->mean.resize(M);
  msg->sigma.resize(M);
  for (double &x : msg->mean) in >> x;
  for (double &x : msg->sigma) in >> x;
  core_nav::SetStopping canned;
  for (double &x : canned.response.PvecData) in >> x;
  for (double &x : canned.response.QvecData) in >> x;
  for (double &x : canned.response.STMvecData) in >> x;
  for (double &x : canned.response.HvecData) in >> x;
  in >> canned.response.PosData.x >> canned.response.PosData.y >> canned.response.PosData.z;
  double arrival = 0, now = 0;
  in >> arrival >> now;
  if (!in) return 3;

  int clock_reads = 0, service_calls = 0;
  bool requested_stopping = false;
  ros::bus().clock = [&]() { return clock_reads++ == 0 ? arrival : now; };
  ros::bus().service = [&](const std::string &, void *p) {
    auto *srv = static_cast<core_nav::SetStopping *>(p);
    requested_stopping = srv->request.stopping;
    srv->response = canned.response;
    ++service_calls;
    return true;
  };
  ros::bus().params = {{"init_llh/x", 0.693457963620326}, {"init_llh/y", -1.39498384275845}, {"init_llh/z", 334.993517334743},
                       {"init_ecef/x", 859153.0153}, {"init_ecef/y", -4836303.7266}, {"init_ecef/z", 4055378.501}};

  ros::NodeHandle nh("");
  GpPredictor gp(nh);                       // gp_predictor.cpp:9-14: subscribes, creates the client, advertises
  const bool params_ok = gp.LoadParameters(nh);
  GpPredictor::Matrix unused;               // the reference's `typedef Eigen::MatrixXd Matrix` (gp_predictor.h:25) exists
  (void)unused;
  for (auto &s : ros::bus().subscriptions) std::printf("subscribed %s queue %d\n", s.first.c_str(), s.second.first);
  for (auto &a : ros::bus().advertised) std::printf("advertised %s queue %d\n", a.first.c_str(), a.second);
  for (auto &c : ros::bus().service_clients) std::printf("client %s\n", c.c_str());
  auto it = ros::bus().subscriptions.find("/core_nav/core_nav/gp_result");
  if (it == ros::bus().subscriptions.end()) return 4;
  core_nav::GP_Output::ConstPtr cp = msg;
  it->second.second(&cp);                   // the middleware delivers the message to GpPredictor::GPCallBack
  std::printf("params_ok %d service_calls %d requested_stopping %d clock_reads %d\n", (int)params_ok, service_calls,
              (int)requested_stopping, clock_reads);
  std::printf("published %zu\n", ros::bus().published.size());
  for (auto &p : ros::bus().published) std::printf("publish %s %.17g\n", p.first.c_str(), p.second);
  std::printf("xy_errSlip %.17g H_0_3 %.17g P_3_3 %.17g\n", gp.xy_errSlip, gp.H_(0, 3), gp.P_pred(3, 3));
  return 0;
}
