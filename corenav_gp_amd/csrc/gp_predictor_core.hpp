// gp_predictor_core.hpp -- ROS-free, Eigen-free arithmetic of the reference's GpPredictor node
// (gp_predictor/src/gp_predictor.cpp).  Plain fixed-size arrays; row-major 15x15 / 4x15.
#pragma once

namespace corenav {

struct StopPrediction {
  bool fired = false;    // xy error crossed the threshold inside the horizon (gp_predictor.cpp:102)
  double stop_cmd = 0.0; // value published on stop_cmd (gp_predictor.cpp:107-118)
  int i = 0;             // odometry-rate steps consumed (member `i`, gp_predictor.cpp:92)
  double xy_err = 0.0;   // last horizontal 3-sigma error (gp_predictor.cpp:99)
};

// GpPredictor::llh_to_enu (gp_predictor.cpp:144-178)
void llh_to_enu(double lat, double lon, double h, const double init_llh[3], const double init_ecef[3],
                double enu[3]);

// Unpack HvecData[60] into a 4x15 row-major H.  bug_compatible reproduces the reference's
// `row1*4+col1` indexing (gp_predictor.cpp:38-42); otherwise row*15+col.
void unpack_H(const double *HvecData, bool bug_compatible, double H[60]);

// Covariance look-ahead of GpPredictor::GPCallBack (gp_predictor.cpp:58-130).
StopPrediction predict_stop(const double *mean, const double *sigma, int M, const double *PvecData,
                            const double *QvecData, const double *STMvecData, const double *HvecData,
                            const double pos_llh[3], double arrival_time, double now, double threshold,
                            bool h_bug_compatible, const double init_llh[3], const double init_ecef[3]);

}  // namespace corenav
