// Test double (see ../README.md): core_nav/SetStopping (srv/SetStopping.srv: bool stopping --- float64[225] x3,
// float64[60] HvecData, geometry_msgs/Point PosData); fixed-size arrays are boost::array in generated code
#pragma once
#include <array>
namespace core_nav {
struct SetStopping {
  struct Request { bool stopping = false; } request;
  struct Response {
    std::array<double, 225> PvecData{}, QvecData{}, STMvecData{};
    std::array<double, 60> HvecData{};
    struct { double x = 0, y = 0, z = 0; } PosData;
  } response;
};
}  // namespace core_nav
