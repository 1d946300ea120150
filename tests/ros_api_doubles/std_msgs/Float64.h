// Test double (see ../README.md): std_msgs/Float64
#pragma once
#include <memory>
namespace std_msgs { struct Float64 { double data = 0; typedef std::shared_ptr<const Float64> ConstPtr; }; }
