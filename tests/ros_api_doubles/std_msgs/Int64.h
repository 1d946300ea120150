// Test double (see ../README.md): std_msgs/Int64
#pragma once
#include <memory>
namespace std_msgs { struct Int64 { long data = 0; typedef std::shared_ptr<const Int64> ConstPtr; }; }
