// Test double (see ../README.md): std_msgs/Bool
#pragma once
#include <memory>
namespace std_msgs { struct Bool { bool data = 0; typedef std::shared_ptr<const Bool> ConstPtr; }; }
