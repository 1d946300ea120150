// Test double (see ../README.md)
#pragma once
#include "ros.h"
#define ROS_INFO(...) ::ros::bus().log.push_back("INFO")
#define ROS_ERROR(...) ::ros::bus().log.push_back("ERROR")
