// Test double (see ../README.md): core_nav/GP_Input (msg/GP_Input.msg: float64[] time_array, float64[] slip_array)
#pragma once
#include <memory>
#include <vector>
namespace core_nav { struct GP_Input { std::vector<double> time_array, slip_array; typedef std::shared_ptr<const GP_Input> ConstPtr; }; }
