// Test double (see ../README.md): core_nav/GP_Output (msg/GP_Output.msg: float64[] mean, float64[] sigma)
#pragma once
#include <memory>
#include <vector>
namespace core_nav { struct GP_Output { std::vector<double> mean, sigma; typedef std::shared_ptr<const GP_Output> ConstPtr; }; }
