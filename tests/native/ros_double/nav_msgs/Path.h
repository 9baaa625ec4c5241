#pragma once
#include "nav_msgs/Odometry.h"
