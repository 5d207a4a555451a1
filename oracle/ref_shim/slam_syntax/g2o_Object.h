// TEST INFRASTRUCTURE: forwards to the stand-in declarations (see slam_standins.hpp)
#pragma once
#include "slam_standins.hpp"
