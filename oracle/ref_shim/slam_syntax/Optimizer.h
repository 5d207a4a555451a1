// TEST INFRASTRUCTURE: the stand-in declarations, then `class Optimizer` as the reference's include/Optimizer.h:33-62 declares it
// (Optimizer_decl.inc is cut from that file by tests/test_adapters.py)
#pragma once
#include "slam_standins.hpp"
namespace ORB_SLAM2 {
#include "Optimizer_decl.inc"
} // namespace ORB_SLAM2
