// TEST INFRASTRUCTURE: the window handle of oracle/ref_shim/ref_graph_api.cpp, shared with adapter_graph_api.cpp (the adapters run over the same stand-in map)
#pragma once
#include <memory>
#include <vector>

#include "slam_graph_standins.hpp"

struct ref_graph {
    std::vector<std::unique_ptr<ORB_SLAM2::KeyFrame>> kfs;
    std::vector<std::unique_ptr<ORB_SLAM2::MapPoint>> mps;
    std::vector<std::unique_ptr<ORB_SLAM2::MapObject>> mos, dets; // landmarks; per-frame detections (KeyFrame::local_cuboids)
    ORB_SLAM2::Map map;
    ORB_SLAM2::EraseLog log;
    std::streambuf *cout_was = nullptr;
};
