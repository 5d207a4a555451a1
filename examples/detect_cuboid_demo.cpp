// detect_cuboid_demo.cpp -- C++ host calling the HIP path through the C-ABI, mirroring the reference demo
// (detect_3d_cuboid/src/main.cpp:27-74) on a raw gray image + edge txt.
//   build: g++ -O2 -std=c++17 examples/detect_cuboid_demo.cpp -Lcube_slam_amd -lcubeslam_hip -Wl,-rpath,$PWD/cube_slam_amd -o /tmp/demo
//   run  : /tmp/demo gray.raw width height edges.txt
#include <cstdio>
#include <fstream>
#include <sstream>

#include "../cube_slam_amd/host/detect_3d_cuboid.hpp"

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: %s gray.raw width height edges.txt\n", argv[0]); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    std::vector<unsigned char> img((size_t)W * H);
    std::ifstream f(argv[1], std::ios::binary);
    if (!f.read((char *)img.data(), (std::streamsize)img.size())) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    std::vector<double> edges;
    { std::ifstream e(argv[4]); double v; while (e >> v) edges.push_back(v); }
    const double Kalib[9] = {529.5, 0, 365.0, 0, 529.5, 265.0, 0, 0, 1.0};                                    // main.cpp:35-38
    const double T[16] = {1, 0.0011, 0.0004, 0, 0, -0.3376, 0.9413, 0, 0.0011, -0.9413, -0.3376, 1.35, 0, 0, 0, 1}; // :40-44
    std::vector<double> boxes = {188 - 1, 189 - 1, 201, 311, 0.88};                                           // :46-48
    cubeslam::Context ctx(0);
    cubeslam::detect_3d_cuboid det(ctx);
    det.set_calibration(Kalib);
    det.whether_sample_bbox_height = false;
    det.whether_sample_cam_roll_pitch = false;
    auto res = det.detect_cuboid(img.data(), W, H, 1, W, T, boxes, edges);
    for (size_t b = 0; b < res.size(); b++)
        for (const cs_cuboid &c : res[b])
            printf("box %zu: pos %.4f %.4f %.4f  rotY %.4f  scale %.4f %.4f %.4f  err %.5f\n", b, c.pos[0], c.pos[1], c.pos[2], c.rotY,
                   c.scale[0], c.scale[1], c.scale[2], c.normalized_error);
    return 0;
}
