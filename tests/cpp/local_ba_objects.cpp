// Driver of cube_slam_amd/host/local_ba_objects.hpp for tests/test_local_ba_objects.py: reads a window dumped as raw arrays, runs the C++
// mirror of Optimizer::LocalBACameraPointObjects on the GPU through the C-ABI and prints what the Python mirror's result is compared with.
#include <cstdio>
#include <fstream>
#include <string>

#include "cube_slam_amd/host/local_ba_objects.hpp"

template <class T> static std::vector<T> rd(const std::string &dir, const char *name) {
    std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
    if (!f) throw std::runtime_error(std::string("missing ") + name);
    f.seekg(0, std::ios::end); const size_t n = (size_t)f.tellg() / sizeof(T); f.seekg(0);
    std::vector<T> v(n);
    f.read((char *)v.data(), (std::streamsize)(n * sizeof(T)));
    return v;
}
static unsigned long long fnv(const void *p, size_t n) { unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) h = (h ^ ((const unsigned char *)p)[i]) * 1099511628211ull; return h; }

int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        cubeslam::Context ctx(0);
        cubeslam::LocalWindow w;
        { auto v = rd<long long>(dir, "kf_id"); w.kf_id.assign(v.begin(), v.end()); }
        w.kf_pose = rd<double>(dir, "kf_pose"); w.mp_pos = rd<double>(dir, "mp_pos"); w.mp_nobs = rd<int>(dir, "mp_nobs");
        w.obs_mp = rd<int>(dir, "obs_mp"); w.obs_kf = rd<int>(dir, "obs_kf"); w.obs_uv = rd<double>(dir, "obs_uv"); w.obs_ur = rd<double>(dir, "obs_ur");
        w.obs_inv_sigma2 = rd<double>(dir, "obs_inv_sigma2");
        w.mo_pose = rd<double>(dir, "mo_pose"); w.mo_scale = rd<double>(dir, "mo_scale"); w.mo_meas_quality = rd<double>(dir, "mo_meas_quality");
        w.mo_largest_point_observations = rd<int>(dir, "mo_largest_point_observations");
        w.up_mo = rd<int>(dir, "up_mo"); w.up_count = rd<int>(dir, "up_count"); w.up_pos = rd<double>(dir, "up_pos");
        w.det_mo = rd<int>(dir, "det_mo"); w.det_kf = rd<int>(dir, "det_kf"); w.det_bbox_2d = rd<int>(dir, "det_bbox_2d"); w.det_left_right_to_car = rd<int>(dir, "det_left_right_to_car");
        w.det_bbox_vec = rd<double>(dir, "det_bbox_vec");
        const auto sc = rd<double>(dir, "scalars"); // n_local, cur_cam_center[3], K[9], img_width, img_height, bf, camera_object_BA_weight
        w.n_local = (int)sc[0];
        for (int i = 0; i < 3; i++) w.cur_cam_center[i] = sc[1 + i];
        cubeslam::LocalBAParams prm;
        for (int i = 0; i < 9; i++) prm.K[i] = sc[4 + i];
        prm.img_width = (int)sc[13]; prm.img_height = (int)sc[14]; prm.bf = sc[15]; prm.camera_object_BA_weight = sc[16];
        cubeslam::LocalBAResult r;
        cubeslam::LocalBACameraPointObjects(ctx, w, prm, r);
        printf("levels %zu %llx %zu %llx %llx\n", r.obs_level.size(), fnv(r.obs_level.data(), r.obs_level.size()), r.cobs_level.size(), fnv(r.cobs_level.data(), r.cobs_level.size()),
               fnv(r.cobs_level2.data(), r.cobs_level2.size()));
        printf("erase %zu", r.erase.size());
        for (auto &e : r.erase) printf(" %d:%d", e.first, e.second);
        printf("\nstats %d %d %.17g %.17g\n", r.st1.iterations, r.st2.iterations, r.st1.chi2_final, r.st2.chi2_final);
        printf("kf"); for (double v : r.kf_pose) printf(" %.17g", v);
        printf("\npoints"); for (double v : r.point_pos) printf(" %.17g", v);
        printf("\nobjects"); for (double v : r.object_pose) printf(" %.17g", v);
        printf("\nunwritten %zu", r.point_unwritten.size()); for (int v : r.point_unwritten) printf(" %d", v);
        printf("\n");
    } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
