// Exercises the C++ host mirrors (cube_slam_amd/host/*.hpp) end to end on one gray frame read from a raw file; prints what the Python
// test compares with the Python mirrors' results: keypoint / descriptor / KeyLine checksums.
#include <cstdio>
#include <cstring>
#include <vector>

#include "cube_slam_amd/host/orb_slam_mirrors.hpp"

static unsigned long long fnv(const void *p, size_t n) { const unsigned char *b = (const unsigned char *)p; unsigned long long h = 1469598103934665603ull; for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } return h; }

template <class T> static std::vector<T> read_bin(const std::string &dir, const char *name) {
    std::vector<T> v;
    FILE *f = fopen((dir + "/" + name + ".bin").c_str(), "rb");
    if (!f) return v;
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)n / sizeof(T));
    if (n && fread(v.data(), 1, (size_t)n, f) != (size_t)n) v.clear();
    fclose(f);
    return v;
}

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    std::vector<uint8_t> img((size_t)W * H);
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(img.data(), 1, img.size(), f) != img.size()) return 3;
    fclose(f);
    try {
        cubeslam::Context ctx(0);
        cubeslam::ORBextractor orb(ctx, 500, 1.2f, 8, 20, 7, W, H);
        std::vector<cs_keypoint> kps; std::vector<uint8_t> desc;
        orb(img.data(), W, kps, desc);
        printf("orb %zu %llx %llx\n", kps.size(), fnv(kps.data(), kps.size() * sizeof(cs_keypoint)), fnv(desc.data(), desc.size()));
        printf("levels %d sf1 %.9g\n", orb.GetLevels(), (double)orb.GetScaleFactors()[1]);
        cubeslam::line_lbd_detect ld(ctx, W, H);
        std::vector<cs_keyline> kl; std::vector<float> lm; std::vector<uint8_t> ldesc;
        ld.detect_raw_lines(img.data(), W, kl);
        ld.detect_filter_lines(img.data(), W, lm);
        ld.get_line_descriptors(img.data(), W, kl, ldesc);
        printf("lines %zu %llx filtered %zu %llx lbd %llx\n", kl.size(), fnv(kl.data(), kl.size() * sizeof(cs_keyline)), lm.size() / 4, fnv(lm.data(), lm.size() * 4), fnv(ldesc.data(), ldesc.size()));
        if (argc > 4) { // a dynamic-BA graph dumped by tests/test_host_cpp_gpu.py: <dir>/<name>.bin raw arrays + <dir>/scalars.bin (doubles)
            const std::string dir = argv[4];
            cubeslam::Optimizer::DynamicGraph g;
            cs_ba_dyn_problem &P = g.P;
            std::vector<uint8_t> cam_fixed = read_bin<uint8_t>(dir, "cam_fixed"), obj_flags = read_bin<uint8_t>(dir, "obj_flags");
            std::vector<double> obj_scale = read_bin<double>(dir, "obj_scale"), obs_uv = read_bin<double>(dir, "obs_uv"), obs_ur = read_bin<double>(dir, "obs_ur"),
                                obs_w = read_bin<double>(dir, "obs_inv_sigma2"), dobs_uv = read_bin<double>(dir, "dobs_uv"), dobs_w = read_bin<double>(dir, "dobs_inv_sigma2"),
                                mot_dt = read_bin<double>(dir, "mot_dt"), cobs_bbox = read_bin<double>(dir, "cobs_bbox"), cobs_info = read_bin<double>(dir, "cobs_info"),
                                pc_points = read_bin<double>(dir, "pc_points"), sc = read_bin<double>(dir, "scalars");
            std::vector<int> obs_cam = read_bin<int>(dir, "obs_cam"), obs_point = read_bin<int>(dir, "obs_point"), dobs_cam = read_bin<int>(dir, "dobs_cam"),
                             dobs_obj = read_bin<int>(dir, "dobs_obj"), dobs_point = read_bin<int>(dir, "dobs_point"), mot_from = read_bin<int>(dir, "mot_from"),
                             mot_to = read_bin<int>(dir, "mot_to"), mot_vel = read_bin<int>(dir, "mot_vel"), cobs_cam = read_bin<int>(dir, "cobs_cam"),
                             cobs_obj = read_bin<int>(dir, "cobs_obj"), pc_obj = read_bin<int>(dir, "pc_obj"), pc_offsets = read_bin<int>(dir, "pc_offsets");
            g.cam_pose = read_bin<double>(dir, "cam_pose"); g.obj_pose = read_bin<double>(dir, "obj_pose"); g.vel = read_bin<double>(dir, "vel");
            g.points = read_bin<double>(dir, "points"); g.dpoints = read_bin<double>(dir, "dpoints");
            P.n_cams = (int)cam_fixed.size(); P.cam_fixed = cam_fixed.data();
            P.n_objs = (int)obj_flags.size(); P.obj_scale = obj_scale.data(); P.obj_flags = obj_flags.data();
            P.n_vels = (int)g.vel.size() / 2; P.n_points = (int)g.points.size() / 3; P.n_dpoints = (int)g.dpoints.size() / 3; P.fix_points = 0;
            P.n_obs = (int)obs_cam.size(); P.obs_cam = obs_cam.data(); P.obs_point = obs_point.data(); P.obs_uv = obs_uv.data(); P.obs_ur = obs_ur.data(); P.obs_inv_sigma2 = obs_w.data();
            P.n_dobs = (int)dobs_cam.size(); P.dobs_cam = dobs_cam.data(); P.dobs_obj = dobs_obj.data(); P.dobs_point = dobs_point.data(); P.dobs_uv = dobs_uv.data();
            P.dobs_inv_sigma2 = dobs_w.data();
            P.n_mot = (int)mot_from.size(); P.mot_from = mot_from.data(); P.mot_to = mot_to.data(); P.mot_vel = mot_vel.data(); P.mot_dt = mot_dt.data();
            P.n_cobs = (int)cobs_cam.size(); P.cobs_cam = cobs_cam.data(); P.cobs_obj = cobs_obj.data(); P.cobs_bbox = cobs_bbox.data(); P.cobs_info = cobs_info.data();
            P.n_pc = (int)pc_obj.size(); P.pc_obj = pc_obj.data(); P.pc_offsets = pc_offsets.data(); P.pc_points = pc_points.data();
            // scalars: fx fy cx cy bf huber_mono huber_stereo ulp_info ulp_scale[3] ulp_ratio K[9] huber_dyn mot_info[3] huber_obj pc_ratio
            const double *q = sc.data();
            P.fx = q[0]; P.fy = q[1]; P.cx = q[2]; P.cy = q[3]; P.bf = q[4]; P.huber_mono = q[5]; P.huber_stereo = q[6]; P.ulp_info = q[7];
            for (int k = 0; k < 3; k++) P.ulp_scale[k] = q[8 + k];
            P.ulp_ratio = q[11];
            for (int k = 0; k < 9; k++) P.K[k] = q[12 + k];
            P.huber_dyn = q[21];
            for (int k = 0; k < 3; k++) P.mot_info[k] = q[22 + k];
            P.huber_obj = q[25]; P.pc_ratio = q[26];
            cs_ba_stats s1, s2;
            cubeslam::Optimizer::LocalBACameraPointObjectsDynamic(ctx, g, nullptr, &s1, &s2);
            long n1[3] = {0, 0, 0};
            for (int o = 0; o < P.n_obs; o++) n1[0] += g.obs_level[o];
            for (int o = 0; o < P.n_dobs; o++) n1[1] += g.dobs_level[o];
            for (int o = 0; o < P.n_cobs; o++) n1[2] += g.cobs_level[o];
            printf("dynba %d %d %.12g %.12g %llx %llx %llx %ld %ld %ld\n", s1.iterations, s2.iterations, s1.chi2_final, s2.chi2_final, fnv(g.obs_level.data(), (size_t)P.n_obs),
                   fnv(g.dobs_level.data(), (size_t)P.n_dobs), fnv(g.cobs_level.data(), (size_t)P.n_cobs), n1[0], n1[1], n1[2]);
            printf("dynpose %.12g %.12g %.12g %.12g\n", g.cam_pose[(size_t)(P.n_cams - 1) * 7], g.obj_pose[0], g.vel[0], g.dpoints[0]);
        }
    } catch (const std::exception &e) { fprintf(stderr, "%s\n", e.what()); return 1; }
    return 0;
}
