// Driver of cube_slam_amd/host/local_ba_dynamic.hpp for tests/test_local_ba_dynamic.py: reads a window dumped as raw arrays (dump_window there) and either
// writes the graph the C++ twin builds (`graph`: no device needed) or runs the whole flow on the GPU through the C-ABI (`run`) and writes what the Python mirror's
// result is compared with.  Output: int32 count, then per array: int32 name length, name, int32 kind (0 f64, 1 i32, 2 u8), int32 count, data.
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>

#include "cube_slam_amd/host/local_ba_dynamic.hpp"

namespace {
struct Out {
    std::ofstream f; int n = 0;
    explicit Out(const char *path) : f(path, std::ios::binary) { int z = 0; f.write((char *)&z, 4); }
    template <class T> void put(const char *name, int kind, const T *p, size_t cnt) {
        const int ln = (int)strlen(name), c = (int)cnt;
        f.write((char *)&ln, 4); f.write(name, ln); f.write((char *)&kind, 4); f.write((char *)&c, 4); f.write((const char *)p, (std::streamsize)(cnt * sizeof(T))); n++;
    }
    void d(const char *name, const std::vector<double> &v) { put(name, 0, v.data(), v.size()); }
    void i(const char *name, const std::vector<int> &v) { put(name, 1, v.data(), v.size()); }
    void b(const char *name, const std::vector<uint8_t> &v) { put(name, 2, v.data(), v.size()); }
    ~Out() { f.seekp(0); f.write((char *)&n, 4); }
};
template <class T> std::vector<T> rd(std::ifstream &f, size_t n) { std::vector<T> v(n); f.read((char *)v.data(), (std::streamsize)(n * sizeof(T))); return v; }
} // namespace

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    try {
        std::ifstream f(argv[2], std::ios::binary);
        if (!f) throw std::runtime_error("missing window file");
        const auto h = rd<int>(f, 14);
        const auto sc = rd<double>(f, 12);
        const size_t n_kf = h[0], n_mp = h[2], n_obs = h[3], n_mo = h[4], n_ov = h[5], n_seq = h[6], n_up = h[7];
        cubeslam::DynamicWindow w;
        cubeslam::DynamicBAParams prm;
        w.n_local = h[1]; prm.img_width = h[8]; prm.img_height = h[9]; prm.build_worldframe_on_ground = h[10] != 0; prm.ba_dyna_pt_obj_cam = h[11] != 0; prm.ba_dyna_obj_velo = h[12] != 0;
        prm.ba_dyna_obj_cam = h[13] != 0;
        for (int i = 0; i < 9; i++) prm.K[i] = sc[i];
        prm.bf = sc[9]; prm.camera_object_BA_weight = sc[10]; prm.object_velocity_BA_weight = sc[11];
        { auto v = rd<long long>(f, n_kf); w.kf_id.assign(v.begin(), v.end()); }
        w.kf_pose = rd<double>(f, n_kf * 7); w.kf_stamp = rd<double>(f, n_kf); w.kf_cam_center = rd<double>(f, n_kf * 3);
        w.mp_pos = rd<double>(f, n_mp * 3); w.mp_nobs = rd<int>(f, n_mp); w.mp_dynamic = rd<uint8_t>(f, n_mp); w.mp_pos_to_obj = rd<double>(f, n_mp * 3); w.mp_best_mo = rd<int>(f, n_mp);
        w.obs_mp = rd<int>(f, n_obs); w.obs_kf = rd<int>(f, n_obs); w.obs_uv = rd<double>(f, n_obs * 2); w.obs_ur = rd<double>(f, n_obs); w.obs_inv_sigma2 = rd<double>(f, n_obs);
        { auto v = rd<long long>(f, n_mo); w.mo_id.assign(v.begin(), v.end()); }
        w.mo_meas_quality = rd<double>(f, n_mo); w.mo_largest_point_observations = rd<int>(f, n_mo); w.mo_velocity = rd<double>(f, n_mo * 2);
        w.ov_mo = rd<int>(f, n_ov); w.ov_kf = rd<int>(f, n_ov); w.ov_pose = rd<double>(f, n_ov * 7); w.ov_bbox_vec = rd<double>(f, n_ov * 4); w.ov_bbox_2d = rd<int>(f, n_ov * 4);
        w.ov_left_right_to_car = rd<int>(f, n_ov);
        w.seq_mo = rd<int>(f, n_seq); w.seq_kf = rd<int>(f, n_seq);
        w.up_mo = rd<int>(f, n_up); w.up_pos = rd<double>(f, n_up * 3); w.up_count = rd<int>(f, n_up);
        if (!f) throw std::runtime_error("short window file");
        Out o(argv[3]);
        if (!strcmp(argv[1], "graph")) {
            const cubeslam::DynamicGraphArrays d = cubeslam::build_dynamic_graph(w, prm);
            o.d("cam_pose", d.cam_pose); o.b("cam_fixed", d.cam_fixed); o.d("obj_pose", d.obj_pose); o.d("obj_scale", d.obj_scale); o.b("obj_flags", d.obj_flags); o.d("vel", d.vel);
            o.d("points", d.points); o.d("dpoints", d.dpoints); o.i("obs_cam", d.obs_cam); o.i("obs_point", d.obs_point); o.d("obs_uv", d.obs_uv); o.d("obs_ur", d.obs_ur);
            o.d("obs_inv_sigma2", d.obs_w); o.i("dobs_cam", d.dobs_cam); o.i("dobs_obj", d.dobs_obj); o.i("dobs_point", d.dobs_point); o.d("dobs_uv", d.dobs_uv); o.d("dobs_inv_sigma2", d.dobs_w);
            o.i("mot_from", d.mot_from); o.i("mot_to", d.mot_to); o.i("mot_vel", d.mot_vel); o.d("mot_dt", d.mot_dt); o.i("cobs_cam", d.cobs_cam); o.i("cobs_obj", d.cobs_obj);
            o.d("cobs_bbox", d.cobs_bbox); o.d("cobs_info", d.cobs_info); o.b("cobs_level", d.cobs_level); o.i("pc_obj", d.pc_obj); o.i("pc_offsets", d.pc_offsets); o.d("pc_points", d.pc_points);
            std::vector<double> s{d.fx, d.fy, d.cx, d.cy, d.bf, d.huber_mono, d.huber_stereo, d.huber_dyn, d.huber_obj, d.ulp_info, d.ulp_ratio, d.pc_ratio};
            s.insert(s.end(), d.ulp_scale, d.ulp_scale + 3); s.insert(s.end(), d.mot_info, d.mot_info + 3); s.insert(s.end(), d.K, d.K + 9);
            o.d("scalars", s);
            o.i("point_rows", d.point_rows); o.i("obs_rows", d.obs_rows); o.i("dpoint_rows", d.dpoint_rows); o.i("dobs_rows", d.dobs_rows); o.i("cobs_rows", d.cobs_rows); o.i("vel_mo", d.vel_mo);
            o.i("up_used", d.up_used); o.i("up_filtered", d.up_filtered);
            return 0;
        }
        cubeslam::Context ctx(0);
        cubeslam::DynamicBAResult r;
        cubeslam::LocalBACameraPointObjectsDynamic(ctx, w, prm, r);
        std::vector<int> er;
        for (auto &e : r.erase) { er.push_back(e.first); er.push_back(e.second); }
        o.i("erase", er); o.b("erase_stereo", r.erase_stereo); o.i("point_unwritten", r.point_unwritten); o.i("point_rows", r.point_rows); o.d("point_pos", r.point_pos); o.d("kf_pose", r.kf_pose);
        o.d("vertex_pose", r.vertex_pose); o.i("object_latest", r.object_latest); o.i("vel_mo", r.vel_mo); o.d("velocity", r.velocity); o.i("dpoint_rows", r.dpoint_rows);
        o.d("dpoint_local", r.dpoint_local); o.i("dworld_rows", r.dworld_rows); o.d("dpoint_world", r.dpoint_world);
    } catch (const std::exception &e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
    return 0;
}
