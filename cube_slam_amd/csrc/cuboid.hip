// cuboid.hip -- detect_3d_cuboid proposal sweep on MI355X (gfx950).
//
// Replaces detect_3d_cuboid::detect_cuboid (reference detect_3d_cuboid/src/box_proposal_detail.cpp:56-557) and its
// callees in object_3d_util.cpp / matrix_utils.cpp.  Built with -ffp-contract=off: the reference is compiled
// without FMA and int() truncation of samples lying exactly on integer box borders depends on it.
//
// Pipeline for a batch of frames (unit = (frame, box, height-sample)):
//   cuboid_frame_prep   per frame : roll/pitch/yaw sample lists (linespace accumulation, matrix_utils.cpp:349-363),
//                                   per-(roll,pitch) camera tables (set_cam_pose :42-54), align_left_right_edges
//   cuboid_unit_lines   per unit  : lines inside ROI, merge_break_lines (object_3d_util.cpp:300-376), angles, midpoints
//   cuboid_canny_nms    per pixel : Sobel + L1 magnitude + direction NMS (cv::Canny 80/200), LDS-tiled
//   cuboid_canny_union/mark/resolve : hysteresis as union-find connected components (same result as the stack flood)
//   cuboid_dt           per unit  : 3x3 chamfer distance transform, one wave per ROI, rows as DPP min-plus scans
//   cuboid_vp           per (unit,roll,pitch,yaw): getVanishingPoints + VP_support_edge_infos (:380-425,:602-607)
//   cuboid_dt_codes     per pixel : the float distances as exact 16-bit (straight, diagonal) step codes for the score kernel
//   cuboid_sweep_corners per hypothesis: corner construction with all reject tests (:254-418), survivors compacted per unit
//   cuboid_score_plan   one workgroup: the units' proposal lists laid on a cost line and cut into one segment per CU
//   cuboid_sweep_score  per surviving proposal: box_edge_sum_dists + box_edge_alignment_angle_error (:427-492), the unit's code map
//                                   resident in LDS (its tail gathered from global memory when it is larger than LDS)
//   cuboid_sweep_score_big          the same from the float map, for units with a pixel too far from every edge for a code
//   cuboid_select       per box   : fuse_normalize_scores_v2 (:495-565) by radix selection, 2D->3D
//                                   (change_2d_corner_to_3d_object :610-648), final ranking (:517-536)
#include "common.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

namespace {

constexpr int ROLL_CAP = 8;
constexpr int RP_CAP = ROLL_CAP * ROLL_CAP;
constexpr int SWEEP_HB = 1024;  // hypotheses per sweep workgroup
constexpr int NMS_TW = 64, NMS_TH = 16;
constexpr int DT_INIT = INT_MAX >> 2;
constexpr int DT_HV = 62587;    // cvRound(0.955f * 65536)
constexpr int DT_DIAG = 89738;  // cvRound(1.3693f * 65536)  (checked against the float product on the host at create())
constexpr double PI = 3.14159265358979323846;
// cuboid_sweep_score (corner construction + edge scoring, the unit's chamfer map resident in LDS as 16-bit codes), see the kernel
constexpr int SC_LDS_BYTES = 160 * 1024;
constexpr int SC_LUT_N = 1024;                                         // residue buckets of 64: (DT_HV + 63) / 64 = 978 used
constexpr int SC_CTRL_BYTES = 64;                                     // control words in front of the table
constexpr int SC_MAP_OFF = SC_CTRL_BYTES + 2048;                       // control words + the encoder's residue table (978 x 2 B, padded)
constexpr int SC_MAP_ENTRIES = (SC_LDS_BYTES - SC_MAP_OFF) / 2;        // 80 864
constexpr float SC_ESC_D = 244.0f;                                     // d < 244 => i <= 255 and j <= 178
constexpr int K_VIS1[9][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {1, 5}, {2, 4}, {3, 7}, {4, 7}, {4, 5}}; // box_proposal_detail.cpp:432
constexpr int K_VIS2[7][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 0}, {1, 5}, {2, 4}, {4, 5}};                 // :442
constexpr int K_VPE1[3][4] = {{0, 1, 7, 4}, {3, 0, 4, 5}, {3, 7, 1, 5}};                               // :434
constexpr int K_VPE2[3][4] = {{0, 1, 2, 3}, {3, 0, 4, 5}, {2, 4, 1, 5}};                               // :444
__host__ __device__ inline bool score_unit_fits(int roi_w, int roi_h) { return (long)roi_w * roi_h + (roi_w + 2 > 8 ? roi_w + 2 : 8) <= (long)SC_MAP_ENTRIES; }


struct Calib { double K[9]; double invK[9]; };

struct FrameInfo {   // host-filled
    double euler[3]; // raw camera roll/pitch/yaw (quat_to_euler_zyx(Quaterniond(R)), computed with the host libm)
    double yaw_src;  // cam_pose.camera_yaw as box_proposal_detail.cpp:126 reads it: the raw yaw, or (cs_cuboid_detect's box-to-box chain, D1) what the previous box left in cam_pose
    double T[16];
    int line_off, n_lines;
};
struct FrameDyn {    // device-filled by cuboid_frame_prep
    int n_roll, n_pitch, n_yaw, pad;
    double roll[ROLL_CAP], pitch[ROLL_CAP];
};
struct CamRP { double T[16]; double KinvR[9]; double gps[4]; double roll, pitch; };

struct Unit {        // host-filled plan of one (frame, box, height-sample)
    int frame, box, hs, n_hs;
    int left, top, right, width_raw, height_raw;
    int down_expand, down_y_expan;
    int n_tops, top_start, top_step;
    unsigned tops_magic; // floor(2^32 / n_tops) + 1 when __umulhi(p, tops_magic) == p / n_tops for every pair index p of the unit, else 0 (divide)
    int roi_x, roi_y, roi_w, roi_h, roi_r, roi_b;
    int hyp_cap;
    int vp_off;      // entries
    int line_off;    // rows in merged-line storage
    double diag;
    long pix_off;
    long hyp_off;
    int nms_off, cc_off, hb_off; // first workgroup of the unit in the grids of cuboid_canny_nms (64 x 16 tiles), cuboid_canny_cc_local / _cc (4 096-pixel bands) and
    int pad_;                    // cuboid_sweep_filter (1 024 hypotheses): a workgroup per piece that EXISTS (grids of max-pieces x units were 40 - 50 % empty workgroups)
};
struct UnitDyn { int n_merged, n_valid, n_kept, branch_b; double pad; };

struct Opts {
    int cfg1, cfg2, sample_rp, max_cuboid_num;
    double nominal_skew_ratio, max_cut_skew, yaw_range_deg, yaw_step_deg;
    int canny_low, canny_high, yaw_cap, pad;
};

struct V2 { double x, y; };

// ------------------------------------------------------------------------------------------------ small device math
__device__ __forceinline__ double cof3(const double *a, int i, int j) {
    int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
    return a[i1 * 3 + j1] * a[i2 * 3 + j2] - a[i1 * 3 + j2] * a[i2 * 3 + j1];
}
__host__ __device__ inline void inv3_cof(const double *a, double *r) { // Eigen fixed 3x3 inverse: cofactors * (1/det)
    auto cf = [&](int i, int j) {
        int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
        return a[i1 * 3 + j1] * a[i2 * 3 + j2] - a[i1 * 3 + j2] * a[i2 * 3 + j1];
    };
    double c00 = cf(0, 0), c10 = cf(1, 0), c20 = cf(2, 0);
    double det = (c00 * a[0] + c10 * a[3]) + c20 * a[6];
    double invdet = 1.0 / det;
    r[0] = c00 * invdet; r[1] = c10 * invdet; r[2] = c20 * invdet;
    r[3] = cf(0, 1) * invdet; r[4] = cf(1, 1) * invdet; r[5] = cf(2, 1) * invdet;
    r[6] = cf(0, 2) * invdet; r[7] = cf(1, 2) * invdet; r[8] = cf(2, 2) * invdet;
}
__device__ __forceinline__ double normalize_to_pi(double a) { // matrix_utils.cpp:326-335
    if (a > PI / 2) return a - PI;
    else if (a < -PI / 2) return a + PI;
    else return a;
}
// normalize_to_pi(atan2(dy, dx)) (matrix_utils.cpp:326-335) to within 1e-16: the level-line angle of an edge in (-pi/2, pi/2].  One
// division, a degree-10 polynomial in z^2 on |z| <= tan(pi/8) (argument reduction (mn - mx) / (mn + mx)), ~45 instructions against the 105
// of the library atan2.  The sign at an exactly vertical edge (+-pi/2) is irrelevant to the callers (min(t, pi - t)).
__device__ __forceinline__ double line_angle_fast(double dy, double dx) {
    const double ax = fabs(dx), ay = fabs(dy);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    const bool big = mn > 0.41421356237309503 * mx;
    const double num = big ? mn - mx : mn;
    const double den = fmax(big ? mn + mx : mx, 1e-300);
    double r = __builtin_amdgcn_rcp(den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    double z = num * r;
    z = __builtin_fma(__builtin_fma(-den, z, num), r, z);
    const double w = z * z;
    double p = 0.021135373157693246;
    p = __builtin_fma(p, w, -0.04348052215716462);
    p = __builtin_fma(p, w, 0.056883492268090106);
    p = __builtin_fma(p, w, -0.06640233930429408);
    p = __builtin_fma(p, w, 0.07689953496306857);
    p = __builtin_fma(p, w, -0.09090773074808414);
    p = __builtin_fma(p, w, 0.11111106180455946);
    p = __builtin_fma(p, w, -0.14285714180976467);
    p = __builtin_fma(p, w, 0.1999999999885511);
    p = __builtin_fma(p, w, -0.3333333333332844);
    double th = __builtin_fma(z * w, p, z);
    th = big ? th + PI / 4 : th;
    th = ay > ax ? PI / 2 - th : th;
    return ((dx < 0) != (dy < 0)) ? -th : th;
}
__device__ __forceinline__ double dist2(V2 a, V2 b) { double dx = a.x - b.x, dy = a.y - b.y; return sqrt(dx * dx + dy * dy); }
__device__ __forceinline__ bool inside_box(V2 p, double l, double t, double r, double b) {
    return l <= p.x && p.x <= r && t <= p.y && p.y <= b;
}
__device__ __forceinline__ V2 seg_hit_boundary(V2 ps, V2 pe, double bx0, double by0, double bx1, double by1) { // object_3d_util.cpp:194-230
    V2 direc{pe.x - ps.x, pe.y - ps.y};
    V2 hit{-1, -1};
    if (by0 == by1) {
        double lambd = (by0 - ps.y) / direc.y;
        if (lambd >= 0) {
            V2 t{ps.x + lambd * direc.x, ps.y + lambd * direc.y};
            if ((bx0 <= t.x) && (t.x <= bx1)) { hit = t; hit.y = by0; }
        }
    }
    if (bx0 == bx1) {
        double lambd = (bx0 - ps.x) / direc.x;
        if (lambd >= 0) {
            V2 t{ps.x + lambd * direc.x, ps.y + lambd * direc.y};
            if ((by0 <= t.y) && (t.y <= by1)) { hit = t; hit.x = bx0; }
        }
    }
    return hit;
}
__device__ __forceinline__ V2 line_intersect_inf(V2 p1s, V2 p1e, V2 p2s, V2 p2e) { // :233-252, infinite_line=true
    double X2_X1 = p1e.x - p1s.x, Y2_Y1 = p1e.y - p1s.y;
    double X4_X3 = p2e.x - p2s.x, Y4_Y3 = p2e.y - p2s.y;
    double X1_X3 = p1s.x - p2s.x, Y1_Y3 = p1s.y - p2s.y;
    double u_a = (X4_X3 * Y1_Y3 - Y4_Y3 * X1_X3) / (Y4_Y3 * X2_X1 - X4_X3 * Y2_Y1);
    double INT_X = p1s.x + X2_X1 * u_a;
    double INT_Y = p1s.y + Y2_Y1 * u_a;
    return V2{INT_X * 1.0, INT_Y * 1.0};
}

// ------------------------------------------------------------------------------------------------ frame prep
__device__ inline int linespace_dev(double starting, double ending, double step, double *res, int cap) { // matrix_utils.cpp:349-363
    int n = 0;
    while (starting <= ending) {
        if (n < cap) res[n] = starting;
        n++;
        starting += step;
        if (n > 1000) break;
    }
    return n < cap ? n : cap;
}

__global__ void __launch_bounds__(64) cuboid_frame_prep(const FrameInfo *fi, FrameDyn *fd, CamRP *cam, double *yaw, Calib cal, Opts o,
                                                        const double *lines_in, double *lines_al) {
    const int f = blockIdx.x, tid = threadIdx.x;
    const FrameInfo &F = fi[f];
    FrameDyn &D = fd[f];
    if (tid == 0) {
        if (o.sample_rp) { // box_proposal_detail.cpp:217-221
            D.n_roll = linespace_dev(F.euler[0] - 6.0 / 180.0 * PI, F.euler[0] + 6.0 / 180.0 * PI, 3.0 / 180.0 * PI, D.roll, ROLL_CAP);
            D.n_pitch = linespace_dev(F.euler[1] - 6.0 / 180.0 * PI, F.euler[1] + 6.0 / 180.0 * PI, 3.0 / 180.0 * PI, D.pitch, ROLL_CAP);
        } else {
            D.n_roll = 1; D.n_pitch = 1; D.roll[0] = F.euler[0]; D.pitch[0] = F.euler[1];
        }
        double yaw_init = F.yaw_src - 90.0 / 180.0 * PI; // :126 (the raw yaw for every box of a batch; cs_cuboid_detect chains the boxes of its frame like the reference)
        D.n_yaw = linespace_dev(yaw_init - o.yaw_range_deg / 180.0 * PI, yaw_init + o.yaw_range_deg / 180.0 * PI,
                                o.yaw_step_deg / 180.0 * PI, yaw + (long)f * o.yaw_cap, o.yaw_cap);
    }
    __syncthreads();
    if (tid < D.n_roll * D.n_pitch) {
        CamRP &C = cam[(long)f * RP_CAP + tid];
        int ri = tid / D.n_pitch, pi = tid % D.n_pitch;
        double R[9];
        if (o.sample_rp) { // euler_zyx_to_rot matrix_utils.cpp:74-89
            double roll = D.roll[ri], pitch = D.pitch[pi], yw = F.euler[2];
            double cp = cos(pitch), sp = sin(pitch), sr = sin(roll), cr = cos(roll), sy = sin(yw), cy = cos(yw);
            R[0] = cp * cy; R[1] = (sr * sp * cy) - (cr * sy); R[2] = (cr * sp * cy) + (sr * sy);
            R[3] = cp * sy; R[4] = (sr * sp * sy) + (cr * cy); R[5] = (cr * sp * sy) - (sr * cy);
            R[6] = -sp;     R[7] = sr * cp;                    R[8] = cr * cp;
        } else
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R[i * 3 + j] = F.T[i * 4 + j];
        for (int i = 0; i < 16; i++) C.T[i] = F.T[i];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) C.T[i * 4 + j] = R[i * 3 + j];
        double invR[9];
        inv3_cof(R, invR);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double s = cal.K[i * 3 + 0] * invR[0 * 3 + j];
                s = s + cal.K[i * 3 + 1] * invR[1 * 3 + j];
                s = s + cal.K[i * 3 + 2] * invR[2 * 3 + j];
                C.KinvR[i * 3 + j] = s;
            }
        for (int i = 0; i < 4; i++) { // transToWolrd^T * (0,0,1,0), :99-100
            double s = C.T[0 * 4 + i] * 0.0;
            s = s + C.T[1 * 4 + i] * 0.0;
            s = s + C.T[2 * 4 + i] * 1.0;
            s = s + C.T[3 * 4 + i] * 0.0;
            C.gps[i] = s;
        }
        C.roll = D.roll[ri]; C.pitch = D.pitch[pi];
    }
    for (int i = tid; i < F.n_lines; i += 64) { // align_left_right_edges object_3d_util.cpp:147-158
        const double *s = lines_in + (long)(F.line_off + i) * 4;
        double *d = lines_al + (long)(F.line_off + i) * 4;
        if (s[2] < s[0]) { d[0] = s[2]; d[1] = s[3]; d[2] = s[0]; d[3] = s[1]; }
        else { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; d[3] = s[3]; }
    }
}

// ------------------------------------------------------------------------------------------------ lines per unit
// One wave per unit.  merge_break_lines is order dependent (restart after every merge, removed row replaced by the
// last one), so each round finds the FIRST (seg1,seg2) in row-major order that merges: lane = seg1, serial seg2.
// The line lists live in DYNAMIC LDS sized by the batch's longest edge list (`cap` rows of 5 doubles; CS_MAX_ROI_LINES at most).  With the static 40 KB for 1024 rows the
// compiler derived "one wave per SIMD" from the LDS size and then RESERVED the register file of such a wave in the kernel descriptor (next_free_vgpr 257 for a kernel that uses 122):
// a wave that asks for 264 registers only starts on a SIMD that holds at most two region-walk waves, and beside the alternating runner's walks -- four waves of 96 registers on
// every SIMD -- this 47 us kernel took 2.4 ms of every step waiting for such SIMDs.
__global__ void __launch_bounds__(64) cuboid_unit_lines(const Unit *units, UnitDyn *ud, const FrameInfo *fi, const double *lines_al,
                                                        double *mlines, double *mangle, double *mmid, int *status, int cap) {
    extern __shared__ double ul_sh[];
    double (*L)[4] = reinterpret_cast<double (*)[4]>(ul_sh);
    double *ang = ul_sh + 4 * (size_t)cap;
    const int u = blockIdx.x, lane = threadIdx.x;
    const Unit &U = units[u];
    const FrameInfo &F = fi[U.frame];
    const double l = U.roi_x, t = U.roi_y, r = U.roi_r, b = U.roi_b;
    int total = 0;
    for (int base = 0; base < F.n_lines; base += 64) { // ordered compaction of the lines inside the expanded box (:166-174)
        int i = base + lane;
        bool in = false;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (i < F.n_lines) {
            const double *s = lines_al + (long)(F.line_off + i) * 4;
            a0 = s[0]; a1 = s[1]; a2 = s[2]; a3 = s[3];
            in = inside_box(V2{a0, a1}, l, t, r, b) && inside_box(V2{a2, a3}, l, t, r, b);
        }
        unsigned long long m = __ballot(in);
        int pos = total + __popcll(m & ((1ull << lane) - 1));
        if (in && pos < cap) { L[pos][0] = a0; L[pos][1] = a1; L[pos][2] = a2; L[pos][3] = a3; } // (cap = min(the longest list, CS_MAX_ROI_LINES) >= this frame's lines unless the list is longer than CS_MAX_ROI_LINES)
        total += __popcll(m);
    }
    if (total > CS_MAX_ROI_LINES) { if (lane == 0) atomicMin(status, CS_ERR_CAPACITY); total = CS_MAX_ROI_LINES; }
    __syncthreads();
    const double angle_thre = 5.0 / 180.0 * PI, dist_thre = 20.0, len_thre = 30.0; // :177-179
    bool can = true;
    int counter = 0;
    while (can && counter < 500) {
        counter++;
        can = false;
        for (int i = lane; i < total; i += 64) ang[i] = atan2(L[i][3] - L[i][1], L[i][2] - L[i][0]);
        __syncthreads();
        int hit1 = -1, hit2 = -1;
        for (int base = 0; base < total - 1; base += 64) {
            int s1 = base + lane, f2 = -1;
            if (s1 < total - 1) {
                double a1 = ang[s1];
                double x10 = L[s1][0], y10 = L[s1][1], x11 = L[s1][2], y11 = L[s1][3];
                for (int s2 = s1 + 1; s2 < total; s2++) {
                    double diff = fabs(a1 - ang[s2]);
                    double angle_diff = fmin(diff, PI - diff);
                    if (angle_diff < angle_thre) {
                        double x20 = L[s2][0], y20 = L[s2][1], x21 = L[s2][2], y21 = L[s2][3];
                        double d12 = dist2(V2{x11, y11}, V2{x20, y20});
                        double d21 = dist2(V2{x21, y21}, V2{x10, y10});
                        if ((d12 < dist_thre) || (d21 < dist_thre)) {
                            V2 ms = (x10 < x20) ? V2{x10, y10} : V2{x20, y20};
                            V2 me = (x11 > x21) ? V2{x11, y11} : V2{x21, y21};
                            double merged_angle = atan2(me.y - ms.y, me.x - ms.x);
                            double temp = fabs(a1 - merged_angle);
                            double mad = fmin(temp, PI - temp);
                            if (mad < angle_thre) { f2 = s2; break; }
                        }
                    }
                }
            }
            unsigned long long m = __ballot(f2 >= 0);
            if (m) {
                int first = __ffsll((long long)m) - 1;
                hit1 = base + first;
                hit2 = __shfl(f2, first);
                break;
            }
        }
        if (hit1 >= 0) {
            if (lane == 0) {
                int s1 = hit1, s2 = hit2;
                V2 ms = (L[s1][0] < L[s2][0]) ? V2{L[s1][0], L[s1][1]} : V2{L[s2][0], L[s2][1]};
                V2 me = (L[s1][2] > L[s2][2]) ? V2{L[s1][2], L[s1][3]} : V2{L[s2][2], L[s2][3]};
                L[s1][0] = ms.x; L[s1][1] = ms.y; L[s1][2] = me.x; L[s1][3] = me.y;
                for (int k = 0; k < 4; k++) L[s2][k] = L[total - 1][k]; // fast_RemoveRow matrix_utils.cpp:172-176
            }
            total--;
            can = true;
        }
        __syncthreads();
    }
    // drop short lines (:358-373), then angles and midpoints (box_proposal_detail.cpp:185-191)
    int nout = 0;
    for (int base = 0; base < total; base += 64) {
        int i = base + lane;
        bool keep = false;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (i < total) {
            a0 = L[i][0]; a1 = L[i][1]; a2 = L[i][2]; a3 = L[i][3];
            double dx = a2 - a0, dy = a3 - a1;
            keep = sqrt(dx * dx + dy * dy) > len_thre;
        }
        unsigned long long m = __ballot(keep);
        int pos = nout + __popcll(m & ((1ull << lane) - 1));
        if (keep) {
            long o = (long)U.line_off + pos;
            mlines[o * 4 + 0] = a0; mlines[o * 4 + 1] = a1; mlines[o * 4 + 2] = a2; mlines[o * 4 + 3] = a3;
            mangle[o] = atan2(a3 - a1, a2 - a0);
            mmid[o * 2 + 0] = (a0 + a2) / 2;
            mmid[o * 2 + 1] = (a1 + a3) / 2;
        }
        nout += __popcll(m);
    }
    if (lane == 0) ud[u].n_merged = nout;
}

// ------------------------------------------------------------------------------------------------ Canny
// NMS codes in emap: 0 none, 1 weak candidate (mag > low, local max), 2 strong candidate (mag > high).
__global__ void __launch_bounds__(256) cuboid_canny_nms(const Unit *units, const int *wg_unit, const uint8_t *gray, int W, int H, uint8_t *emap, int *lab,
                                                        int low, int high) {
    const Unit &U = units[__builtin_amdgcn_readfirstlane(wg_unit[blockIdx.x])];
    const int tiles_x = (U.roi_w + NMS_TW - 1) / NMS_TW, tile = (int)blockIdx.x - U.nms_off;
    const int tx0 = (tile % tiles_x) * NMS_TW, ty0 = (tile / tiles_x) * NMS_TH;
    __shared__ __attribute__((aligned(4))) uint8_t g[NMS_TH + 4][NMS_TW + 4];
    __shared__ short mg[NMS_TH + 2][NMS_TW + 2];
    const uint8_t *img = gray + (long)U.frame * W * H;
    const int tid = threadIdx.x;
    // four pixels per lane and step (unaligned dwords); a dword that touches the image border is built from clamped bytes (BORDER_REPLICATE at the image
    // border; real pixels outside the ROI view)
    static_assert((NMS_TW + 4) % 4 == 0, "tile rows are whole dwords");
    for (int i = tid; i < (NMS_TH + 4) * ((NMS_TW + 4) / 4); i += 256) {
        const int ly = i / ((NMS_TW + 4) / 4), k4 = i - ly * ((NMS_TW + 4) / 4);
        const int X0 = U.roi_x + tx0 + 4 * k4 - 2;
        int Y = U.roi_y + ty0 + ly - 2;
        Y = Y < 0 ? 0 : (Y >= H ? H - 1 : Y);
        const uint8_t *row = img + (long)Y * W;
        uint32_t v;
        if (X0 >= 0 && X0 + 3 < W) v = load_u32_unaligned(row + X0);
        else {
            v = 0;
            for (int c = 0; c < 4; c++) { int X = X0 + c; X = X < 0 ? 0 : (X >= W ? W - 1 : X); v |= (uint32_t)row[X] << (8 * c); }
        }
        reinterpret_cast<uint32_t *>(&g[ly][0])[k4] = v;
    }
    __syncthreads();
    for (int i = tid; i < (NMS_TH + 2) * (NMS_TW + 2); i += 256) {
        int ly = i / (NMS_TW + 2), lx = i % (NMS_TW + 2);
        int x = tx0 + lx - 1, y = ty0 + ly - 1; // ROI coordinates
        int m = 0;
        if (x >= 0 && x < U.roi_w && y >= 0 && y < U.roi_h) {
            int gy = ly + 1, gx = lx + 1;
            int dx = (g[gy - 1][gx + 1] + 2 * g[gy][gx + 1] + g[gy + 1][gx + 1]) - (g[gy - 1][gx - 1] + 2 * g[gy][gx - 1] + g[gy + 1][gx - 1]);
            int dy = (g[gy + 1][gx - 1] + 2 * g[gy + 1][gx] + g[gy + 1][gx + 1]) - (g[gy - 1][gx - 1] + 2 * g[gy - 1][gx] + g[gy - 1][gx + 1]);
            m = abs(dx) + abs(dy);
        }
        mg[ly][lx] = (short)m;
    }
    __syncthreads();
    const int lx = tid & 63;
    for (int ly = tid >> 6; ly < NMS_TH; ly += 4) {
        int x = tx0 + lx, y = ty0 + ly;
        if (x >= U.roi_w || y >= U.roi_h) continue;
        int gy = ly + 2, gx = lx + 2, my = ly + 1, mx = lx + 1;
        int m = mg[my][mx];
        uint8_t code = 0;
        if (m > low) {
            int xs = (g[gy - 1][gx + 1] + 2 * g[gy][gx + 1] + g[gy + 1][gx + 1]) - (g[gy - 1][gx - 1] + 2 * g[gy][gx - 1] + g[gy + 1][gx - 1]);
            int ys = (g[gy + 1][gx - 1] + 2 * g[gy + 1][gx] + g[gy + 1][gx + 1]) - (g[gy - 1][gx - 1] + 2 * g[gy - 1][gx] + g[gy - 1][gx + 1]);
            const int TG22 = 13573; // (int)(0.41421356237309504 * (1<<15) + 0.5)
            int ax = abs(xs), ay = abs(ys) << 15;
            int tg22x = ax * TG22;
            bool keep;
            if (ay < tg22x) keep = m > mg[my][mx - 1] && m >= mg[my][mx + 1];
            else {
                int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) keep = m > mg[my - 1][mx] && m >= mg[my + 1][mx];
                else {
                    int s = (xs ^ ys) < 0 ? -1 : 1;
                    keep = m > mg[my - 1][mx - s] && m > mg[my + 1][mx + s];
                }
            }
            if (keep) code = m > high ? 2 : 1;
        }
        long p = (long)y * U.roi_w + x;
        // the map is zero-filled before the launch; only edge candidates are written (byte stores of whole rows were the bulk of this kernel's time)
        if (code) { emap[U.pix_off + p] = code; lab[U.pix_off + p] = code == 2 ? (int)p : (int)(p + (long)U.roi_w * U.roi_h); }
    }
}

__device__ __forceinline__ int lab_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// Union-find over candidate pixels.  Node id of pixel p: p if strong (mag > high), p + A if weak; lab[p] holds the parent
// id; the root of a component is its smallest id, so a component contains a strong pixel iff its root id < A.
__device__ __forceinline__ int uf_find(int *lab, int x, int A) { // with path halving (ids only ever decrease along a path)
    while (true) {
        int *px = lab + (x >= A ? x - A : x);
        int v = lab_load(px);
        if (v == x) return x;
        int g = lab_load(lab + (v >= A ? v - A : v));
        if (g == v) return v;
        atomicMin(px, g);
        x = g;
    }
}
__device__ inline void uf_union(int *lab, int a, int b, int A) {
    while (true) {
        a = uf_find(lab, a, A);
        b = uf_find(lab, b, A);
        if (a == b) return;
        if (a < b) { int t = a; a = b; b = t; } // a > b: hook a under b
        int old = atomicMin(lab + (a >= A ? a - A : a), b);
        if (old == a) return;
        a = old; // a was no longer a root: keep uniting (old, b)
    }
}

// Hysteresis = connected components of the candidate pixels (8-neighbourhood); a component is an edge iff it holds a strong pixel.  Three kernels,
// the kernel boundaries are the global syncs:
//   cuboid_canny_cc_local   a workgroup takes 4096 consecutive pixels of the ROI (a band of whole rows): union-find of the band's candidates in LDS
//                           (LDS atomics: W / NW / N / NE neighbours inside the band), then every candidate's global parent = its band-local root
//   cuboid_canny_cc_border  the first w + 1 pixels of every band: unions with the neighbours in the band above (global atomics, few pixels)
//   cuboid_canny_cc         candidates become 255 iff their root is a strong pixel (paths are at most band root -> chain of band roots)
// Node ids, locally and globally: pixel index if strong, + the pixel count if weak; the root of a component is its smallest id, so it is strong iff
// the id is below the count.  A band's row-major order is the ROI's, so a band-local root is the global minimum of its part of the component.
constexpr int CC_BAND = 4096;
__device__ __forceinline__ int ufl_find(int *par, int x) { // LDS, path halving
    while (true) {
        int *px = par + (x & (CC_BAND - 1));
        const int v = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (v == x) return x;
        const int g = __hip_atomic_load(par + (v & (CC_BAND - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (g == v) return v;
        atomicMin(px, g);
        x = g;
    }
}
__device__ inline void ufl_union(int *par, int a, int b) {
    while (true) {
        a = ufl_find(par, a);
        b = ufl_find(par, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; } // a > b: hook a under b
        const int old = atomicMin(par + (a & (CC_BAND - 1)), b);
        if (old == a) return;
        a = old;
    }
}
__global__ void __launch_bounds__(256) cuboid_canny_cc_local(const Unit *units, const int *wg_unit, const uint8_t *emap, int *lab) {
    __shared__ int s_par[CC_BAND];      // parent id: local index (strong) or local index + CC_BAND (weak)
    __shared__ uint8_t s_code[CC_BAND];
    __shared__ unsigned short s_list[CC_BAND];
    __shared__ int s_n;
    const Unit &U = units[__builtin_amdgcn_readfirstlane(wg_unit[blockIdx.x])];
    const int A = U.roi_w * U.roi_h, w = U.roi_w;
    const int p0b = ((int)blockIdx.x - U.cc_off) * CC_BAND;
    const uint8_t *em = emap + U.pix_off;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    {
        const int l0 = threadIdx.x * 16, p0 = p0b + l0;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (p0 < A) v = *reinterpret_cast<const uint4 *>(em + p0); // pix_off and the arena padding are multiples of 64
        *reinterpret_cast<uint4 *>(s_code + l0) = v;
        if ((v.x | v.y | v.z | v.w) != 0) {
            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 16; k++) {
                const unsigned c = (wd[k >> 2] >> ((k & 3) * 8)) & 255u;
                if (c && p0 + k < A) { s_par[l0 + k] = c == 2 ? l0 + k : l0 + k + CC_BAND; s_list[atomicAdd(&s_n, 1)] = (unsigned short)(l0 + k); }
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    auto nid = [&](int l) -> int { const uint8_t c = s_code[l]; return c == 0 ? -1 : (c == 2 ? l : l + CC_BAND); };
    for (int e = threadIdx.x; e < n; e += 256) {
        const int l = s_list[e], p = p0b + l, x = p % w, id = nid(l);
        int nb;
        if (x > 0 && l >= 1 && (nb = nid(l - 1)) >= 0) ufl_union(s_par, id, nb);
        if (l >= w) { // the row above is inside the band (p >= w then as well)
            const int q = l - w;
            if (x > 0 && q >= 1 && (nb = nid(q - 1)) >= 0) ufl_union(s_par, id, nb);
            if ((nb = nid(q)) >= 0) ufl_union(s_par, id, nb);
            if (x + 1 < w && (nb = nid(q + 1)) >= 0) ufl_union(s_par, id, nb);
        } else if (l + 1 >= w && x + 1 < w && l - w + 1 >= 0) { // only the NE neighbour is inside (first row of the band, shifted by one)
            if ((nb = nid(l - w + 1)) >= 0) ufl_union(s_par, id, nb);
        }
    }
    __syncthreads();
    int *lb = lab + U.pix_off;
    for (int e = threadIdx.x; e < n; e += 256) {
        const int l = s_list[e], r = ufl_find(s_par, nid(l));
        lb[p0b + l] = r >= CC_BAND ? p0b + (r - CC_BAND) + A : p0b + r;
    }
}
__global__ void __launch_bounds__(256) cuboid_canny_cc_border(const Unit *units, const uint8_t *emap, int *lab) { // one workgroup per unit: all its band heads
    const Unit &U = units[blockIdx.x];
    const int A = U.roi_w * U.roi_h, w = U.roi_w;
    const uint8_t *em = emap + U.pix_off;
    int *lb = lab + U.pix_off;
    auto nid = [&](int q) { const uint8_t cq = em[q]; return cq == 0 ? -1 : (cq == 2 ? q : q + A); };
    const int n_heads = (A - 1) / CC_BAND; // bands 1 .. n_heads start inside the ROI (band 0 has nothing above it)
    for (int i = blockIdx.y * 256 + threadIdx.x; i < n_heads * (w + 1); i += 256 * gridDim.y) {
        const int band = i / (w + 1) + 1, k = i - (band - 1) * (w + 1); // the band's first w + 1 pixels are the ones with a neighbour in front of the band
        const int p0b = band * CC_BAND, p = p0b + k;
        if (p >= A) continue;
        const uint8_t c = em[p];
        if (!c) continue;
        const int id = c == 2 ? p : p + A, x = p % w;
        int nb;
        if (x > 0 && p - 1 < p0b && (nb = nid(p - 1)) >= 0) uf_union(lb, id, nb, A);
        const int q = p - w; // (>= 0 unless a row is longer than a band)
        if (q < 0) continue;
        if (x > 0 && q - 1 < p0b && (nb = nid(q - 1)) >= 0) uf_union(lb, id, nb, A);
        if (q < p0b && (nb = nid(q)) >= 0) uf_union(lb, id, nb, A);
        if (x + 1 < w && q + 1 < p0b && (nb = nid(q + 1)) >= 0) uf_union(lb, id, nb, A);
    }
}
// candidates become 255 iff their root is a strong pixel.  16 pixels per thread (one 16-byte load; edge maps are sparse).
__global__ void __launch_bounds__(256) cuboid_canny_cc(const Unit *units, const int *wg_unit, uint8_t *emap, int *lab) {
    __shared__ int s_list[4096];
    __shared__ int s_n;
    const Unit &U = units[__builtin_amdgcn_readfirstlane(wg_unit[blockIdx.x])];
    const long A = (long)U.roi_w * U.roi_h;
    const int cb = (int)blockIdx.x - U.cc_off;
    uint8_t *em = emap + U.pix_off;
    int *lb = lab + U.pix_off;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const long p0 = ((long)cb * 256 + threadIdx.x) * 16;
    if (p0 < A) {
        uint4 v = *reinterpret_cast<const uint4 *>(em + p0); // pix_off and the arena padding are multiples of 64
        if ((v.x | v.y | v.z | v.w) != 0) {
            const unsigned wd[4] = {v.x, v.y, v.z, v.w};
            for (int k = 0; k < 16; k++) {
                unsigned c = (wd[k >> 2] >> ((k & 3) * 8)) & 255u;
                if (c && p0 + k < A) s_list[atomicAdd(&s_n, 1)] = (int)(p0 + k) * 2 + (c == 2 ? 1 : 0);
            }
        }
    }
    __syncthreads();
    const int n = s_n, iA = (int)A;
    for (int e = threadIdx.x; e < n; e += 256) { // one candidate per thread: the dependent L2 round trips run in parallel
        const int p = s_list[e] >> 1;
        const int idp = (s_list[e] & 1) ? p : p + iA;
        em[p] = uf_find(lb, idp, iA) < iA ? 255 : 0;
    }
}

// ------------------------------------------------------------------------------------------------ distance transform
// cv::distanceTransform(255 - canny, CV_DIST_L2, 3) == distanceTransform_3x3: forward/backward raster passes of
//   t[j] = min(c[j], t[j-1] + HV)  ==  j*HV + prefix_min(c[k] - k*HV)   (integers, so any evaluation order is exact)
// One wave per ROI; rows sequential, columns in 64-wide segments scanned with DPP.  The int map of the forward pass
// and the final float map share the `dist` arena (in place).
__global__ void __launch_bounds__(256) cuboid_dt(const Unit *units, int n_units, const uint8_t *emap, float *dist, int wbuf) {
    extern __shared__ int s_rows[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int u = blockIdx.x * 4 + wave;
    if (u >= n_units) return;
    const Unit &U = units[u];
    const int w = U.roi_w, h = U.roi_h;
    int *bufA = s_rows + (long)wave * 2 * wbuf, *bufB = bufA + wbuf; // index j+1; [0] and [w+1] are the INIT border columns
    const uint8_t *em = emap + U.pix_off;
    int *tmp = (int *)(dist + U.pix_off);
    float *out = dist + U.pix_off;
    for (int j = lane; j < w + 2; j += 64) { bufA[j] = DT_INIT; bufB[j] = DT_INIT; }
    __builtin_amdgcn_wave_barrier();
    int *up = bufA, *cur = bufB;
    for (int i = 0; i < h; i++) {
        int carry = DT_INIT + DT_HV; // tmp[-1] - (-1)*HV
        for (int j0 = 0; j0 < w; j0 += 64) {
            int j = j0 + lane;
            int uval = INT_MAX;
            if (j < w) {
                int c = 0;
                if (em[(long)i * w + j] == 0) { // not an edge pixel -> src != 0
                    c = up[j] + DT_DIAG;
                    int t = up[j + 1] + DT_HV; if (c > t) c = t;
                    t = up[j + 2] + DT_DIAG; if (c > t) c = t;
                }
                uval = c - j * DT_HV;
            }
            int s = wave_incl_min_scan(uval);
            int v = min(s, carry);
            carry = __builtin_amdgcn_readlane(v, 63);
            if (j < w) { int t = v + j * DT_HV; cur[j + 1] = t; tmp[(long)i * w + j] = t; }
        }
        __builtin_amdgcn_wave_barrier();
        int *sw = up; up = cur; cur = sw;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    for (int j = lane; j < w + 2; j += 64) { bufA[j] = DT_INIT; bufB[j] = DT_INIT; }
    __builtin_amdgcn_wave_barrier();
    int *down = bufA; cur = bufB;
    const float scale = 1.f / 65536.f;
    const int nseg = (w + 63) / 64;
    for (int i = h - 1; i >= 0; i--) {
        int carry = DT_INIT + w * DT_HV; // tmp[w] + w*HV
        for (int sg = nseg - 1; sg >= 0; sg--) {
            int j = sg * 64 + (63 - lane); // descending j along the lanes: suffix scan == prefix scan
            int uval = INT_MAX;
            if (j < w) {
                int c = tmp[(long)i * w + j];
                int t = down[j + 2] + DT_DIAG; if (c > t) c = t;
                t = down[j + 1] + DT_HV; if (c > t) c = t;
                t = down[j] + DT_DIAG; if (c > t) c = t;
                uval = c + j * DT_HV;
            }
            int s = wave_incl_min_scan(uval);
            int v = min(s, carry);
            carry = __builtin_amdgcn_readlane(v, 63);
            if (j < w) { int t = v - j * DT_HV; cur[j + 1] = t; out[(long)i * w + j] = (float)t * scale; }
        }
        __builtin_amdgcn_wave_barrier();
        int *sw = down; down = cur; cur = sw;
    }
}

// Workgroup-per-ROI variant (ROI width <= 1024): thread = column, rows sequential, ONE barrier per row.  Per row every
// wave publishes its segment minimum and its first raw value; after the barrier each wave rebuilds its carry-in and the
// two boundary neighbours of the row it just finished from those, so the up-row lives in registers (DPP wave shifts).
// BWD runs the same recurrence on the mirrored image (scan position jj = tid <-> column W64-1-tid).
template <bool BWD>
__device__ __forceinline__ void dt_block_pass(const Unit &U, const uint8_t *em, int *tmp, float *out, int (*s_tot)[16], int (*s_first)[16]) {
    const int w = U.roi_w, h = U.roi_h;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6, W64 = nw * 64;
    const int lo = BWD ? W64 - w : 0, hi = BWD ? W64 : w;
    const bool active = tid >= lo && tid < hi;
    const int j = BWD ? W64 - 1 - tid : tid;
    const int init_carry = DT_INIT - (lo - 1) * DT_HV;
    const int jl = 64 * wave - 1, jr = 64 * (wave + 1);
    const bool l_act = jl >= lo && jl < hi, r_act = jr >= lo && jr < hi;
    const float scale = 1.f / 65536.f;
    int prev = DT_INIT, left_b = DT_INIT, right_b = DT_INIT;
    auto row_of = [&](int r) { return BWD ? h - 1 - r : r; };
    auto fetch = [&](int r) -> int { // input of scan row r for this thread's column (0 when out of range)
        if (!active || r >= h) return 0;
        long o = (long)row_of(r) * w + j;
        return BWD ? tmp[o] : (int)em[o];
    };
    auto do_row = [&](int r, int curv) {
        const int i = row_of(r);
        int nl = __builtin_amdgcn_update_dpp(left_b, prev, 0x138, 0xf, 0xf, false);  // wave_shr:1 -> value of lane-1
        int nr = __builtin_amdgcn_update_dpp(right_b, prev, 0x130, 0xf, 0xf, false); // wave_shl:1 -> value of lane+1
        int uval = INT_MAX;
        if (active) {
            int c;
            if (!BWD && curv != 0) c = 0; // edge pixel: source
            else {
                c = nl + DT_DIAG;
                int t = prev + DT_HV; if (c > t) c = t;
                t = nr + DT_DIAG; if (c > t) c = t;
                if (BWD && c > curv) c = curv;
            }
            uval = c - tid * DT_HV;
        }
        int sc = wave_incl_min_scan(uval);
        const int b = r & 1;
        if (lane == 63) s_tot[b][wave] = sc;
        if (lane == 0) s_first[b][wave] = uval;
        lds_barrier(); // LDS-only: __syncthreads() would also wait for the previous row's global store and the prefetched loads
        int t = (lane < wave) ? s_tot[b][lane] : INT_MAX;
        t = wave_incl_min_scan(t);
        int carry = min(init_carry, __builtin_amdgcn_readlane(t, 63));
        int v = min(sc, carry);
        int val = DT_INIT;
        if (active) {
            val = v + tid * DT_HV;
            if (BWD) out[(long)i * w + j] = (float)val * scale; else tmp[(long)i * w + j] = val;
        }
        prev = val;
        left_b = l_act ? carry + jl * DT_HV : DT_INIT;
        if (r_act) { int m2 = min(carry, s_tot[b][wave]); m2 = min(m2, s_first[b][wave + 1]); right_b = m2 + jr * DT_HV; }
        else right_b = DT_INIT;
    };
    // rows in groups of 4 with the next group's inputs already in flight (global latency >> one row step)
    int c0 = fetch(0), c1 = fetch(1), c2 = fetch(2), c3 = fetch(3);
    for (int r = 0; r < h; r += 4) {
        int n0 = fetch(r + 4), n1 = fetch(r + 5), n2 = fetch(r + 6), n3 = fetch(r + 7);
        do_row(r, c0);
        if (r + 1 < h) do_row(r + 1, c1);
        if (r + 2 < h) do_row(r + 2, c2);
        if (r + 3 < h) do_row(r + 3, c3);
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    }
}
__global__ void __launch_bounds__(1024) cuboid_dt_block(const Unit *units, const uint8_t *emap, float *dist) {
    __shared__ int s_tot[2][16], s_first[2][16];
    const Unit &U = units[blockIdx.x];
    const uint8_t *em = emap + U.pix_off;
    int *tmp = (int *)(dist + U.pix_off);
    dt_block_pass<false>(U, em, tmp, nullptr, s_tot, s_first);
    __threadfence_block();
    __syncthreads();
    dt_block_pass<true>(U, em, tmp, dist + U.pix_off, s_tot, s_first);
}

// Wave-per-ROI variant, no barrier at all: a lane owns C CONSECUTIVE scan positions (s = lane*C + k), so a row needs one wave scan
// (of the lanes' local minima) instead of one per 64-column segment, the previous row stays in registers, and the only cross-lane
// traffic per row is two DPP wave shifts (neighbours) plus the scan.  Rows are sequential; the inputs of the next G rows are
// already in flight.  The int map between the passes lives in its own arena in a lane-major layout ([row][k][lane]), so both
// passes touch it with fully coalesced accesses: the backward pass runs the same recurrence on the mirrored image and its lane
// l, slot k is the forward pass's lane 63-l, slot C-1-k.  The edge map is read with unaligned dwords.
template <int C, bool BWD>
__device__ __forceinline__ void dt_wave_pass(const Unit &U, const uint8_t *em, int *tmp, float *out, float *lbuf) {
    const int w = U.roi_w, h = U.roi_h, lane = threadIdx.x & 63;
    constexpr int WC = 64 * C, NW = (C + 3) / 4 + 1; // dwords that cover C bytes at any byte offset
    // scan position s <-> column j: forward j = s, backward j = WC-1-s.  Active: j < w.
    int jcol[C]; bool act[C]; int sHV[C];
#pragma unroll
    for (int k = 0; k < C; k++) { const int sp = lane * C + k; jcol[k] = BWD ? WC - 1 - sp : sp; act[k] = jcol[k] < w; sHV[k] = sp * DT_HV; }
    const int lo = BWD ? WC - w : 0;
    const int init_carry = DT_INIT - (lo - 1) * DT_HV;
    const bool any_act = BWD ? act[C - 1] : act[0];
    const float scale = 1.f / 65536.f;
    int P[C];
#pragma unroll
    for (int k = 0; k < C; k++) P[k] = DT_INIT;
    constexpr int G = BWD ? 4 : 8;
    // forward input: bytes em[i*w + lane*C .. +C-1] (unaligned dwords); backward input: tmp[(i*C + C-1-k)*64 + 63-lane]
    uint32_t fin[G][NW] = {}; int bin[G][C] = {};
    auto fetch = [&](int r0) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int r = r0 + g;
            if (r < h) {
                const int i = BWD ? h - 1 - r : r;
                if (!BWD) {
                    if (any_act) {
                        const uint8_t *p = em + (long)i * w + lane * C;
#pragma unroll
                        for (int q = 0; q < NW; q++) fin[g][q] = (q * 4 < C) ? load_u32_unaligned(p + q * 4) : 0u;
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < C; k++) bin[g][k] = tmp[((long)i * C + (C - 1 - k)) * 64 + (63 - lane)];
                }
            }
        }
    };
    fetch(0);
    // backward pass output: a lane's C floats are strided against its neighbours' (C dword stores per row, each a 64-line gather for the
    // texture path); the row goes through a per-wave LDS line instead and leaves as C coalesced stores one row later -- the LDS reads of
    // the previous row are issued before this row's arithmetic and consumed after it
    float ob[C];
    int oi = -1; // image row held in ob / in the LDS line (oi & 1)
    auto flush_store = [&]() {
        if (oi < 0) return;
#pragma unroll
        for (int q = 0; q < C; q++) { const int j = q * 64 + lane; if (j < w) out[(long)oi * w + j] = ob[q]; }
    };
    int rows_done = 0;
    for (int r0 = 0; r0 < h; r0 += G) {
        uint32_t cf[G][NW]; int cb[G][C];
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int q = 0; q < NW; q++) cf[g][q] = fin[g][q];
#pragma unroll
            for (int k = 0; k < C; k++) cb[g][k] = bin[g][k];
        }
        fetch(r0 + G);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const int r = r0 + g;
            if (r >= h) break;
            const int i = BWD ? h - 1 - r : r;
            if (BWD && rows_done > 0) { // previous row: LDS line -> registers (waited for at the end of this row)
                const float *lb = lbuf + ((rows_done - 1) & 1) * WC;
#pragma unroll
                for (int q = 0; q < C; q++) ob[q] = lb[q * 64 + lane];
                oi = h - 1 - (r - 1);
            }
            const int left = __builtin_amdgcn_update_dpp(DT_INIT, P[C - 1], 0x138, 0xf, 0xf, false); // wave_shr:1 -> lane-1's last position
            const int right = __builtin_amdgcn_update_dpp(DT_INIT, P[0], 0x130, 0xf, 0xf, false);    // wave_shl:1 -> lane+1's first position
            int u[C], m = INT_MAX;
#pragma unroll
            for (int k = 0; k < C; k++) {
                const int pl = k == 0 ? left : P[k - 1], pr = k == C - 1 ? right : P[k + 1];
                int c = min(min(pl + DT_DIAG, P[k] + DT_HV), pr + DT_DIAG);
                if (!BWD) { const int byte = (cf[g][k >> 2] >> (8 * (k & 3))) & 255; if (byte != 0) c = 0; } // edge pixel: source
                else c = min(c, cb[g][k]);
                u[k] = act[k] ? c - sHV[k] : INT_MAX;
                m = min(m, u[k]);
            }
            const int incl = wave_incl_min_scan(m);
            const int excl = __builtin_amdgcn_update_dpp(INT_MAX, incl, 0x138, 0xf, 0xf, false);
            int run = min(excl, init_carry);
#pragma unroll
            for (int k = 0; k < C; k++) {
                run = min(run, u[k]);
                const int t = run + sHV[k];
                P[k] = act[k] ? t : DT_INIT;
                if (!BWD) tmp[((long)i * C + k) * 64 + lane] = t;
                else lbuf[(rows_done & 1) * WC + jcol[k]] = (float)t * scale; // inactive positions land beyond column w of the line: never stored
            }
            if (BWD) { flush_store(); rows_done++; }
        }
    }
    if (BWD && rows_done > 0) { // the last row
        const float *lb = lbuf + ((rows_done - 1) & 1) * WC;
#pragma unroll
        for (int q = 0; q < C; q++) ob[q] = lb[q * 64 + lane];
        oi = 0;
        flush_store();
    }
}
template <int C>
__global__ void __launch_bounds__(256) cuboid_dt_wave(const Unit *units, int n_units, const uint8_t *emap, int *tmp_arena, const long *tmp_off, float *dist) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (u >= n_units) return;
    const Unit &U = units[u];
    int *tmp = tmp_arena + tmp_off[u];
    __shared__ float s_line[4][2 * 64 * C];
    dt_wave_pass<C, false>(U, emap + U.pix_off, tmp, nullptr, nullptr);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // the wave's own stores, read back by other lanes of the wave
    dt_wave_pass<C, true>(U, nullptr, tmp, dist + U.pix_off, s_line[threadIdx.x >> 6]);
}

// ------------------------------------------------------------------------------------------------ vanishing points
struct VPEntry { double vp[6]; double ang[6]; }; // vp1.x vp1.y vp2.x ... ; per VP two boundary-edge angles (NaN = none)

__global__ void __launch_bounds__(256) cuboid_vp(const Unit *units, const UnitDyn *ud, const FrameDyn *fd, const CamRP *cam, const double *yaw,
                                                 Opts o, const double *mangle, const double *mmid, VPEntry *vpt) {
    const Unit &U = units[blockIdx.y];
    const FrameDyn &D = fd[U.frame];
    const int n_rp = D.n_roll * D.n_pitch;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_rp * D.n_yaw) return;
    const int rp = e / D.n_yaw, yi = e % D.n_yaw;
    const CamRP &C = cam[(long)U.frame * RP_CAP + rp];
    const double y = yaw[(long)U.frame * o.yaw_cap + yi];
    const double cy = cos(y), sy = sin(y);
    VPEntry E;
    { // getVanishingPoints object_3d_util.cpp:602-607
        const double d[3][3] = {{cy, sy, 0}, {-sy, cy, 0}, {0, 0, 1}};
        for (int k = 0; k < 3; k++) {
            double r[3];
            for (int i = 0; i < 3; i++) {
                double s = C.KinvR[i * 3 + 0] * d[k][0];
                s = s + C.KinvR[i * 3 + 1] * d[k][1];
                s = s + C.KinvR[i * 3 + 2] * d[k][2];
                r[i] = s;
            }
            E.vp[k * 2 + 0] = r[0] / r[2];
            E.vp[k * 2 + 1] = r[1] / r[2];
        }
    }
    const int n = ud[blockIdx.y].n_merged;
    const double *ang = mangle + U.line_off, *mid = mmid + (long)U.line_off * 2;
    for (int vp = 0; vp < 3; vp++) { // VP_support_edge_infos :380-425
        double thre = (vp != 2 ? 15.0 : 10.0) / 180.0 * PI; // box_proposal_detail.cpp:79-80
        double vx = E.vp[vp * 2], vy = E.vp[vp * 2 + 1];
        int cnt = 0, lo = -1, hi = -1;
        double base = 0, vlo = 0, vhi = 0;
        for (int k = 0; k < n; k++) {
            double a_raw = atan2(mid[k * 2 + 1] - vy, mid[k * 2] - vx);
            double a_norm = normalize_to_pi(a_raw);
            double dd = fabs(ang[k] - a_norm);
            dd = fmin(dd, PI - dd);
            if (dd < thre) {
                if (cnt == 0) base = a_raw;
                double sh = a_raw; // smooth_jump_angles :175-189
                if ((a_raw - base) < -PI) sh = a_raw + 2 * PI;
                else if ((a_raw - base) > PI) sh = a_raw - 2 * PI;
                if (cnt == 0) { lo = hi = k; vlo = vhi = sh; }
                else { if (sh > vlo) { vlo = sh; lo = k; } if (sh < vhi) { vhi = sh; hi = k; } }
                cnt++;
            }
        }
        if (cnt > 0) {
            int low_id = lo, top_id = hi;
            if (vp > 0) { int t = low_id; low_id = top_id; top_id = t; }
            E.ang[vp * 2] = ang[low_id];
            E.ang[vp * 2 + 1] = ang[top_id];
        } else {
            E.ang[vp * 2] = __longlong_as_double(0x7ff8000000000000ll);
            E.ang[vp * 2 + 1] = __longlong_as_double(0x7ff8000000000000ll);
        }
    }
    vpt[(long)U.vp_off + e] = E;
}

// ------------------------------------------------------------------------------------------------ sweep: filter + score
// hypothesis index h = ((rp*n_yaw + yaw)*n_tops + top)*2 + (cfg-1)   (the reference's loop nest :229-285); q = rp*n_yaw + yaw indexes the VP table
// per global hypothesis index g = hyp_off + h: flag[g] u8 (0 rejected, 1/2 = vp_1_position), derr[g], aerr[g] (survivors only).
// The corners of a hypothesis are a pure function of (unit, q, top sample, configuration): they are never stored -- the filter, the score
// kernel and the selection each build the ones they need from the same function, so all three see the same doubles.
__device__ __forceinline__ void hyp_decode(const Unit &U, int h, int &q, int &ti) {
    const unsigned p = (unsigned)h >> 1;
    q = U.tops_magic ? (int)__umulhi(p, U.tops_magic) : (int)(p / (unsigned)U.n_tops);
    ti = (int)p - q * U.n_tops;
}

// seg_hit_boundary (object_3d_util.cpp:194-230) against a vertical boundary x = bx, by0 <= y <= by1 (its `bx0 == bx1` branch; the `by0 == by1`
// branch needs a box of height zero, whose top samples can never produce the t.x == bx it asks for) and against a horizontal one y = by.
// A miss is (-1, -1) like the reference's initial value.
__device__ __forceinline__ V2 hit_vertical(V2 ps, V2 pe, double bx, double by0, double by1) {
    const double dx = pe.x - ps.x, dy = pe.y - ps.y;
    const double lambd = (bx - ps.x) / dx;
    const double ty = ps.y + lambd * dy;
    const bool hit = lambd >= 0 && by0 <= ty && ty <= by1;
    return V2{hit ? bx : -1.0, hit ? ty : -1.0};
}
__device__ __forceinline__ V2 hit_horizontal(V2 ps, V2 pe, double by, double bx0, double bx1) {
    const double dx = pe.x - ps.x, dy = pe.y - ps.y;
    const double lambd = (by - ps.y) / dy;
    const double tx = ps.x + lambd * dx;
    const bool hit = lambd >= 0 && bx0 <= tx && tx <= bx1;
    return V2{hit ? tx : -1.0, hit ? by : -1.0};
}

// Corner construction of box_proposal_detail.cpp:254-418 for one hypothesis, branch-free across the two configurations: configuration 1 finds
// corner 4 from corner 1 and intersects for corner 3, configuration 2 finds corner 3 from corner 2 and intersects for corner 4 -- the same
// two steps on (S, other) = (c1, c2) / (c2, c1), the arguments of line_intersect_inf picked per lane.  TESTS: with every reject test of the
// reference (returns vp_1_position, 0 = rejected; a wave leaves as soon as all its lanes are dead); without them for a hypothesis already
// known to survive.  N78 = false skips corners 7 and 8 (no edge of configuration 2 uses them).  Same operations in the same order as the
// reference: the result does not depend on TESTS.
template <bool TESTS, bool N78>
__device__ __forceinline__ int corners_build(const Unit &U, const double *vp, int top_x, int cfg, V2 (&c)[8], bool live = true) {
    // `dist(a, b) < 20` (shorted_edge_thre :81) without the square root: sqrt is monotone and correctly rounded, and the largest double
    // whose root rounds below 20 is pred(pred(400)), so sqrt(d2) < 20 <=> d2 < pred(400) = 0x4078ffffffffffff (tests/test_cabi.py checks it)
    auto short_edge = [](V2 a, V2 b) { const double dx = a.x - b.x, dy = a.y - b.y; return dx * dx + dy * dy < __longlong_as_double(0x4078ffffffffffffll); };
    const V2 vp_1{vp[0], vp[1]}, vp_2{vp[2], vp[3]}, vp_3{vp[4], vp[5]};
    const double left = U.left, right = U.right, top = U.top, down = U.down_y_expan;
    const V2 c1{(double)top_x, top};
    const V2 c2r = hit_vertical(vp_1, c1, right, top, down), c2l = hit_vertical(vp_1, c1, left, top, down); // :257-268: the right edge first
    const int pos = c2r.x != -1 ? 1 : (c2l.x != -1 ? 2 : 0);
    const V2 c2 = pos == 1 ? c2r : c2l;
    bool alive = live && pos > 0;
    if (TESTS) { alive = alive && !short_edge(c1, c2); if (!__any(alive)) return 0; }
    const bool k1 = cfg == 1;
    const V2 S = k1 ? c1 : c2, other = k1 ? c2 : c1;
    const V2 X = hit_vertical(vp_2, S, pos == 1 ? left : right, top, down); // :296-300 / :331-335
    if (TESTS) { alive = alive && X.y != -1 && !short_edge(S, X); if (!__any(alive)) return 0; }
    const V2 Y = k1 ? line_intersect_inf(vp_2, other, vp_1, X) : line_intersect_inf(vp_1, X, vp_2, other); // :306 / :342
    if (TESTS) { // configuration 2 tests corner 4 against the expanded y-range (:347), configuration 1 corner 3 against the box (:311)
        alive = alive && inside_box(Y, left, k1 ? top : (double)U.roi_y, right, k1 ? down : (double)U.roi_b) && !short_edge(Y, X) && !short_edge(Y, other);
        if (!__any(alive)) return 0;
    }
    const V2 c3 = k1 ? Y : X, c4 = k1 ? X : Y;
    const double el = U.roi_x, et = U.roi_y, er = U.roi_r, eb = U.roi_b;
    const V2 c5 = hit_horizontal(vp_3, c3, down, left, right);
    if (TESTS) { alive = alive && c5.y != -1 && !short_edge(c3, c5); if (!__any(alive)) return 0; }
    const V2 c6 = line_intersect_inf(vp_2, c5, vp_3, c2);
    if (TESTS) { alive = alive && inside_box(c6, el, et, er, eb) && !short_edge(c6, c2) && !short_edge(c6, c5); if (!__any(alive)) return 0; }
    c[0] = c1; c[1] = c2; c[2] = c3; c[3] = c4; c[4] = c5; c[5] = c6;
    if (TESTS || N78) {
        const V2 c7 = line_intersect_inf(vp_1, c6, vp_3, c1);
        if (TESTS) { alive = alive && inside_box(c7, el, et, er, eb) && !short_edge(c7, c1) && !short_edge(c7, c6); if (!__any(alive)) return 0; }
        const V2 c8 = line_intersect_inf(vp_1, c5, vp_2, c7);
        if (TESTS) alive = alive && inside_box(c8, el, et, er, eb) && !short_edge(c8, c4) && !short_edge(c8, c5) && !short_edge(c8, c7);
        c[6] = c7; c[7] = c8;
    } else { c[6] = V2{0, 0}; c[7] = V2{0, 0}; }
    return alive ? pos : 0;
}

// cuboid_sweep_filter: the reject tests of every hypothesis (SWEEP_HB per workgroup, one per lane).  The surviving hypotheses are appended to
// the unit's two proposal lists -- configuration 1 grows from vlist[hyp_off] upwards, configuration 2 from vlist[hyp_off + hyp_cap - 1]
// downwards, counts in vcount[2u], vcount[2u+1] -- so that the scoring tasks are configuration-uniform.  Ordered compaction (ballot ranks, no
// atomics inside the workgroup): a list is in hypothesis order inside every workgroup's stretch, neighbouring lanes of a scoring task sample
// neighbouring pixels.  Nothing else leaves the kernel: 1 B of flag per hypothesis, 4 B per survivor.
__global__ void __launch_bounds__(256) cuboid_sweep_filter(const Unit *units, const int *wg_unit, const FrameDyn *fd, Opts o,
                                                           const VPEntry *vpt, uint8_t *flag, int *vcount, int *vlist) {
    __shared__ int s_list[2][SWEEP_HB / 2];
    __shared__ int s_wc[2][SWEEP_HB / 64]; // survivors per (round, wave) and configuration
    __shared__ int s_base[2];
    const int u = __builtin_amdgcn_readfirstlane(wg_unit[blockIdx.x]);
    const Unit &U = units[u];
    const int blk = (int)blockIdx.x - U.hb_off;
    const FrameDyn &D = fd[U.frame];
    const int n_hyp = D.n_roll * D.n_pitch * D.n_yaw * U.n_tops * 2;
    const int h0 = blk * SWEEP_HB;
    if (h0 >= U.hyp_cap) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int rank[SWEEP_HB / 256]; // rank among the survivors of the same configuration in this wave and round, -1 = rejected
    for (int r = 0; r < SWEEP_HB / 256; r++) {
        const int h = h0 + r * 256 + threadIdx.x;
        int pos = 0;
        if (h0 + r * 256 + (wave << 6) < n_hyp) { // (wave-uniform: the early exits of corners_build vote over the wave)
            const int cfg = (h & 1) + 1;
            const bool on = h < n_hyp && ((cfg == 1 && o.cfg1) || (cfg == 2 && o.cfg2));
            int q, ti;
            hyp_decode(U, on ? h : h0 + r * 256 + (wave << 6), q, ti);
            const double *vpq = vpt[(long)U.vp_off + q].vp;
            double vp[6];
#pragma unroll
            for (int k = 0; k < 6; k++) vp[k] = vpq[k];
            V2 c[8];
            pos = corners_build<true, true>(U, vp, U.top_start + ti * U.top_step, cfg, c, on);
        }
        if (h < U.hyp_cap) flag[U.hyp_off + h] = (uint8_t)pos;
        // h0 and r*256 are even, so lane parity = configuration: even lanes are configuration 1
        const unsigned long long m = __ballot(pos != 0);
        const unsigned long long mine = (lane & 1) ? (m & 0xAAAAAAAAAAAAAAAAull) : (m & 0x5555555555555555ull);
        rank[r] = pos ? __popcll(mine & ((1ull << lane) - 1)) : -1;
        if (lane < 2) s_wc[lane][r * 4 + wave] = __popcll(m & (lane ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull));
    }
    __syncthreads();
    // ordered compaction: list position = survivors of the same configuration in earlier (round, wave) pairs + rank
    const int c = lane & 1;
    int before = 0, total[2] = {0, 0};
    for (int i = 0; i < SWEEP_HB / 64; i++) { total[0] += s_wc[0][i]; total[1] += s_wc[1][i]; }
    for (int r = 0; r < SWEEP_HB / 256; r++) {
        for (int w = 0; w < 4; w++) { if (w == wave && rank[r] >= 0) s_list[c][before + rank[r]] = h0 + r * 256 + threadIdx.x; before += s_wc[c][r * 4 + w]; }
    }
    const int c1 = total[0], c2 = total[1];
    if (c1 + c2 == 0) return;
    if (threadIdx.x < 2 && total[threadIdx.x] > 0) s_base[threadIdx.x] = atomicAdd(&vcount[2 * u + threadIdx.x], total[threadIdx.x]);
    __syncthreads();
    int *d1 = vlist + U.hyp_off + s_base[0], *d2 = vlist + U.hyp_off + U.hyp_cap - 1 - s_base[1];
    for (int s = threadIdx.x; s < c1; s += 256) d1[s] = s_list[0][s];
    for (int s = threadIdx.x; s < c2; s += 256) d2[-s] = s_list[1][s];
}

// test / debug access (cs_cuboid_batch_unit): the 16 corner coordinates of every surviving hypothesis of one unit, planes of n_hyp doubles
__global__ void __launch_bounds__(256) cuboid_unit_corners(const Unit *units, int u, int n_hyp, const VPEntry *vpt, const uint8_t *flag, double *out) {
    const Unit &U = units[u];
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= n_hyp || !(flag[U.hyp_off + h] & 3)) return;
    int q, ti;
    hyp_decode(U, h, q, ti);
    V2 c[8];
    corners_build<false, true>(U, vpt[(long)U.vp_off + q].vp, U.top_start + ti * U.top_step, (h & 1) + 1, c);
    for (int k = 0; k < 8; k++) { out[(long)k * n_hyp + h] = c[k].x; out[(long)(8 + k) * n_hyp + h] = c[k].y; }
}

// the padding of every unit's map slice (its A pixels are rounded up to 64) back to zero: cuboid_sweep_score copies and encodes whole groups of eight floats, and after
// cs_cuboid_batch_set_scene the slices lie where other units' pixels were
__global__ void __launch_bounds__(256) cuboid_clear_pad(const Unit *units, int n_units, float *dist) {
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (u >= n_units) return;
    const long A = (long)units[u].roi_w * units[u].roi_h, Ap = (A + 63) / 64 * 64;
    if (A + lane < Ap) dist[units[u].pix_off + A + lane] = 0.0f;
}

// ---- cuboid_sweep_score: corner construction + edge scoring of the surviving proposals, the unit's distance map resident in LDS -------------
// Why LDS: scoring is a gather (99 / 77 samples per proposal at positions that differ from lane to lane); through global memory every
// gathered lane costs the texture-address path about one CU-cycle (round 1: TA_BUSY 80 %).  LDS serves 32 lanes per cycle.
//
// Representation.  The chamfer values are t * 2^-16 with t = i*DT_HV + j*DT_DIAG (i straight and j diagonal steps of the shortest path,
// unique for t < 2^24), so a pixel less than 244 px from the nearest edge (i <= 255, j <= 178) is exactly the 16-bit code i | j << 8, and
// 80 864 codes fit one CU's 160 KB.  The workgroup reads the unit's FLOAT map once (the distance transform's output: 4 B per ROI pixel, the
// bytes SURVEY 8d counts) and encodes while it copies:
//   encode: t = d * 65536 exactly; qn = floor(t / HV) from one float FMA (the fractional part of t / HV is (j * DIAG mod HV) / HV: 0 or in
//     [0.00155, 0.99845], a bias of 0.0005 absorbs the rounding); the residue q = t - qn * HV identifies j (the 256 residues j * DIAG mod
//     HV are >= 97 apart: one per 64-wide bucket, tests/test_cabi.py) and code = qn + lut[q >> 6] with lut = (j << 8) - floor(j * DIAG /
//     HV), because i = qn - floor(j * DIAG / HV).
//   decode: d = fma(float(j), DIAG * 2^-16, float(i) * HV * 2^-16) -- the product is exact (< 2^24) and the fused sum rounds t once, like
//     the distance transform's int -> float conversion: v_cvt_f32_ubyte0, v_cvt_f32_ubyte1, v_mul_f32, v_fma_f32.
// A unit with a pixel farther than 244 px from every edge (an edge-free disc: never on real scenes) samples the float map in global memory
// instead; a unit whose ROI is larger than LDS keeps the head of its map resident and reads the float map for samples past it.
// One lane scores one proposal (the float sum of box_edge_sum_dists is a chain of 99 / 77 ordered additions, so a proposal cannot be
// spread over lanes without paying for a transpose); the lane builds the proposal's corners from (VP entry, top sample) -- no corner ever
// touches HBM -- and keeps them in registers, all samples unrolled with constant corner indices.
// D2 (unchecked dist_map.at on x == w / y == h) without a per-sample clamp: corners are inside the ROI inclusive, so the flat index is at
// most w*h + w, and the w + 2 entries after the map repeat its last pixel -- the value the clamp of round 1 produced.
//
// Work distribution.  Persistent workgroups, one per CU (the map takes the whole LDS), pull (unit, slice) items from a global cursor; the host
// lists the units by falling cost estimate, so the big ones go first and the tail is made of small ones.  With fewer units than CUs a unit
// is cut into n_slices items (task t belongs to slice t % n_slices; every slice copies the map).  Inside a unit the waves pull tasks of 64
// proposals of one configuration from an LDS counter.
__device__ __forceinline__ int sc_tasks(int c) { return (c + 63) >> 6; }

// two pixels at a time (packed f32 multiply / fma); a pixel without a code (d >= SC_ESC_D, or not a number) sets `esc` and its code is garbage:
// the unit is then scored from the float map and its codes are never read.  The table has 1024 entries, so a 10-bit field of the residue
// cannot leave it.
typedef float sc_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned sc_encode2(float d0, float d1, const unsigned short *lut, bool &esc) {
    const sc_f2 d = {d0, d1};
    const sc_f2 tf = d * 65536.0f;
    const sc_f2 qf = __builtin_elementwise_fma(tf, (sc_f2){1.0f / (float)DT_HV, 1.0f / (float)DT_HV}, (sc_f2){0.0005f, 0.0005f});
    const int t0 = (int)tf.x, t1 = (int)tf.y, q0n = (int)qf.x, q1n = (int)qf.y;
    const unsigned r0 = (unsigned)(t0 - __mul24(q0n, DT_HV)), r1 = (unsigned)(t1 - __mul24(q1n, DT_HV));
    const unsigned e0 = lut[(r0 >> 6) & (SC_LUT_N - 1)], e1 = lut[(r1 >> 6) & (SC_LUT_N - 1)];
    esc = esc || !(d0 < SC_ESC_D) || !(d1 < SC_ESC_D);
    return ((unsigned)q0n + e0) | (((unsigned)q1n + e1) << 16);
}

// box_edge_alignment_angle_error (object_3d_util.cpp:455-492), branch-free: a NaN boundary angle never wins `t < best`
__device__ __forceinline__ double angle_best(double ang, double a0, double a1) {
    double bst = 100;
    double t = fabs(ang - a0); t = fmin(t, PI - t); bst = t < bst ? t : bst;
    t = fabs(ang - a1); t = fmin(t, PI - t); bst = t < bst ? t : bst;
    return bst;
}
template <int CFG> __device__ __forceinline__ double edge_angle_error_reg(const double *ang6, const double (&cx)[8], const double (&cy)[8]) {
    double total = 0;
    const double not_found_penalty = 30.0 / 180.0 * PI * 2;
#pragma unroll
    for (int vp = 0; vp < 3; vp++) {
        const double a0 = ang6[vp * 2], a1 = ang6[vp * 2 + 1];
        const bool any = !isnan(a0) || !isnan(a1);
        double best[2];
#pragma unroll
        for (int ee = 0; ee < 2; ee++) {
            const int a = CFG == 1 ? K_VPE1[vp][2 * ee] : K_VPE2[vp][2 * ee], b = CFG == 1 ? K_VPE1[vp][2 * ee + 1] : K_VPE2[vp][2 * ee + 1];
            best[ee] = angle_best(line_angle_fast(cy[b] - cy[a], cx[b] - cx[a]), a0, a1);
        }
        const double x = total + (any ? best[0] : not_found_penalty);
        total = any ? x + best[1] : x;
    }
    return total;
}

// box_edge_sum_dists (object_3d_util.cpp:427-453) with ROI-relative corners in registers and the code map in LDS.  Sample s of an edge is
// s/10 * p1 + (1 - s/10) * p2 in double, in the reference's operation order; s = 0 and s = 10 reproduce the corners exactly (0*p1 + 1*p2),
// and every corner ends two or three edges: the corner pixels are decoded once.  The cfg-2 weights `dist*3.0/2.0` and `dist*2.0` (float ->
// double -> float) equal the float products dist*1.5f and dist*2.0f bit for bit (3*x and x/2 are exact in double: one rounding of 1.5*x).
// HYB: only the first n_res pixels are resident (n_res = 0: none, the unit holds a pixel without a code); a sample past them is read from the
// float map in global memory (index clamped to the slice: D2).
// LEAN (1024-thread workgroups, 128 registers per lane): no cross-edge prefetch and the samples' address computations kept in source order, so
// that the live set stays at the corners + one edge of codes; the other three waves of the SIMD cover the LDS latency instead.
template <int CFG, bool HYB, bool LEAN> __device__ __forceinline__ float edge_sum_dists_code(const double (&rx)[8], const double (&ry)[8], int w, const unsigned short *lmap, const float *gdist,
                                                                               int n_res, int a_last) {
    constexpr int NE = CFG == 1 ? 9 : 7;
    constexpr float HVS = (float)DT_HV / 65536.0f, DGS = (float)DT_DIAG / 65536.0f;
    auto decode = [&](unsigned c) -> float { return __builtin_fmaf((float)((c >> 8) & 0xffu), DGS, (float)(c & 0xffu) * HVS); };
    // the raw code (its consumer comes a whole edge later); HYB: the float's bits
    auto gather = [&](double px, double py) -> unsigned {
        const int idx = __mul24(int(py), w) + int(px);
        if (!HYB) return *reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(lmap) + ((unsigned)idx << 1));
        float v;
        if (idx < n_res) v = decode(*reinterpret_cast<const unsigned short *>(reinterpret_cast<const char *>(lmap) + ((unsigned)idx << 1)));
        else v = gdist[min(idx, a_last)];
        return __float_as_uint(v);
    };
    auto value = [&](unsigned c) -> float { return HYB ? __uint_as_float(c) : decode(c); };
    auto gather_edge = [&](int e, unsigned (&c)[9]) {
        const int ia = CFG == 1 ? K_VIS1[e][0] : K_VIS2[e][0], ib = CFG == 1 ? K_VIS1[e][1] : K_VIS2[e][1];
        double x1 = rx[ia], y1 = ry[ia], x2 = rx[ib], y2 = ry[ib];
        // LEAN: the products s/10 * corner are shared by the edges that meet in the corner, and the compiler would keep all of them alive across
        // the edges (242 registers); an opaque copy per edge keeps the live set at one edge
        if (LEAN) asm volatile("" : "+v"(x1), "+v"(y1), "+v"(x2), "+v"(y2));
#pragma unroll
        for (int si = 1; si < 10; si++) {
            const double s = (double)si;
            c[si - 1] = gather(s / 10.0 * x1 + (1 - s / 10.0) * x2, s / 10.0 * y1 + (1 - s / 10.0) * y2);
            if (LEAN) __builtin_amdgcn_sched_barrier(0);
        }
    };
    // software pipeline, written out because the compiler keeps the source order of this (fully unrolled) block: the 9 LDS gathers of
    // edge e+1 are issued before the codes of edge e are decoded and added
    unsigned cc[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { cc[k] = (CFG == 2 && k >= 6) ? 0u : gather(rx[k], ry[k]); if (LEAN) __builtin_amdgcn_sched_barrier(0); }
    unsigned cur[9], nxt[9];
    if (!LEAN) gather_edge(0, cur);
    __builtin_amdgcn_sched_barrier(0);
    float cp[8];
#pragma unroll
    for (int k = 0; k < 8; k++) cp[k] = value(cc[k]);
    float sum_dist = 0;
#pragma unroll
    for (int e = 0; e < NE; e++) {
        if (LEAN) gather_edge(e, cur);
        else if (e + 1 < NE) gather_edge(e + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        const int ia = CFG == 1 ? K_VIS1[e][0] : K_VIS2[e][0], ib = CFG == 1 ? K_VIS1[e][1] : K_VIS2[e][1];
        const float wgt = (CFG == 2 && (e == 4 || e == 5)) ? 1.5f : ((CFG == 2 && e == 6) ? 2.0f : 1.0f);
#pragma unroll
        for (int si = 0; si < 11; si++) {
            float d = si == 0 ? cp[ib] : (si == 10 ? cp[ia] : value(cur[si - 1]));
            if (CFG == 2) d = d * wgt;
            sum_dist = sum_dist + d;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!LEAN) {
#pragma unroll
            for (int k = 0; k < 9; k++) cur[k] = nxt[k];
        }
    }
    return sum_dist;
}

template <int CFG, bool HYB, bool LEAN> __device__ __forceinline__ void sc_score_task(const Unit &U, int first, int count, int lane, const VPEntry *vpt, const int *vlist,
                                                                           const unsigned short *lmap, const float *gdist, int n_res, double *derr, double *aerr) {
    const int s = first + lane;
    const bool live = s < count;
    const int sc = live ? s : count - 1;
    const int h = CFG == 1 ? vlist[U.hyp_off + sc] : vlist[U.hyp_off + U.hyp_cap - 1 - sc];
    const long g = U.hyp_off + h;
    int q, ti;
    hyp_decode(U, h, q, ti);
    const VPEntry &E = vpt[(long)U.vp_off + q];
    double cx[8], cy[8];
    {
        double vp[6];
#pragma unroll
        for (int k = 0; k < 6; k++) vp[k] = E.vp[k];
        V2 c[8];
        corners_build<false, CFG == 1>(U, vp, U.top_start + ti * U.top_step, CFG, c); // :254-418, the corners of a survivor
#pragma unroll
        for (int k = 0; k < 8; k++) { cx[k] = c[k].x; cy[k] = c[k].y; }
    }
    double ang[6];
#pragma unroll
    for (int k = 0; k < 6; k++) ang[k] = E.ang[k];
#ifdef NO_ANGLE
    const double ae = ang[0];
#else
    const double ae = edge_angle_error_reg<CFG>(ang, cx, cy);
#endif
    const double rx = (double)U.roi_x, ry = (double)U.roi_y;
#pragma unroll
    for (int k = 0; k < 8; k++) { cx[k] = cx[k] - rx; cy[k] = cy[k] - ry; } // :423-425
#ifdef NO_SUM
    const float sum_dist = (float)(cx[0] + cy[1] + cx[2] + cy[3] + cx[4] + cy[5] + cx[6] + cy[7]);
#else
    const float sum_dist = edge_sum_dists_code<CFG, HYB, LEAN>(cx, cy, U.roi_w, lmap, gdist, n_res, U.roi_w * U.roi_h - 1);
#endif
    if (live) {
        derr[g] = double(sum_dist) / U.diag; // :451
        aerr[g] = ae;
    }
}

// NT = 256 is the variant that fits BESIDE a CU's sixteen lsd_rg_seq waves (4 x 96 registers per SIMD leave 128, no LDS in use there): four waves of at most 128
// registers and the whole LDS -- the score kernel of the alternating runner, where region walks hold most CUs all the time.
template <int NT>
__global__ void __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 4 : 1, NT == 256 ? 4 : 8))) cuboid_sweep_score(const Unit *units, const int *order, int n_items, int n_slices, int *cursor, const VPEntry *vpt, const float *dist,
                                                         const int *vcount, const int *vlist, int *uflag, double *derr, double *aerr, unsigned long long *prof) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sc_mem[];
    // (development, CUBESLAM_SCORE_PROF) wall-clock ticks of this wave in the three parts of a unit: copy + encode, scoring tasks, waiting at the barriers
    unsigned long long t_copy = 0, t_task = 0, t_wait = 0, t_mark = prof ? wall_clock64() : 0, n_task = 0;
    auto lap = [&](unsigned long long &acc) { if (prof) { const unsigned long long t = wall_clock64(); acc += t - t_mark; t_mark = t; } };
    int *ctrl = reinterpret_cast<int *>(sc_mem); // control words, see the loop
    unsigned short *lut = reinterpret_cast<unsigned short *>(sc_mem + SC_CTRL_BYTES);
    unsigned short *lmap = reinterpret_cast<unsigned short *>(sc_mem + SC_MAP_OFF);
    const int tid = threadIdx.x, lane = tid & 63;
    // beside a CU's region walks (the 256-thread shape) this wave is one of five on its SIMD and the only one somebody waits for: it goes first
    if (NT == 256) __builtin_amdgcn_s_setprio(3);
    for (int i = tid; i < SC_LUT_N; i += NT) lut[i] = 0;
    if (tid == 0) { ctrl[2] = 0; ctrl[3] = 0; ctrl[5] = atomicAdd(cursor, 1); }
    __syncthreads();
    if (tid < 256) { const int j = tid, r = (j * DT_DIAG) % DT_HV; lut[r >> 6] = (unsigned short)((j << 8) - (j * DT_DIAG) / DT_HV); }
    __syncthreads();
    int item = __builtin_amdgcn_readfirstlane(ctrl[5]); // (uniform by construction; tell the compiler: the unit's fields then live in scalar registers)
    constexpr bool LEAN = NT != 512; // (512: 2 waves per SIMD with 256 registers each; every other shape lives on 128)
    constexpr int NPF = NT >= 768 ? 4 : 8; // groups of 8 pixels (two 16-byte loads) per thread in flight while a map is copied
    auto unit_of = [&](int itm, int &uu, int &sl) { uu = __builtin_amdgcn_readfirstlane(order[n_slices == 1 ? itm : itm / n_slices]); sl = n_slices == 1 ? 0 : itm % n_slices; };
    for (int it = 0; item < n_items; it++) {
        int u, slice;
        unit_of(item, u, slice);
        const Unit &U = units[u];
        const int c1 = vcount[2 * u], c2 = vcount[2 * u + 1];
        const int n1 = sc_tasks(c1), nt = n1 + sc_tasks(c2);
        const bool work = slice < nt;
        // ctrl: [0] task counter; [2], [3] "a pixel without a code" of the unit being copied, by iteration parity; [4], [5] the next item, likewise
        if (tid == 0) { ctrl[4 + (it & 1)] = atomicAdd(cursor, 1); ctrl[0] = slice; ctrl[2 + ((it + 1) & 1)] = 0; }
        const int A = U.roi_w * U.roi_h;
        const bool fits = score_unit_fits(U.roi_w, U.roi_h); // else: the first n_res pixels are resident, the tail is read from the float map
        int n_res = fits ? A : (SC_MAP_ENTRIES & ~7);
        const float *gdist = dist + U.pix_off;
        if (work) { // copy + encode the float map, NPF groups per thread in flight
            const int A8 = (n_res + 7) >> 3; // slices are padded to 64 pixels
            const float4 *dm4 = reinterpret_cast<const float4 *>(gdist);
            uint4 *lm4 = reinterpret_cast<uint4 *>(lmap);
            bool esc = false;
            // D2 (unchecked dist_map.at on x == w / y == h): the w + 2 entries after the map repeat its last pixel, written in this same phase
            const float dlast = gdist[A - 1];
            const unsigned lastc = sc_encode2(dlast, dlast, lut, esc);
            auto consume = [&](const float4 (&pa)[NPF], const float4 (&pb)[NPF], int k0) {
#pragma unroll
                for (int r = 0; r < NPF; r++) {
                    const int k = min(k0 + r * NT, A8 - 1);
                    // (the padding of a slice past its A pixels is zero: cs_cuboid_batch_create clears the arena once and nothing writes there)
                    uint4 c = make_uint4(sc_encode2(pa[r].x, pa[r].y, lut, esc), sc_encode2(pa[r].z, pa[r].w, lut, esc), sc_encode2(pb[r].x, pb[r].y, lut, esc), sc_encode2(pb[r].z, pb[r].w, lut, esc));
                    if (fits && 8 * k + 8 > A) { // the group that holds the end of the map: the entries past it are the last pixel's
                        unsigned w4[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                        for (int i = 0; i < 8; i++) if (8 * k + i >= A) w4[i >> 1] = (i & 1) ? (w4[i >> 1] & 0xffffu) | (lastc & 0xffff0000u) : (w4[i >> 1] & 0xffff0000u) | (lastc & 0xffffu);
                        c = make_uint4(w4[0], w4[1], w4[2], w4[3]);
                    }
                    lm4[k] = c;
                }
            };
            // (requesting the next unit's first groups before the end-of-unit barrier was measured: the copy phase fell from 135 to 118 us per wave and the
            // tasks grew by as much -- the phase is bound by the encoder's VALU + table work and the CU's share of the memory system, not by one round trip)
#pragma unroll 1
            for (int k0 = tid; k0 < A8; k0 += NPF * NT) {
                float4 pa[NPF], pb[NPF];
#pragma unroll
                for (int r = 0; r < NPF; r++) { const int k = min(k0 + r * NT, A8 - 1); pa[r] = dm4[2 * k]; pb[r] = dm4[2 * k + 1]; } // branch-free: a tail thread re-copies the last group
                consume(pa, pb, k0);
            }
            if (fits) for (int k = 8 * A8 + tid; k < A + max(U.roi_w + 2, 8); k += NT) lmap[k] = (unsigned short)lastc;
            if (esc) atomicOr(&ctrl[2 + (it & 1)], 1);
        }
        lap(t_copy);
        __syncthreads();
        lap(t_wait);
        const int next = __builtin_amdgcn_readfirstlane(ctrl[4 + (it & 1)]);
        const bool escape = __builtin_amdgcn_readfirstlane(ctrl[2 + (it & 1)]) != 0; // a pixel without a code: the whole unit samples the float map
        if (work) {
            if (tid == 0 && slice == 0) uflag[u] = escape ? 1 : 0;
            if (escape) n_res = 0;
#ifdef NO_HYB
            const bool hyb = false;
#else
            const bool hyb = escape || !fits;
#endif
            for (;;) { // waves pull tasks of 64 proposals of one configuration
                int t = 0;
                if (lane == 0) t = atomicAdd(&ctrl[0], n_slices);
                t = __builtin_amdgcn_readfirstlane(t);
                if (t >= nt) break;
                n_task++;
                if (!hyb) {
                    if (t < n1) sc_score_task<1, false, LEAN>(U, t << 6, c1, lane, vpt, vlist, lmap, gdist, n_res, derr, aerr);
                    else sc_score_task<2, false, LEAN>(U, (t - n1) << 6, c2, lane, vpt, vlist, lmap, gdist, n_res, derr, aerr);
                } else {
                    if (t < n1) sc_score_task<1, true, LEAN>(U, t << 6, c1, lane, vpt, vlist, lmap, gdist, n_res, derr, aerr);
                    else sc_score_task<2, true, LEAN>(U, (t - n1) << 6, c2, lane, vpt, vlist, lmap, gdist, n_res, derr, aerr);
                }
            }
            lap(t_task);
            __syncthreads(); // every wave is done with this unit's map
            lap(t_wait);
        }
        item = next;
    }
    if (prof && lane == 0) { unsigned long long *o = prof + ((size_t)blockIdx.x * (NT / 64) + (tid >> 6)) * 4; o[0] = t_copy; o[1] = t_task; o[2] = t_wait; o[3] = n_task; }
}

// ------------------------------------------------------------------------------------------------ selection
__device__ __forceinline__ unsigned long long dkey(double x) { // order-preserving map double -> u64
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// The valid hypotheses of a unit are compacted (in order) into key arrays; every thread caches its first SEL_R entries
// (entry e = tid + 256*k) in registers so the 8-bit radix-select passes run out of registers + LDS histograms.
constexpr int SEL_R = 8;
struct KeyCache { unsigned long long v[SEL_R]; };
template <class F> __device__ __forceinline__ void for_keys(const KeyCache &c, const unsigned long long *gl, int n, F f) {
#pragma unroll
    for (int k = 0; k < SEL_R; k++) { int e = threadIdx.x + 256 * k; if (e < n) f(c.v[k], e); }
    for (int e = threadIdx.x + 256 * SEL_R; e < n; e += 256) f(gl[e], e);
}
// k-th smallest (0-based) of n keys; MSB-first 8-bit radix select.  All threads return it.  blockDim.x == 256.
__device__ unsigned long long block_select_kth(const KeyCache &c, const unsigned long long *gl, int n, int k, int *hist /*256*/, int *s_misc) {
    unsigned long long prefix = 0, mask = 0;
    const int lane = threadIdx.x & 63;
    for (int pass = 0; pass < 8; pass++) {
        int shift = 56 - 8 * pass;
        hist[threadIdx.x] = 0;
        __syncthreads();
        for_keys(c, gl, n, [&](unsigned long long key, int) { if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1); });
        __syncthreads();
        if (threadIdx.x < 64) { // wave 0: 4 bins per lane, inclusive scan over lanes, locate the bin holding rank k
            int h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
            int sum = h0 + h1 + h2 + h3, inc = sum;
            for (int off = 1; off < 64; off <<= 1) { int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
            unsigned long long m = __ballot(inc > k);
            int first = __ffsll((long long)m) - 1;
            if (lane == first) {
                int acc = inc - sum, bin = 4 * lane;
                if (acc + h0 > k) {} else { acc += h0; bin++; if (acc + h1 > k) {} else { acc += h1; bin++; if (acc + h2 > k) {} else { acc += h2; bin++; } } }
                s_misc[0] = bin; s_misc[1] = k - acc;
            }
        }
        __syncthreads();
        prefix |= (unsigned long long)s_misc[0] << shift;
        mask |= 0xffull << shift;
        k = s_misc[1];
    }
    return prefix;
}
// ordered exclusive count of predicate over [0,n) (in index order); returns total; cb(i, rank) for elements with pred
template <class P, class F> __device__ int block_ordered_visit(int n, P pred, F cb, int *s_wave /*>= nwaves+1*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    int running = 0;
    for (int base = 0; base < n; base += blockDim.x) {
        int i = base + threadIdx.x;
        bool p = i < n && pred(i);
        unsigned long long m = __ballot(p);
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = running, tot = 0;
        for (int w2 = 0; w2 < nw; w2++) { int c = s_wave[w2]; if (w2 < wave) off += c; tot += c; }
        if (p) cb(i, off + __popcll(m & ((1ull << lane) - 1)));
        running += tot;
        __syncthreads();
    }
    return running;
}

struct Conv3D { double pos[3], scale[3]; };
__device__ inline void plane_hit(const double *T, const double *invK, const double *pl, double px, double py, double *o3) { // object_3d_util.cpp:574-585
    double ray[3];
    for (int i = 0; i < 3; i++) {
        double s = invK[i * 3] * px;
        s = s + invK[i * 3 + 1] * py;
        s = s + invK[i * 3 + 2] * 1.0;
        ray[i] = s;
    }
    double den = pl[0] * ray[0];
    den = den + pl[1] * ray[1];
    den = den + pl[2] * ray[2];
    double frac = -pl[3] / den;
    double s4[4] = {frac * ray[0], frac * ray[1], frac * ray[2], 1.0};
    double w[4];
    for (int i = 0; i < 4; i++) {
        double a = T[i * 4] * s4[0];
        a = a + T[i * 4 + 1] * s4[1];
        a = a + T[i * 4 + 2] * s4[2];
        a = a + T[i * 4 + 3] * s4[3];
        w[i] = a;
    }
    o3[0] = w[0] / w[3]; o3[1] = w[1] / w[3]; o3[2] = w[2] / w[3];
}
// change_2d_corner_to_3d_object :610-648 (position and scale part)
__device__ inline void corners_to_3d(const double *cx, const double *cy, const CamRP &C, const double *invK, Conv3D &o) {
    double g[4][3];
    for (int i = 0; i < 4; i++) plane_hit(C.T, invK, C.gps, cx[4 + i], cy[4 + i], g[i]);
    auto nrm = [](const double *a, const double *b) {
        double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        return sqrt(dx * dx + dy * dy + dz * dz);
    };
    double length_half = nrm(g[0], g[3]) / 2;
    double width_half = nrm(g[0], g[1]) / 2;
    double d[3] = {g[0][0] - g[1][0], g[0][1] - g[1][1], g[0][2] - g[1][2]};
    double nw[3] = {d[1] * 1.0 - d[2] * 0.0, d[2] * 0.0 - d[0] * 1.0, d[0] * 0.0 - d[1] * 0.0};
    double nn = sqrt(nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2]);
    nw[0] /= nn; nw[1] /= nn; nw[2] /= nn;
    double dist = -((nw[0] * g[0][0] + nw[1] * g[0][1]) + nw[2] * g[0][2]);
    double pw[4] = {nw[0], nw[1], nw[2], dist};
    if (dist < 0) for (int i = 0; i < 4; i++) pw[i] = -pw[i];
    double ps[4];
    for (int i = 0; i < 4; i++) {
        double s = C.T[0 * 4 + i] * pw[0];
        s = s + C.T[1 * 4 + i] * pw[1];
        s = s + C.T[2 * 4 + i] * pw[2];
        s = s + C.T[3 * 4 + i] * pw[3];
        ps[i] = s;
    }
    double top[3];
    plane_hit(C.T, invK, ps, cx[1], cy[1], top);
    double height_half = top[2] / 2;
    o.pos[0] = (((g[0][0] + g[1][0]) + g[2][0]) + g[3][0]) / 4.0;
    o.pos[1] = (((g[0][1] + g[1][1]) + g[2][1]) + g[3][1]) / 4.0;
    o.pos[2] = height_half;
    o.scale[0] = length_half; o.scale[1] = width_half; o.scale[2] = height_half;
}

// flag bits after selection: bits 0-1 vp_1_position, bit 2 kept by fuse_normalize, bit 3 candidate for final ranking
// Register budget: two waves per SIMD (225 registers) is the kernel's best alone (0.66 ms per 3 072 boxes) -- and a workgroup that needs 4 x 225 registers free on one CU
// waits for them when region walks hold 2-3 x 96 on every SIMD: 9.4 ms per launch in the batch runner.  Three waves (168 registers, 216 B of scratch): 2.9 ms there.
#ifndef SELECT_WAVES
#define SELECT_WAVES 3
#endif
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(SELECT_WAVES, SELECT_WAVES))) cuboid_select(const Unit *units, UnitDyn *ud, const int *box_first_unit, const FrameDyn *fd,
                                                     const FrameInfo *fi, const CamRP *cam, const double *yaw, Calib cal, Opts o,
                                                     uint8_t *flag, const double *derr, const double *aerr, const VPEntry *vpt,
                                                     double *score, double *nscore, unsigned long long *ckey_d, unsigned long long *ckey_a,
                                                     int *cidx, cs_cuboid *out, int *counts, int *carry_rp) {
    __shared__ int hist[256];
    __shared__ int s_last; // entry (rank among the valid proposals) of the proposal the reference's loop :479-546 processes last
    __shared__ int s_misc[8];
    __shared__ int s_wave[8];
    __shared__ double s_red[4][4];
    __shared__ double s_best[4];
    __shared__ int s_bi[4][2];
    __shared__ int s_branch[4];
    const int box = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int u0 = box_first_unit[box];
    const int n_hs = units[u0].n_hs;
    const FrameDyn &D = fd[units[u0].frame];
    const FrameInfo &FI = fi[units[u0].frame];
    const int n_rp = D.n_roll * D.n_pitch;
    const double weight_vp_angle = 0.8, weight_skew_error = 1.5; // box_proposal_detail.cpp:86-87
    int n_cand_total = 0;

    for (int hs = 0; hs < n_hs; hs++) {
        const int u = u0 + hs;
        const Unit &U = units[u];
        const int n_hyp = n_rp * D.n_yaw * U.n_tops * 2;
        uint8_t *fl = flag + U.hyp_off;
        const double *de = derr + U.hyp_off, *ae = aerr + U.hyp_off;
        // ---- fuse_normalize_scores_v2 (object_3d_util.cpp:495-565)
        unsigned long long *ckd = ckey_d + U.hyp_off, *cka = ckey_a + U.hyp_off;
        int *ch = cidx + U.hyp_off;
        const int n = block_ordered_visit(n_hyp, [&](int i) { return fl[i] != 0; },
                                          [&](int i, int r) { ckd[r] = dkey(de[i]); cka[r] = dkey(ae[i]); ch[r] = i; }, s_wave);
        __syncthreads();
        KeyCache kd, ka;
#pragma unroll
        for (int k = 0; k < SEL_R; k++) { int e = tid + 256 * k; kd.v[k] = e < n ? ckd[e] : 0; ka.v[k] = e < n ? cka[e] : 0; }
        int branch_b = 0, n_kept = 0;
        if (tid == 0) s_last = n - 1; // n <= 4: all of them, in index order
        if (n > 4) {
            int breaking_num = (int)round(float(n) / 3.0 * 2.0);
            int k = breaking_num - 1; // elements kept per criterion
            unsigned long long vd = block_select_kth(kd, ckd, n, k - 1, hist, s_misc);
            unsigned long long va = block_select_kth(ka, cka, n, k - 1, hist, s_misc);
            unsigned long long va1 = block_select_kth(ka, cka, n, k, hist, s_misc);
            __syncthreads();
            if (tid == 0) { s_misc[2] = 0; s_misc[3] = 0; s_misc[4] = INT_MAX; }
            __syncthreads();
            {
                int nl = 0, ne = 0;
                for_keys(kd, ckd, n, [&](unsigned long long key, int) { nl += key < vd; ne += key == vd; });
                for (int off = 32; off > 0; off >>= 1) { nl += __shfl_xor(nl, off); ne += __shfl_xor(ne, off); }
                if (lane == 0) { atomicAdd(&s_misc[2], nl); atomicAdd(&s_misc[3], ne); }
            }
            __syncthreads();
            const int need_eq = k - s_misc[2]; // ties at the threshold are kept in index order (std::partial_sort tie order pinned, D3)
            if (need_eq < s_misc[3] && tid == 0) { // rare: more ties than room -> keep the first need_eq of them
                int seen = 0;
                for (int e = 0; e < n; e++) if (ckd[e] == vd && ++seen == need_eq) { s_misc[4] = e; break; }
            }
            __syncthreads();
            const int e_star = s_misc[4];
            const bool use_angle = va1 > va; // angle criterion active: keep set = {angle <= a_k}, intersect
            if (!use_angle) branch_b = 1;    // final_keep_inds = dist_keep_inds, in (dist, index) order
#pragma unroll
            for (int kk = 0; kk < SEL_R; kk++) {
                int e = tid + 256 * kk;
                if (e < n) {
                    bool ok = kd.v[kk] < vd || (kd.v[kk] == vd && e <= e_star);
                    if (use_angle) ok = ok && ka.v[kk] <= va;
                    if (ok) fl[ch[e]] |= 4;
                }
            }
            for (int e = tid + 256 * SEL_R; e < n; e += 256) {
                bool ok = ckd[e] < vd || (ckd[e] == vd && e <= e_star);
                if (use_angle) ok = ok && cka[e] <= va;
                if (ok) fl[ch[e]] |= 4;
            }
            if (hs == n_hs - 1 && carry_rp) { // (only for cs_cuboid_detect's chain) the last entry of final_keep_inds: the largest index of the intersection, or the last of the (distance, index) order
                __syncthreads();
                if (tid == 0) s_last = -1;
                __syncthreads();
                int mine = -1;
                for (int e = tid; e < n; e += 256) {
                    const bool kept = (ckd[e] < vd || (ckd[e] == vd && e <= e_star)) && (!use_angle || cka[e] <= va);
                    if (kept && (use_angle || ckd[e] == vd)) mine = e; // (ascending e per thread: the last one stays)
                }
                if (mine >= 0) atomicMax(&s_last, mine);
                __syncthreads();
            }
        } else {
            for (int e = tid; e < n; e += 256) fl[ch[e]] |= 4;
        }
        __syncthreads();
        if (hs == n_hs - 1 && carry_rp) { // what cam_pose holds after this box: the roll / pitch sample of the last good proposal (:481-487), or of the last loop pass (:233-239)
            __syncthreads();
            if (tid == 0) {
                int rp = n_rp - 1;
                if (n > 0 && s_last >= 0) { int q, ti; hyp_decode(U, ch[s_last], q, ti); rp = q / D.n_yaw; }
                carry_rp[box] = rp;
            }
        }
        // min / max over the kept set
        double mn_d = 1e6, mx_d = -1, mn_a = 1e6, mx_a = -1;
        int cntk = 0;
        for (int e = tid; e < n; e += 256) {
            const int i = ch[e];
            if (fl[i] & 4) {
                double td = de[i], ta = ae[i];
                mn_d = fmin(mn_d, td); mx_d = fmax(mx_d, td); mn_a = fmin(mn_a, ta); mx_a = fmax(mx_a, ta);
                cntk++;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            mn_d = fmin(mn_d, __shfl_xor(mn_d, off)); mx_d = fmax(mx_d, __shfl_xor(mx_d, off));
            mn_a = fmin(mn_a, __shfl_xor(mn_a, off)); mx_a = fmax(mx_a, __shfl_xor(mx_a, off));
            cntk += __shfl_xor(cntk, off);
        }
        if (lane == 0) { s_red[wave][0] = mn_d; s_red[wave][1] = mx_d; s_red[wave][2] = mn_a; s_red[wave][3] = mx_a; s_wave[wave] = cntk; }
        __syncthreads();
        mn_d = fmin(fmin(s_red[0][0], s_red[1][0]), fmin(s_red[2][0], s_red[3][0]));
        mx_d = fmax(fmax(s_red[0][1], s_red[1][1]), fmax(s_red[2][1], s_red[3][1]));
        mn_a = fmin(fmin(s_red[0][2], s_red[1][2]), fmin(s_red[2][2], s_red[3][2]));
        mx_a = fmax(fmax(s_red[0][3], s_red[1][3]), fmax(s_red[2][3], s_red[3][3]));
        n_kept = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
        // normalised score, 2D->3D, skew penalty, final-ranking candidate score
        int ncand = 0;
        for (int e = tid; e < n; e += 256) {
            const int i = ch[e];
            if (!(fl[i] & 4)) continue;
            double dk = de[i], ak = ae[i], comb;
            if (n_kept > 1) { // whether_normalize_two_errors = true (:85)
                comb = (dk - mn_d) / (mx_d - mn_d);
                if ((mx_a - mn_a) > 0) ak = (ak - mn_a) / (mx_a - mn_a);
                comb = (comb + weight_vp_angle * ak) / (1 + weight_vp_angle);
            } else
                comb = (dk + weight_vp_angle * ak) / (1 + weight_vp_angle);
            long g = U.hyp_off + i;
            int q, ti;
            hyp_decode(U, i, q, ti);
            int rp = q / D.n_yaw;
            double cx[8], cy[8];
            { // the proposal's corners, rebuilt (box_proposal_detail.cpp:254-418): nothing stores them
                V2 c[8];
                corners_build<false, true>(U, vpt[(long)U.vp_off + q].vp, U.top_start + ti * U.top_step, (i & 1) + 1, c);
                for (int k = 0; k < 8; k++) { cx[k] = c[k].x; cy[k] = c[k].y; }
            }
            Conv3D c3;
            corners_to_3d(cx, cy, cam[(long)U.frame * RP_CAP + rp], cal.invK, c3);
            if (c3.scale[0] < 0 || c3.scale[1] < 0 || c3.scale[2] < 0) continue; // :493
            double mxs = c3.scale[0] < c3.scale[1] ? c3.scale[1] : c3.scale[0];   // Eigen maxCoeff / minCoeff
            double mns = c3.scale[1] < c3.scale[0] ? c3.scale[1] : c3.scale[0];
            double skew_ratio = mxs / mns;
            double dsk = skew_ratio - o.nominal_skew_ratio;
            double skew_error = weight_skew_error * ((dsk < 0.0) ? 0.0 : dsk); // std::max(a,0.0): (a<b)?b:a
            if (skew_ratio > o.max_cut_skew) skew_error = 100;
            nscore[g] = comb;
            score[g] = comb + weight_skew_error * skew_error; // :526
            fl[i] |= 8;
            ncand++;
        }
        for (int off = 32; off > 0; off >>= 1) ncand += __shfl_xor(ncand, off);
        if (lane == 0) s_wave[wave] = ncand;
        __syncthreads();
        n_cand_total += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        if (tid == 0) { ud[u].n_valid = n; ud[u].n_kept = n_kept; ud[u].branch_b = branch_b; s_branch[hs] = branch_b; }
        __syncthreads();
    }

    // ---- final ranking :517-536: best max_cuboid_num by (score, position in raw_obj_proposals)
    const int kout = n_cand_total < o.max_cuboid_num ? n_cand_total : o.max_cuboid_num;
    for (int pick = 0; pick < kout; pick++) {
        double bs = 0, bd = 0; int bhs = -1, bi = -1;
        for (int hs = 0; hs < n_hs; hs++) {
            const Unit &U = units[u0 + hs];
            const int n_hyp = n_rp * D.n_yaw * U.n_tops * 2;
            const uint8_t *fl = flag + U.hyp_off;
            const int bb = s_branch[hs];
            for (int i = tid; i < n_hyp; i += 256) {
                if (!(fl[i] & 8)) continue;
                double s = score[U.hyp_off + i], dd = bb ? derr[U.hyp_off + i] : 0.0;
                bool better = bi < 0 || s < bs || (s == bs && (hs < bhs || (hs == bhs && (dd < bd || (dd == bd && i < bi)))));
                if (better) { bs = s; bd = dd; bhs = hs; bi = i; }
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            double s2 = __shfl_xor(bs, off), d2 = __shfl_xor(bd, off);
            int hs2 = __shfl_xor(bhs, off), i2 = __shfl_xor(bi, off);
            bool better = i2 >= 0 && (bi < 0 || s2 < bs || (s2 == bs && (hs2 < bhs || (hs2 == bhs && (d2 < bd || (d2 == bd && i2 < bi))))));
            if (better) { bs = s2; bd = d2; bhs = hs2; bi = i2; }
        }
        if (lane == 0) { s_best[wave] = bs; s_red[wave][0] = bd; s_bi[wave][0] = bhs; s_bi[wave][1] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w2 = 1; w2 < 4; w2++) {
                double s2 = s_best[w2], d2 = s_red[w2][0];
                int hs2 = s_bi[w2][0], i2 = s_bi[w2][1];
                bool better = i2 >= 0 && (bi < 0 || s2 < bs || (s2 == bs && (hs2 < bhs || (hs2 == bhs && (d2 < bd || (d2 == bd && i2 < bi))))));
                if (better) { bs = s2; bd = d2; bhs = hs2; bi = i2; }
            }
            const Unit &U = units[u0 + bhs];
            long g = U.hyp_off + bi;
            flag[g] &= ~8;
            // full record: change_2d_corner_to_3d_object + fields (:489-513)
            int cfg = (bi & 1) + 1, q, ti;
            hyp_decode(U, bi, q, ti);
            int yi = q % D.n_yaw, rp = q / D.n_yaw;
            const CamRP &C = cam[(long)U.frame * RP_CAP + rp];
            double cx[8], cy[8];
            {
                V2 c[8];
                corners_build<false, true>(U, vpt[(long)U.vp_off + q].vp, U.top_start + ti * U.top_step, cfg, c);
                for (int k = 0; k < 8; k++) { cx[k] = c[k].x; cy[k] = c[k].y; }
            }
            Conv3D c3;
            corners_to_3d(cx, cy, C, cal.invK, c3);
            cs_cuboid &O = out[(long)box * o.max_cuboid_num + pick];
            for (int k = 0; k < 3; k++) { O.pos[k] = c3.pos[k]; O.scale[k] = c3.scale[k]; }
            O.rotY = yaw[(long)U.frame * o.yaw_cap + yi];
            int vp1 = flag[g] & 3;
            O.box_config_type[0] = cfg; O.box_config_type[1] = vp1;
            const int map1[8] = {6, 5, 8, 7, 2, 3, 4, 1}, map2[8] = {5, 6, 7, 8, 3, 2, 1, 4}; // object_3d_util.cpp:637-640
            for (int k = 0; k < 8; k++) {
                int src = (vp1 == 1 ? map1[k] : map2[k]) - 1;
                O.box_corners_2d[k] = (int)cx[src];
                O.box_corners_2d[8 + k] = (int)cy[src];
            }
            { // compute3D_BoxCorner :41-50
                const double body[3][8] = {{1, 1, -1, -1, 1, 1, -1, -1}, {1, -1, -1, 1, 1, -1, -1, 1}, {-1, -1, -1, -1, 1, 1, 1, 1}};
                double cyw = cos(O.rotY), syw = sin(O.rotY);
                double rot[3][3] = {{cyw, -syw, 0}, {syw, cyw, 0}, {0, 0, 1}};
                double rs[3][3];
                for (int i = 0; i < 3; i++)
                    for (int j = 0; j < 3; j++) {
                        double s = rot[i][0] * (j == 0 ? O.scale[0] : 0.0);
                        s = s + rot[i][1] * (j == 1 ? O.scale[1] : 0.0);
                        s = s + rot[i][2] * (j == 2 ? O.scale[2] : 0.0);
                        rs[i][j] = s;
                    }
                for (int k = 0; k < 8; k++) {
                    double wv[4];
                    for (int i = 0; i < 3; i++) {
                        double s = rs[i][0] * body[0][k];
                        s = s + rs[i][1] * body[1][k];
                        s = s + rs[i][2] * body[2][k];
                        s = s + O.pos[i] * 1.0;
                        wv[i] = s;
                    }
                    wv[3] = ((0.0 * body[0][k] + 0.0 * body[1][k]) + 0.0 * body[2][k]) + 1.0;
                    for (int i = 0; i < 3; i++) O.box_corners_3d_world[i * 8 + k] = wv[i] / wv[3];
                }
            }
            O.rect_detect_2d[0] = U.left; O.rect_detect_2d[1] = U.top; O.rect_detect_2d[2] = U.width_raw; O.rect_detect_2d[3] = U.height_raw;
            O.edge_distance_error = derr[g]; O.edge_angle_error = aerr[g];
            O.normalized_error = nscore[g];
            double mxs = c3.scale[0] < c3.scale[1] ? c3.scale[1] : c3.scale[0];
            double mns = c3.scale[1] < c3.scale[0] ? c3.scale[1] : c3.scale[0];
            O.skew_ratio = mxs / mns;
            O.down_expand_height = U.down_expand;
            if (o.sample_rp) { O.camera_roll_delta = C.roll - FI.euler[0]; O.camera_pitch_delta = C.pitch - FI.euler[1]; }
            else { O.camera_roll_delta = 0; O.camera_pitch_delta = 0; }
        }
        __syncthreads();
    }
    if (tid == 0) counts[box] = kout;
}

__global__ void cuboid_bgr2gray(const uint8_t *bgr, int stride, int w, int h, uint8_t *gray) { // cvtColor(CV_BGR2GRAY), :62-66
    int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= w) return;
    const uint8_t *p = bgr + (long)y * stride + 3 * x;
    gray[(long)y * w + x] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
}

// ------------------------------------------------------------------------------------------------ host side
static void host_euler_from_T(const double *T, double *euler) { // set_cam_pose :45-48 with the host libm
    double m[3][3];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) m[i][j] = T[i * 4 + j];
    double qw, qx, qy, qz, q[3];
    double t = m[0][0] + m[1][1] + m[2][2];
    if (t > 0) { // Eigen::Quaterniond(Matrix3d)
        t = std::sqrt(t + 1.0);
        qw = 0.5 * t; t = 0.5 / t;
        qx = (m[2][1] - m[1][2]) * t; qy = (m[0][2] - m[2][0]) * t; qz = (m[1][0] - m[0][1]) * t;
    } else {
        int i = 0;
        if (m[1][1] > m[0][0]) i = 1;
        if (m[2][2] > m[i][i]) i = 2;
        int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(m[i][i] - m[j][j] - m[k][k] + 1.0);
        q[i] = 0.5 * t; t = 0.5 / t;
        qw = (m[k][j] - m[j][k]) * t;
        q[j] = (m[j][i] + m[i][j]) * t;
        q[k] = (m[k][i] + m[i][k]) * t;
        qx = q[0]; qy = q[1]; qz = q[2];
    }
    euler[0] = std::atan2(2 * (qw * qx + qy * qz), 1 - 2 * (qx * qx + qy * qy)); // quat_to_euler_zyx matrix_utils.cpp:35-46
    euler[1] = std::asin(2 * (qw * qy - qz * qx));
    euler[2] = std::atan2(2 * (qw * qz + qx * qy), 1 - 2 * (qy * qy + qz * qz));
}

} // namespace

struct cs_cuboid_batch {
    int n_frames = 0, W = 0, H = 0, n_boxes = 0, n_units = 0;
    Opts o{};
    Calib cal{};
    std::vector<Unit> units;
    std::vector<FrameInfo> fi;   // host copy (cs_cuboid_batch_set_lines rewrites the line ranges)
    long cap_lines_in = 0, cap_line_rows = 0;
    // capacities of the plan-sized device arrays (cs_cuboid_batch_set_scene grows them when another set of boxes needs more) and the pinned staging of its uploads
    long cap_pix = 0, cap_hyp = 0, cap_vp = 0, cap_dttmp = 0; int cap_units = 0, cap_boxes = 0, sample_bbox_height = 0;
    uint8_t *h_stage[2] = {nullptr, nullptr}; size_t stage_cap[2] = {0, 0}; hipEvent_t stage_ev[2] = {nullptr, nullptr}; int stage_k = 0;
    std::vector<int> box_first_unit;
    long pix_total = 0, hyp_total = 0, vp_total = 0, line_rows = 0;
    int max_tiles = 0, max_cc_blocks = 0, max_vp_blocks = 0, blocks_per_unit = 0, max_roi_w = 0;
    int n_nms = 0, n_cc = 0, n_hb = 0, cap_wgmap = 0; // workgroups of the compact grids; d_wgmap = [unit of every NMS tile | of every 4 096-pixel band | of every hypothesis block]
    int *d_wgmap = nullptr;
    // the two branches of a run that read nothing of each other -- poses / edge lists / vanishing points, and the image: Canny, components, distance transform -- on two
    // streams, joined before the sweep (cs_cuboid_batch_run): a small batch (one frame of a drop-in call, config 4's 512 units) does not fill the chip and its kernels are
    // per-unit latency chains, so the branches cost their sum when queued behind each other
    hipEvent_t ev_fork = nullptr, ev_join = nullptr; // (the second stream is the context's: cs_ctx::aux_stream)
    // device
    uint8_t *d_gray = nullptr, *d_emap = nullptr, *d_flag = nullptr;
    int *d_lab = nullptr; // aliases d_dist
    float *d_dist = nullptr;
    int *d_dttmp = nullptr; long *d_dttmp_off = nullptr;
    int *d_order = nullptr, *d_cursor = nullptr; // cuboid_sweep_score: units by falling cost estimate, the work cursor
    int *d_uflag = nullptr;                      // per unit: 1 = a pixel without a 16-bit code (scored from the float map)
    unsigned long long *d_prof = nullptr;        // CUBESLAM_SCORE_PROF: per wave {copy, tasks, barrier wait} ticks and task count of the last launch
    int score_G = 256;      // workgroups of cuboid_sweep_score (one per CU)
    int score_T = 512;      // threads per workgroup (CUBESLAM_SCORE_THREADS = 256 | 512 | 768 | 1024: 1 / 2 / 3 / 4 waves per SIMD with 128 / 256 / 168 / 128 registers)
    bool score_T_forced = false; // set by the environment: cs_cuboid_batch_set_shared_gpu leaves it alone
    int score_slices = 1;   // items per unit (more than one when there are fewer units than CUs)
    int dt_C = 0; // wave-per-ROI distance transform: int map between the passes, lane-major
    FrameInfo *d_fi = nullptr; FrameDyn *d_fd = nullptr; CamRP *d_cam = nullptr;
    double *d_yaw = nullptr, *d_lines_in = nullptr, *d_lines_al = nullptr, *d_mlines = nullptr, *d_mangle = nullptr, *d_mmid = nullptr;
    Unit *d_units = nullptr; UnitDyn *d_ud = nullptr; int *d_box_first = nullptr, *d_status = nullptr, *d_counts = nullptr, *d_carry = nullptr;
    bool want_carry = false; // cs_cuboid_detect's box-to-box chain asks cuboid_select which roll / pitch sample the last kept proposal carries
    VPEntry *d_vp = nullptr;
    int *d_vcount = nullptr, *d_vlist = nullptr; // per unit: number of surviving proposals and their hypothesis indices
    double *d_derr = nullptr, *d_aerr = nullptr, *d_score = nullptr, *d_nscore = nullptr;
    unsigned long long *d_ckd = nullptr, *d_cka = nullptr; int *d_cidx = nullptr;
    cs_cuboid *d_out = nullptr;
};


// The plan of a batch: one Unit per (frame, box, height sample) with its ROI, top samples and its slices of the pixel / hypothesis / vanishing-point / merged-line arenas
// (box_proposal_detail.cpp:107-161), and the per-frame pose record.  Built on the host from the boxes and poses alone (3 072 units: ~0.2 ms), by cs_cuboid_batch_create and
// again by cs_cuboid_batch_set_scene when a step brings other boxes.  keep_lines: the frames' line ranges when the edge lists stay (line_offsets NULL).
struct Plan {
    std::vector<Unit> units; std::vector<FrameInfo> fi; std::vector<int> box_first_unit;
    long pix_total = 0, hyp_total = 0, vp_total = 0, line_rows = 0;
    int max_tiles = 0, max_cc_blocks = 0, max_vp_blocks = 0, blocks_per_unit = 0, max_roi_w = 0, n_boxes = 0;
    int n_nms = 0, n_cc = 0, n_hb = 0; // workgroups of the compact grids
};
static int plan_build(const Opts &o, int sample_bbox_height, int n_frames, int width, int height, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets,
                      const std::vector<FrameInfo> *keep_lines, Plan &P) {
    P.n_boxes = box_offsets[n_frames];
    std::vector<FrameInfo> &fi = P.fi;
    fi.resize(n_frames);
    for (int f = 0; f < n_frames; f++) {
        for (int i = 0; i < 16; i++) fi[f].T[i] = Twc[(long)f * 16 + i];
        host_euler_from_T(fi[f].T, fi[f].euler);
        fi[f].yaw_src = fi[f].euler[2];
        if (line_offsets) { fi[f].line_off = line_offsets[f]; fi[f].n_lines = line_offsets[f + 1] - line_offsets[f]; }
        else { fi[f].line_off = (*keep_lines)[f].line_off; fi[f].n_lines = (*keep_lines)[f].n_lines; }
    }
    const int rp_cap = o.sample_rp ? 25 : 1; // 5x5 at most (linespace +-6 deg step 3 deg gives 4 or 5 per axis)
    for (int f = 0; f < n_frames; f++)
        for (int bi = box_offsets[f]; bi < box_offsets[f + 1]; bi++) {
            const double *bb = boxes + (long)bi * 5;
            // box_proposal_detail.cpp:107-123
            int left_x_raw = (int)bb[0], top_y_raw = (int)bb[1], obj_width_raw = (int)bb[2], obj_height_raw = (int)bb[3];
            int right_x_raw = (int)(left_x_raw + bb[2]);
            int hs_list[3], n_hs = 0;
            hs_list[n_hs++] = 0;
            if (sample_bbox_height) {
                int r = std::max(std::min(20, obj_height_raw - 90), 20);
                r = std::min(r, height - top_y_raw - obj_height_raw - 1);
                if (r > 10) hs_list[n_hs++] = (int)std::round(r / 2);
                hs_list[n_hs++] = r;
            }
            P.box_first_unit.push_back((int)P.units.size());
            for (int hs = 0; hs < n_hs; hs++) {
                Unit U{};
                U.frame = f; U.box = bi; U.hs = hs; U.n_hs = n_hs;
                U.left = left_x_raw; U.top = top_y_raw; U.right = right_x_raw; U.width_raw = obj_width_raw; U.height_raw = obj_height_raw;
                U.down_expand = hs_list[hs];
                int obj_height_expan = obj_height_raw + U.down_expand; // :139-141
                U.down_y_expan = top_y_raw + obj_height_expan;
                U.diag = std::sqrt((double)(obj_width_raw * obj_width_raw + obj_height_expan * obj_height_expan));
                int res = (int)std::round((double)std::min(20, obj_width_raw / 10)); // :144-146
                {
                    int s = left_x_raw + 5, e = right_x_raw - 5, n = 0;
                    while (s <= e) { n++; s += res; if (n > 1000) break; }
                    U.n_tops = n; U.top_start = left_x_raw + 5; U.top_step = res;
                }
                if (U.n_tops > 1) { // pair index / n_tops by one multiply when that is exact for every pair index of the unit (p * e < 2^32, e = magic * n - 2^32)
                    const unsigned long long magic = (1ull << 32) / (unsigned)U.n_tops + 1, e = magic * (unsigned)U.n_tops - (1ull << 32);
                    const unsigned long long pmax = (unsigned long long)rp_cap * o.yaw_cap * U.n_tops;
                    U.tops_magic = (magic < (1ull << 32) && pmax * e < (1ull << 32)) ? (unsigned)magic : 0u;
                }
                int ew = std::min(std::max(std::min(20, obj_width_raw - 100), 10), std::max(std::min(20, obj_height_expan - 100), 10)); // :155
                U.roi_x = std::max(0, left_x_raw - ew);
                U.roi_r = std::min(width - 1, right_x_raw + ew);
                U.roi_y = std::max(0, top_y_raw - ew);
                U.roi_b = std::min(height - 1, U.down_y_expan + ew);
                U.roi_w = U.roi_r - U.roi_x; U.roi_h = U.roi_b - U.roi_y;
                if (U.roi_w <= 0 || U.roi_h <= 0 || U.roi_x + U.roi_w > width || U.roi_y + U.roi_h > height) return CS_ERR_BAD_ARG;
                U.pix_off = P.pix_total; P.pix_total += ((long)U.roi_w * U.roi_h + 63) / 64 * 64;
                U.hyp_cap = rp_cap * o.yaw_cap * U.n_tops * 2;
                U.hyp_off = P.hyp_total; P.hyp_total += ((long)U.hyp_cap + 63) / 64 * 64;
                U.vp_off = (int)P.vp_total; P.vp_total += (long)rp_cap * o.yaw_cap;
                U.line_off = (int)P.line_rows; P.line_rows += std::min(fi[f].n_lines, CS_MAX_ROI_LINES);
                U.nms_off = P.n_nms; P.n_nms += ((U.roi_w + NMS_TW - 1) / NMS_TW) * ((U.roi_h + NMS_TH - 1) / NMS_TH);
                U.cc_off = P.n_cc; P.n_cc += (int)(((long)U.roi_w * U.roi_h + 4095) / 4096);
                U.hb_off = P.n_hb; P.n_hb += (U.hyp_cap + SWEEP_HB - 1) / SWEEP_HB;
                P.max_roi_w = std::max(P.max_roi_w, U.roi_w);
                P.max_tiles = std::max(P.max_tiles, ((U.roi_w + NMS_TW - 1) / NMS_TW) * ((U.roi_h + NMS_TH - 1) / NMS_TH));
                P.max_cc_blocks = std::max(P.max_cc_blocks, (int)(((long)U.roi_w * U.roi_h + 4095) / 4096));
                P.blocks_per_unit = std::max(P.blocks_per_unit, (U.hyp_cap + SWEEP_HB - 1) / SWEEP_HB);
                P.units.push_back(U);
            }
        }
    P.max_vp_blocks = (rp_cap * o.yaw_cap + 255) / 256;
    if (P.vp_total > INT_MAX || P.line_rows > INT_MAX) return CS_ERR_CAPACITY;
    return CS_OK;
}
static void plan_commit(cs_cuboid_batch *b, Plan &P) {
    b->units.swap(P.units); b->fi.swap(P.fi); b->box_first_unit.swap(P.box_first_unit);
    b->n_boxes = P.n_boxes; b->n_units = (int)b->units.size();
    b->pix_total = P.pix_total; b->hyp_total = P.hyp_total; b->vp_total = P.vp_total; b->line_rows = P.line_rows;
    b->n_nms = P.n_nms; b->n_cc = P.n_cc; b->n_hb = P.n_hb;
    b->max_tiles = P.max_tiles; b->max_cc_blocks = P.max_cc_blocks; b->max_vp_blocks = P.max_vp_blocks; b->blocks_per_unit = P.blocks_per_unit; b->max_roi_w = P.max_roi_w;
}
// the distance transform's column count per lane, its scratch offsets, the score kernel's item split and the units by falling cost estimate: functions of the plan
static void plan_derived(cs_cuboid_batch *b, std::vector<long> &dt_off, std::vector<int> &order, std::vector<int> &wgmap) {
    wgmap.resize((size_t)b->n_nms + b->n_cc + b->n_hb);
    for (size_t u = 0; u < b->units.size(); u++) {
        const Unit &U = b->units[u];
        const int e_nms = u + 1 < b->units.size() ? b->units[u + 1].nms_off : b->n_nms, e_cc = u + 1 < b->units.size() ? b->units[u + 1].cc_off : b->n_cc, e_hb = u + 1 < b->units.size() ? b->units[u + 1].hb_off : b->n_hb;
        for (int k = U.nms_off; k < e_nms; k++) wgmap[(size_t)k] = (int)u;
        for (int k = U.cc_off; k < e_cc; k++) wgmap[(size_t)b->n_nms + k] = (int)u;
        for (int k = U.hb_off; k < e_hb; k++) wgmap[(size_t)b->n_nms + b->n_cc + k] = (int)u;
    }
    static const int CS_[] = {4, 5, 6, 8, 10, 12, 16, 20};
    const int need = (b->max_roi_w + 63) / 64;
    b->dt_C = 0;
    for (int c : CS_) if (c >= need) { b->dt_C = c; break; }
    const char *dte = getenv("CUBESLAM_DT"); // "block": the workgroup-per-ROI variant
    if (dte && !strcmp(dte, "block")) b->dt_C = 0;
    dt_off.assign(b->units.size() + 1, 0);
    if (b->dt_C) for (size_t u = 0; u < b->units.size(); u++) dt_off[u + 1] = dt_off[u] + (long)b->units[u].roi_h * b->dt_C * 64;
    b->score_slices = b->n_units >= 2 * b->score_G ? 1 : std::min(16, (2 * b->score_G + b->n_units - 1) / std::max(1, b->n_units));
    const char *se = getenv("CUBESLAM_SCORE_SLICES"); // tuning knob / tests: items per unit
    if (se && atoi(se) > 0) b->score_slices = std::min(64, atoi(se));
    // units by falling cost estimate (map copy ~ pixels, scoring ~ hypotheses): the persistent workgroups take the big ones first
    order.resize(b->units.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    auto cost = [&](int u) { const Unit &U = b->units[u]; return 0.19 * (double)U.roi_w * U.roi_h + 10.8 * (double)U.hyp_cap; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) { return cost(a) > cost(c); });
}

extern "C" {

void cs_cuboid_default_opts(cs_cuboid_opts *o) {
    if (!o) return;
    o->consider_config_1 = 1; o->consider_config_2 = 1;
    o->whether_sample_cam_roll_pitch = 0; o->whether_sample_bbox_height = 0;
    o->max_cuboid_num = 1; o->nominal_skew_ratio = 1; o->max_cut_skew = 3;
    o->yaw_range_deg = 45; o->yaw_step_deg = 6; o->canny_low = 80; o->canny_high = 200;
}

void cs_cuboid_batch_destroy(cs_ctx *ctx, cs_cuboid_batch *b) {
    if (!b) return;
    if (ctx) { hipSetDevice(ctx->device); hipStreamSynchronize(ctx->stream); }
    void *ptrs[] = {b->d_gray, b->d_emap, b->d_flag, b->d_dist, b->d_dttmp, b->d_dttmp_off, b->d_fi, b->d_fd, b->d_cam, b->d_yaw, b->d_lines_in, b->d_lines_al,
                    b->d_mlines, b->d_mangle, b->d_mmid, b->d_units, b->d_ud, b->d_box_first, b->d_status, b->d_counts, b->d_carry, b->d_vp,
                    b->d_derr, b->d_aerr, b->d_score, b->d_nscore, b->d_ckd, b->d_cka, b->d_cidx, b->d_out, b->d_vcount, b->d_vlist, b->d_order, b->d_cursor, b->d_uflag, b->d_prof, b->d_wgmap};
    for (void *p : ptrs) cs_dfree(ctx, p);
    for (int k = 0; k < 2; k++) { if (b->h_stage[k]) hipHostFree(b->h_stage[k]); if (b->stage_ev[k]) hipEventDestroy(b->stage_ev[k]); }
    if (b->ev_fork) hipEventDestroy(b->ev_fork);
    if (b->ev_join) hipEventDestroy(b->ev_join);
    delete b;
}

int cs_cuboid_batch_create(cs_ctx *ctx, int n_frames, int width, int height, const uint8_t *gray, const double *K, const double *Twc,
                           const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines,
                           const cs_cuboid_opts *opts, cs_cuboid_batch **out) {
    if (!ctx || !out || n_frames <= 0 || width <= 0 || height <= 0 || !gray || !K || !Twc || !box_offsets || !line_offsets || !opts)
        return CS_ERR_BAD_ARG;
    if (opts->max_cuboid_num < 1 || !(opts->yaw_step_deg > 0)) return CS_ERR_BAD_ARG;
    *out = nullptr;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    cs_cuboid_batch *b = new (std::nothrow) cs_cuboid_batch();
    if (!b) return CS_ERR_NOMEM;
    b->n_frames = n_frames; b->W = width; b->H = height;
    Opts &o = b->o;
    o.cfg1 = opts->consider_config_1; o.cfg2 = opts->consider_config_2; o.sample_rp = opts->whether_sample_cam_roll_pitch;
    o.max_cuboid_num = opts->max_cuboid_num; o.nominal_skew_ratio = opts->nominal_skew_ratio; o.max_cut_skew = opts->max_cut_skew;
    o.yaw_range_deg = opts->yaw_range_deg; o.yaw_step_deg = opts->yaw_step_deg;
    o.canny_low = opts->canny_low; o.canny_high = opts->canny_high;
    {
        double cap = std::floor(2.0 * opts->yaw_range_deg / opts->yaw_step_deg) + 3;
        o.yaw_cap = (int)std::min(1001.0, std::max(1.0, cap));
    }
    for (int i = 0; i < 9; i++) b->cal.K[i] = K[i];
    inv3_cof(b->cal.K, b->cal.invK); // set_calibration box_proposal_detail.cpp:36-40
    if ((int)std::lrint((double)(0.955f * 65536)) != DT_HV || (int)std::lrint((double)(1.3693f * 65536)) != DT_DIAG) {
        ctx->err = "chamfer constants mismatch"; delete b; return CS_ERR_BAD_ARG;
    }
    const int n_boxes = box_offsets[n_frames], n_lines = line_offsets[n_frames];
    b->sample_bbox_height = opts->whether_sample_bbox_height;
    {
        Plan P;
        const int status = plan_build(o, b->sample_bbox_height, n_frames, width, height, Twc, box_offsets, boxes, line_offsets, nullptr, P);
        if (status == CS_ERR_CAPACITY) { delete b; return status; }
        if (status != CS_OK) { ctx->err = "box ROI outside the image"; delete b; return status; }
        plan_commit(b, P);
    }
    b->cap_pix = b->pix_total; b->cap_hyp = b->hyp_total; b->cap_vp = b->vp_total; b->cap_units = b->n_units; b->cap_boxes = n_boxes;

#define A_(call) do { int r__ = (call); if (r__ != CS_OK) { cs_cuboid_batch_destroy(ctx, b); return r__; } } while (0)
    const size_t npx = (size_t)n_frames * width * height;
    A_(cs_dalloc(ctx, &b->d_gray, npx));
    A_(cs_dalloc(ctx, &b->d_emap, (size_t)b->pix_total + 256)); // slack: the wave distance transform reads whole dwords at the row ends
    {
        for (const void *fn : {reinterpret_cast<const void *>(cuboid_sweep_score<256>), reinterpret_cast<const void *>(cuboid_sweep_score<512>), reinterpret_cast<const void *>(cuboid_sweep_score<768>),
                               reinterpret_cast<const void *>(cuboid_sweep_score<1024>)}) { // (per call: the attribute is per device)
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, SC_LDS_BYTES);
            if (e != hipSuccess) { ctx->err = hipGetErrorString(e); cs_cuboid_batch_destroy(ctx, b); return CS_ERR_HIP; }
        }
        {
            static int cu_of_device[64]; // (hipGetDeviceProperties is slow enough to show in the per-frame call: asked once per device)
            const int dv = ctx->device & 63;
            int cus = __atomic_load_n(&cu_of_device[dv], __ATOMIC_RELAXED);
            if (cus == 0 && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device) == hipSuccess && cus > 0) __atomic_store_n(&cu_of_device[dv], cus, __ATOMIC_RELAXED);
            if (cus > 0) b->score_G = cus;
        }
        const char *ge = getenv("CUBESLAM_SCORE_SEGMENTS"); // tuning knob: workgroups of cuboid_sweep_score
        if (ge && atoi(ge) > 0) b->score_G = atoi(ge);
        const char *te = getenv("CUBESLAM_SCORE_THREADS"); // tuning knob: 512 (2 waves per SIMD, 256 registers) or 1024 (4 waves per SIMD, 128 registers)
        if (te && (atoi(te) == 256 || atoi(te) == 512 || atoi(te) == 768 || atoi(te) == 1024)) { b->score_T = atoi(te); b->score_T_forced = true; }
        std::vector<long> dt_off; std::vector<int> order, wgmap;
        plan_derived(b, dt_off, order, wgmap);
        b->cap_wgmap = (int)std::max<size_t>(wgmap.size(), 1);
        A_(cs_dalloc(ctx, &b->d_wgmap, (size_t)b->cap_wgmap));
        A_(cs_h2d(ctx, b->d_wgmap, wgmap.data(), wgmap.size()));
        b->cap_dttmp = std::max<long>(dt_off.back(), 1);
        A_(cs_dalloc(ctx, &b->d_dttmp, (size_t)b->cap_dttmp));
        A_(cs_dalloc(ctx, &b->d_dttmp_off, dt_off.size()));
        A_(cs_h2d(ctx, b->d_dttmp_off, dt_off.data(), dt_off.size()));
        A_(cs_dalloc(ctx, &b->d_order, std::max<size_t>(1, order.size())));
        A_(cs_h2d(ctx, b->d_order, order.data(), order.size()));
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (the two vectors are locals)
        A_(cs_dalloc(ctx, &b->d_cursor, (size_t)1));
        A_(cs_dalloc(ctx, &b->d_uflag, (size_t)std::max(1, b->n_units)));
        if (getenv("CUBESLAM_SCORE_PROF")) A_(cs_dalloc(ctx, &b->d_prof, (size_t)b->score_G * 16 * 4));
    }
    A_(cs_dalloc(ctx, &b->d_dist, (size_t)b->pix_total));
    CS_HIP(ctx, hipMemsetAsync(b->d_dist, 0, sizeof(float) * (size_t)b->pix_total, ctx->stream)); // the slices' padding (to 64 pixels) stays zero: cuboid_sweep_score encodes whole groups of 8
    b->d_lab = (int *)b->d_dist;
    A_(cs_dalloc(ctx, &b->d_fi, (size_t)n_frames));
    A_(cs_dalloc(ctx, &b->d_fd, (size_t)n_frames));
    A_(cs_dalloc(ctx, &b->d_cam, (size_t)n_frames * RP_CAP));
    A_(cs_dalloc(ctx, &b->d_yaw, (size_t)n_frames * o.yaw_cap));
    A_(cs_dalloc(ctx, &b->d_lines_in, (size_t)n_lines * 4));
    A_(cs_dalloc(ctx, &b->d_lines_al, (size_t)n_lines * 4));
    A_(cs_dalloc(ctx, &b->d_mlines, (size_t)b->line_rows * 4));
    A_(cs_dalloc(ctx, &b->d_mangle, (size_t)b->line_rows));
    A_(cs_dalloc(ctx, &b->d_mmid, (size_t)b->line_rows * 2));
    b->cap_lines_in = n_lines; b->cap_line_rows = b->line_rows;
    A_(cs_dalloc(ctx, &b->d_units, (size_t)b->n_units));
    A_(cs_dalloc(ctx, &b->d_ud, (size_t)b->n_units));
    A_(cs_dalloc(ctx, &b->d_box_first, (size_t)n_boxes));
    A_(cs_dalloc(ctx, &b->d_status, (size_t)1));
    A_(cs_dalloc(ctx, &b->d_counts, (size_t)n_boxes));
    A_(cs_dalloc(ctx, &b->d_carry, (size_t)n_boxes));
    A_(cs_dalloc(ctx, &b->d_vp, (size_t)b->vp_total));
    A_(cs_dalloc(ctx, &b->d_flag, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_vcount, (size_t)std::max<size_t>(1, 2 * b->units.size())));
    A_(cs_dalloc(ctx, &b->d_vlist, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_derr, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_aerr, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_score, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_nscore, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_ckd, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_cka, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_cidx, (size_t)b->hyp_total));
    A_(cs_dalloc(ctx, &b->d_out, (size_t)n_boxes * o.max_cuboid_num));
    A_(cs_h2d(ctx, b->d_gray, gray, npx));
    A_(cs_h2d(ctx, b->d_fi, b->fi.data(), (size_t)n_frames));
    A_(cs_h2d(ctx, b->d_lines_in, lines, (size_t)n_lines * 4));
    A_(cs_h2d(ctx, b->d_units, b->units.data(), (size_t)b->n_units));
    A_(cs_h2d(ctx, b->d_box_first, b->box_first_unit.data(), (size_t)n_boxes));
    { hipError_t e = hipStreamSynchronize(ctx->stream); if (e != hipSuccess) { ctx->err = hipGetErrorString(e); cs_cuboid_batch_destroy(ctx, b); return CS_ERR_HIP; } }
#undef A_
    *out = b;
    return CS_OK;
}

// New edge lists for the frames of an existing batch (same frames, boxes and options): what a step of the chain detect_filter_lines ->
// detect_cuboid (main_obj.cpp:428-449) hands over when the frames stay resident.  Only the line ranges of the plan change.
int cs_cuboid_batch_n_frames(const cs_cuboid_batch *b) { return b ? b->n_frames : -1; }
int cs_cuboid_batch_n_boxes(const cs_cuboid_batch *b) { return b ? b->n_boxes : -1; }
int cs_cuboid_batch_geometry(const cs_cuboid_batch *b, int *width, int *height, int *n_frames) { // (library-internal, see cs_orb_geometry)
    if (!b) return CS_ERR_BAD_ARG;
    *width = b->W; *height = b->H; *n_frames = b->n_frames;
    return CS_OK;
}
// new pixels for the frames of the batch (same boxes, poses and plan) from DEVICE memory: a copy on the context's stream, nothing waits
int cs_cuboid_batch_set_gray_device(cs_ctx *ctx, cs_cuboid_batch *b, const uint8_t *d_gray) {
    if (!ctx || !b || !d_gray) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    CS_HIP(ctx, hipMemcpyAsync(b->d_gray, d_gray, (size_t)b->n_frames * b->W * b->H, hipMemcpyDeviceToDevice, ctx->stream));
    return CS_OK;
}
int cs_cuboid_batch_set_lines(cs_ctx *ctx, cs_cuboid_batch *b, const int *line_offsets, const double *lines) {
    if (!ctx || !b || !line_offsets || (line_offsets[b->n_frames] > 0 && !lines)) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream)); // (a run still reading the old lists)
    const int n_lines = line_offsets[b->n_frames];
    // the new plan is worked out beside the old one and committed only when every check and allocation has succeeded: a failure leaves the batch as it was
    std::vector<int> unit_off(b->units.size());
    long rows = 0;
    for (size_t u = 0; u < b->units.size(); u++) {
        const int f = b->units[u].frame;
        unit_off[u] = (int)rows; rows += std::min(line_offsets[f + 1] - line_offsets[f], CS_MAX_ROI_LINES);
        if (rows > INT_MAX) return CS_ERR_CAPACITY;
    }
    int r;
    if (n_lines > b->cap_lines_in) {
        const size_t cap = (size_t)n_lines + n_lines / 4 + 64;
        double *a = nullptr, *c = nullptr;
        r = cs_dalloc(ctx, &a, cap * 4); if (r) return r;
        r = cs_dalloc(ctx, &c, cap * 4); if (r) { cs_dfree(ctx, a); return r; }
        cs_dfree(ctx, b->d_lines_in); cs_dfree(ctx, b->d_lines_al);
        b->d_lines_in = a; b->d_lines_al = c; b->cap_lines_in = (long)cap;
    }
    if (rows > b->cap_line_rows) {
        const size_t cap = (size_t)rows + rows / 4 + 64;
        decltype(b->d_mlines) m1 = nullptr; decltype(b->d_mangle) m2 = nullptr; decltype(b->d_mmid) m3 = nullptr;
        r = cs_dalloc(ctx, &m1, cap * 4); if (r) return r;
        r = cs_dalloc(ctx, &m2, cap); if (r) { cs_dfree(ctx, m1); return r; }
        r = cs_dalloc(ctx, &m3, cap * 2); if (r) { cs_dfree(ctx, m1); cs_dfree(ctx, m2); return r; }
        cs_dfree(ctx, b->d_mlines); cs_dfree(ctx, b->d_mangle); cs_dfree(ctx, b->d_mmid);
        b->d_mlines = m1; b->d_mangle = m2; b->d_mmid = m3; b->cap_line_rows = (long)cap;
    }
    for (int f = 0; f < b->n_frames; f++) { b->fi[f].line_off = line_offsets[f]; b->fi[f].n_lines = line_offsets[f + 1] - line_offsets[f]; }
    for (size_t u = 0; u < b->units.size(); u++) b->units[u].line_off = unit_off[u];
    b->line_rows = rows;
    if (n_lines > 0) { r = cs_h2d(ctx, b->d_lines_in, lines, (size_t)n_lines * 4); if (r) return r; }
    r = cs_h2d(ctx, b->d_fi, b->fi.data(), (size_t)b->n_frames); if (r) return r;
    r = cs_h2d(ctx, b->d_units, b->units.data(), (size_t)b->n_units); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

// Other 2-D boxes, camera poses and (optionally) edge lists for the frames of an existing batch -- what every call of detect_cuboid brings with its pixels
// (detect_3d_cuboid.h:62-63, main_obj.cpp:420-449).  The plan (ROIs, top samples, arena slices: box_proposal_detail.cpp:107-161) is rebuilt on the host -- it is a function
// of the boxes and poses alone, ~0.2 ms for 3 072 units -- and goes to the device from pinned staging as copies on the context's stream behind whatever run still reads the
// old plan: nothing waits unless an arena has to grow.  line_offsets NULL: the edge lists stay.  The frame count, the
// image size and the options are the batch's.
int cs_cuboid_batch_set_scene(cs_ctx *ctx, cs_cuboid_batch *b, const double *Twc, const int *box_offsets, const double *boxes, const int *line_offsets, const double *lines) {
    if (!ctx || !b || !Twc || !box_offsets || (box_offsets[b->n_frames] > 0 && !boxes) || (line_offsets && line_offsets[b->n_frames] > 0 && !lines)) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    Plan P;
    int r = plan_build(b->o, b->sample_bbox_height, b->n_frames, b->W, b->H, Twc, box_offsets, boxes, line_offsets, &b->fi, P);
    if (r == CS_ERR_BAD_ARG) ctx->err = "box ROI outside the image";
    if (r != CS_OK) return r; // (the batch is as it was)
    const int n_lines = line_offsets ? line_offsets[b->n_frames] : 0, n_units = (int)P.units.size(), n_boxes = P.n_boxes;
    // arenas that are too small for this plan: wait for the runs that read them, then grow (a quarter of headroom: a stream's box sizes wander)
    bool synced = false;
    auto grow = [&](auto **p, size_t n) -> int { if (!synced) { CS_HIP(ctx, hipStreamSynchronize(ctx->stream)); synced = true; } cs_dfree(ctx, *p); *p = nullptr; return cs_dalloc(ctx, p, n); };
#define G_(call) do { r = (call); if (r != CS_OK) return r; } while (0)
    if (P.pix_total > b->cap_pix) {
        const long c = P.pix_total + P.pix_total / 4;
        G_(grow(&b->d_emap, (size_t)c + 256)); G_(grow(&b->d_dist, (size_t)c)); b->d_lab = (int *)b->d_dist; b->cap_pix = c;
    }
    if (P.hyp_total > b->cap_hyp) {
        const long c = P.hyp_total + P.hyp_total / 4;
        G_(grow(&b->d_flag, (size_t)c)); G_(grow(&b->d_vlist, (size_t)c)); G_(grow(&b->d_derr, (size_t)c)); G_(grow(&b->d_aerr, (size_t)c)); G_(grow(&b->d_score, (size_t)c)); G_(grow(&b->d_nscore, (size_t)c));
        G_(grow(&b->d_ckd, (size_t)c)); G_(grow(&b->d_cka, (size_t)c)); G_(grow(&b->d_cidx, (size_t)c)); b->cap_hyp = c;
    }
    if (P.vp_total > b->cap_vp) { const long c = P.vp_total + P.vp_total / 4; G_(grow(&b->d_vp, (size_t)c)); b->cap_vp = c; }
    if (n_units > b->cap_units) {
        const int c = n_units + n_units / 4;
        G_(grow(&b->d_units, (size_t)c)); G_(grow(&b->d_ud, (size_t)c)); G_(grow(&b->d_order, (size_t)c)); G_(grow(&b->d_uflag, (size_t)c)); G_(grow(&b->d_vcount, (size_t)2 * c)); G_(grow(&b->d_dttmp_off, (size_t)c + 1));
        b->cap_units = c;
    }
    if (n_boxes > b->cap_boxes) {
        const int c = n_boxes + n_boxes / 4;
        G_(grow(&b->d_box_first, (size_t)c)); G_(grow(&b->d_counts, (size_t)c)); G_(grow(&b->d_carry, (size_t)c)); G_(grow(&b->d_out, (size_t)c * b->o.max_cuboid_num)); b->cap_boxes = c;
    }
    if (line_offsets && n_lines > b->cap_lines_in) { const long c = (long)n_lines + n_lines / 4 + 64; G_(grow(&b->d_lines_in, (size_t)c * 4)); G_(grow(&b->d_lines_al, (size_t)c * 4)); b->cap_lines_in = c; }
    if (P.line_rows > b->cap_line_rows) { const long c = P.line_rows + P.line_rows / 4 + 64; G_(grow(&b->d_mlines, (size_t)c * 4)); G_(grow(&b->d_mangle, (size_t)c)); G_(grow(&b->d_mmid, (size_t)c * 2)); b->cap_line_rows = c; }
    plan_commit(b, P);
    std::vector<long> dt_off; std::vector<int> order, wgmap;
    plan_derived(b, dt_off, order, wgmap);
    if ((int)wgmap.size() > b->cap_wgmap) { const int c = (int)(wgmap.size() + wgmap.size() / 4); G_(grow(&b->d_wgmap, (size_t)c)); b->cap_wgmap = c; }
    if (std::max<long>(dt_off.back(), 1) > b->cap_dttmp) { const long c = dt_off.back() + dt_off.back() / 4; G_(grow(&b->d_dttmp, (size_t)c)); b->cap_dttmp = c; }
    // one pinned block: units | frame records | first unit of every box | order | scratch offsets | edge lists; two blocks alternate, a block is reused once its copies are through
    const int k = b->stage_k; b->stage_k ^= 1;
    auto al = [](size_t x) { return (x + 63) & ~(size_t)63; };
    const size_t o_units = 0, o_fi = al(o_units + sizeof(Unit) * (size_t)n_units), o_bf = al(o_fi + sizeof(FrameInfo) * (size_t)b->n_frames), o_ord = al(o_bf + sizeof(int) * (size_t)n_boxes),
                 o_dt = al(o_ord + sizeof(int) * (size_t)n_units), o_wg = al(o_dt + sizeof(long) * ((size_t)n_units + 1)), o_ln = al(o_wg + sizeof(int) * wgmap.size()), total = al(o_ln + sizeof(double) * 4 * (size_t)n_lines);
    if (!b->stage_ev[k]) CS_HIP(ctx, hipEventCreateWithFlags(&b->stage_ev[k], hipEventDisableTiming));
    else CS_HIP(ctx, hipEventSynchronize(b->stage_ev[k]));
    if (total > b->stage_cap[k]) {
        if (b->h_stage[k]) hipHostFree(b->h_stage[k]);
        b->h_stage[k] = nullptr; b->stage_cap[k] = 0;
        CS_HIP(ctx, hipHostMalloc((void **)&b->h_stage[k], total + total / 4, hipHostMallocDefault));
        b->stage_cap[k] = total + total / 4;
    }
    uint8_t *h = b->h_stage[k];
    memcpy(h + o_units, b->units.data(), sizeof(Unit) * (size_t)n_units); memcpy(h + o_fi, b->fi.data(), sizeof(FrameInfo) * (size_t)b->n_frames);
    memcpy(h + o_bf, b->box_first_unit.data(), sizeof(int) * (size_t)n_boxes); memcpy(h + o_ord, order.data(), sizeof(int) * (size_t)n_units);
    memcpy(h + o_dt, dt_off.data(), sizeof(long) * ((size_t)n_units + 1));
    memcpy(h + o_wg, wgmap.data(), sizeof(int) * wgmap.size());
    if (n_lines) memcpy(h + o_ln, lines, sizeof(double) * 4 * (size_t)n_lines);
    G_(cs_h2d(ctx, b->d_units, reinterpret_cast<const Unit *>(h + o_units), (size_t)n_units));
    G_(cs_h2d(ctx, b->d_fi, reinterpret_cast<const FrameInfo *>(h + o_fi), (size_t)b->n_frames));
    G_(cs_h2d(ctx, b->d_box_first, reinterpret_cast<const int *>(h + o_bf), (size_t)n_boxes));
    G_(cs_h2d(ctx, b->d_order, reinterpret_cast<const int *>(h + o_ord), (size_t)n_units));
    G_(cs_h2d(ctx, b->d_dttmp_off, reinterpret_cast<const long *>(h + o_dt), (size_t)n_units + 1));
    G_(cs_h2d(ctx, b->d_wgmap, reinterpret_cast<const int *>(h + o_wg), wgmap.size()));
    if (n_lines) G_(cs_h2d(ctx, b->d_lines_in, reinterpret_cast<const double *>(h + o_ln), (size_t)n_lines * 4));
    CS_HIP(ctx, hipEventRecord(b->stage_ev[k], ctx->stream));
    if (n_units) CS_LAUNCH(ctx, "cuboid_clear_pad", cuboid_clear_pad, dim3((n_units + 3) / 4), dim3(256), 0, b->d_units, n_units, b->d_dist);
#undef G_
    return CS_OK;
}

// shared != 0: kernels that hold whole CUs for a long time run beside this batch (the line detectors' one-wave-per-frame region walks of the alternating
// front-end runner): cuboid_sweep_score then takes the shape that fits into what those leave of a CU (256 threads of at most 128 registers + the LDS they do
// not use) instead of waiting for CUs to come free.  Same results either way.
int cs_cuboid_batch_set_shared_gpu(cs_cuboid_batch *b, int shared) {
    if (!b) return CS_ERR_BAD_ARG;
    if (!b->score_T_forced) b->score_T = shared ? 256 : 512;
    return CS_OK;
}

int cs_cuboid_batch_run(cs_ctx *ctx, cs_cuboid_batch *b) {
    if (!ctx || !b) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    CS_HIP(ctx, hipMemsetAsync(b->d_status, 0, sizeof(int), ctx->stream));
    CS_HIP(ctx, hipMemsetAsync(b->d_out, 0, sizeof(cs_cuboid) * (size_t)std::max(1, b->n_boxes * b->o.max_cuboid_num), ctx->stream));
    if (b->n_units == 0) return CS_OK;
    const int U = b->n_units;
    int ul_cap = 1;
    for (const FrameInfo &fi_ : b->fi) ul_cap = std::max(ul_cap, std::min(fi_.n_lines, CS_MAX_ROI_LINES));
    // (per-kernel event timing reads one stream: a timed run keeps everything on it)
    const bool fork = !ctx->timing;
    if (fork) {
        if (!ctx->aux_stream) CS_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
        hipStream_t side = ctx->aux_stream;
        if (!b->ev_fork) {
            CS_HIP(ctx, hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
            CS_HIP(ctx, hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
        }
        CS_HIP(ctx, hipEventRecord(b->ev_fork, ctx->stream)); // behind the two memsets above, the batch's uploads and the last run's readers
        CS_HIP(ctx, hipStreamWaitEvent(side, b->ev_fork, 0));
        hipLaunchKernelGGL(cuboid_frame_prep, dim3(b->n_frames), dim3(64), 0, side, b->d_fi, b->d_fd, b->d_cam, b->d_yaw, b->cal, b->o, b->d_lines_in, b->d_lines_al);
        hipLaunchKernelGGL(cuboid_unit_lines, dim3(U), dim3(64), sizeof(double) * 5 * (size_t)ul_cap, side, b->d_units, b->d_ud, b->d_fi, b->d_lines_al, b->d_mlines, b->d_mangle, b->d_mmid, b->d_status, ul_cap);
        hipLaunchKernelGGL(cuboid_vp, dim3(b->max_vp_blocks, U), dim3(256), 0, side, b->d_units, b->d_ud, b->d_fd, b->d_cam, b->d_yaw, b->o, b->d_mangle, b->d_mmid, b->d_vp);
        CS_HIP(ctx, hipEventRecord(b->ev_join, side));
    } else {
        CS_LAUNCH(ctx, "cuboid_frame_prep", cuboid_frame_prep, dim3(b->n_frames), dim3(64), 0, b->d_fi, b->d_fd, b->d_cam, b->d_yaw, b->cal, b->o,
                  b->d_lines_in, b->d_lines_al);
        CS_LAUNCH(ctx, "cuboid_unit_lines", cuboid_unit_lines, dim3(U), dim3(64), sizeof(double) * 5 * (size_t)ul_cap, b->d_units, b->d_ud, b->d_fi, b->d_lines_al, b->d_mlines,
                  b->d_mangle, b->d_mmid, b->d_status, ul_cap);
    }
    CS_HIP(ctx, hipMemsetAsync(b->d_emap, 0, (size_t)b->pix_total, ctx->stream));
    CS_LAUNCH(ctx, "cuboid_canny_nms", cuboid_canny_nms, dim3(b->n_nms), dim3(256), 0, b->d_units, b->d_wgmap, b->d_gray, b->W, b->H, b->d_emap,
              b->d_lab, b->o.canny_low, b->o.canny_high);
    CS_LAUNCH(ctx, "cuboid_canny_cc_local", cuboid_canny_cc_local, dim3(b->n_cc), dim3(256), 0, b->d_units, b->d_wgmap + b->n_nms, b->d_emap, b->d_lab);
    if (b->max_cc_blocks > 1)
        CS_LAUNCH(ctx, "cuboid_canny_cc_border", cuboid_canny_cc_border, dim3(U, 4), dim3(256), 0, b->d_units, b->d_emap, b->d_lab);
    CS_LAUNCH(ctx, "cuboid_canny_cc", cuboid_canny_cc, dim3(b->n_cc), dim3(256), 0, b->d_units, b->d_wgmap + b->n_nms, b->d_emap, b->d_lab);
    if (b->dt_C) {
        const dim3 g((U + 3) / 4), t(256);
        switch (b->dt_C) {
#define DTW_(c) case c: CS_LAUNCH(ctx, "cuboid_dt", cuboid_dt_wave<c>, g, t, 0, b->d_units, U, b->d_emap, b->d_dttmp, b->d_dttmp_off, b->d_dist); break;
            DTW_(4) DTW_(5) DTW_(6) DTW_(8) DTW_(10) DTW_(12) DTW_(16) DTW_(20)
#undef DTW_
        }
    } else if (b->max_roi_w <= 1024) {
        CS_LAUNCH(ctx, "cuboid_dt", cuboid_dt_block, dim3(U), dim3(64 * ((b->max_roi_w + 63) / 64)), 0, b->d_units, b->d_emap, b->d_dist);
    } else { // very wide ROIs: one wave per ROI, segments scanned serially
        const int wbuf = b->W + 2;
        CS_LAUNCH(ctx, "cuboid_dt", cuboid_dt, dim3((U + 3) / 4), dim3(256), (size_t)wbuf * 2 * 4 * sizeof(int), b->d_units, U, b->d_emap, b->d_dist, wbuf);
    }
    if (fork) CS_HIP(ctx, hipStreamWaitEvent(ctx->stream, b->ev_join, 0));
    else
        CS_LAUNCH(ctx, "cuboid_vp", cuboid_vp, dim3(b->max_vp_blocks, U), dim3(256), 0, b->d_units, b->d_ud, b->d_fd, b->d_cam, b->d_yaw, b->o,
                  b->d_mangle, b->d_mmid, b->d_vp);
    CS_HIP(ctx, hipMemsetAsync(b->d_vcount, 0, sizeof(int) * 2 * (size_t)U, ctx->stream));
    CS_LAUNCH(ctx, "cuboid_sweep_filter", cuboid_sweep_filter, dim3(b->n_hb), dim3(256), 0, b->d_units, b->d_wgmap + b->n_nms + b->n_cc, b->d_fd, b->o, b->d_vp,
              b->d_flag, b->d_vcount, b->d_vlist);
    CS_HIP(ctx, hipMemsetAsync(b->d_cursor, 0, sizeof(int), ctx->stream));
    CS_HIP(ctx, hipMemsetAsync(b->d_uflag, 0, sizeof(int) * (size_t)U, ctx->stream));
    {
        const int items = U * b->score_slices, grid = std::min(b->score_G, items);
        if (b->score_T == 256)
            CS_LAUNCH(ctx, "cuboid_sweep_score", cuboid_sweep_score<256>, dim3(grid), dim3(256), SC_LDS_BYTES, b->d_units, b->d_order, items, b->score_slices, b->d_cursor, b->d_vp,
                      b->d_dist, b->d_vcount, b->d_vlist, b->d_uflag, b->d_derr, b->d_aerr, b->d_prof);
        else if (b->score_T == 512)
            CS_LAUNCH(ctx, "cuboid_sweep_score", cuboid_sweep_score<512>, dim3(grid), dim3(512), SC_LDS_BYTES, b->d_units, b->d_order, items, b->score_slices, b->d_cursor, b->d_vp,
                      b->d_dist, b->d_vcount, b->d_vlist, b->d_uflag, b->d_derr, b->d_aerr, b->d_prof);
        else if (b->score_T == 768)
            CS_LAUNCH(ctx, "cuboid_sweep_score", cuboid_sweep_score<768>, dim3(grid), dim3(768), SC_LDS_BYTES, b->d_units, b->d_order, items, b->score_slices, b->d_cursor, b->d_vp,
                      b->d_dist, b->d_vcount, b->d_vlist, b->d_uflag, b->d_derr, b->d_aerr, b->d_prof);
        else
            CS_LAUNCH(ctx, "cuboid_sweep_score", cuboid_sweep_score<1024>, dim3(grid), dim3(1024), SC_LDS_BYTES, b->d_units, b->d_order, items, b->score_slices, b->d_cursor, b->d_vp,
                      b->d_dist, b->d_vcount, b->d_vlist, b->d_uflag, b->d_derr, b->d_aerr, b->d_prof);
    }
    CS_LAUNCH(ctx, "cuboid_select", cuboid_select, dim3(b->n_boxes), dim3(256), 0, b->d_units, b->d_ud, b->d_box_first, b->d_fd, b->d_fi,
              b->d_cam, b->d_yaw, b->cal, b->o, b->d_flag, b->d_derr, b->d_aerr, b->d_vp, b->d_score, b->d_nscore,
              b->d_ckd, b->d_cka, b->d_cidx, b->d_out, b->d_counts, b->want_carry ? b->d_carry : nullptr);
    CS_HIP(ctx, hipGetLastError());
    return CS_OK;
}

int cs_cuboid_batch_read(cs_ctx *ctx, cs_cuboid_batch *b, cs_cuboid *out, int *counts) {
    if (!ctx || !b || !out || !counts) return CS_ERR_BAD_ARG;
    int status = 0;
    int r = cs_d2h(ctx, out, b->d_out, (size_t)b->n_boxes * b->o.max_cuboid_num); if (r) return r;
    r = cs_d2h(ctx, counts, b->d_counts, (size_t)b->n_boxes); if (r) return r;
    r = cs_d2h(ctx, &status, b->d_status, 1); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (status != 0) { ctx->err = "more than CS_MAX_ROI_LINES lines inside one box"; return status; }
    return CS_OK;
}

// cs_cuboid_batch_read's copies on a stream of the caller's choice, nothing waits; *status_out is valid once the stream has passed them
int cs_cuboid_batch_read_on(cs_cuboid_batch *b, void *stream, cs_cuboid *out, int *counts, int *status_out) {
    if (!b || !out || !counts || !status_out) return CS_ERR_BAD_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemcpyAsync(out, b->d_out, sizeof(cs_cuboid) * (size_t)b->n_boxes * b->o.max_cuboid_num, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(counts, b->d_counts, sizeof(int) * (size_t)b->n_boxes, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(status_out, b->d_status, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return CS_ERR_HIP;
    return CS_OK;
}

int cs_cuboid_batch_stats(cs_ctx *ctx, cs_cuboid_batch *b, long *n_units, long *roi_pixels, long *n_hypotheses, long *n_valid) {
    if (!ctx || !b) return CS_ERR_BAD_ARG;
    std::vector<UnitDyn> ud(b->n_units);
    std::vector<FrameDyn> fd(b->n_frames);
    int r = cs_d2h(ctx, ud.data(), b->d_ud, ud.size()); if (r) return r;
    r = cs_d2h(ctx, fd.data(), b->d_fd, fd.size()); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    long px = 0, hy = 0, va = 0;
    for (int u = 0; u < b->n_units; u++) {
        const Unit &U = b->units[u];
        const FrameDyn &D = fd[U.frame];
        px += (long)U.roi_w * U.roi_h;
        hy += (long)D.n_roll * D.n_pitch * D.n_yaw * U.n_tops * 2;
        va += ud[u].n_valid;
    }
    if (n_units) *n_units = b->n_units;
    if (roi_pixels) *roi_pixels = px;
    if (n_hypotheses) *n_hypotheses = hy;
    if (n_valid) *n_valid = va;
    return CS_OK;
}

int cs_cuboid_batch_score_stats(cs_ctx *ctx, cs_cuboid_batch *b, long out[6]) {
    if (!ctx || !b || !out) return CS_ERR_BAD_ARG;
    if (b->d_prof) { // (development) where the waves of the last cuboid_sweep_score launch spent their time
        const int grid = std::min(b->score_G, b->n_units * b->score_slices), nw = grid * (b->score_T / 64);
        std::vector<unsigned long long> h((size_t)nw * 4);
        hipStreamSynchronize(ctx->stream);
        if (hipMemcpy(h.data(), b->d_prof, h.size() * 8, hipMemcpyDeviceToHost) == hipSuccess) {
            double c = 0, t = 0, w = 0, n = 0, mx = 0;
            for (int i = 0; i < nw; i++) { c += (double)h[4 * i]; t += (double)h[4 * i + 1]; w += (double)h[4 * i + 2]; n += (double)h[4 * i + 3]; mx = std::max(mx, (double)(h[4 * i] + h[4 * i + 1] + h[4 * i + 2])); }
            fprintf(stderr, "[score prof] %d waves: copy %.1f us, tasks %.1f us (%.1f tasks, %.2f us each), barrier wait %.1f us per wave; longest wave %.1f us\n", nw, c / nw / 100, t / nw / 100, n / nw,
                    n > 0 ? t / n / 100 : 0.0, w / nw / 100, mx / 100);
        }
    }
    std::vector<int> vc(2 * (size_t)b->n_units + 1), uf((size_t)b->n_units + 1);
    int r = cs_d2h(ctx, vc.data(), b->d_vcount, 2 * (size_t)b->n_units); if (r) return r;
    r = cs_d2h(ctx, uf.data(), b->d_uflag, (size_t)b->n_units); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 6; i++) out[i] = 0;
    for (int u = 0; u < b->n_units; u++) {
        const Unit &U = b->units[u];
        const int k = !(uf[u] & 1) ? 0 : 3; // (oversize units keep the head of their map in LDS and gather the tail: same kernel)
        out[k] += 1; out[k + 1] += (long)U.roi_w * U.roi_h; out[k + 2] += vc[2 * u] + vc[2 * u + 1];
    }
    return CS_OK;
}

int cs_cuboid_batch_unit(cs_ctx *ctx, cs_cuboid_batch *b, int unit, int dims[12], uint8_t *edges, float *dist, double *rows, long rows_cap,
                         double *merged, long merged_cap) {
    if (!ctx || !b || unit < 0 || unit >= b->n_units || !dims) return CS_ERR_BAD_ARG;
    const Unit &U = b->units[unit];
    UnitDyn ud; FrameDyn fd; FrameInfo fi;
    int r = cs_d2h(ctx, &ud, b->d_ud + unit, 1); if (r) return r;
    r = cs_d2h(ctx, &fd, b->d_fd + U.frame, 1); if (r) return r;
    r = cs_d2h(ctx, &fi, b->d_fi + U.frame, 1); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const long A = (long)U.roi_w * U.roi_h;
    dims[0] = U.roi_x; dims[1] = U.roi_y; dims[2] = U.roi_w; dims[3] = U.roi_h; dims[4] = U.hyp_cap; dims[5] = ud.n_valid;
    dims[6] = ud.n_merged; dims[7] = fd.n_yaw; dims[8] = U.frame; dims[9] = U.box; dims[10] = U.hs; dims[11] = U.n_hs;
    if (edges) { r = cs_d2h(ctx, edges, b->d_emap + U.pix_off, (size_t)A); if (r) return r; }
    if (dist) { r = cs_d2h(ctx, dist, b->d_dist + U.pix_off, (size_t)A); if (r) return r; }
    if (merged) { r = cs_d2h(ctx, merged, b->d_mlines + (long)U.line_off * 4, (size_t)std::min<long>(merged_cap, ud.n_merged) * 4); if (r) return r; }
    if (rows) {
        const int n_hyp = fd.n_roll * fd.n_pitch * fd.n_yaw * U.n_tops * 2;
        std::vector<uint8_t> fl(n_hyp);
        std::vector<double> de(n_hyp), ae(n_hyp), co((size_t)n_hyp * 16), yw(b->o.yaw_cap);
        r = cs_d2h(ctx, fl.data(), b->d_flag + U.hyp_off, (size_t)n_hyp); if (r) return r;
        r = cs_d2h(ctx, de.data(), b->d_derr + U.hyp_off, (size_t)n_hyp); if (r) return r;
        r = cs_d2h(ctx, ae.data(), b->d_aerr + U.hyp_off, (size_t)n_hyp); if (r) return r;
        { // the corners are not stored anywhere: rebuilt for this unit by the function the kernels use
            double *d_co = nullptr;
            CS_HIP(ctx, hipMalloc((void **)&d_co, sizeof(double) * 16 * (size_t)std::max(1, n_hyp)));
            hipMemsetAsync(d_co, 0, sizeof(double) * 16 * (size_t)std::max(1, n_hyp), ctx->stream);
            if (n_hyp > 0) hipLaunchKernelGGL(cuboid_unit_corners, dim3((n_hyp + 255) / 256), dim3(256), 0, ctx->stream, b->d_units, unit, n_hyp, b->d_vp, b->d_flag, d_co);
            r = cs_d2h(ctx, co.data(), d_co, (size_t)n_hyp * 16);
            hipStreamSynchronize(ctx->stream);
            hipFree(d_co);
            if (r) return r;
        }
        r = cs_d2h(ctx, yw.data(), b->d_yaw + (long)U.frame * b->o.yaw_cap, (size_t)b->o.yaw_cap); if (r) return r;
        CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
        long nr = 0;
        for (int h = 0; h < n_hyp && nr < rows_cap; h++) {
            if (!(fl[h] & 3)) continue;
            int cfg = (h & 1) + 1, q = h >> 1;
            int ti = q % U.n_tops; q /= U.n_tops;
            int yi = q % fd.n_yaw, rp = q / fd.n_yaw;
            double *row = rows + nr * 25;
            row[0] = cfg; row[1] = fl[h] & 3; row[2] = yw[yi]; row[3] = ti; row[4] = de[h]; row[5] = ae[h]; row[6] = U.down_expand;
            row[7] = b->o.sample_rp ? fd.roll[rp / fd.n_pitch] : fi.euler[0];
            row[8] = b->o.sample_rp ? fd.pitch[rp % fd.n_pitch] : fi.euler[1];
            for (int p = 0; p < 16; p++) row[9 + p] = co[(size_t)p * n_hyp + h];
            nr++;
        }
    }
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_cuboid_detect(cs_ctx *ctx, const uint8_t *img, int width, int height, int channels, int stride, const double *K, const double *Twc,
                     const double *boxes, int n_boxes, const double *lines, int n_lines, const cs_cuboid_opts *opts, cs_cuboid *out,
                     int *counts) {
    if (!ctx || !img || width <= 0 || height <= 0 || (channels != 1 && channels != 3) || stride < width * channels || n_boxes < 0 ||
        n_lines < 0 || !opts || !out || !counts)
        return CS_ERR_BAD_ARG;
    if (n_boxes == 0) return CS_OK;
    std::vector<uint8_t> gray((size_t)width * height);
    if (channels == 1) {
        for (int y = 0; y < height; y++) memcpy(&gray[(size_t)y * width], img + (size_t)y * stride, (size_t)width);
    } else {
        CS_HIP(ctx, hipSetDevice(ctx->device));
        uint8_t *d_bgr = nullptr, *d_g = nullptr;
        CS_HIP(ctx, hipMalloc((void **)&d_bgr, (size_t)stride * height));
        if (hipMalloc((void **)&d_g, (size_t)width * height) != hipSuccess) { hipFree(d_bgr); return CS_ERR_NOMEM; }
        hipError_t e = hipMemcpyAsync(d_bgr, img, (size_t)stride * height, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            ctx->begin("cuboid_bgr2gray");
            hipLaunchKernelGGL(cuboid_bgr2gray, dim3((width + 255) / 256, height), dim3(256), 0, ctx->stream, d_bgr, stride, width, height, d_g);
            ctx->end();
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(gray.data(), d_g, (size_t)width * height, hipMemcpyDeviceToHost, ctx->stream);
        const hipError_t es = hipStreamSynchronize(ctx->stream);
        hipFree(d_bgr); hipFree(d_g); // on every path
        if (e == hipSuccess) e = es;
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); return CS_ERR_HIP; }
    }
    int lo[2] = {0, n_lines};
    double dummy[4] = {0, 0, 0, 0};
    // The reference reads cam_pose.camera_yaw at :126 and, while it samples camera roll / pitch, leaves in cam_pose the pose of the last proposal it turned into a cuboid
    // (:481-487; of the last loop pass :233-239 when there was none): box b + 1 of a frame starts its yaw samples from what box b left (pin D1 of DESIGN.md:
    // object_slam/src/main_obj.cpp:442 samples on every frame but the first).  A frame's boxes are therefore chained here, one after the other; without sampling, or with one
    // box, they go as one batch.
    const bool chained = opts->whether_sample_cam_roll_pitch && n_boxes > 1;
    int r = CS_OK;
    double yaw_src = 0; bool have_src = false;
    for (int first = 0; first < n_boxes && r == CS_OK; first += chained ? 1 : n_boxes) {
        const int nb = chained ? 1 : n_boxes;
        int bo[2] = {0, nb};
        cs_cuboid_batch *b = nullptr;
        // the frame's batch is built from (and dropped back into) the context's block pool: a camera stream asks for the same sizes frame after frame
        ctx->pooling = true;
        r = cs_cuboid_batch_create(ctx, 1, width, height, gray.data(), K, Twc, bo, boxes + (size_t)first * 5, lo, n_lines ? lines : dummy, opts, &b);
        ctx->pooling = false;
        if (r != CS_OK) return r;
        b->want_carry = chained;
        if (have_src) { // what the previous box left in cam_pose
            b->fi[0].yaw_src = yaw_src;
            r = cs_h2d(ctx, b->d_fi, b->fi.data(), 1);
        }
        if (r == CS_OK) r = cs_cuboid_batch_run(ctx, b);
        if (r == CS_OK) r = cs_cuboid_batch_read(ctx, b, out + (size_t)first * opts->max_cuboid_num, counts + first);
        if (r == CS_OK && chained && first + 1 < n_boxes) {
            int rp = 0; FrameDyn fdh;
            r = cs_d2h(ctx, &rp, b->d_carry, 1);
            if (r == CS_OK) r = cs_d2h(ctx, &fdh, b->d_fd, 1);
            if (r == CS_OK) {
                CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
                const double roll = fdh.roll[rp / fdh.n_pitch], pitch = fdh.pitch[rp % fdh.n_pitch], yw = b->fi[0].euler[2];
                // euler_zyx_to_rot (matrix_utils.cpp:74-89) with the host libm, then set_cam_pose's way back to the yaw (:45-48)
                const double cp = std::cos(pitch), sp = std::sin(pitch), sr = std::sin(roll), cr = std::cos(roll), sy = std::sin(yw), cy = std::cos(yw);
                double Tn[16] = {0}, e3[3];
                Tn[0] = cp * cy; Tn[1] = (sr * sp * cy) - (cr * sy); Tn[2] = (cr * sp * cy) + (sr * sy);
                Tn[4] = cp * sy; Tn[5] = (sr * sp * sy) + (cr * cy); Tn[6] = (cr * sp * sy) - (sr * cy);
                Tn[8] = -sp;     Tn[9] = sr * cp;                    Tn[10] = cr * cp;
                host_euler_from_T(Tn, e3);
                yaw_src = e3[2]; have_src = true;
            }
        }
        cs_cuboid_batch_destroy(ctx, b);
    }
    return r;
}

} // extern "C"
