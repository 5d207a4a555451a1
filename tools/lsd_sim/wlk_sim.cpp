// wlk_sim: the two-role region stage of the line detector (cube_slam_amd/csrc/lsd_rg_wlk.h, the very source the kernel compiles: a walker wave with one
// lane per frame that seeds and grows, rectangle waves that take the regions it parks) run on the host with the 64 lanes as loops and the mailbox
// served after every walker iteration, against the sequential algorithm of the oracle: for every frame of the wave the same `used` map at the end and
// the same rectangles handed to rect_improve, bit for bit and in the same order.  -DWLK_ACC=n: accepted pixels per iteration (the kernel's template argument).
//   g++ -O2 -std=c++17 -o /tmp/wlk_sim tools/lsd_sim/wlk_sim.cpp && /tmp/wlk_sim 640 480 a.raw b.raw ...   (up to 64 frames: one walker wave)
#include "../../oracle/lsd_oracle.cpp"
#include "../../cube_slam_amd/csrc/lsd_rg_wlk.h"
#ifndef WLK_ACC
#define WLK_ACC 1
#endif
struct HostMail { // the mailbox policy of the host model: one walker wave, its posts served at the end of the iteration that made them
    rgw::Mail<64> m;
    long served = 0;
    rgw::Mail<64> &mail() { return m; }
    int slot_of(int l) const { return l; }
    int ring_size() const { return 64; }
    void after_iteration(const rgl::Batch &B) { while (m.head < m.tail) { rgw::serve_ticket<rgs::Wave, 64>(B, m, m.head); m.head++; served++; } }
};
#include <cstdio>
using namespace std;
int main(int argc, char **argv) {
    const int W = atoi(argv[1]), H = atoi(argv[2]), NF = argc - 3;
    if (NF < 1 || NF > 64) return 1;
    vector<LSD> Ls(NF);
    vector<vector<Rect>> rect_true(NF);
    int w = 0, h = 0;
    for (int f = 0; f < NF; f++) {
        vector<uint8_t> gray((size_t)W * H);
        FILE *fp = fopen(argv[3 + f], "rb"); if (!fp || fread(gray.data(), 1, gray.size(), fp) != gray.size()) return 1; fclose(fp);
        LSD &L = Ls[f]; L.prepare(gray.data(), W, H);
        w = L.w; h = L.h;
    }
    const int N = w * h;
    const double prec = PI * Ls[0].ANG_TH / 180, p = Ls[0].ANG_TH / 180;
    const double LOG_NT = 5 * (log10(double(w)) + log10(double(h))) / 2 + log10(11.0);
    const int min_reg_size = int(-LOG_NT / log10(p));
    const int HEAD = w + 8, STRIDE = N + 272; // the layout of the library: a head of undefined pixels, padded frames
    vector<float> ang((size_t)HEAD + (size_t)NF * STRIDE + 16, rgs::NOTDEF_F);
    vector<int> caddr, frame_base(NF + 1, 0);

    vector<double> mod((size_t)NF * N);
    vector<float> seed_cs;
    for (int f = 0; f < NF; f++) {
        LSD &L = Ls[f]; L.LOG_NT = LOG_NT;
        { // the sequential algorithm
            L.used.assign(N, 0);
            vector<RegionPoint> reg(N);
            for (int adx : L.order) if (L.used[adx] == 0 && L.angles[adx] != NOTDEF) {
                int reg_size; double reg_angle;
                L.region_grow(adx % w, adx / w, reg, reg_size, reg_angle, prec);
                if (reg_size < min_reg_size) continue;
                Rect rec; L.region2rect(reg, reg_size, reg_angle, prec, p, rec);
                if (!L.refine(reg, reg_size, reg_angle, prec, p, rec, L.DENSITY_TH)) continue;
                rect_true[f].push_back(rec);
            }
        }
        for (int q = 0; q < N; q++) {
            mod[(size_t)f * N + q] = L.modgrad[q];
            if (L.angles[q] == NOTDEF) continue;
            const double a = L.angles[q];
            float d = (float)(a / DEG_TO_RADS);
            if ((double)d * DEG_TO_RADS != a) { const float up = nextafterf(d, 1e9f), dn = nextafterf(d, -1e9f); d = ((double)up * DEG_TO_RADS == a) ? up : dn; }
            if ((double)d * DEG_TO_RADS != a) { printf("angle %d is not a float degree\n", q); return 3; }
            ang[(size_t)HEAD + (size_t)f * STRIDE + q] = d;
            seed_cs.push_back(float(cos(a))); seed_cs.push_back(float(sin(a)));
            bool alone = true;
            const int x = q % w, y = q / w;
            for (int yy = max(y - 1, 0); yy <= min(y + 1, h - 1); yy++) for (int xx = max(x - 1, 0); xx <= min(x + 1, w - 1); xx++) {
                if (xx == x && yy == y) continue;
                const double b = L.angles[xx + yy * w];
                if (b == NOTDEF) continue;
                double nt = a - b; if (nt < 0) nt = -nt;
                if (nt > (3 * PI) / 2) { nt -= 2 * PI; if (nt < 0) nt = -nt; }
                if (nt <= prec) alone = false;
            }
            caddr.push_back(alone ? (q | (int)0x80000000) : q);
        }
        frame_base[f + 1] = (int)caddr.size();
    }
    int list_cap = rgl::CAP;
    if (const char *e = getenv("GRP_CAP")) list_cap = atoi(e);
    vector<rgl::Ent> list((size_t)NF * (list_cap + 16) + 16);
    const int cand_cap = 4096;
    vector<double> rect((size_t)NF * cand_cap * 12);
    vector<int> cand_cnt(NF, 0), status(4 * NF, 0);
    rgl::Batch B;
    B.F = NF; B.ang_head = HEAD; B.ang_stride = STRIDE; B.list_stride = list_cap + 16; B.rect_stride = cand_cap * 12; B.order = nullptr; B.w = w; B.h = h; B.npx = N; B.caddr = caddr.data(); B.frame_base = frame_base.data(); B.ang = ang.data(); B.mod = mod.data(); B.seed_cs = seed_cs.data();
    B.list = list.data(); B.list_cap = list_cap; B.rect = rect.data(); B.cand_cap = cand_cap; B.cand_cnt = cand_cnt.data(); B.status = status.data(); B.min_reg_size = min_reg_size; B.max_iters = 64 * N;
    for (int q = 0; q < 8; q++) caddr.push_back(0); // (the slack the seed batches read)
    B.caddr = caddr.data();
    HostMail mp; memset(&mp.m, 0, sizeof(mp.m));
    rgw::run_walker<rgl::LWave, WLK_ACC>(B, 0, mp);
    printf("rectangle jobs %ld\n", mp.served);
    int rc = 0;
    for (int f = 0; f < NF; f++) {
        const LSD &L = Ls[f];
        long wrong_used = 0, bad_c = 0;
        for (int q = 0; q < N; q++) if (L.angles[q] != NOTDEF) { const bool u = ang[(size_t)HEAD + (size_t)f * STRIDE + q] == rgs::NOTDEF_F; if (u != (L.used[q] != 0)) wrong_used++; }
        static_assert(sizeof(Rect) == 12 * sizeof(double), "");
        for (int k = 0; k < cand_cnt[f] && k < (int)rect_true[f].size(); k++) if (memcmp(&rect_true[f][k], &rect[((size_t)f * cand_cap + k) * 12], sizeof(Rect)) != 0) bad_c++;
        const bool eq = wrong_used == 0 && bad_c == 0 && cand_cnt[f] == (int)rect_true[f].size() && !status[4 * f + 1];
        printf("frame %d: defined %d, grows %d, rectangle stage %d, iterations %d, fail %d; used map wrong %ld, candidates %d vs %zu, differing %ld -> %s\n", f, frame_base[f + 1] - frame_base[f], status[4 * f], status[4 * f + 2],
               status[4 * f + 3], status[4 * f + 1], wrong_used, cand_cnt[f], rect_true[f].size(), bad_c, eq ? "EQUAL" : "DIFFERENT");
        if (!eq) rc = 2;
    }
    return rc;
}
