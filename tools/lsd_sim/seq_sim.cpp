// seq_sim: the wave-per-frame region stage of the line detector (cube_slam_amd/csrc/lsd_rg_seq.h, the very source the kernel compiles) run on
// the host with its 64 lanes as loops, against the sequential algorithm of the oracle: same `used` map at the end, same rectangles handed to rect_improve, bit for bit and in
// the same order.  Also counts the neighbourhood fetches (memory round trips of a wave).
//   g++ -O2 -std=c++17 -o /tmp/seq_sim tools/lsd_sim/seq_sim.cpp && /tmp/seq_sim frame.raw 640 480
#include "../../oracle/lsd_oracle.cpp"
#include "../../cube_slam_amd/csrc/lsd_rg_seq.h"
#include <cstdio>
#include <map>
using namespace std;
int main(int argc, char **argv) {
    const int W = atoi(argv[2]), H = atoi(argv[3]);
    vector<uint8_t> gray((size_t)W * H);
    FILE *f = fopen(argv[1], "rb"); if (!f || fread(gray.data(), 1, gray.size(), f) != gray.size()) return 1; fclose(f);
    LSD L; L.prepare(gray.data(), W, H);
    const int w = L.w, h = L.h, N = w * h;
    const double prec = PI * L.ANG_TH / 180, p = L.ANG_TH / 180;
    L.LOG_NT = 5 * (log10(double(w)) + log10(double(h))) / 2 + log10(11.0);
    const int min_reg_size = int(-L.LOG_NT / log10(p));
    vector<int> caddr, rank_at(N, -1);
    for (int q = 0; q < N; q++) if (L.angles[q] != NOTDEF) { rank_at[q] = (int)caddr.size(); caddr.push_back(q); }
    const int ne = (int)caddr.size();
    // the sequential algorithm: candidates (rank -> pixel list, region angle) and the used map at the end
    vector<Rect> rect_true;
    long seq_pix = 0, seq_grows = 0;
    {
        L.used.assign(N, 0);
        vector<RegionPoint> reg(N);
        for (int adx : L.order) if (L.used[adx] == 0 && L.angles[adx] != NOTDEF) {
            int reg_size; double reg_angle;
            L.region_grow(adx % w, adx / w, reg, reg_size, reg_angle, prec);
            seq_pix += reg_size; seq_grows++;
            if (reg_size < min_reg_size) continue;
            Rect rec; L.region2rect(reg, reg_size, reg_angle, prec, p, rec);
            if (!L.refine(reg, reg_size, reg_angle, prec, p, rec, L.DENSITY_TH)) continue;
            rect_true.push_back(rec);
        }
    }
    // what lsd_emit hands over: the map of float degrees (the walk's own copy and the read-only one), the seeds' cos / sin, the "stays alone" flag
    vector<float> fre(N, rgs::NOTDEF_F), ang(N, rgs::NOTDEF_F);
    vector<float> seed_cs(2 * (size_t)caddr.size());
    for (int i = 0; i < ne; i++) {
        const int q = caddr[i]; const double a = L.angles[q];
        float d = (float)(a / DEG_TO_RADS);
        if ((double)d * DEG_TO_RADS != a) { const float up = nextafterf(d, 1e9f), dn = nextafterf(d, -1e9f); d = ((double)up * DEG_TO_RADS == a) ? up : dn; }
        if ((double)d * DEG_TO_RADS != a) { printf("angle %d is not a float degree\n", q); return 3; }
        fre[q] = d; ang[q] = d;
        seed_cs[2 * i] = float(cos(a)); seed_cs[2 * i + 1] = float(sin(a));
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
        if (alone) caddr[i] |= (int)0x80000000;
    }
    vector<int> status(4, 0), Lglob(rgs::CAP);
    rgs::List Llist; Llist.glob = Lglob.data();
    vector<double> rect((size_t)12 * ne); int cand_cnt = 0;
    rgs::Frame F;
    F.w = w; F.h = h; F.ne = ne; F.caddr = caddr.data(); F.fre = fre.data(); F.ang = ang.data(); F.mod = L.modgrad.data(); F.seed_cs = seed_cs.data(); F.rect = rect.data(); F.cand_cap = ne; F.cand_cnt = &cand_cnt;
    F.status = status.data(); F.min_reg_size = min_reg_size; F.list_cap = rgs::CAP;
    rgs::run_frame<rgs::Wave>(F, Llist);
    long wrong_used = 0;
    for (int q = 0; q < N; q++) if (L.angles[q] != NOTDEF) { const bool u = fre[q] == rgs::NOTDEF_F; if (u != (L.used[q] != 0)) wrong_used++; }
    long n_c = cand_cnt, bad_c = 0;
    static_assert(sizeof(Rect) == 12 * sizeof(double), "");
    for (int k = 0; k < cand_cnt && k < (int)rect_true.size(); k++) if (memcmp(&rect_true[k], &rect[(size_t)12 * k], sizeof(Rect)) != 0) bad_c++; // the rectangles, bit for bit, in seed order
    printf("defined %d, grows %d (sequential %ld), region pixels %ld, rectangle stage %d, fetches %d, overflow %d\n", ne, status[0], seq_grows, seq_pix, status[2], status[3], status[1]);
    printf("used map wrong %ld, candidates %ld vs %zu, differing %ld -> %s\n", wrong_used, n_c, rect_true.size(), bad_c, (wrong_used == 0 && bad_c == 0 && n_c == (long)rect_true.size() && !status[1]) ? "EQUAL" : "DIFFERENT");
    return (wrong_used == 0 && bad_c == 0 && n_c == (long)rect_true.size() && !status[1]) ? 0 : 2;
}
