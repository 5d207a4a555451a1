// txn_sim: the device region stage of the line detector (cube_slam_amd/csrc/lsd_rg_txn.h, the very source the kernel compiles) run on the
// host with LANES transactions interleaved step by step, against the sequential algorithm of the oracle.  Checks that the fixed point is
// the sequential owner map and the same set of line candidates, and counts rounds / executions / steps.
//   g++ -O2 -std=c++17 -o /tmp/txn_sim tools/lsd_sim/txn_sim.cpp && /tmp/txn_sim frame.raw 640 480 [lanes]
#include "../../oracle/lsd_oracle.cpp"
#include "../../cube_slam_amd/csrc/lsd_rg_txn.h"
#include <cstdio>
#include <set>
using namespace std;
int main(int argc, char **argv) {
    const int W = atoi(argv[2]), H = atoi(argv[3]), LANES = argc > 4 ? atoi(argv[4]) : 1024;
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
    // the sequential algorithm: owner map (ranks) and the seeds whose region reaches rect_improve
    vector<int> own_true(N, INT_MAX); set<int> cand_true; long seq_tests = 0;
    {
        L.used.assign(N, 0);
        vector<RegionPoint> reg(N);
        for (int adx : L.order) if (L.used[adx] == 0 && L.angles[adx] != NOTDEF) {
            int reg_size; double reg_angle;
            vector<uint8_t> before = L.used;
            L.region_grow(adx % w, adx / w, reg, reg_size, reg_angle, prec);
            seq_tests += 9L * reg_size;
            if (reg_size >= min_reg_size) { Rect rec; L.region2rect(reg, reg_size, reg_angle, prec, p, rec); if (L.refine(reg, reg_size, reg_angle, prec, p, rec, L.DENSITY_TH)) cand_true.insert(rank_at[adx]); }
            for (int q = 0; q < N; q++) if (L.used[q] && !before[q]) own_true[q] = rank_at[adx];
        }
    }
    rg::Frame F;
    const int tw = (w + rg::TILE - 1) / rg::TILE, th = (h + rg::TILE - 1) / rg::TILE;
    vector<rg::u64> own(N, rg::FREE);
    vector<int> fp_off(ne, 0), fp_cnt(ne, 0), fp_cap(ne, 0), fp_nt(ne, 0), pool(64 * ne + (1 << 20)), chg(tw * th, INT_MAX), status(4, 0);
    vector<unsigned> execs(ne, 0); vector<uint8_t> flag(ne, 0); vector<double> ra(ne, 0);
    int pool_head = 0;
    F.w = w; F.h = h; F.ne = ne; F.caddr = caddr.data(); F.ang = L.angles.data(); F.mod = L.modgrad.data(); F.own = own.data();
    F.fp_off = fp_off.data(); F.fp_cnt = fp_cnt.data(); F.fp_cap = fp_cap.data(); F.execs = execs.data(); F.fp_nt = fp_nt.data(); F.flag = flag.data(); F.reg_angle = ra.data();
    F.pool = pool.data(); F.pool_cap = (int)pool.size(); F.pool_head = &pool_head; F.chg = chg.data(); F.tw = tw; F.status = status.data(); F.min_reg_size = min_reg_size;
    vector<rg::Txn> T(LANES);
    vector<vector<int>> scratch(LANES, vector<int>(4 * rg::CAP));
    for (int l = 0; l < LANES; l++) { T[l].L = scratch[l].data(); T[l].E = T[l].L + rg::CAP; T[l].P1 = T[l].E + rg::CAP; T[l].TL = T[l].P1 + rg::CAP; }
    vector<int> dirty, next;
    for (int i = 0; i < ne; i++) if (rg::is_initial(F, i)) dirty.push_back(i);
    long total_exec = 0, total_steps = 0, wave_steps = 0;
    int round = 0;
    while (!dirty.empty() && round < 2000) {
        round++;
        int ctl[4] = {0, 0, 0, 0};
        for (auto &t : T) t.phase = rg::PH_IDLE;
        long steps = 0, rsteps = 0;
        for (;;) {
            bool any = false;
            for (int l = 0; l < LANES; l++) if (T[l].phase != rg::PH_DONE) { rg::step(F, T[l], dirty.data(), (int)dirty.size(), ctl); any = true; steps++; }
            if (!any) break;
            rsteps++;
        }
        total_steps += steps; wave_steps += rsteps; total_exec += (long)dirty.size();
        next.clear();
        for (int i = 0; i < ne; i++) if (rg::is_dirty(F, i)) next.push_back(i);
        long wrong = 0; for (int q = 0; q < N; q++) { const unsigned r = rg::rank_of(own[q]); if ((r == 0xFFFFFFFFu ? INT_MAX : (int)r) != own_true[q]) wrong++; }
        if (round <= 40 || round % 20 == 0 || next.empty()) printf("round %d: executed %zu lane-steps %ld lockstep-iterations %ld wrong %ld next %zu\n", round, dirty.size(), steps, rsteps, wrong, next.size());
        fill(chg.begin(), chg.end(), INT_MAX);
        dirty.swap(next);
    }
    set<int> cand;
    for (int i = 0; i < ne; i++) if ((flag[i] & 3) == 3) cand.insert(i);
    long wrong = 0; for (int q = 0; q < N; q++) { const unsigned r = rg::rank_of(own[q]); if ((r == 0xFFFFFFFFu ? INT_MAX : (int)r) != own_true[q]) wrong++; }
    for (int q = 0; q < N; q++) { const unsigned r = rg::rank_of(own[q]); const int o = r == 0xFFFFFFFFu ? INT_MAX : (int)r; if (o != own_true[q] && getenv("SHOW")) printf("  pixel %d (%d,%d): owner %d (seed at %d, flag %d, fp %d, execs %u) truth %d (flag %d, fp %d)\n", q, q % w, q / w, o == INT_MAX ? -1 : o, o == INT_MAX ? -1 : caddr[o], o == INT_MAX ? -1 : flag[o], o == INT_MAX ? -1 : fp_cnt[o], o == INT_MAX ? 0 : execs[o], own_true[q] == INT_MAX ? -1 : own_true[q], own_true[q] == INT_MAX ? -1 : flag[own_true[q]], own_true[q] == INT_MAX ? -1 : fp_cnt[own_true[q]]); }
    printf("defined %d, rounds %d, executions %ld, lane-steps %ld (sequential pixel steps %ld), lockstep iterations %ld, pool %d, status %d\n", ne, round, total_exec, total_steps, seq_tests / 9, wave_steps, pool_head, status[1]);
    printf("owner map wrong %ld, candidates %zu vs %zu %s\n", wrong, cand.size(), cand_true.size(), cand == cand_true ? "EQUAL" : "DIFFERENT");
    return (wrong == 0 && cand == cand_true) ? 0 : 2;
}
