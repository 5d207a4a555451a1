// dep_sim: how much of the line detector's region stage (lsd.cpp:464-536, 637-871) is a dependent chain?  (the host-model gate for a tile-parallel stage.)
// The reference visits the seeds in raster order and a growth reads and writes the `used` map, so growth j depends on growth i < j exactly when one of them WRITES a pixel whose
// `used` value the other READS (or writes).  The tool runs the oracle's sequential algorithm on a frame with the read and write sets of every growth recorded (region_grow, the
// re-grow of refine() and the releases of refine() / reduce_region_radius()) and reports
//   serial      the steps of today's formulation: one per list pixel visited (+ one per pixel of a region2rect pass), all growths one behind the other;
//   critical    the longest dependent chain through the conflict graph (unbounded lanes, dependencies known in advance): a lower bound for ANY exact schedule;
//   lanes P     a list schedule on P lanes that take the seeds in order and wait for their dependencies (still with the dependencies known in advance);
//   tiles T     the same with a lane per T x T tile (a growth runs on the lane of its seed's tile): the "tile-parallel batches" of VERDICT's item, without the cost of
//               finding the conflicts (a real stage has to speculate and re-run what a neighbour invalidated: count 2 - 3 x on top);
//   window K    growths of K consecutive live seeds started together against the state before the window, the longest conflict-free prefix committed per round
//               (what a speculative stage without a dependency oracle can do), steps = the sum over rounds of the longest growth of the round.
//   g++ -O2 -std=c++17 -o /tmp/dep_sim tools/lsd_sim/dep_sim.cpp && /tmp/dep_sim frames.raw 640 480 n_frames
#include "../../oracle/lsd_oracle.cpp"
#include <algorithm>
#include <cstdio>
#include <queue>
using namespace std;

struct Growth { int seed; long cost; vector<int> rd, wr; };

struct LSDI : LSD {
    vector<int> *rd = nullptr, *wr = nullptr; long steps = 0;
    void grow_i(int sx, int sy, vector<RegionPoint> &reg, int &reg_size, double &reg_angle, double prec) {
        reg_size = 1;
        int addr = sx + sy * w;
        reg[0] = RegionPoint{sx, sy, angles[addr], modgrad[addr]};
        reg_angle = angles[addr];
        float sumdx = float(cos(reg_angle)), sumdy = float(sin(reg_angle));
        used[addr] = 1; wr->push_back(addr);
        for (int i = 0; i < reg_size; ++i) {
            steps++;
            const int px = reg[i].x, py = reg[i].y;
            int xx_min = max(px - 1, 0), xx_max = min(px + 1, w - 1), yy_min = max(py - 1, 0), yy_max = min(py + 1, h - 1);
            for (int yy = yy_min; yy <= yy_max; ++yy) {
                int c_addr = xx_min + yy * w;
                for (int xx = xx_min; xx <= xx_max; ++xx, ++c_addr) {
                    if (angles[c_addr] != NOTDEF) rd->push_back(c_addr); // (a pixel without an angle can never be taken: its `used` value decides nothing)
                    if ((used[c_addr] != 1) && isAligned(c_addr, reg_angle, prec)) {
                        used[c_addr] = 1; wr->push_back(c_addr);
                        const double angle = angles[c_addr];
                        reg[reg_size] = RegionPoint{xx, yy, angle, modgrad[c_addr]};
                        ++reg_size;
                        sumdx += cos(float(angle)); sumdy += sin(float(angle));
                        reg_angle = fastAtan2(sumdy, sumdx) * DEG_TO_RADS;
                    }
                }
            }
        }
    }
    bool reduce_i(vector<RegionPoint> &reg, int &reg_size, double reg_angle, double prec, double p, Rect &rec, double density, double density_th) {
        const double xc = double(reg[0].x), yc = double(reg[0].y);
        auto dsq = [](double x1, double y1, double x2, double y2) { return (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1); };
        const double r1 = dsq(xc, yc, rec.x1, rec.y1), r2 = dsq(xc, yc, rec.x2, rec.y2);
        double radSq = r1 > r2 ? r1 : r2;
        while (density < density_th) {
            radSq *= 0.75 * 0.75;
            for (int i = 0; i < reg_size; ++i)
                if (dsq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) { used[reg[i].x + reg[i].y * w] = 0; wr->push_back(reg[i].x + reg[i].y * w); swap(reg[i], reg[reg_size - 1]); --reg_size; --i; }
            steps += reg_size;
            if (reg_size < 2) return false;
            region2rect(reg, reg_size, reg_angle, prec, p, rec); steps += reg_size;
            density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        }
        return true;
    }
    bool refine_i(vector<RegionPoint> &reg, int &reg_size, double reg_angle, double prec, double p, Rect &rec, double density_th) {
        double density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= density_th) return true;
        const double xc = double(reg[0].x), yc = double(reg[0].y), ang_c = reg[0].angle;
        double sum = 0, s_sum = 0; int n = 0;
        for (int i = 0; i < reg_size; ++i) {
            used[reg[i].x + reg[i].y * w] = 0; wr->push_back(reg[i].x + reg[i].y * w);
            if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) { const double ang_d = angle_diff_signed(reg[i].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
        }
        steps += reg_size;
        const double mean_angle = sum / double(n);
        const double tau = 2.0 * sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
        grow_i(reg[0].x, reg[0].y, reg, reg_size, reg_angle, tau);
        if (reg_size < 2) return false;
        region2rect(reg, reg_size, reg_angle, prec, p, rec); steps += reg_size;
        density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density < density_th) return reduce_i(reg, reg_size, reg_angle, prec, p, rec, density, density_th);
        return true;
    }
};

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: dep_sim frames.raw W H n_frames\n"); return 2; }
    const int W = atoi(argv[2]), H = atoi(argv[3]), NF = atoi(argv[4]);
    FILE *f = fopen(argv[1], "rb"); if (!f) return 1;
    vector<uint8_t> gray((size_t)W * H);
    const int Ps[] = {16, 64, 256, 1024}, Ts[] = {16, 32, 64, 128}, Ks[] = {64, 256, 1024};
    double acc_serial = 0, acc_crit = 0, acc_g = 0, acc_big = 0, acc_P[4] = {0, 0, 0, 0}, acc_T[4] = {0, 0, 0, 0}, acc_K[3] = {0, 0, 0}, acc_Kr[3] = {0, 0, 0}, acc_selfdep = 0;
    printf("frame growths big serial critical | lanes16 lanes64 lanes256 lanes1024 | tile16 tile32 tile64 tile128 | win64 (rounds) win256 win1024\n");
    for (int fr = 0; fr < NF; fr++) {
        if (fread(gray.data(), 1, gray.size(), f) != gray.size()) return 1;
        LSDI L; L.prepare(gray.data(), W, H);
        const int w = L.w, h = L.h, N = w * h;
        const double prec = PI * L.ANG_TH / 180, p = L.ANG_TH / 180;
        L.LOG_NT = 5 * (log10(double(w)) + log10(double(h))) / 2 + log10(11.0);
        const int min_reg_size = int(-L.LOG_NT / log10(p));
        L.used.assign(N, 0);
        vector<RegionPoint> reg(N);
        vector<Growth> G;
        int n_big = 0;
        for (int adx : L.order) if (L.used[adx] == 0 && L.angles[adx] != NOTDEF) {
            G.emplace_back(); Growth &g = G.back(); g.seed = adx;
            L.rd = &g.rd; L.wr = &g.wr; L.steps = 0;
            g.rd.push_back(adx);
            int reg_size; double reg_angle;
            L.grow_i(adx % w, adx / w, reg, reg_size, reg_angle, prec);
            if (reg_size >= min_reg_size) {
                n_big++;
                Rect rec; L.region2rect(reg, reg_size, reg_angle, prec, p, rec); L.steps += reg_size;
                L.refine_i(reg, reg_size, reg_angle, prec, p, rec, L.DENSITY_TH);
            }
            g.cost = L.steps;
            sort(g.rd.begin(), g.rd.end()); g.rd.erase(unique(g.rd.begin(), g.rd.end()), g.rd.end());
            sort(g.wr.begin(), g.wr.end()); g.wr.erase(unique(g.wr.begin(), g.wr.end()), g.wr.end());
        }
        const int ng = (int)G.size();
        long serial = 0; for (auto &g : G) serial += g.cost;
        // dependency time of a growth given finish times of the earlier ones: max over conflicts
        auto schedule = [&](auto lane_of, int n_lanes) { // lane_of(g) < 0: any lane (take the earliest free one)
            vector<long> wT(N, 0), rT(N, 0), lane_free(max(n_lanes, 1), 0);
            priority_queue<long, vector<long>, greater<long>> freeq;
            if (n_lanes > 0) for (int i = 0; i < n_lanes; i++) freeq.push(0);
            long finish = 0;
            for (int j = 0; j < ng; j++) {
                const Growth &g = G[j];
                long dep = 0;
                for (int q : g.rd) dep = max(dep, wT[q]);
                for (int q : g.wr) dep = max(dep, max(rT[q], wT[q]));
                long start = dep;
                const int ln = lane_of(j);
                if (n_lanes > 0) { if (ln < 0) { start = max(start, freeq.top()); freeq.pop(); } else start = max(start, lane_free[ln]); }
                const long T = start + g.cost;
                if (n_lanes > 0) { if (ln < 0) freeq.push(T); else lane_free[ln] = T; }
                for (int q : g.wr) wT[q] = T;
                for (int q : g.rd) rT[q] = max(rT[q], T);
                finish = max(finish, T);
            }
            return finish;
        };
        const long crit = schedule([](int) { return -1; }, 0);
        long lanesP[4], tilesT[4];
        for (int k = 0; k < 4; k++) lanesP[k] = schedule([](int) { return -1; }, Ps[k]);
        for (int k = 0; k < 4; k++) { const int T = Ts[k], tw = (w + T - 1) / T, th = (h + T - 1) / T; tilesT[k] = schedule([&](int j) { return (G[j].seed / w / T) * tw + (G[j].seed % w) / T; }, tw * th); }
        // speculative windows without a dependency oracle: K growths at once, the longest prefix without a conflict inside it commits, the rest runs again (sets taken from the
        // sequential run: a re-run growth touches what it touches in the sequence once everything before it is final, which is when it commits)
        long winK[3], winR[3];
        {
            vector<int> stampW(N, -1), stampR(N, -1);
            for (int k = 0; k < 3; k++) {
                const int K = Ks[k];
                long steps = 0, rounds = 0;
                int j0 = 0;
                while (j0 < ng) {
                    const int j1 = min(ng, j0 + K);
                    long longest = 0; int commit = j1;
                    for (int j = j0; j < j1; j++) {
                        longest = max(longest, G[j].cost);
                        bool conflict = false;
                        for (int q : G[j].rd) if (stampW[q] >= j0 && stampW[q] < j) { conflict = true; break; }
                        if (!conflict) for (int q : G[j].wr) if ((stampW[q] >= j0 && stampW[q] < j) || (stampR[q] >= j0 && stampR[q] < j)) { conflict = true; break; }
                        if (conflict) { commit = j; break; }
                        for (int q : G[j].wr) stampW[q] = j;
                        for (int q : G[j].rd) stampR[q] = j;
                    }
                    long lmax = 0; for (int j = j0; j < j1; j++) lmax = max(lmax, G[j].cost); // every lane of the window ran
                    steps += lmax; rounds++;
                    j0 = max(commit, j0 + 1);
                }
                winK[k] = steps; winR[k] = rounds;
                fill(stampW.begin(), stampW.end(), -1); fill(stampR.begin(), stampR.end(), -1);
            }
        }
        printf("%5d %7d %4d %7ld %8ld | %7ld %7ld %7ld %7ld | %7ld %7ld %7ld %7ld | %8ld (%ld) %8ld %8ld\n", fr, ng, n_big, serial, crit, lanesP[0], lanesP[1], lanesP[2], lanesP[3], tilesT[0], tilesT[1], tilesT[2], tilesT[3],
               winK[0], winR[0], winK[1], winK[2]);
        acc_serial += serial; acc_crit += crit; acc_g += ng; acc_big += n_big;
        for (int k = 0; k < 4; k++) { acc_P[k] += lanesP[k]; acc_T[k] += tilesT[k]; }
        for (int k = 0; k < 3; k++) { acc_K[k] += winK[k]; acc_Kr[k] += winR[k]; }
    }
    (void)acc_selfdep;
    printf("mean  %7.0f %4.0f %7.0f %8.0f | %7.0f %7.0f %7.0f %7.0f | %7.0f %7.0f %7.0f %7.0f | %8.0f (%.0f) %8.0f %8.0f\n", acc_g / NF, acc_big / NF, acc_serial / NF, acc_crit / NF, acc_P[0] / NF, acc_P[1] / NF, acc_P[2] / NF,
           acc_P[3] / NF, acc_T[0] / NF, acc_T[1] / NF, acc_T[2] / NF, acc_T[3] / NF, acc_K[0] / NF, acc_Kr[0] / NF, acc_K[1] / NF, acc_K[2] / NF);
    return 0;
}
