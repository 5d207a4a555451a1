// async_sim: first model of the owner-map scheme with whole transactions executed one after another inside a round (superseded by txn_sim.cpp, which interleaves the steps of the real transaction source; kept for the round / work counts quoted in DESIGN.md)
#include "../../oracle/lsd_oracle.cpp"
#include <cstdio>
#include <climits>
#include <random>
using namespace std;
static const int INF = INT_MAX;
static int WATCH = -1, ROUND = 0, WSEED = -1;
static std::vector<int> WSNAP;
#define LOGW(q, what, who) do { if ((q) == WATCH) printf("  [round %d] pixel %d %s by %d\n", ROUND, (q), what, (who)); } while (0)
struct Sim2 : LSD {
    vector<int> own;            // rank (= address) of the claiming seed, INF free
    vector<int> chg;            // per tile: min rank that changed something this round
    int tw, th;                 // tiles (8x8)
    vector<vector<int>> fp;     // per seed: footprint of the last execution
    vector<int> bx0, by0, bx1, by1; // read bbox of last execution
    vector<char> active;
    long touched = 0;
    void mark(int q, int lo) { int t = (q / w / 8) * tw + (q % w) / 8; if (lo < chg[t]) chg[t] = lo; }
    // region_grow with owner semantics
    void grow(int s, vector<int> &L, double &reg_angle, double prec, int &x0, int &y0, int &x1, int &y1) {
        L.clear(); L.push_back(s);
        reg_angle = angles[s];
        float sumdx = float(std::cos(reg_angle)), sumdy = float(std::sin(reg_angle));
        for (size_t i = 0; i < L.size(); ++i) {
            const int px = L[i] % w, py = L[i] / w;
            int xx_min = std::max(px - 1, 0), xx_max = std::min(px + 1, w - 1), yy_min = std::max(py - 1, 0), yy_max = std::min(py + 1, h - 1);
            x0 = min(x0, xx_min); x1 = max(x1, xx_max); y0 = min(y0, yy_min); y1 = max(y1, yy_max);
            for (int yy = yy_min; yy <= yy_max; ++yy)
                for (int xx = xx_min; xx <= xx_max; ++xx) {
                    const int c = xx + yy * w;
                    touched++;
                    if (own[c] <= s) continue; // mine or earlier
                    if (!isAligned(c, reg_angle, prec)) continue;
                    if (own[c] != INF) { mark(c, s); LOGW(c, "stolen", s); } else LOGW(c, "claimed", s);
                    own[c] = s;
                    L.push_back(c);
                    const double angle = angles[c];
                    sumdx += std::cos(float(angle)); sumdy += std::sin(float(angle));
                    reg_angle = fastAtan2(sumdy, sumdx) * DEG_TO_RADS;
                }
        }
    }
    void to_points(const vector<int> &L, vector<RegionPoint> &reg) { for (size_t i = 0; i < L.size(); i++) reg[i] = RegionPoint{L[i] % w, L[i] / w, angles[L[i]], modgrad[L[i]]}; }
    void execute(int s, double prec, double p, int min_reg_size, vector<RegionPoint> &reg, vector<int> &L) {
        vector<int> old; old.swap(fp[s]);
        { vector<int> eff; for (int q : old) if (own[q] == s) { own[q] = INF; eff.push_back(q); LOGW(q, "released(start)", s); } old.swap(eff); } // what it really still held
        active[s] = 0;
        if (own[s] < s) { for (int q : old) mark(q, s); return; }
        if (own[s] != INF) mark(s, s);
        own[s] = s;
        int x0 = w, y0 = h, x1 = -1, y1 = -1;
        double reg_angle;
        grow(s, L, reg_angle, prec, x0, y0, x1, y1);
        int reg_size = (int)L.size();
        if (reg_size >= min_reg_size) {
            to_points(L, reg);
            Rect rec;
            region2rect(reg, reg_size, reg_angle, prec, p, rec);
            double density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
            if (density < DENSITY_TH) {
                // refine: release all, regrow with tau, maybe reduce radius
                const double xc = double(reg[0].x), yc = double(reg[0].y), ang_c = reg[0].angle;
                double sum = 0, s_sum = 0; int n = 0;
                for (int i = 0; i < reg_size; ++i) {
                    if (own[L[i]] == s) { own[L[i]] = INF; LOGW(L[i], "released(refine)", s); }
                    if (dist(xc, yc, reg[i].x, reg[i].y) < rec.width) { const double ang_d = angle_diff_signed(reg[i].angle, ang_c); sum += ang_d; s_sum += ang_d * ang_d; ++n; }
                }
                const double mean_angle = sum / double(n);
                const double tau = 2.0 * std::sqrt((s_sum - 2.0 * mean_angle * sum) / double(n) + mean_angle * mean_angle);
                vector<int> Lold = L;
                if (own[s] < s) { /* lost the seed meanwhile: cannot happen in the sequential sim */ }
                own[s] = s;
                grow(s, L, reg_angle, tau, x0, y0, x1, y1);
                reg_size = (int)L.size();
                if (reg_size >= 2) {
                    to_points(L, reg);
                    region2rect(reg, reg_size, reg_angle, prec, p, rec);
                    density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                    if (density < DENSITY_TH) { // reduce_region_radius with own-map release
                        auto dsq = [](double x1_, double y1_, double x2_, double y2_) { return (x2_ - x1_) * (x2_ - x1_) + (y2_ - y1_) * (y2_ - y1_); };
                        const double r1 = dsq(xc, yc, rec.x1, rec.y1), r2 = dsq(xc, yc, rec.x2, rec.y2);
                        double radSq = r1 > r2 ? r1 : r2;
                        while (density < DENSITY_TH) {
                            radSq *= 0.75 * 0.75;
                            for (int i = 0; i < reg_size; ++i)
                                if (dsq(xc, yc, double(reg[i].x), double(reg[i].y)) > radSq) {
                                    if (own[L[i]] == s) { own[L[i]] = INF; LOGW(L[i], "released(radius)", s); }
                                    std::swap(reg[i], reg[reg_size - 1]); std::swap(L[i], L[reg_size - 1]);
                                    --reg_size; --i;
                                }
                            if (reg_size < 2) break;
                            region2rect(reg, reg_size, reg_angle, prec, p, rec);
                            density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
                        }
                    }
                }
                // pixels of the first growth that are not re-claimed stay released
                (void)Lold;
            }
        }
        L.resize(reg_size);
        // new footprint = entries still mine
        for (int q : L) if (own[q] == s) fp[s].push_back(q);
        if (fp[s] != old) { for (int q : old) mark(q, s); for (int q : fp[s]) mark(q, s); } // the footprint changed: everything it touched or left is news for higher ranks
        // pixels of the first growth released by refine and not re-taken: they were mine only transiently within this execution unless in `old`
        bx0[s] = x0; by0[s] = y0; bx1[s] = x1; by1[s] = y1;
        active[s] = 1;
        if (s == WSEED) WSNAP = own;
    }
};
int main(int argc, char **argv) {
    int W = atoi(argv[2]), H = atoi(argv[3]);
    vector<uint8_t> gray((size_t)W * H);
    FILE *f = fopen(argv[1], "rb"); fread(gray.data(), 1, gray.size(), f); fclose(f);
    Sim2 L; L.prepare(gray.data(), W, H);
    const int w = L.w, h = L.h, N = w * h;
    const double prec = PI * L.ANG_TH / 180, p = L.ANG_TH / 180;
    L.LOG_NT = 5 * (log10(double(w)) + log10(double(h))) / 2 + log10(11.0);
    const int min_reg_size = int(-L.LOG_NT / log10(p));
    // ground truth owner map
    vector<int> own_true(N, INF);
    {
        L.used.assign(N, 0);
        vector<RegionPoint> reg(N);
        for (int adx : L.order) if (L.used[adx] == 0 && L.angles[adx] != NOTDEF) {
            int reg_size; double reg_angle;
            vector<uint8_t> before = L.used;
            L.region_grow(adx % w, adx / w, reg, reg_size, reg_angle, prec);
            if (reg_size >= min_reg_size) { Rect rec; L.region2rect(reg, reg_size, reg_angle, prec, p, rec); L.refine(reg, reg_size, reg_angle, prec, p, rec, L.DENSITY_TH); }
            for (int q = 0; q < N; q++) if (L.used[q] && !before[q]) own_true[q] = adx;
        }
    }
    L.own.assign(N, INF); L.tw = (w + 7) / 8; L.th = (h + 7) / 8; L.chg.assign(L.tw * L.th, INF);
    L.fp.assign(N, {}); L.bx0.assign(N, 0); L.by0.assign(N, 0); L.bx1.assign(N, 0); L.by1.assign(N, 0); L.active.assign(N, 0);
    vector<RegionPoint> reg(N); vector<int> Lst;
    mt19937 rng(1);
    vector<int> dirty;
    // round 1: seeds without an aligned lower-rank defined neighbour (likely true seeds)
    for (int s : L.order) {
        if (L.angles[s] == NOTDEF) continue;
        const int x = s % w, y = s / w;
        bool has = false;
        const int nb[4][2] = {{-1, -1}, {0, -1}, {1, -1}, {-1, 0}};
        for (auto &d : nb) { int xx = x + d[0], yy = y + d[1]; if (xx < 0 || yy < 0 || xx >= w) continue; int c = xx + yy * w; if (L.angles[c] != NOTDEF && L.isAligned(c, L.angles[s], prec)) has = true; }
        if (!has) dirty.push_back(s);
    }
    long total_exec = 0;
    if (argc > 4) WATCH = atoi(argv[4]);
    if (argc > 5) WSEED = atoi(argv[5]);
    for (int round = 1; round <= 300; round++) {
        ROUND = round;
        shuffle(dirty.begin(), dirty.end(), rng); // arbitrary order inside a round (the GPU runs them concurrently)
        L.touched = 0;
        for (int s : dirty) L.execute(s, prec, p, min_reg_size, reg, Lst);
        total_exec += dirty.size();
        long wrong = 0; for (int q = 0; q < N; q++) if (L.own[q] != own_true[q]) wrong++;
        size_t nd = dirty.size();
        // next dirty set
        dirty.clear();
        for (int s : L.order) {
            if (L.angles[s] == NOTDEF) continue;
            if (L.active[s]) {
                bool d = L.own[s] != s;
                for (int ty = L.by0[s] / 8; ty <= L.by1[s] / 8 && !d; ty++) for (int tx = L.bx0[s] / 8; tx <= L.bx1[s] / 8; tx++) if (L.chg[ty * L.tw + tx] < s) { d = true; break; }
                if (d) dirty.push_back(s);
            } else if (L.own[s] > s) dirty.push_back(s); // a defined pixel that is free or held by a HIGHER rank is a seed
            else if (!L.fp[s].empty()) dirty.push_back(s);  // lost its seed: must release its footprint
        }
        if (WSEED >= 0 && !WSNAP.empty()) {
            int s = WSEED;
            for (int y = L.by0[s]; y <= L.by1[s]; y++) for (int x = L.bx0[s]; x <= L.bx1[s]; x++) {
                int q = x + y * w; int a = WSNAP[q], b2 = L.own[q];
                if (a != b2 && (std::min(a, b2) < s)) printf("  watch round %d: pixel %d (%d,%d) own %d -> %d tile mark %d\n", round, q, x, y, a == INF ? -1 : a, b2 == INF ? -1 : b2, L.chg[(y / 8) * L.tw + x / 8] == INF ? -1 : L.chg[(y / 8) * L.tw + x / 8]);
            }
            WSNAP = L.own;
        }
        if (getenv("CHECK")) { // find regions that are inconsistent but were not marked dirty
            vector<char> isd(N, 0); for (int s : dirty) isd[s] = 1;
            int found = 0;
            for (int s : L.order) {
                if (L.angles[s] == NOTDEF || isd[s] || !L.active[s]) continue;
                // save state
                vector<int> own_save = L.own; vector<int> chg_save = L.chg; auto fp_save = L.fp[s]; char act = L.active[s];
                int b0 = L.bx0[s], b1 = L.by0[s], b2 = L.bx1[s], b3 = L.by1[s];
                vector<int> before = L.fp[s]; sort(before.begin(), before.end());
                L.execute(s, prec, p, min_reg_size, reg, Lst);
                vector<int> after = L.fp[s]; sort(after.begin(), after.end());
                L.own = own_save; L.chg = chg_save; L.fp[s] = fp_save; L.active[s] = act; L.bx0[s] = b0; L.by0[s] = b1; L.bx1[s] = b2; L.by1[s] = b3;
                if (before != after && found < 3) {
                    printf("  !! round %d: seed %d inconsistent but not dirty (fp %zu -> %zu), bbox %d..%d x %d..%d, tiles:", round, s, before.size(), after.size(), b0, b2, b1, b3);
                    for (int ty = b1 / 8; ty <= b3 / 8; ty++) for (int tx = b0 / 8; tx <= b2 / 8; tx++) printf(" %d", L.chg[ty * L.tw + tx] == INF ? -1 : L.chg[ty * L.tw + tx]);
                    printf("\n");
                    for (int q : after) if (!binary_search(before.begin(), before.end(), q)) printf("     gained %d (own now %d)\n", q, L.own[q]);
                    for (int q : before) if (!binary_search(after.begin(), after.end(), q)) printf("     lost %d\n", q);
                    found++;
                }
            }
        }
        fill(L.chg.begin(), L.chg.end(), INF);
        printf("round %d: executed %zu touched %ld wrong %ld next dirty %zu\n", round, nd, L.touched, wrong, dirty.size());
        if (dirty.empty()) break;
    }
    printf("total executions %ld\n", total_exec);
    // diagnostic: which regions are inconsistent at the fixed point?  re-execute each one in rank order and compare footprints
    int shown = 0;
    for (int s2 : L.order) {
        if (L.angles[s2] == NOTDEF) continue;
        if (!(L.active[s2] || L.own[s2] == INF || !L.fp[s2].empty())) continue;
        vector<int> before = L.fp[s2]; sort(before.begin(), before.end());
        char was = L.active[s2];
        L.execute(s2, prec, p, min_reg_size, reg, Lst);
        vector<int> after = L.fp[s2]; sort(after.begin(), after.end());
        if (before != after && shown < 6) {
            for (int q : after) if (!binary_search(before.begin(), before.end(), q)) printf("   gained %d\n", q);
            for (int q : before) if (!binary_search(after.begin(), after.end(), q)) printf("   lost %d\n", q);
            printf("inconsistent seed %d (x=%d y=%d) was_active=%d: footprint %zu -> %zu; own_true[s]=%d\n", s2, s2 % w, s2 / w, (int)was, before.size(), after.size(), own_true[s2]);
            shown++;
        }
    }
    long wrong = 0; for (int q = 0; q < N; q++) if (L.own[q] != own_true[q]) wrong++;
    printf("after one forced sweep in rank order: wrong %ld\n", wrong);
}
