// jacobi_sim.cpp -- experiment (round 2): region growing as a speculative fixed point.  Every defined pixel runs its region transaction
// (grow, rectangle, refine) against the owner map of the previous round (a pixel is taken iff a LOWER-ranked seed claimed it); the claims
// are merged with min().  Prints rounds, executions and touched pixels per round, and the distance to the sequential result (0 at the
// fixed point: the scheme is exact).  g++ -O2 -std=c++17 -o sim jacobi_sim.cpp && ./sim frame.raw 640 480   
#include "../../oracle/lsd_oracle.cpp"
#include <cstdio>
#include <climits>
using namespace std;
struct Sim : LSD {
    // returns footprint (pixels marked used at the end of transaction) for seed adx given base `used`
    void transaction(int adx, vector<RegionPoint> &reg, vector<int> &touched, double prec, double p, int min_reg_size) {
        int reg_size; double reg_angle;
        region_grow(adx % w, adx / w, reg, reg_size, reg_angle, prec);
        for (int i = 0; i < reg_size; i++) touched.push_back(reg[i].x + reg[i].y * w);
        if (reg_size < min_reg_size) return;
        Rect rec;
        region2rect(reg, reg_size, reg_angle, prec, p, rec);
        // refine may release + regrow
        double density = double(reg_size) / (dist(rec.x1, rec.y1, rec.x2, rec.y2) * rec.width);
        if (density >= DENSITY_TH) return;
        refine(reg, reg_size, reg_angle, prec, p, rec, DENSITY_TH);
        for (int i = 0; i < reg_size; i++) touched.push_back(reg[i].x + reg[i].y * w);
    }
};
int main(int argc, char **argv) {
    // input: raw gray file W H
    int W = atoi(argv[2]), H = atoi(argv[3]);
    vector<uint8_t> gray((size_t)W * H);
    FILE *f = fopen(argv[1], "rb"); fread(gray.data(), 1, gray.size(), f); fclose(f);
    Sim L; L.prepare(gray.data(), W, H);
    const int w = L.w, h = L.h, N = w * h;
    const double prec = PI * L.ANG_TH / 180, p = L.ANG_TH / 180;
    L.LOG_NT = 5 * (log10(double(w)) + log10(double(h))) / 2 + log10(11.0);
    const int min_reg_size = int(-L.LOG_NT / log10(p));
    // sequential ground truth: owner map
    vector<int> own_true(N, INT_MAX);
    {
        L.used.assign(N, 0);
        vector<RegionPoint> reg(N); vector<int> touched;
        long nseeds = 0, npx = 0; int maxreg = 0;
        for (int adx : L.order) {
            if (L.used[adx] == 0 && L.angles[adx] != NOTDEF) {
                touched.clear();
                vector<uint8_t> before; // not needed
                L.transaction(adx, reg, touched, prec, p, min_reg_size);
                int cnt = 0;
                for (int q : touched) if (L.used[q] == 1 && own_true[q] == INT_MAX) { own_true[q] = adx; cnt++; }
                nseeds++; npx += cnt; if (cnt > maxreg) maxreg = cnt;
            }
        }
        long ndef = 0; for (int i = 0; i < N; i++) if (L.angles[i] != NOTDEF) ndef++;
        printf("scaled %dx%d defined %ld true seeds %ld footprint px %ld max region %d\n", w, h, ndef, nseeds, npx, maxreg);
    }
    // Jacobi
    vector<int> own_prev(N, INT_MAX), own_next(N);
    vector<RegionPoint> reg(N); vector<int> touched;
    for (int round = 1; round <= 400; round++) {
        fill(own_next.begin(), own_next.end(), INT_MAX);
        // base used state for seed s: own_prev[q] < s.  sweep s increasing; pixels become used when s passes own_prev[q]
        vector<pair<int,int>> byowner; byowner.reserve(N);
        for (int q = 0; q < N; q++) if (own_prev[q] != INT_MAX) byowner.push_back({own_prev[q], q});
        sort(byowner.begin(), byowner.end());
        size_t bp = 0;
        L.used.assign(N, 0);
        long execs = 0, work = 0;
        for (int s : L.order) {
            while (bp < byowner.size() && byowner[bp].first < s) { L.used[byowner[bp].second] = 1; bp++; }
            if (L.angles[s] == NOTDEF) continue;
            if (L.used[s]) continue; // own_prev[s] < s
            touched.clear();
            L.transaction(s, reg, touched, prec, p, min_reg_size);
            execs++; work += touched.size();
            for (int q : touched) {
                if (L.used[q] == 1 && !(own_prev[q] < s)) { if (s < own_next[q]) own_next[q] = s; }
                L.used[q] = (own_prev[q] < s) ? 1 : 0; // restore base (for later seeds the sweep pointer re-adds what is needed)
            }
            // pixels with own_prev[q] in [.. s] : sweep handles < s'; own_prev[q]==s itself becomes used for s' > s via the sweep
        }
        long diff = 0, wrong = 0;
        for (int q = 0; q < N; q++) { if (own_next[q] != own_prev[q]) diff++; if (own_next[q] != own_true[q]) wrong++; }
        printf("round %d: executions %ld touched %ld changed px %ld wrong vs sequential %ld\n", round, execs, work, diff, wrong);
        own_prev = own_next;
        if (diff == 0) break;
    }
}
