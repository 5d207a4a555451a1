// CPU emulation of the list-free formulation of DistributeOctTree that the device kernel orb_quadtree uses (a split pass =
// [children of the last expandable node n4..n1, ..., of the first] ++ [non-expandable nodes in order]; second phase with a push
// stack + deletion marks), checked against the sequential host restatement (cs_orb_host::QuadTree) on random inputs.
#include "cube_slam_amd/csrc/orb_quadtree.h"
#include <cstdio>
#include <cstdlib>
#include <random>
using cs_orb_host::Cand;

struct PNode { int x0, y0, x1, y1, begin, end; bool no_more; };

static void par_distribute(const Cand *K, int n, int minX, int maxX, int minY, int maxY, int N, std::vector<int> &result) {
    result.clear();
    const int nIni = (int)std::round(static_cast<float>(maxX - minX) / (maxY - minY));
    if (nIni < 1 || n == 0) return;
    const float hX = static_cast<float>(maxX - minX) / nIni;
    std::vector<PNode> nodes;
    std::vector<int> perm(n), tmp(n);
    // roots: stable bucket by x / hX
    std::vector<int> cnt(nIni + 1, 0), bk(n);
    for (int i = 0; i < n; i++) { int b = (int)(K[i].x / hX); if (b >= nIni) b = nIni - 1; bk[i] = b; cnt[b + 1]++; }
    for (int b = 0; b < nIni; b++) cnt[b + 1] += cnt[b];
    { std::vector<int> pos(cnt.begin(), cnt.end() - 1); for (int i = 0; i < n; i++) perm[pos[bk[i]]++] = i; }
    std::vector<int> list; // node ids in list order
    for (int b = 0; b < nIni; b++) {
        PNode nd{(int)(hX * static_cast<float>(b)), 0, (int)(hX * static_cast<float>(b + 1)), maxY - minY, cnt[b], cnt[b + 1], false};
        if (nd.end == nd.begin) continue;
        nd.no_more = (nd.end - nd.begin) == 1;
        nodes.push_back(nd); list.push_back((int)nodes.size() - 1);
    }
    auto split = [&](int id, int cid[4], int csz[4]) { // creates children (ids in n1..n4 order, -1 if empty)
        const PNode nd = nodes[id];
        const int halfX = (int)std::ceil(static_cast<float>(nd.x1 - nd.x0) / 2), halfY = (int)std::ceil(static_cast<float>(nd.y1 - nd.y0) / 2);
        const int mx = nd.x0 + halfX, my = nd.y0 + halfY;
        int c[4] = {0, 0, 0, 0};
        std::vector<int> q(nd.end - nd.begin);
        for (int p = nd.begin; p < nd.end; p++) { const Cand &k = K[perm[p]]; q[p - nd.begin] = (k.x < mx) ? ((k.y < my) ? 0 : 2) : ((k.y < my) ? 1 : 3); c[q[p - nd.begin]]++; }
        int off[4] = {nd.begin, nd.begin + c[0], nd.begin + c[0] + c[1], nd.begin + c[0] + c[1] + c[2]}, pos[4] = {off[0], off[1], off[2], off[3]};
        for (int p = nd.begin; p < nd.end; p++) tmp[pos[q[p - nd.begin]]++] = perm[p];
        for (int p = nd.begin; p < nd.end; p++) perm[p] = tmp[p];
        const int bx[4][4] = {{nd.x0, nd.y0, mx, my}, {mx, nd.y0, nd.x1, my}, {nd.x0, my, mx, nd.y1}, {mx, my, nd.x1, nd.y1}};
        for (int k = 0; k < 4; k++) {
            cid[k] = -1; csz[k] = c[k];
            if (!c[k]) continue;
            nodes.push_back(PNode{bx[k][0], bx[k][1], bx[k][2], bx[k][3], off[k], off[k] + c[k], c[k] == 1});
            cid[k] = (int)nodes.size() - 1;
        }
    };
    std::vector<std::pair<int, int>> vsize; // (size, id) in creation order
    bool finish = false;
    while (!finish) {
        const int prev_size = (int)list.size();
        // ---- phase-1 pass, data-parallel form: new list = [children of the last expandable node reversed, ..., of the first reversed] ++ [non-expandable nodes in order]
        std::vector<int> expandable, keep;
        for (int id : list) (nodes[id].no_more ? keep : expandable).push_back(id);
        vsize.clear();
        int n_to_expand = 0;
        std::vector<std::vector<int>> kids(expandable.size());
        for (size_t e = 0; e < expandable.size(); e++) { // creation order = list order of the parents, n1..n4
            int cid[4], csz[4];
            split(expandable[e], cid, csz);
            for (int k = 0; k < 4; k++) if (cid[k] >= 0) { kids[e].push_back(cid[k]); if (csz[k] > 1) { n_to_expand++; vsize.push_back({csz[k], cid[k]}); } }
        }
        std::vector<int> nl;
        for (int e = (int)expandable.size() - 1; e >= 0; e--) for (int k = (int)kids[e].size() - 1; k >= 0; k--) nl.push_back(kids[e][k]);
        nl.insert(nl.end(), keep.begin(), keep.end());
        list.swap(nl);
        const int size = (int)list.size();
        if (size >= N || size == prev_size) finish = true;
        else if (size + n_to_expand * 3 > N) {
            // ---- phase 2: sequential, largest first.  list as: front (pushed children, newest first) ++ base with deletions
            std::vector<int> front; // push order; list order = reversed(front) ++ base(alive)
            std::vector<char> dead(nodes.size() * 8 + 64, 0);
            int cur = size;
            while (!finish) {
                const int ps = cur;
                auto prev = vsize;
                vsize.clear();
                std::sort(prev.begin(), prev.end());
                for (int j = (int)prev.size() - 1; j >= 0; j--) {
                    int cid[4], csz[4];
                    const int id = prev[j].second;
                    split(id, cid, csz);
                    if (dead.size() < nodes.size() + 8) dead.resize(nodes.size() * 2 + 64, 0);
                    for (int k = 0; k < 4; k++) if (cid[k] >= 0) { front.push_back(cid[k]); cur++; if (csz[k] > 1) vsize.push_back({csz[k], cid[k]}); }
                    dead[id] = 1; cur--;
                    if (cur >= N) break;
                }
                if (cur >= N || cur == ps) finish = true;
            }
            std::vector<int> fl;
            for (int k = (int)front.size() - 1; k >= 0; k--) if (!dead[front[k]]) fl.push_back(front[k]);
            for (int id : list) if (!dead[id]) fl.push_back(id);
            list.swap(fl);
        }
    }
    for (int id : list) {
        const PNode &nd = nodes[id];
        int best = perm[nd.begin]; float mxr = K[best].response;
        for (int p = nd.begin + 1; p < nd.end; p++) if (K[perm[p]].response > mxr) { best = perm[p]; mxr = K[best].response; }
        result.push_back(best);
    }
}

int main() {
    std::mt19937 rng(7);
    cs_orb_host::QuadTree qt;
    int bad = 0;
    for (int trial = 0; trial < 3000; trial++) {
        const int W = 100 + rng() % 1200, H = 60 + rng() % 500;
        const int n = rng() % 3 == 0 ? rng() % 40 : rng() % 6000;
        const int N = 1 + rng() % 600;
        std::vector<Cand> K(n);
        const bool clustered = rng() % 2;
        for (auto &k : K) {
            if (clustered && rng() % 3) { k.x = (float)(W / 3 + rng() % std::max(1, W / 10)); k.y = (float)(H / 2 + rng() % std::max(1, H / 10)); }
            else { k.x = (float)(rng() % W); k.y = (float)(rng() % H); }
            k.response = (float)(rng() % 50);
        }
        std::vector<int> a, b;
        qt.distribute(K.data(), n, 16, 16 + W, 16, 16 + H, N, a);
        par_distribute(K.data(), n, 16, 16 + W, 16, 16 + H, N, b);
        if (a != b) { if (bad < 5) printf("MISMATCH trial %d: n=%d N=%d W=%d H=%d  sizes %zu %zu\n", trial, n, N, W, H, a.size(), b.size()); bad++; }
    }
    printf("%s (%d mismatches)\n", bad ? "FAIL" : "all equal", bad);
    return bad != 0;
}
