// ba_cr.hip -- the reduced camera system of the object BA solved by nested dissection of the camera chain (block cyclic reduction), the
// parallel replacement of the two serial band-Cholesky chains of ba.hip (LinearSolverEigen's role, vendored g2o
// solvers/linear_solver_eigen.h:94-124 behind block_solver.hpp:354-486; any exact factorisation of the same SPD system serves).
//
// The band matrix (block (i, k) non-zero only for |i - k| <= Bc cameras) is block TRIDIAGONAL in super-blocks of Bc cameras (NB = 6 Bc
// unknowns).  Level l eliminates the super-blocks i = 2^l (2m + 1) all at once, one workgroup each: their neighbours a = i - 2^l and
// b = i + 2^l survive to the next level, so after ceil(log2 M) levels only super-block 0 is left (M = 100 super-blocks for 1000 key frames
// at Bc = 10: 7 levels + the root instead of two chains of 495 columns).  For an eliminated block with diagonal D, couplings L (to a) and
// R (to b) and right-hand side r:
//      D = G G^T,   Xa = G^-1 L,   Xb = G^-1 R^T,   y = G^-1 r,   Ginv = G^-1
//      D_a -= Xa^T Xa,   D_b -= Xb^T Xb,   new coupling (b, a) = -Xb^T Xa,   r_a -= Xa^T y,   r_b -= Xb^T y
// and on the way back  x_i = Ginv^T (y - Xa x_a - Xb x_b).
//
// Kernel ba_cr_eliminate<BC>: 4 NB + 1 threads, thread j holds COLUMN j of the panel [D | L | R^T | r | I] in registers (NB doubles).  Step k:
// the owner of column k publishes its multipliers (column k below the pivot, divided by it) in LDS, one barrier, every other thread
// subtracts multiplier x its own row-k value -- an LDL^T elimination whose rows are scaled by 1/sqrt(pivot) at the end, which is the
// Cholesky factor and G^-1 applied to everything right of it.  One broadcast LDS read per multiply-add, one barrier per pivot.  The three
// NB x NB products run on the matrix cores (v_mfma_f64_16x16x4_f64 tiles over the scaled panel staged in LDS).  A block reads its
// diagonal as the assembled D minus every update earlier levels left for it (two per level), so there is no separate "apply" pass:
// levels + 1 launches forward, levels + 1 back.
#include "ba_cr.h"

#include <algorithm>
#include <cmath>

namespace {
typedef double cr_v4d __attribute__((ext_vector_type(4)));

struct CrView { // device buffers; every per-node array is indexed by the super-block
    int C, Bc, Bt, M, NB;       // Bc: cameras per super-block; Bt: half bandwidth of the stored band (Bt <= Bc)
    const double *bandA, *brhs; // assembled band (C columns x (Bc+1) blocks of 36) and right-hand side (6C)
    double *D, *E, *ET, *r;     // M x NB^2 (dense symmetric), (M-1) x NB^2 (rows of super k+1, columns of super k) and its transpose, M x NB
    double *Ginv, *XaT, *XbT, *y; // per eliminated node: G^-1 (lower, [k][c]), (G^-1 L)^T and (G^-1 R^T)^T ([c][k]), G^-1 r
    double *accL, *accR, *accrL, *accrR, *V, *VT; // per node: updates accumulated from its right / left eliminated neighbours (Xa^T Xa, Xb^T Xb; Xa^T y + Xb^T y), and Xb^T Xa
    double *x;                  // solution, 6 per camera (padded to M NB)
    const double *zero, *one;   // a 0.0 and a 1.0 in device memory (branch-free panel loads)
    int *status;
    int *done; int epoch;    // the chained way back: done[i] == epoch once x of node i is in memory (this solve)
    unsigned long long *dbg; // CUBESLAM_CR_PROF: wall-clock stamps of block 0 at the phase boundaries of every level (100 MHz)
};

__global__ void __launch_bounds__(256) ba_cr_assemble(CrView W) {
    const int NB = W.NB, nb2 = NB * NB;
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const int bs = (W.Bt + 1) * 36;
    if (t < (long)W.M * nb2) {
        const int k = (int)(t / nb2), e = (int)(t % nb2), r = e / NB, c = e % NB, ci = k * W.Bc + r / 6, cj = k * W.Bc + c / 6;
        double v;
        if (ci >= W.C || cj >= W.C) v = r == c ? 1.0 : 0.0; // padding of the last super-block
        else if (ci >= cj) v = ci - cj <= W.Bt ? W.bandA[(long)cj * bs + (ci - cj) * 36 + (r % 6) * 6 + c % 6] : 0.0;
        else v = cj - ci <= W.Bt ? W.bandA[(long)ci * bs + (cj - ci) * 36 + (c % 6) * 6 + r % 6] : 0.0;
        W.D[t] = v;
        if (k + 1 < W.M) { // E_k: rows of super k+1, columns of super k
            const int ei = (k + 1) * W.Bc + r / 6, d = ei - cj;
            const double ev = (ei < W.C && d <= W.Bt) ? W.bandA[(long)cj * bs + d * 36 + (r % 6) * 6 + c % 6] : 0.0;
            W.E[t] = ev; W.ET[(long)k * nb2 + c * NB + r] = ev;
        }
    }
    if (t < (long)W.M * NB) { const int cam = (int)(t / 6); W.r[t] = cam < W.C ? W.brhs[t] : 0.0; }
}

__device__ __forceinline__ double cr_rcp(double p) { // 1 / p: hardware estimate + two Newton steps (|error| below 1 ulp for the pivots seen here)
    double r = __builtin_amdgcn_rcp(p);
    r = fma(fma(-p, r, 1.0), r, r);
    r = fma(fma(-p, r, 1.0), r, r);
    return r;
}

// one workgroup per eliminated super-block of the level (blockIdx.x = m, node i = step (2m + 1); root: node 0 alone)
template <int BC> __global__ void __launch_bounds__(256) ba_cr_eliminate(CrView W, int level, int root) {
    constexpr int NB = 6 * BC, NBP = (NB + 15) / 16 * 16, NCOL = 4 * NB + 1, NT = 256, LDW = 2 * NBP + 1;
    constexpr int TR = 8, TC = 32, RPT = (NB + TR - 1) / TR, CPT = (NCOL + TC - 1) / TC; // thread (tr, tc) owns rows tr + TR i, columns tc + TC ci
    extern __shared__ double cr_sh[];
    double *cbuf = cr_sh;                   // 2 x 2 x TR RPT: columns k, k+1 of the panel (double-buffered)
    double *vbuf = cbuf + 4 * TR * RPT;     // 2 x 2 x TC CPT: rows k, k+1
    double *piv = vbuf + 4 * TC * CPT;      // NB pivots
    double *Wx = piv + NB;                  // NBP x LDW: scaled [Xa | Xb] rows k, then y in column 2 NBP
    const int tid = threadIdx.x, nb2 = NB * NB, tc = tid & (TC - 1), tr = tid / TC;
    const int step = 1 << level, i = root ? 0 : step * (2 * (int)blockIdx.x + 1);
    const int a = root ? -1 : i - step, b = (root || i + step >= W.M) ? -1 : i + step;
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 0] = wall_clock64();
    // ---- this thread's elements of the panel [D | L | R^T | r | I] (NB rows): every global read runs along a row
    double P[RPT][CPT];
    // every element is  s1 * p1[o1] - p2[o2] - p3[o3]  with the pointers chosen per region and aimed at a zero word where a term is absent:
    // no branch around a load, so all of a thread's requests are in flight together
    const double *Lsrc = a < 0 ? nullptr : (level == 0 ? W.E + (long)a * nb2 : W.V + (long)(i - step / 2) * nb2);   // [row of i][column of a]
    const double *Rsrc = b < 0 ? nullptr : (level == 0 ? W.ET + (long)i * nb2 : W.VT + (long)(i + step / 2) * nb2); // transposed: [row of i][column of b]
    const double sgn = level == 0 ? 1.0 : -1.0;
    const double *p1[RPT][CPT], *p2[RPT][CPT], *p3[RPT][CPT];
    double s1[RPT][CPT];
#pragma unroll
    for (int ri = 0; ri < RPT; ri++) {
        const int row = tr + TR * ri;
#pragma unroll
        for (int ci = 0; ci < CPT; ci++) {
            const int col = tc + TC * ci;
            const double *q1 = W.zero, *q2 = W.zero, *q3 = W.zero;
            double sg = 1.0;
            if (row < NB && col < NCOL) {
                if (col < NB) { const long o = (long)i * nb2 + row * NB + col; q1 = W.D + o; q2 = W.accL + o; q3 = W.accR + o; }
                else if (col < 2 * NB) { if (Lsrc) { q1 = Lsrc + row * NB + col - NB; sg = sgn; } }
                else if (col < 3 * NB) { if (Rsrc) { q1 = Rsrc + row * NB + col - 2 * NB; sg = sgn; } }
                else if (col == 3 * NB) { const long o = (long)i * NB + row; q1 = W.r + o; q2 = W.accrL + o; q3 = W.accrR + o; }
                else if (col - 3 * NB - 1 == row) q1 = W.one;
            } else if (row >= NB && col == row) q1 = W.one; // (padding rows never pivot: unit diagonal keeps them inert)
            p1[ri][ci] = q1; p2[ri][ci] = q2; p3[ri][ci] = q3; s1[ri][ci] = sg;
        }
    }
#pragma unroll
    for (int ri = 0; ri < RPT; ri++)
#pragma unroll
        for (int ci = 0; ci < CPT; ci++) P[ri][ci] = s1[ri][ci] * *p1[ri][ci] - *p2[ri][ci] - *p3[ri][ci];
    __syncthreads();
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 1] = wall_clock64();
    // ---- elimination (LDL^T, rows scaled afterwards): step k publishes row k and column k of the running Schur complement, one barrier, then the
    // rank-one update of every element below row k.  8 + 8 values read per thread and step for 64 multiply-adds.
    // Two pivots per barrier: rows / columns k and k+1 are published as they stand before step k; every thread derives row k+1 and column k+1
    // after step k from them (l = P[k+1][k] / p1) and applies both rank-one updates at once.
    bool fail = false;
#pragma unroll
    for (int ki = 0; ki < RPT; ki++) {       // k = TR ki + kr: the register row ki and the register column ki / (TC / TR) are compile-time
        constexpr int CPR = TC / TR;
        const int kci = ki / CPR;
#pragma unroll 1
        for (int kr = 0; kr < TR; kr += 2) {
            const int k = TR * ki + kr, kc = (ki % CPR) * TR + kr, buf = (kr >> 1) & 1;
            if (k >= NB) break;
            double *vb = vbuf + buf * 2 * TC * CPT, *cb = cbuf + buf * 2 * TR * RPT;
            if (tr == kr || tr == kr + 1) {
                double *dst = vb + (tr - kr) * TC * CPT;
#pragma unroll
                for (int ci = 0; ci < CPT; ci++) dst[tc + TC * ci] = P[ki][ci];
            }
            if (tc == kc || tc == kc + 1) {
                double *dst = cb + (tc - kc) * TR * RPT;
#pragma unroll
                for (int ri = 0; ri < RPT; ri++) dst[tr + TR * ri] = P[ri][kci];
            }
            __syncthreads();
            const double p1 = vb[k], a12 = vb[k + 1], a22 = vb[TC * CPT + k + 1]; // P[k][k], P[k][k+1], P[k+1][k+1]
            const double rp1 = cr_rcp(p1), l = a12 * rp1, p2 = fma(-l, a12, a22), rp2 = cr_rcp(p2);
            if (!(p1 > 0) || !(p2 > 0)) fail = true;
            double v1[CPT], v2[CPT], c1[RPT], c2[RPT];
#pragma unroll
            for (int ci = 0; ci < CPT; ci++) { const double r1 = vb[tc + TC * ci], r2 = vb[TC * CPT + tc + TC * ci]; v1[ci] = r1 * rp1; v2[ci] = fma(-l, r1, r2) * rp2; }
#pragma unroll
            for (int ri = ki; ri < RPT; ri++) { // (register rows below ki hold rows < TR ki <= k: finished)
                const double q1 = cb[tr + TR * ri], q2 = cb[TR * RPT + tr + TR * ri]; const int row = tr + TR * ri;
                c1[ri] = row > k ? q1 : 0.0; c2[ri] = row > k + 1 ? fma(-l, q1, q2) : 0.0;
            }
#pragma unroll
            for (int ri = ki; ri < RPT; ri++)
#pragma unroll
                for (int ci = 0; ci < CPT; ci++) P[ri][ci] = fma(-c2[ri], v2[ci], fma(-c1[ri], v1[ci], P[ri][ci]));
            if (tid == 0) { piv[k] = p1; piv[k + 1] = p2; }
        }
    }
    if (fail && tid == 0) *W.status = 1;
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 2] = wall_clock64();
    // ---- scale the rows by 1 / sqrt(pivot): [G^T | Xa | Xb | y | Ginv]; stage Xa, Xb, y in LDS for the products; Ginv and y go out (streaming stores:
    // only the way back reads them)
    for (int t = tid; t < NBP * LDW; t += NT) Wx[t] = 0.0;
    __syncthreads();
    if (tid < NB) piv[tid] = 1.0 / sqrt(piv[tid]);
    __syncthreads();
#pragma unroll
    for (int ri = 0; ri < RPT; ri++) {
        const int row = tr + TR * ri;
        if (row >= NB) continue;
        const double sc = piv[row];
#pragma unroll
        for (int ci = 0; ci < CPT; ci++) {
            const int col = tc + TC * ci;
            if (col < NB || col >= NCOL) continue;
            const double v = P[ri][ci] * sc;
            if (col < 2 * NB) Wx[row * LDW + col - NB] = v;
            else if (col < 3 * NB) Wx[row * LDW + NBP + col - 2 * NB] = v;
            else if (col == 3 * NB) { Wx[row * LDW + 2 * NBP] = v; __builtin_nontemporal_store(v, &W.y[(long)i * NB + row]); }
            else __builtin_nontemporal_store(v, &W.Ginv[(long)i * nb2 + row * NB + col - 3 * NB - 1]);
        }
    }
    __syncthreads();
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 3] = wall_clock64();
    if (root) return;
    // ---- the way back reads Xa, Xb by columns: stored transposed ([c][k]), coalesced, from the staged copy
    for (int t = tid; t < 2 * nb2; t += NT) {
        const int m = t >= nb2, e = t - m * nb2, c = e / NB, k = e - c * NB;
        __builtin_nontemporal_store(Wx[k * LDW + m * NBP + c], &(m ? W.XbT : W.XaT)[(long)i * nb2 + e]);
    }
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 4] = wall_clock64();
    // ---- r_a -= Xa^T y, r_b -= Xb^T y (accumulated: one producer per array, target and level)
    if (tid < 2 * NB) {
        const int m = tid >= NB, c = tid - m * NB, tgt = m ? b : a;
        if (tgt >= 0) {
            double *dst = (m ? W.accrR : W.accrL) + (long)tgt * NB + c;
            const double was = *dst;
            double v = 0, v2 = 0;
#pragma unroll 6
            for (int k = 0; k < NB; k += 2) { v += Wx[k * LDW + m * NBP + c] * Wx[k * LDW + 2 * NBP]; v2 += Wx[(k + 1) * LDW + m * NBP + c] * Wx[(k + 1) * LDW + 2 * NBP]; }
            *dst = was + (v + v2);
        }
    }
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 5] = wall_clock64();
    // ---- D_a -= Xa^T Xa, D_b -= Xb^T Xb (accumulated in accL[a] / accR[b]), V = Xb^T Xa (and its transpose) on the matrix cores: 16x16 tiles, K = NBP
    // A operand: lane l holds A[row = l & 15][k = l >> 4]; B operand: B[k = l >> 4][col = l & 15]; result reg g: row = (l >> 4) + 4 g, col = l & 15
    constexpr int TL = NBP / 16, NTILE = 3 * TL * TL, NW = NT / 64, TPW = (NTILE + NW - 1) / NW;
    const int wave = tid >> 6, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    constexpr int TG = 4; // tiles per group: their running sums are requested together, before the group's products
    for (int u0 = 0; u0 < TPW; u0 += TG) {
        double old[TG][4];
#pragma unroll
        for (int u = 0; u < TG; u++) {
            const int t = wave + (u0 + u) * NW, prod = t / (TL * TL), tt = t % (TL * TL), tr2 = tt / TL, tc2 = tt % TL;
            const bool on = t < NTILE && prod < 2 && (prod == 0 ? a >= 0 : b >= 0);
            const double *src = prod == 0 ? W.accL + (long)max(a, 0) * nb2 : W.accR + (long)max(b, 0) * nb2;
#pragma unroll
            for (int g = 0; g < 4; g++) { const int row = tr2 * 16 + lk + 4 * g, cl = tc2 * 16 + lr; old[u][g] = (on && row < NB && cl < NB) ? src[row * NB + cl] : 0.0; }
        }
#pragma unroll
        for (int u = 0; u < TG; u++) {
            const int t = wave + (u0 + u) * NW, prod = t / (TL * TL), tt = t % (TL * TL), tr2 = tt / TL, tc2 = tt % TL;
            if (t >= NTILE || (prod == 0 && a < 0) || (prod >= 1 && b < 0)) continue; // (V needs both neighbours; without b it is never read)
            const int lo = prod == 0 ? 0 : NBP, ro = prod == 1 ? NBP : 0; // out[c1][c2] = sum_k Left[k][c1] Right[k][c2]:  Xa,Xa   Xb,Xb   Xb,Xa
            double *out = prod == 0 ? W.accL + (long)a * nb2 : prod == 1 ? W.accR + (long)b * nb2 : W.V + (long)i * nb2;
            cr_v4d acc = {0, 0, 0, 0};
#pragma unroll
            for (int k0 = 0; k0 < NBP; k0 += 4) {
                const double av = Wx[(k0 + lk) * LDW + lo + tr2 * 16 + lr], bv = Wx[(k0 + lk) * LDW + ro + tc2 * 16 + lr];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int row = tr2 * 16 + lk + 4 * g, cl = tc2 * 16 + lr;
                if (row < NB && cl < NB) { out[row * NB + cl] = old[u][g] + acc[g]; if (prod == 2) W.VT[(long)i * nb2 + cl * NB + row] = acc[g]; }
            }
        }
    }
    __syncthreads();
    if (W.dbg && blockIdx.x == 0 && tid == 0) W.dbg[(root ? 31 : level) * 8 + 6] = wall_clock64();
}

// x_i = Ginv^T (y - Xa x_a - Xb x_b) for the nodes eliminated at `level` (root: node 0).  Everything that does not depend on the neighbours'
// solutions is requested first.
template <int BC> __global__ void __launch_bounds__(64 * ((6 * BC + 63) / 64)) ba_cr_back(CrView W, int level, int root) {
    constexpr int NB = 6 * BC;
    __shared__ double tv[NB], xn[2 * NB];
    const int tid = threadIdx.x, nb2 = NB * NB;
    const int step = 1 << level, i = root ? 0 : step * (2 * (int)blockIdx.x + 1);
    const int a = root ? -1 : i - step, b = (root || i + step >= W.M) ? -1 : i + step;
    const int t = min(tid, NB - 1);
    double xa_col[NB], xb_col[NB], g_col[NB];
    const double *XA = W.XaT + (long)i * nb2 + t, *XB = W.XbT + (long)i * nb2 + t, *G = W.Ginv + (long)i * nb2 + t;
    const double y = W.y[(long)i * NB + t];
#pragma unroll
    for (int c = 0; c < NB; c++) { xa_col[c] = a >= 0 ? XA[c * NB] : 0.0; xb_col[c] = b >= 0 ? XB[c * NB] : 0.0; g_col[c] = c >= t ? G[(long)c * NB] : 0.0; }
    if (tid < NB) { xn[tid] = a >= 0 ? W.x[(long)a * NB + tid] : 0.0; xn[NB + tid] = b >= 0 ? W.x[(long)b * NB + tid] : 0.0; }
    __syncthreads();
    double v = y, v2 = 0;
#pragma unroll
    for (int c = 0; c < NB; c++) { v -= xa_col[c] * xn[c]; v2 -= xb_col[c] * xn[NB + c]; }
    if (tid < NB) tv[tid] = v + v2;
    __syncthreads();
    double x0 = 0, x1 = 0;
#pragma unroll
    for (int k = 0; k < NB; k += 2) { x0 += g_col[k] * tv[k]; x1 += g_col[k + 1] * tv[k + 1]; } // Ginv is lower triangular (zeros loaded above the diagonal)
    if (tid < NB) W.x[(long)i * NB + tid] = x0 + x1;
}

// The whole way back as ONE launch: a workgroup per node, in dependency order (root, then the levels from the top: a node's two neighbours were eliminated later
// than it, so their workgroups have smaller indices and never wait for a larger one -- no deadlock whatever the GPU holds at a time).  A node requests everything
// of its own first, then waits for done[a] / done[b] to show this solve's epoch, reads the neighbours' solutions past the caches (another XCD's L2 may have
// written them), and publishes its own behind an agent-scope fence.  Nine dependent launches of ~11 us become one whose levels cost a flag round trip each.
template <int BC> __global__ void __launch_bounds__(64 * ((6 * BC + 63) / 64)) ba_cr_back_chain(CrView W, int levels) {
    constexpr int NB = 6 * BC;
    __shared__ double tv[NB], xn[2 * NB];
    const int tid = threadIdx.x, nb2 = NB * NB;
    int level = levels, m = 0, i = 0;
    bool root = blockIdx.x == 0;
    if (!root) {
        int left = (int)blockIdx.x - 1;
        for (level = levels - 1; level >= 0; level--) {
            const int step = 1 << level, n = (W.M - step + 2 * step - 1) / (2 * step);
            if (left < n) break;
            left -= n;
        }
        m = left; i = (1 << level) * (2 * m + 1);
    }
    const int step = root ? 0 : 1 << level;
    const int a = root ? -1 : i - step, b = (root || i + step >= W.M) ? -1 : i + step;
    const int t = min(tid, NB - 1);
    double xa_col[NB], xb_col[NB], g_col[NB];
    const double *XA = W.XaT + (long)i * nb2 + t, *XB = W.XbT + (long)i * nb2 + t, *G = W.Ginv + (long)i * nb2 + t;
    const double y = W.y[(long)i * NB + t];
#pragma unroll
    for (int c = 0; c < NB; c++) { xa_col[c] = a >= 0 ? XA[c * NB] : 0.0; xb_col[c] = b >= 0 ? XB[c * NB] : 0.0; g_col[c] = c >= t ? G[(long)c * NB] : 0.0; }
    // the hand-over follows the write-through recipe (MI355X_MICROARCH.md, "handoff-flag"): the producer's solution goes out as sc1 (agent-scope relaxed atomic) stores, every
    // storing wave drains them (s_waitcnt vmcnt(0)), one barrier, then ONE relaxed agent-scope flag store; the consumer polls that flag with relaxed loads and reads the
    // solution with sc1 loads, which are served past its L1 and its XCD's L2 copy -- no buffer_wbl2 / buffer_inv per level (a __threadfence() is ~3.5 us, an acquire poll
    // invalidates the L1 on every turn: the way back was 66 us for eight levels of flag round trips)
    if (tid == 0) {
        // (a node only waits for nodes with SMALLER workgroup indices, and the dispatcher hands workgroups out in index order, so a waiting workgroup never holds a slot its
        //  producer needs; should that ever not hold -- the programming model does not promise it -- the spin is bounded by the wall clock (100 MHz: 0.5 s), then the solve
        //  is flagged as failed (status 2) instead of hanging the stream)
        const unsigned long long t_give_up = wall_clock64() + 50000000ull;
        bool late = false;
        if (a >= 0) while (__hip_atomic_load(&W.done[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != W.epoch && !(late = wall_clock64() > t_give_up)) __builtin_amdgcn_s_sleep(1);
        if (b >= 0) while (__hip_atomic_load(&W.done[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != W.epoch && !(late = late || wall_clock64() > t_give_up)) __builtin_amdgcn_s_sleep(1);
        if (late) *W.status = 2;
    }
    __syncthreads();
    if (tid < NB) {
        xn[tid] = a >= 0 ? __hip_atomic_load(&W.x[(long)a * NB + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        xn[NB + tid] = b >= 0 ? __hip_atomic_load(&W.x[(long)b * NB + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
    }
    __syncthreads();
    double v = y, v2 = 0;
#pragma unroll
    for (int c = 0; c < NB; c++) { v -= xa_col[c] * xn[c]; v2 -= xb_col[c] * xn[NB + c]; }
    if (tid < NB) tv[tid] = v + v2;
    __syncthreads();
    double x0 = 0, x1 = 0;
#pragma unroll
    for (int k = 0; k < NB; k += 2) { x0 += g_col[k] * tv[k]; x1 += g_col[k + 1] * tv[k + 1]; }
    if (tid < NB) __hip_atomic_store(&W.x[(long)i * NB + tid], x0 + x1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The drained-store hand-over relies on gfx9-family behaviour (stores counted by vmcnt, sc1 stores written through to memory-side coherence): it is what this library
    // is built for (gfx942 / gfx950).  Any other target -- the Makefile's ARCH can be overridden -- and -DCR_BACK_FENCE take the memory model's own release fence.
#if defined(CR_BACK_FENCE) || !(defined(__gfx942__) || defined(__gfx950__))
    __threadfence();
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // this wave's write-through stores have left
#endif
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&W.done[i], W.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int BC> int cr_run(cs_ctx *ctx, const CrView &W) {
    constexpr int NB = 6 * BC, NBP = (NB + 15) / 16 * 16, NCOL = 4 * NB + 1, NT = 256, LDW = 2 * NBP + 1, RPT = (NB + 7) / 8, CPT = (NCOL + 31) / 32;
    const size_t lds = sizeof(double) * (4 * 8 * (size_t)RPT + 4 * 32 * (size_t)CPT + (size_t)NB + (size_t)NBP * LDW);
    CS_HIP(ctx, hipFuncSetAttribute((const void *)ba_cr_eliminate<BC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); // per device, so on every call (a cheap host-side setting)
    int levels = 0;
    while ((1 << levels) < W.M) levels++;
    for (int l = 0; l < levels; l++) {
        const int step = 1 << l, n = (W.M - step + 2 * step - 1) / (2 * step); // nodes step (2m + 1) < M
        if (n > 0) CS_LAUNCH(ctx, "ba_cr_eliminate", ba_cr_eliminate<BC>, dim3(n), dim3(NT), lds, W, l, 0);
    }
    CS_LAUNCH(ctx, "ba_cr_eliminate", ba_cr_eliminate<BC>, dim3(1), dim3(NT), lds, W, levels, 1);
    constexpr int BT = 64 * ((NB + 63) / 64);
    static const bool chain = !(getenv("CUBESLAM_CR_CHAIN") && atoi(getenv("CUBESLAM_CR_CHAIN")) == 0);
    if (chain) { // (CUBESLAM_CR_CHAIN=0: a launch per level, the cross-check)
        CS_LAUNCH(ctx, "ba_cr_back", ba_cr_back_chain<BC>, dim3(W.M), dim3(BT), 0, W, levels);
        return CS_OK;
    }
    CS_LAUNCH(ctx, "ba_cr_back", ba_cr_back<BC>, dim3(1), dim3(BT), 0, W, levels, 1);
    for (int l = levels - 1; l >= 0; l--) {
        const int step = 1 << l, n = (W.M - step + 2 * step - 1) / (2 * step);
        if (n > 0) CS_LAUNCH(ctx, "ba_cr_back", ba_cr_back<BC>, dim3(n), dim3(BT), 0, W, l, 0);
    }
    return CS_OK;
}
} // namespace

struct BaCr {
    int C = 0, Bc = 0, BCT = 0, M = 0;
    double *buf = nullptr;
    int *done = nullptr; int epoch = 0; // ba_cr_back_chain's flags (zeroed once; a solve's epoch never repeats)
};
void ba_cr_destroy(BaCr *w) { if (w) { if (w->buf) hipFree(w->buf); if (w->done) hipFree(w->done); delete w; } }
// the elimination kernel of the super-block size the solve would instantiate keeps NBP * LDW + ... doubles in dynamic LDS (77 KB at 10 cameras per
// super-block): a device that does not offer that much per workgroup takes the band solver instead
bool ba_cr_supported(int C, int Bc) {
    if (!(Bc >= 1 && Bc <= 10 && C >= 4 * Bc)) return false;
    const int BCT = Bc <= 2 ? 2 : Bc <= 4 ? 4 : Bc <= 6 ? 6 : Bc <= 8 ? 8 : 10;
    const int NB = 6 * BCT, NBP = (NB + 15) / 16 * 16, NCOL = 4 * NB + 1, LDW = 2 * NBP + 1, RPT = (NB + 7) / 8, CPT = (NCOL + 31) / 32;
    const size_t lds = sizeof(double) * (4 * 8 * (size_t)RPT + 4 * 32 * (size_t)CPT + (size_t)NB + (size_t)NBP * LDW);
    int dev = 0, cap = 0, optin = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cap, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
    if (hipDeviceGetAttribute(&optin, hipDeviceAttributeSharedMemPerBlockOptin, dev) != hipSuccess) optin = 0; // what hipFuncSetAttribute can raise a kernel to
    return lds <= (size_t)std::max(cap, optin);
}

// Solves the band system (A, rhs) of C cameras, half bandwidth Bc, into x (6C); *status is set to 1 on a non-positive pivot.
int ba_cr_solve(cs_ctx *ctx, BaCr **handle, int C, int Bc, const double *d_bandA, const double *d_brhs, double *d_x, int *d_status) {
    if (!ba_cr_supported(C, Bc)) return CS_ERR_BAD_ARG;
    const int BCT = Bc <= 2 ? 2 : Bc <= 4 ? 4 : Bc <= 6 ? 6 : Bc <= 8 ? 8 : 10; // instantiated super-block sizes (a wider super-block only adds explicit zeros)
    const int NB = 6 * BCT, M = (C + BCT - 1) / BCT;
    BaCr *w = *handle;
    const size_t nb2 = (size_t)NB * NB, per = 10 * nb2 + 5 * (size_t)NB; // D E ET Ginv XaT XbT accL accR V VT;  r y accrL accrR x
    if (!w || w->C != C || w->Bc != Bc) {
        ba_cr_destroy(w);
        w = new BaCr(); *handle = w;
        w->C = C; w->Bc = Bc; w->BCT = BCT; w->M = M;
        int rc = cs_dalloc(ctx, &w->buf, per * (size_t)M + 64); if (rc) return rc;
        rc = cs_dalloc(ctx, &w->done, (size_t)M + 1); if (rc) return rc;
        CS_HIP(ctx, hipMemsetAsync(w->done, 0, sizeof(int) * ((size_t)M + 1), ctx->stream));
    }
    CrView V;
    V.C = C; V.Bc = BCT; V.Bt = Bc; V.M = M; V.NB = NB; V.bandA = d_bandA; V.brhs = d_brhs; V.status = d_status;
    V.done = w->done; V.epoch = ++w->epoch;
    double *p = w->buf;
    V.accL = p; p += nb2 * M; V.accR = p; p += nb2 * M; V.accrL = p; p += (size_t)NB * M; V.accrR = p; p += (size_t)NB * M; // zeroed every solve
    const size_t zero_bytes = sizeof(double) * (size_t)(p - w->buf);
    V.D = p; p += nb2 * M; V.E = p; p += nb2 * M; V.ET = p; p += nb2 * M; V.VT = p; p += nb2 * M; V.Ginv = p; p += nb2 * M; V.XaT = p; p += nb2 * M; V.XbT = p; p += nb2 * M; V.V = p; p += nb2 * M;
    V.r = p; p += (size_t)NB * M; V.y = p; p += (size_t)NB * M; V.x = p;
    CS_HIP(ctx, hipMemsetAsync(w->buf, 0, zero_bytes, ctx->stream));
    static unsigned long long *d_dbg = nullptr;
    const bool prof = getenv("CUBESLAM_CR_PROF") != nullptr;
    if (prof && !d_dbg) { hipMalloc((void **)&d_dbg, 32 * 8 * sizeof(unsigned long long)); }
    if (prof) hipMemsetAsync(d_dbg, 0, 32 * 8 * sizeof(unsigned long long), ctx->stream);
    V.dbg = prof ? d_dbg : nullptr;
    V.zero = V.accrR; // (zeroed above; the first entry belongs to node 0, which has no left neighbour and is never written)
    V.one = p + (size_t)NB * M;
    { static const double one = 1.0; CS_HIP(ctx, hipMemcpyAsync((void *)V.one, &one, sizeof one, hipMemcpyHostToDevice, ctx->stream)); }
    CS_LAUNCH(ctx, "ba_cr_assemble", ba_cr_assemble, dim3((unsigned)(((size_t)M * nb2 + 255) / 256)), dim3(256), 0, V);
    int rc = CS_OK;
    switch (BCT) {
    case 2: rc = cr_run<2>(ctx, V); break;
    case 4: rc = cr_run<4>(ctx, V); break;
    case 6: rc = cr_run<6>(ctx, V); break;
    case 8: rc = cr_run<8>(ctx, V); break;
    default: rc = cr_run<10>(ctx, V); break;
    }
    if (rc) return rc;
    CS_HIP(ctx, hipMemcpyAsync(d_x, V.x, sizeof(double) * (size_t)C * 6, hipMemcpyDeviceToDevice, ctx->stream));
    if (prof) {
        unsigned long long h[32 * 8];
        hipMemcpy(h, d_dbg, sizeof h, hipMemcpyDeviceToHost);
        for (int l = 0; l < 32; l++) if (h[l * 8]) { fprintf(stderr, "[cr prof] level %2d:", l); for (int q = 1; q < 7; q++) fprintf(stderr, " %6.2f", h[l * 8 + q] ? (h[l * 8 + q] - h[l * 8 + q - 1]) / 100.0 : 0.0); fprintf(stderr, " us (load, eliminate, scale+stage, transposed store, rhs, products)\n"); }
    }
    return CS_OK;
}
