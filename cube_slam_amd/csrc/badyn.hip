// Dynamic-object bundle adjustment on the GPU: the graph Optimizer::LocalBACameraPointObjectsDynamic (orb_object_slam/src/Optimizer.cc:
// 1537-2573) hands to g2o's BlockSolverX + LinearSolverDense + Levenberg.  Per-item math lives in badyn_math.h; this file holds the kernels,
// the dense solve of the reduced pose system and the host-side LM control loop (OptimizationAlgorithmLevenberg::solve :61-164).
//
//   badyn_errors      thread per edge: computeError + the edge's share of activeRobustChi2, block partial sums
//   badyn_linearize   thread per edge: Jacobians (analytic for the reprojection edges, central differences for the CubeSLAM ones),
//                     robustified J^T W J / J^T W e accumulated with fp64 atomics into the dense pose system, the landmark blocks and
//                     the edge's own pose-landmark slot(s)
//   badyn_schur_init  S = Hpp + lambda I, bs = bp
//   badyn_dinv / badyn_bd / badyn_schur_blocks / badyn_schur_rhs   the Schur complement as gathers: 3x3 inverses per landmark, B D^-1 per
//                     slot, one wave per 6x6 target block summing over the slot pairs the host listed for it, one thread per reduced
//                     right-hand-side row (a first version with fp64 atomics from a thread per landmark took 21 ms on 2 200 landmarks:
//                     all landmarks of a camera pair hit the same 36 words)
//   badyn_chol_diag / badyn_chol_panel / badyn_chol_update / badyn_chol_tri   blocked right-looking Cholesky of S (32 columns per step:
//                     one workgroup factors the 32x32 pivot block in LDS, the rows of the panel are solved against it one per thread, then
//                     32x32 tiles of the trailing matrix are updated by independent workgroups), then one workgroup for both triangular solves (a single-workgroup
//                     column-by-column version -- badyn_chol_solve, CUBESLAM_BADYN_CHOL=simple -- took 26 ms at 840 unknowns: every
//                     trailing update waits for its own global load)
//   badyn_backsub     thread per landmark, badyn_update thread per vertex (oplus), badyn_diag gathers diag(H) for computeLambdaInit
// The windows this runs on are small (10-30 key frames, a few object tracks): the pose system has a few hundred to a few thousand scalars,
// so everything is latency-bound; the dense factorisation is the only super-linear step and stays on one CU.
#include "common.h"
#include "badyn_math.h"
#include "badyn_lists.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <vector>

namespace {

constexpr int DYN_MAX_NP = 4000; // two LDS vectors of NP doubles in badyn_chol_solve

__device__ inline void dyn_block_sum_store(double v, double *partials) {
    __shared__ double s[4];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}
__global__ void __launch_bounds__(256) badyn_errors(DynG G, int n_edges, double *partials) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    dyn_block_sum_store(e < n_edges ? dyn_error_item(G, e) : 0.0, partials);
}
template <int ONLY> __global__ void __launch_bounds__(64) badyn_linearize(DynG G, int e0, int n) { // edges [e0, e0 + n) of the concatenated list
    const int e = blockIdx.x * 64 + threadIdx.x;
    if (e < n) dyn_lin_item<ONLY>(G, e0 + e);
}
// ---- deterministic build: the per-edge pieces (DynG::stage) added per target in edge order, one lane per element, eight loads in flight
__global__ void __launch_bounds__(64) badyn_gather_blocks(DynG G) { // one wave per pose x pose block
    const int t = blockIdx.x, lane = threadIdx.x, a = lane / 6, c = lane % 6;
    const int lo = G.pb_lo[t], hi = G.pb_hi[t], dr = G.pb_dim[t] & 0xff, dc = G.pb_dim[t] >> 8;
    if (lane >= 36 || a >= dr || c >= dc) return;
    double acc = 0;
    const int q0 = G.pb_start[t], q1 = G.pb_start[t + 1];
    for (int q = q0; q < q1; q += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int s = G.pb_src[min(q + u, q1 - 1)], tr = s & 1; // (edge * 6 + pair) * 2 + transposed
            v[u] = G.stage[(long)(s >> 1) / 6 * DYN_STAGE + 18 + ((s >> 1) % 6) * 36 + (tr ? c * 6 + a : a * 6 + c)];
        }
#pragma unroll
        for (int u = 0; u < 8; u++) if (q + u < q1) acc += v[u];
    }
    G.Hpp[(long)(lo + a) * G.NP + hi + c] = acc;
    if (lo != hi) G.Hpp[(long)(hi + c) * G.NP + lo + a] = acc;
}
__global__ void __launch_bounds__(64) badyn_gather_grad(DynG G) { // a lane per row of a pose vertex' gradient, 10 vertices per workgroup
    const int t = blockIdx.x * 10 + threadIdx.x / 6, a = threadIdx.x % 6;
    if (threadIdx.x >= 60 || t >= G.n_pg || a >= G.pg_dim[t]) return;
    double acc = 0;
    for (int q = G.pg_start[t]; q < G.pg_start[t + 1]; q++) { const int s = G.pg_src[q]; acc += G.stage[(long)(s / 3) * DYN_STAGE + (s % 3) * 6 + a]; }
    G.bp[G.pg_off[t] + a] = acc;
}
__global__ void __launch_bounds__(64) badyn_gather_lm(DynG G) { // a thread per landmark: 3x3 block and gradient
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= G.L) return;
    double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
    for (int q = G.lg_start[l]; q < G.lg_start[l + 1]; q++) {
        const int s = G.lg_src[q], e = s / 3, i = s % 3;
        const double *st = G.stage + (long)e * DYN_STAGE;
        const double *hb = st + 18 + dyn_pair_index(i, i) * 36;
        for (int r = 0; r < 3; r++) { g[r] += st[i * 6 + r]; for (int c = 0; c < 3; c++) H[r * 3 + c] += hb[r * 6 + c]; }
    }
    for (int k = 0; k < 9; k++) G.Hll[(long)l * 9 + k] = H[k];
    for (int k = 0; k < 3; k++) G.bl[(long)l * 3 + k] = g[k];
}
__global__ void __launch_bounds__(256) badyn_schur_init(DynG G, double lambda) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x, n2 = (long)G.NP * G.NP;
    if (i < n2) { const int r = (int)(i / G.NP), c = (int)(i % G.NP); G.S[i] = G.Hpp[i] + (r == c ? lambda : 0.0); }
    if (i < G.NP) G.bs[i] = G.bp[i];
}
__global__ void __launch_bounds__(64) badyn_dinv(DynG G, double lambda) {
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li < G.L) dyn_dinv_item(G, li, lambda);
}
__global__ void __launch_bounds__(64) badyn_bd(DynG G, int n_slots) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s < n_slots) dyn_bd_item(G, s);
}
__global__ void __launch_bounds__(256) badyn_schur_blocks(DynG G) { // one workgroup per target block: 7 groups of 36 lanes share its pair list
    __shared__ double part[7][36];
    const int g = threadIdx.x / 36, e = threadIdx.x % 36;
    if (g < 7) part[g][e] = dyn_schur_block_partial(G, blockIdx.x, e, g, 7);
    __syncthreads();
    if (threadIdx.x < 36) {
        double acc = 0;
        for (int k = 0; k < 7; k++) acc += part[k][e];
        dyn_schur_block_store(G, blockIdx.x, e, acc);
    }
}
__global__ void __launch_bounds__(64) badyn_schur_rhs(DynG G) { // one wave per pose vertex: 10 groups of 6 lanes share its slot list
    __shared__ double part[10][6];
    const int g = threadIdx.x / 6, a = threadIdx.x % 6;
    if (g < 10) part[g][a] = dyn_rhs_partial(G, blockIdx.x, a, g, 10);
    __syncthreads();
    if (threadIdx.x < 6) {
        double acc = 0;
        for (int k = 0; k < 10; k++) acc += part[k][a];
        dyn_rhs_store(G, blockIdx.x, a, acc);
    }
}
__global__ void __launch_bounds__(64) badyn_backsub(DynG G) {
    const int li = blockIdx.x * 64 + threadIdx.x;
    if (li < G.L) dyn_backsub_item(G, li);
}
__global__ void __launch_bounds__(64) badyn_update(DynG G, int n_vertices) {
    const int v = blockIdx.x * 64 + threadIdx.x;
    if (v < n_vertices) dyn_update_item(G, v);
}
__global__ void __launch_bounds__(256) badyn_diag(DynG G, double *diag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < G.NP) diag[i] = G.Hpp[(long)i * G.NP + i];
    else if (i < G.NP + 3 * G.L) { const int l = (i - G.NP) / 3, k = (i - G.NP) % 3; diag[i] = G.Hll[(long)l * 9 + k * 4]; }
}

// A (n x n, row-major, lower triangle used) -> L in place; x: right-hand side -> solution.  Column j is scaled into LDS, the trailing rows are
// updated one row per wave (contiguous in k), and the forward substitution y_j = b_j / L_jj, b_i -= L_ij y_j rides on the same column; the
// backward substitution walks the rows of L upwards: x_i = y_i / L_ii, y_k -= L_ik x_i (k < i).
__global__ void __launch_bounds__(1024) badyn_chol_solve(int n, double *A, double *x, int *status) {
    extern __shared__ double lds[];
    double *col = lds, *rhs = lds + n;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    for (int i = tid; i < n; i += T) rhs[i] = x[i];
    __syncthreads();
    for (int j = 0; j < n; j++) {
        const double d = A[(long)j * n + j];
        if (!(d > 0)) { if (tid == 0) *status = 1; return; } // uniform: every thread reads the same pivot
        const double r = sqrt(d);
        for (int i = j + tid; i < n; i += T) {
            const double v = (i == j) ? r : A[(long)i * n + j] / r;
            A[(long)i * n + j] = v;
            col[i] = v;
        }
        __syncthreads();
        const double yj = rhs[j] / r;
        for (int i = j + 1 + wave; i < n; i += nw) {
            const double lij = col[i];
            double *row = A + (long)i * n;
            for (int k = j + 1 + lane; k <= i; k += 64) row[k] -= lij * col[k];
            if (lane == 0) rhs[i] -= lij * yj;
        }
        __syncthreads();
        if (tid == 0) rhs[j] = yj;
    }
    __syncthreads();
    for (int i = n - 1; i >= 0; i--) {
        const double *row = A + (long)i * n;
        const double xi = rhs[i] / row[i];
        __syncthreads();
        for (int k = tid; k < i; k += T) rhs[k] -= row[k] * xi;
        if (tid == 0) rhs[i] = xi;
        __syncthreads();
    }
    for (int i = tid; i < n; i += T) x[i] = rhs[i];
}


// ---- blocked Cholesky: A (n x n row-major, lower triangle) -> strictly-lower panels in place, pivot blocks in Dg (one 32x32 row-major
// block per step, the upper part zero).
constexpr int CB = 32;
// pivot block: one workgroup, one thread per element, three barriers per column (as a redundant prologue of every panel workgroup, 128
// threads looping over the elements, this took 60 us per step: the panel kernel's whole duration)
__global__ void __launch_bounds__(CB * CB) badyn_chol_diag(int n, int jb, const double *A, double *Dg, double *rd, int *status) {
    __shared__ double Ld[CB][CB + 1];
    const int tid = threadIdx.x, i = tid / CB, k = tid % CB, nb = min(CB, n - jb);
    Ld[i][k] = (i < nb && k <= i) ? A[(long)(jb + i) * n + jb + k] : (i == k ? 1.0 : 0.0); // a ragged last block is padded with the identity
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < CB; c++) {
        const double d = Ld[c][c];
        if (!(d > 0)) { if (tid == 0) *status = 1; return; } // uniform
        const double r = sqrt(d);
        __syncthreads();
        if (k == c && i >= c) Ld[i][c] = (i == c) ? r : Ld[i][c] / r;
        __syncthreads();
        if (k > c && k <= i) Ld[i][k] -= Ld[i][c] * Ld[k][c];
        __syncthreads();
    }
    Dg[(long)(jb / CB) * CB * CB + tid] = k <= i ? Ld[i][k] : 0.0;
    if (tid < CB) rd[(jb / CB) * CB + tid] = 1.0 / Ld[tid][tid];
}
constexpr int CP_T = 128; // rows (= threads) per panel workgroup
// the rows behind the pivot block: x L^T = a, right-looking, the thread's row in an LDS column (rolled loops: an unrolled register version
// spilled 620 VGPRs)
__global__ void __launch_bounds__(CP_T) badyn_chol_panel(int n, int jb, double *A, const double *Dg, const double *rd) {
    __shared__ double Ld[CB][CB + 1];
    __shared__ double xs[CB][CP_T];
    __shared__ double rdiag[CB];
    const int tid = threadIdx.x, nb = min(CB, n - jb);
    for (int e = tid; e < CB * CB; e += CP_T) Ld[e / CB][e % CB] = Dg[(long)(jb / CB) * CB * CB + e];
    if (tid < CB) rdiag[tid] = rd[(jb / CB) * CB + tid];
    __syncthreads();
    const int i = jb + nb + blockIdx.x * CP_T + tid;
    if (i >= n) return;
    double *row = A + (long)i * n + jb;
#pragma unroll 1
    for (int c = 0; c < nb; c++) xs[c][tid] = row[c];
#pragma unroll 1
    for (int t = 0; t < nb; t++) {
        const double xt = xs[t][tid] * rdiag[t];
        row[t] = xt;
#pragma unroll 1
        for (int c = t + 1; c < nb; c++) xs[c][tid] -= xt * Ld[c][t];
    }
}
// trailing update: tile (ti, tk), tk <= ti, of the rows / columns behind the panel: A_ik -= P_i P_k^T
__global__ void __launch_bounds__(256) badyn_chol_update(int n, int jb, double *A) {
    const int ti = blockIdx.y, tk = blockIdx.x;
    if (tk > ti) return;
    __shared__ double Pi[CB][CB + 1], Pk[CB][CB + 1];
    const int tid = threadIdx.x, nb = min(CB, n - jb), r0 = jb + nb + ti * CB, k0 = jb + nb + tk * CB;
    for (int e = tid; e < CB * CB; e += 256) {
        const int r = e / CB, c = e % CB;
        Pi[r][c] = (r0 + r < n && c < nb) ? A[(long)(r0 + r) * n + jb + c] : 0.0;
        Pk[r][c] = (k0 + r < n && c < nb) ? A[(long)(k0 + r) * n + jb + c] : 0.0;
    }
    __syncthreads();
    const int r = tid >> 3, c4 = (tid & 7) * 4;
    double acc[4] = {0, 0, 0, 0};
#pragma unroll 8
    for (int t = 0; t < CB; t++) {
        const double pi = Pi[r][t];
        for (int q = 0; q < 4; q++) acc[q] += pi * Pk[c4 + q][t];
    }
    const int i = r0 + r;
    if (i >= n) return;
    for (int q = 0; q < 4; q++) { const int k = k0 + c4 + q; if (k <= i) A[(long)i * n + k] -= acc[q]; }
}
// both triangular solves with the factor left by the two kernels above: L y = b (blocks ascending), L^T x = y (descending)
__global__ void __launch_bounds__(1024) badyn_chol_tri(int n, const double *A, const double *Dg, double *x) {
    extern __shared__ double lds[];
    double *rhs = lds;               // n
    double *Ld = lds + n;            // CB x (CB + 1)
    const int tid = threadIdx.x, T = blockDim.x;
    for (int i = tid; i < n; i += T) rhs[i] = x[i];
    const int nblk = (n + CB - 1) / CB;
    for (int bi = 0; bi < nblk; bi++) {
        const int jb = bi * CB, nb = min(CB, n - jb);
        __syncthreads();
        for (int e = tid; e < CB * CB; e += T) Ld[(e / CB) * (CB + 1) + e % CB] = Dg[(long)bi * CB * CB + e];
        __syncthreads();
        if (tid < 64) { // pivot block: lane c owns y_c, the solved values travel by shuffle
            double acc = tid < nb ? rhs[jb + tid] : 0.0;
            const double rd = tid < nb ? 1.0 / Ld[tid * (CB + 1) + tid] : 0.0;
            for (int t = 0; t < nb; t++) {
                const double yt = __shfl(acc, t) * __shfl(rd, t);
                if (tid == t) acc = yt;
                else if (tid > t && tid < nb) acc -= Ld[tid * (CB + 1) + t] * yt;
            }
            if (tid < nb) rhs[jb + tid] = acc;
        }
        __syncthreads();
        for (int i = jb + nb + tid; i < n; i += T) {
            const double *row = A + (long)i * n + jb;
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0; // four short chains instead of one of 32 dependent FMAs
            int c = 0;
            for (; c + 3 < nb; c += 4) { p0 += row[c] * rhs[jb + c]; p1 += row[c + 1] * rhs[jb + c + 1]; p2 += row[c + 2] * rhs[jb + c + 2]; p3 += row[c + 3] * rhs[jb + c + 3]; }
            for (; c < nb; c++) p0 += row[c] * rhs[jb + c];
            rhs[i] -= (p0 + p1) + (p2 + p3);
        }
    }
    for (int bi = nblk - 1; bi >= 0; bi--) {
        const int jb = bi * CB, nb = min(CB, n - jb);
        __syncthreads();
        for (int e = tid; e < CB * CB; e += T) Ld[(e / CB) * (CB + 1) + e % CB] = Dg[(long)bi * CB * CB + e];
        __syncthreads();
        if (tid < 64) {
            double acc = tid < nb ? rhs[jb + tid] : 0.0;
            const double rd = tid < nb ? 1.0 / Ld[tid * (CB + 1) + tid] : 0.0;
            for (int t = nb - 1; t >= 0; t--) {
                const double xt = __shfl(acc, t) * __shfl(rd, t);
                if (tid == t) acc = xt;
                else if (tid < t) acc -= Ld[t * (CB + 1) + tid] * xt; // L^T: column tid of row t
            }
            if (tid < nb) rhs[jb + tid] = acc;
        }
        __syncthreads();
        for (int k = tid; k < jb; k += T) {
            double p0 = 0, p1 = 0, p2 = 0, p3 = 0;
            int c = 0;
            for (; c + 3 < nb; c += 4) {
                p0 += A[(long)(jb + c) * n + k] * rhs[jb + c]; p1 += A[(long)(jb + c + 1) * n + k] * rhs[jb + c + 1];
                p2 += A[(long)(jb + c + 2) * n + k] * rhs[jb + c + 2]; p3 += A[(long)(jb + c + 3) * n + k] * rhs[jb + c + 3];
            }
            for (; c < nb; c++) p0 += A[(long)(jb + c) * n + k] * rhs[jb + c];
            rhs[k] -= (p0 + p1) + (p2 + p3);
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += T) x[i] = rhs[i];
}

} // namespace

struct cs_ba_dyn {
    DynG G;
    const volatile unsigned char *stop8 = nullptr; // the caller's bool (setForceStopFlag)
    std::vector<void *> bufs;
    int n_edges = 0, n_vertices = 0, max_part = 0, n_slots = 0, simple_chol = 0;
    double *d_Dg = nullptr, *d_rd = nullptr;
    size_t state_doubles = 0;
    double *d_state = nullptr, *d_bak = nullptr, *d_partials = nullptr, *d_diag = nullptr;
    int *d_status = nullptr;
    std::vector<double> h_partials, h_x, h_b;
};

namespace {

template <class T> int dyn_upload(cs_ctx *ctx, cs_ba_dyn *b, T **d, const T *h, size_t n) {
    T *p = nullptr;
    int r = cs_dalloc(ctx, &p, n);
    if (r) return r;
    b->bufs.push_back(p);
    if (h && n) r = cs_h2d(ctx, p, h, n);
    else if (!h) { hipError_t e = hipMemsetAsync(p, 0, std::max<size_t>(n, 1) * sizeof(T), ctx->stream); if (e != hipSuccess) { ctx->err = hipGetErrorString(e); r = CS_ERR_HIP; } }
    *d = p;
    return r;
}
template <class T> int dyn_upload(cs_ctx *ctx, cs_ba_dyn *b, const T **d, const T *h, size_t n) { return dyn_upload(ctx, b, const_cast<T **>(d), h, n); }

int dyn_errors(cs_ctx *ctx, cs_ba_dyn *b, double *chi) {
    const int nb = (b->n_edges + 255) / 256;
    *chi = 0;
    if (nb == 0) return CS_OK;
    CS_LAUNCH(ctx, "badyn_errors", badyn_errors, dim3(nb), dim3(256), 0, b->G, b->n_edges, b->d_partials);
    b->h_partials.resize(nb);
    int r = cs_d2h(ctx, b->h_partials.data(), b->d_partials, (size_t)nb);
    if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double s = 0;
    for (int i = 0; i < nb; i++) s += b->h_partials[i];
    *chi = s;
    return CS_OK;
}
int dyn_build(cs_ctx *ctx, cs_ba_dyn *b) { // BlockSolver::buildSystem: one thread per edge writes its pieces, then they are added per target in edge order
    const DynG &G = b->G;
    const int n_rest = b->n_edges - G.n_obs - G.n_dobs; // motion, camera-object, point-object, local-point edges: few, numeric Jacobians
    if (G.n_obs) CS_LAUNCH(ctx, "badyn_linearize", badyn_linearize<0>, dim3((G.n_obs + 63) / 64), dim3(64), 0, G, 0, G.n_obs);
    if (G.n_dobs) CS_LAUNCH(ctx, "badyn_linearize_dyn", badyn_linearize<1>, dim3((G.n_dobs + 63) / 64), dim3(64), 0, G, G.n_obs, G.n_dobs);
    if (n_rest) CS_LAUNCH(ctx, "badyn_linearize_num", badyn_linearize<-1>, dim3((n_rest + 63) / 64), dim3(64), 0, G, G.n_obs + G.n_dobs, n_rest);
    // Hpp / bp / Hll / bl were zeroed once at creation: the targets below are the only words ever written, and every one of them is rewritten here
    if (G.n_pb) CS_LAUNCH(ctx, "badyn_gather_blocks", badyn_gather_blocks, dim3(G.n_pb), dim3(64), 0, G);
    if (G.n_pg) CS_LAUNCH(ctx, "badyn_gather_grad", badyn_gather_grad, dim3((G.n_pg + 9) / 10), dim3(64), 0, G);
    if (G.L) CS_LAUNCH(ctx, "badyn_gather_lm", badyn_gather_lm, dim3((G.L + 63) / 64), dim3(64), 0, G);
    return CS_OK;
}
int dyn_reduce(cs_ctx *ctx, cs_ba_dyn *b, double lambda) { // Schur complement of the marginalised points
    const DynG &G = b->G;
    const long n2 = std::max<long>((long)G.NP * G.NP, G.NP);
    if (n2 > 0) CS_LAUNCH(ctx, "badyn_schur_init", badyn_schur_init, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, G, lambda);
    if (G.L > 0) {
        CS_LAUNCH(ctx, "badyn_dinv", badyn_dinv, dim3((G.L + 63) / 64), dim3(64), 0, G, lambda);
        if (b->n_slots > 0) CS_LAUNCH(ctx, "badyn_bd", badyn_bd, dim3((b->n_slots + 63) / 64), dim3(64), 0, G, b->n_slots);
        if (G.n_blocks > 0) CS_LAUNCH(ctx, "badyn_schur_blocks", badyn_schur_blocks, dim3(G.n_blocks), dim3(256), 0, G);
        if (G.n_vtx > 0) CS_LAUNCH(ctx, "badyn_schur_rhs", badyn_schur_rhs, dim3(G.n_vtx), dim3(64), 0, G);
    }
    return CS_OK;
}
int dyn_solve(cs_ctx *ctx, cs_ba_dyn *b, double lambda) { // BlockSolver::solve: reduce, dense solve, back substitution; status on the device
    const DynG &G = b->G;
    CS_HIP(ctx, hipMemsetAsync(b->d_status, 0, sizeof(int), ctx->stream));
    int r = dyn_reduce(ctx, b, lambda);
    if (r) return r;
    if (G.NP > 0) {
        CS_HIP(ctx, hipMemcpyAsync(G.xp, G.bs, sizeof(double) * G.NP, hipMemcpyDeviceToDevice, ctx->stream));
        if (b->simple_chol) {
            CS_LAUNCH(ctx, "badyn_chol_solve", badyn_chol_solve, dim3(1), dim3(1024), sizeof(double) * 2 * (size_t)G.NP, G.NP, G.S, G.xp, b->d_status);
        } else {
            const int n = G.NP;
            for (int jb = 0; jb < n; jb += CB) {
                const int nb = std::min(CB, n - jb), m = n - jb - nb; // rows behind the panel
                CS_LAUNCH(ctx, "badyn_chol_diag", badyn_chol_diag, dim3(1), dim3(CB * CB), 0, n, jb, G.S, b->d_Dg, b->d_rd, b->d_status);
                if (m > 0) CS_LAUNCH(ctx, "badyn_chol_panel", badyn_chol_panel, dim3((m + CP_T - 1) / CP_T), dim3(CP_T), 0, n, jb, G.S, b->d_Dg, b->d_rd);
                if (m > 0) { const int T = (m + CB - 1) / CB; CS_LAUNCH(ctx, "badyn_chol_update", badyn_chol_update, dim3(T, T), dim3(256), 0, n, jb, G.S); }
            }
            CS_LAUNCH(ctx, "badyn_chol_tri", badyn_chol_tri, dim3(1), dim3(1024), sizeof(double) * ((size_t)n + CB * (CB + 1)), n, G.S, b->d_Dg, G.xp);
        }
    }
    if (G.L > 0) CS_LAUNCH(ctx, "badyn_backsub", badyn_backsub, dim3((G.L + 63) / 64), dim3(64), 0, G);
    return CS_OK;
}

} // namespace

extern "C" {

void cs_ba_dyn_destroy(cs_ctx *ctx, cs_ba_dyn *b) {
    if (!b) return;
    if (ctx) hipSetDevice(ctx->device);
    for (void *p : b->bufs) if (p) hipFree(p);
    delete b;
}

int cs_ba_dyn_create(cs_ctx *ctx, const cs_ba_dyn_problem *p, cs_ba_dyn **out) {
    if (!ctx || !p || !out || p->n_cams < 1 || p->n_objs < 0 || p->n_vels < 0 || p->n_points < 0 || p->n_dpoints < 0 || p->n_obs < 0 || p->n_dobs < 0 || p->n_mot < 0 ||
        p->n_cobs < 0 || p->n_pc < 0 || !p->cam_pose || !p->cam_fixed)
        return CS_ERR_BAD_ARG;
    if ((p->n_objs && (!p->obj_pose || !p->obj_scale || !p->obj_flags)) || (p->n_vels && !p->vel) || (p->n_points && !p->points) || (p->n_dpoints && !p->dpoints) ||
        (p->n_obs && (!p->obs_cam || !p->obs_point || !p->obs_uv || !p->obs_inv_sigma2)) ||
        (p->n_dobs && (!p->dobs_cam || !p->dobs_obj || !p->dobs_point || !p->dobs_uv || !p->dobs_inv_sigma2)) ||
        (p->n_mot && (!p->mot_from || !p->mot_to || !p->mot_vel || !p->mot_dt)) || (p->n_cobs && (!p->cobs_cam || !p->cobs_obj || !p->cobs_bbox || !p->cobs_info)) ||
        (p->n_pc && (!p->pc_obj || !p->pc_offsets || !p->pc_points)))
        return CS_ERR_BAD_ARG;
    auto in = [](int v, int n) { return v >= 0 && v < n; };
    for (int o = 0; o < p->n_obs; o++) if (!in(p->obs_cam[o], p->n_cams) || !in(p->obs_point[o], p->n_points)) return CS_ERR_BAD_ARG;
    for (int o = 0; o < p->n_dobs; o++) if (!in(p->dobs_cam[o], p->n_cams) || !in(p->dobs_obj[o], p->n_objs) || !in(p->dobs_point[o], p->n_dpoints)) return CS_ERR_BAD_ARG;
    for (int o = 0; o < p->n_mot; o++) if (!in(p->mot_from[o], p->n_objs) || !in(p->mot_to[o], p->n_objs) || !in(p->mot_vel[o], p->n_vels)) return CS_ERR_BAD_ARG;
    for (int o = 0; o < p->n_cobs; o++) if (!in(p->cobs_cam[o], p->n_cams) || !in(p->cobs_obj[o], p->n_objs)) return CS_ERR_BAD_ARG;
    for (int o = 0; o < p->n_pc; o++) if (!in(p->pc_obj[o], p->n_objs) || p->pc_offsets[o + 1] < p->pc_offsets[o]) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    cs_ba_dyn *b = new cs_ba_dyn();
    DynG &G = b->G;
    memset(&G, 0, sizeof(G));
    G.n_cams = p->n_cams; G.n_objs = p->n_objs; G.n_vels = p->n_vels; G.n_pts = p->n_points; G.n_dpts = p->n_dpoints; G.fix_points = p->fix_points ? 1 : 0;
    G.n_obs = p->n_obs; G.n_dobs = p->n_dobs; G.n_mot = p->n_mot; G.n_cobs = p->n_cobs; G.n_pc = p->n_pc;
    G.fx = p->fx; G.fy = p->fy; G.cx = p->cx; G.cy = p->cy; G.bf = p->bf; G.huber_mono = p->huber_mono; G.huber_stereo = p->huber_stereo; G.huber_dyn = p->huber_dyn;
    G.huber_obj = p->huber_obj; G.ulp_info = p->ulp_info; G.ulp_ratio = p->ulp_ratio; G.pc_ratio = p->pc_ratio;
    for (int k = 0; k < 9; k++) G.K[k] = p->K[k];
    for (int k = 0; k < 3; k++) { G.ulp_scale[k] = p->ulp_scale[k]; G.mot_info[k] = p->mot_info[k]; }
    DynLists X; // pose-system offsets, pose-landmark slots, Schur target blocks
    dyn_build_lists(p, X);
    const int NP = X.NP;
    if (NP > DYN_MAX_NP) { delete b; ctx->err = "dynamic BA: pose system larger than DYN_MAX_NP scalars"; return CS_ERR_BAD_ARG; }
    G.NP = NP; G.L = X.L;
    const int n_slots = b->n_slots = X.n_slots;
    G.n_blocks = (int)X.blk_ou.size(); G.n_vtx = (int)X.vtx_off.size();
    { const char *ce = getenv("CUBESLAM_BADYN_CHOL"); b->simple_chol = ce && !strcmp(ce, "simple"); }
    b->n_edges = p->n_obs + p->n_dobs + p->n_mot + p->n_cobs + p->n_pc + p->n_dpoints;
    b->n_vertices = p->n_cams + p->n_objs + p->n_vels + p->n_points + p->n_dpoints;
    b->max_part = std::max(1, (b->n_edges + 255) / 256);
#define D_(call) do { int r__ = (call); if (r__ != CS_OK) { cs_ba_dyn_destroy(ctx, b); return r__; } } while (0)
    // estimates: one buffer so that push() / pop() of the LM trial are one copy each
    const size_t o_cam = 0, o_obj = o_cam + (size_t)p->n_cams * 7, o_vel = o_obj + (size_t)p->n_objs * 7, o_pts = o_vel + (size_t)p->n_vels * 2,
                 o_dp = o_pts + (size_t)p->n_points * 3;
    b->state_doubles = o_dp + (size_t)p->n_dpoints * 3;
    std::vector<double> st(std::max<size_t>(b->state_doubles, 1), 0.0);
    for (int i = 0; i < p->n_cams; i++) { SE3 T = se3_load(p->cam_pose + (size_t)i * 7); normalize_rotation(T); se3_store(T, &st[o_cam + (size_t)i * 7]); } // SE3Quat(q, t) normalises
    for (int i = 0; i < p->n_objs; i++) { SE3 T = se3_load(p->obj_pose + (size_t)i * 7); normalize_rotation(T); se3_store(T, &st[o_obj + (size_t)i * 7]); }
    if (p->n_vels) memcpy(&st[o_vel], p->vel, sizeof(double) * (size_t)p->n_vels * 2);
    if (p->n_points) memcpy(&st[o_pts], p->points, sizeof(double) * (size_t)p->n_points * 3);
    if (p->n_dpoints) memcpy(&st[o_dp], p->dpoints, sizeof(double) * (size_t)p->n_dpoints * 3);
    D_(dyn_upload(ctx, b, &b->d_state, st.data(), st.size()));
    D_(dyn_upload(ctx, b, &b->d_bak, (const double *)nullptr, st.size()));
    G.cam = b->d_state + o_cam; G.obj = b->d_state + o_obj; G.vel = b->d_state + o_vel; G.pts = b->d_state + o_pts; G.dpts = b->d_state + o_dp;
    D_(dyn_upload(ctx, b, &G.obj_scale, p->obj_scale, (size_t)p->n_objs * 3)); D_(dyn_upload(ctx, b, &G.obj_flags, p->obj_flags, (size_t)p->n_objs));
    D_(dyn_upload(ctx, b, &G.cam_off, X.cam_off.data(), X.cam_off.size())); D_(dyn_upload(ctx, b, &G.obj_off, X.obj_off.data(), X.obj_off.size()));
    D_(dyn_upload(ctx, b, &G.vel_off, X.vel_off.data(), X.vel_off.size()));
    D_(dyn_upload(ctx, b, &G.o_cam, p->obs_cam, (size_t)p->n_obs)); D_(dyn_upload(ctx, b, &G.o_pt, p->obs_point, (size_t)p->n_obs));
    D_(dyn_upload(ctx, b, &G.o_uv, p->obs_uv, (size_t)p->n_obs * 2)); D_(dyn_upload(ctx, b, &G.o_w, p->obs_inv_sigma2, (size_t)p->n_obs));
    if (p->obs_ur && p->n_obs) D_(dyn_upload(ctx, b, &G.o_ur, p->obs_ur, (size_t)p->n_obs));
    if (p->obs_level && p->n_obs) D_(dyn_upload(ctx, b, &G.o_lvl, p->obs_level, (size_t)p->n_obs));
    D_(dyn_upload(ctx, b, &G.d_cam, p->dobs_cam, (size_t)p->n_dobs)); D_(dyn_upload(ctx, b, &G.d_obj, p->dobs_obj, (size_t)p->n_dobs));
    D_(dyn_upload(ctx, b, &G.d_pt, p->dobs_point, (size_t)p->n_dobs)); D_(dyn_upload(ctx, b, &G.d_uv, p->dobs_uv, (size_t)p->n_dobs * 2));
    D_(dyn_upload(ctx, b, &G.d_w, p->dobs_inv_sigma2, (size_t)p->n_dobs));
    if (p->dobs_level && p->n_dobs) D_(dyn_upload(ctx, b, &G.d_lvl, p->dobs_level, (size_t)p->n_dobs));
    D_(dyn_upload(ctx, b, &G.m_from, p->mot_from, (size_t)p->n_mot)); D_(dyn_upload(ctx, b, &G.m_to, p->mot_to, (size_t)p->n_mot));
    D_(dyn_upload(ctx, b, &G.m_vel, p->mot_vel, (size_t)p->n_mot)); D_(dyn_upload(ctx, b, &G.m_dt, p->mot_dt, (size_t)p->n_mot));
    D_(dyn_upload(ctx, b, &G.c_cam, p->cobs_cam, (size_t)p->n_cobs)); D_(dyn_upload(ctx, b, &G.c_obj, p->cobs_obj, (size_t)p->n_cobs));
    D_(dyn_upload(ctx, b, &G.c_bbox, p->cobs_bbox, (size_t)p->n_cobs * 4)); D_(dyn_upload(ctx, b, &G.c_info, p->cobs_info, (size_t)p->n_cobs * 4));
    if (p->cobs_level && p->n_cobs) D_(dyn_upload(ctx, b, &G.c_lvl, p->cobs_level, (size_t)p->n_cobs));
    D_(dyn_upload(ctx, b, &G.pc_obj, p->pc_obj, (size_t)p->n_pc));
    { const int zero2[2] = {0, 0}; D_(dyn_upload(ctx, b, &G.pc_off, p->n_pc ? p->pc_offsets : zero2, (size_t)p->n_pc + 1)); }
    D_(dyn_upload(ctx, b, &G.pc_pts, p->pc_points, p->n_pc ? (size_t)p->pc_offsets[p->n_pc] * 3 : 0));
    D_(dyn_upload(ctx, b, &G.e_obs, (const double *)nullptr, (size_t)p->n_obs * 3)); D_(dyn_upload(ctx, b, &G.e_dobs, (const double *)nullptr, (size_t)p->n_dobs * 2));
    D_(dyn_upload(ctx, b, &G.e_mot, (const double *)nullptr, (size_t)p->n_mot * 3)); D_(dyn_upload(ctx, b, &G.e_cobs, (const double *)nullptr, (size_t)p->n_cobs * 4));
    D_(dyn_upload(ctx, b, &G.e_pc, (const double *)nullptr, (size_t)p->n_pc * 3)); D_(dyn_upload(ctx, b, &G.e_ulp, (const double *)nullptr, (size_t)p->n_dpoints * 3));
    D_(dyn_upload(ctx, b, &G.Hpp, (const double *)nullptr, (size_t)NP * NP)); D_(dyn_upload(ctx, b, &G.S, (const double *)nullptr, (size_t)NP * NP));
    D_(dyn_upload(ctx, b, &G.bp, (const double *)nullptr, (size_t)NP)); D_(dyn_upload(ctx, b, &G.bs, (const double *)nullptr, (size_t)NP));
    D_(dyn_upload(ctx, b, &G.xp, (const double *)nullptr, (size_t)NP));
    D_(dyn_upload(ctx, b, &G.Hll, (const double *)nullptr, (size_t)G.L * 9)); D_(dyn_upload(ctx, b, &G.Dinv, (const double *)nullptr, (size_t)G.L * 9));
    D_(dyn_upload(ctx, b, &G.bl, (const double *)nullptr, (size_t)G.L * 3)); D_(dyn_upload(ctx, b, &G.xl, (const double *)nullptr, (size_t)G.L * 3));
    D_(dyn_upload(ctx, b, &G.Bslot, (const double *)nullptr, (size_t)n_slots * 18));
    D_(dyn_upload(ctx, b, &G.stage, (const double *)nullptr, (size_t)b->n_edges * DYN_STAGE));
    G.n_pb = (int)X.pb_lo.size(); G.n_pg = (int)X.pg_off.size();
    D_(dyn_upload(ctx, b, &G.pb_lo, X.pb_lo.data(), X.pb_lo.size())); D_(dyn_upload(ctx, b, &G.pb_hi, X.pb_hi.data(), X.pb_hi.size())); D_(dyn_upload(ctx, b, &G.pb_dim, X.pb_dim.data(), X.pb_dim.size()));
    D_(dyn_upload(ctx, b, &G.pb_start, X.pb_start.data(), X.pb_start.size())); D_(dyn_upload(ctx, b, &G.pb_src, X.pb_src.data(), X.pb_src.size()));
    D_(dyn_upload(ctx, b, &G.pg_off, X.pg_off.data(), X.pg_off.size())); D_(dyn_upload(ctx, b, &G.pg_dim, X.pg_dim.data(), X.pg_dim.size()));
    D_(dyn_upload(ctx, b, &G.pg_start, X.pg_start.data(), X.pg_start.size())); D_(dyn_upload(ctx, b, &G.pg_src, X.pg_src.data(), X.pg_src.size()));
    D_(dyn_upload(ctx, b, &G.lg_start, X.lg_start.data(), X.lg_start.size())); D_(dyn_upload(ctx, b, &G.lg_src, X.lg_src.data(), X.lg_src.size()));
    D_(dyn_upload(ctx, b, &G.slot_off, X.slot_off.data(), X.slot_off.size())); D_(dyn_upload(ctx, b, &G.lm_start, X.lm_start.data(), X.lm_start.size()));
    D_(dyn_upload(ctx, b, &G.lm_slots, X.lm_slots.data(), X.lm_slots.size())); D_(dyn_upload(ctx, b, &G.slot_lm, X.slot_lm.data(), X.slot_lm.size()));
    D_(dyn_upload(ctx, b, &G.BD, (const double *)nullptr, (size_t)n_slots * 18)); D_(dyn_upload(ctx, b, &G.bsub, (const double *)nullptr, (size_t)n_slots * 6));
    D_(dyn_upload(ctx, b, &G.blk_ou, X.blk_ou.data(), X.blk_ou.size())); D_(dyn_upload(ctx, b, &G.blk_ot, X.blk_ot.data(), X.blk_ot.size()));
    D_(dyn_upload(ctx, b, &G.blk_start, X.blk_start.data(), X.blk_start.size())); D_(dyn_upload(ctx, b, &G.pair_u, X.pair_u.data(), X.pair_u.size()));
    D_(dyn_upload(ctx, b, &G.pair_t, X.pair_t.data(), X.pair_t.size())); D_(dyn_upload(ctx, b, &G.vtx_off, X.vtx_off.data(), X.vtx_off.size()));
    D_(dyn_upload(ctx, b, &G.vtx_start, X.vtx_start.data(), X.vtx_start.size())); D_(dyn_upload(ctx, b, &G.vtx_slots, X.vtx_slots.data(), X.vtx_slots.size()));
    D_(dyn_upload(ctx, b, &b->d_Dg, (const double *)nullptr, (size_t)((NP + CB - 1) / CB) * CB * CB));
    D_(dyn_upload(ctx, b, &b->d_rd, (const double *)nullptr, (size_t)((NP + CB - 1) / CB) * CB));
    D_(dyn_upload(ctx, b, &b->d_partials, (const double *)nullptr, (size_t)b->max_part));
    D_(dyn_upload(ctx, b, &b->d_diag, (const double *)nullptr, (size_t)NP + 3 * (size_t)G.L));
    D_(dyn_upload(ctx, b, &b->d_status, (const int *)nullptr, 1));
#undef D_
    if (2 * (size_t)NP * sizeof(double) > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(badyn_chol_solve), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * (size_t)NP * sizeof(double)));
        if (e != hipSuccess) { ctx->err = hipGetErrorString(e); cs_ba_dyn_destroy(ctx, b); return CS_ERR_HIP; }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream); // the host vectors above go out of scope
    if (e != hipSuccess) { ctx->err = hipGetErrorString(e); cs_ba_dyn_destroy(ctx, b); return CS_ERR_HIP; }
    *out = b;
    return CS_OK;
}

int cs_ba_dyn_set_stop_flag_bool(cs_ba_dyn *b, const volatile unsigned char *flag) { if (!b) return CS_ERR_BAD_ARG; b->stop8 = flag; return CS_OK; }

int cs_ba_dyn_optimize(cs_ctx *ctx, cs_ba_dyn *b, int iterations, const volatile int *stop_flag, cs_ba_stats *st) {
    if (!ctx || !b || iterations < 0) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const DynG &G = b->G;
    cs_ba_stats S;
    memset(&S, 0, sizeof(S));
    double lambda = 0, ni = 2, currentChi = 0;
    int nBad = 0, r;
    bool accepted = false;
    const size_t nx = (size_t)G.NP + 3 * (size_t)G.L;
    auto terminate = [&]() { return (stop_flag && *stop_flag) || (b->stop8 && *b->stop8); };
    for (int it = 0; it < iterations && !terminate(); it++) { // OptimizationAlgorithmLevenberg::solve :61-164
        double tempChi;
        if (it == 0 || !accepted) { r = dyn_errors(ctx, b, &currentChi); if (r) return r; } // otherwise the residuals of the accepted trial are current
        const double iniChi = currentChi;
        if (it == 0) S.chi2_init = currentChi;
        r = dyn_build(ctx, b); if (r) return r;
        b->h_b.resize(std::max<size_t>(nx, 1));
        if (G.NP) { r = cs_d2h(ctx, b->h_b.data(), G.bp, (size_t)G.NP); if (r) return r; }
        if (G.L) { r = cs_d2h(ctx, b->h_b.data() + G.NP, G.bl, (size_t)G.L * 3); if (r) return r; }
        if (it == 0) { // computeLambdaInit :166-180
            std::vector<double> diag(std::max<size_t>(nx, 1), 0.0);
            if (nx) {
                CS_LAUNCH(ctx, "badyn_diag", badyn_diag, dim3((unsigned)((nx + 255) / 256)), dim3(256), 0, G, b->d_diag);
                r = cs_d2h(ctx, diag.data(), b->d_diag, nx); if (r) return r;
            }
            CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
            double mx = 0;
            for (size_t i = 0; i < nx; i++) mx = std::max(mx, std::fabs(diag[i]));
            lambda = 1e-5 * mx; ni = 2; nBad = 0;
        }
        double rho = 0;
        int qmax = 0;
        do {
            CS_HIP(ctx, hipMemcpyAsync(b->d_bak, b->d_state, sizeof(double) * std::max<size_t>(b->state_doubles, 1), hipMemcpyDeviceToDevice, ctx->stream)); // push()
            r = dyn_solve(ctx, b, lambda); if (r) return r;
            CS_LAUNCH(ctx, "badyn_update", badyn_update, dim3((b->n_vertices + 63) / 64), dim3(64), 0, G, b->n_vertices);
            int status = 0;
            b->h_x.resize(std::max<size_t>(nx, 1));
            r = cs_d2h(ctx, &status, b->d_status, 1); if (r) return r;
            if (G.NP) { r = cs_d2h(ctx, b->h_x.data(), G.xp, (size_t)G.NP); if (r) return r; }
            if (G.L) { r = cs_d2h(ctx, b->h_x.data() + G.NP, G.xl, (size_t)G.L * 3); if (r) return r; }
            r = dyn_errors(ctx, b, &tempChi); if (r) return r; // synchronises
            if (status != 0) tempChi = std::numeric_limits<double>::max();
            rho = (currentChi - tempChi);
            double scale = 0; // computeScale :182-190
            if (status == 0) for (size_t j = 0; j < nx; j++) scale += b->h_x[j] * (lambda * b->h_x[j] + b->h_b[j]);
            scale += 1e-3;
            rho /= scale;
            if (rho > 0 && std::isfinite(tempChi)) {
                double alpha = 1. - std::pow((2 * rho - 1), 3);
                alpha = (std::min)(alpha, 2. / 3.);
                const double scaleFactor = (std::max)(1. / 3., alpha);
                lambda *= scaleFactor;
                ni = 2;
                currentChi = tempChi;
                accepted = true;
            } else {
                accepted = false;
                lambda *= ni;
                ni *= 2;
                CS_HIP(ctx, hipMemcpyAsync(b->d_state, b->d_bak, sizeof(double) * std::max<size_t>(b->state_doubles, 1), hipMemcpyDeviceToDevice, ctx->stream)); // pop()
            }
            qmax++;
            S.lm_trials++;
        } while (rho < 0 && qmax < 10 && !terminate());
        S.iterations = it + 1;
        if (it < 64) S.chi2_trace[it] = currentChi;
        S.chi2_final = currentChi;
        if (qmax == 10 || rho == 0) break;
        if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
        if (nBad >= 3) break;
    }
    S.lambda_final = lambda;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (st) *st = S;
    return CS_OK;
}

int cs_ba_dyn_read(cs_ctx *ctx, cs_ba_dyn *b, double *cam_pose, double *obj_pose, double *vel, double *points, double *dpoints) {
    if (!ctx || !b) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const DynG &G = b->G;
    int r = CS_OK;
    if (cam_pose) r = cs_d2h(ctx, cam_pose, G.cam, (size_t)G.n_cams * 7);
    if (!r && obj_pose) r = cs_d2h(ctx, obj_pose, G.obj, (size_t)G.n_objs * 7);
    if (!r && vel) r = cs_d2h(ctx, vel, G.vel, (size_t)G.n_vels * 2);
    if (!r && points) r = cs_d2h(ctx, points, G.pts, (size_t)G.n_pts * 3);
    if (!r && dpoints) r = cs_d2h(ctx, dpoints, G.dpts, (size_t)G.n_dpts * 3);
    if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_ba_dyn_errors(cs_ctx *ctx, cs_ba_dyn *b, double *chi2, double *e_obs, double *e_dobs, double *e_mot, double *e_cobs, double *e_pc, double *e_ulp) {
    if (!ctx || !b) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const DynG &G = b->G;
    double chi = 0;
    int r = dyn_errors(ctx, b, &chi);
    if (r) return r;
    if (chi2) *chi2 = chi;
    if (e_obs) r = cs_d2h(ctx, e_obs, G.e_obs, (size_t)G.n_obs * 3);
    if (!r && e_dobs) r = cs_d2h(ctx, e_dobs, G.e_dobs, (size_t)G.n_dobs * 2);
    if (!r && e_mot) r = cs_d2h(ctx, e_mot, G.e_mot, (size_t)G.n_mot * 3);
    if (!r && e_cobs) r = cs_d2h(ctx, e_cobs, G.e_cobs, (size_t)G.n_cobs * 4);
    if (!r && e_pc) r = cs_d2h(ctx, e_pc, G.e_pc, (size_t)G.n_pc * 3);
    if (!r && e_ulp) r = cs_d2h(ctx, e_ulp, G.e_ulp, (size_t)G.n_dpts * 3);
    if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

int cs_ba_dyn_reduced_dense(cs_ctx *ctx, cs_ba_dyn *b, double lambda, double *H, double *bvec, int *n) {
    if (!ctx || !b || !n) return CS_ERR_BAD_ARG;
    CS_HIP(ctx, hipSetDevice(ctx->device));
    const DynG &G = b->G;
    *n = G.NP;
    if (!H || !bvec || G.NP == 0) return CS_OK;
    double chi = 0;
    int r = dyn_errors(ctx, b, &chi); if (r) return r;
    r = dyn_build(ctx, b); if (r) return r;
    r = dyn_reduce(ctx, b, lambda); if (r) return r;
    r = cs_d2h(ctx, H, G.S, (size_t)G.NP * G.NP); if (r) return r;
    r = cs_d2h(ctx, bvec, G.bs, (size_t)G.NP); if (r) return r;
    CS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return CS_OK;
}

} // extern "C"
