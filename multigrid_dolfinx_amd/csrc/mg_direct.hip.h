// Exact coarsest-level solve on the device: block-tridiagonal LU of the lexicographically ordered grid
// matrix (stands in for `spsolve(A_sp_dict[coarsest_level][0], f_h)`, multigrid.py:239-241).
//
// After the internal renumbering every row couples to its own grid plane and the two adjacent ones, so
// with g consecutive planes per block the matrix is block tridiagonal with dense-ish p x p diagonal
// blocks D_k (p = g * plane, 289 .. ~1100) and very sparse couplings L_k, U_k.  Set-up (once per
// hierarchy) forms the Schur complements S_0 = D_0, S_k = D_k - L_k S_{k-1}^-1 U_{k-1} and stores
// T_k = S_k^-1 densely (Gauss-Jordan without pivoting: the S_k of an SPD / M-matrix need none).  A solve
// is then nb forward steps y_k = b_k - L_k T_{k-1} y_{k-1} and nb backward steps
// x_k = T_k (y_k - U_k x_{k+1}): 2 nb small launches, no iteration, no host round trip, bit-reproducible.
#pragma once
#include "mg_kernels.hip.h"

namespace mgk {

struct EllView {
    const double* vals;
    const int* cols;
    const unsigned long long* codes;
    const int* offsets;
    int W, R, coded;
    int64_t n;          // rows of the level (lead == 0: the coarsest level is never distributed)
};

// k-th stored entry of `row`: column (row-based index) and value
__device__ __forceinline__ void ell_get(const EllView& e, int64_t row, int k, int64_t* col, double* val) {
    const int64_t S = (int64_t)WAVE * e.R;
    const int64_t slice = row / S, within = row % S;
    const size_t idx = ((size_t)slice * e.W + k) * S + within;
    *val = e.vals[idx];
    if (e.coded) {
        const int CW = (e.W + 7) / 8;
        const unsigned long long w = e.codes[((size_t)slice * CW + k / 8) * S + within];
        *col = row + e.offsets[(int)((w >> (8 * (k % 8))) & 0xffull)];
    } else {
        *col = e.cols[idx];
    }
}

struct BtGeom {
    int64_t n;      // rows of the level
    int p;          // rows per block (g planes)
    int plane;      // rows per grid plane
    int nb;         // blocks
    int W;          // coupling entries kept per row (= ELL width)
};

// D_k as a dense row-major p x p matrix (rows past the end of the level become identity rows);
// status |= 1 if a row couples beyond the adjacent blocks.
__global__ void bt_extract_dense(EllView e, BtGeom g, int k, double* __restrict__ M, int* status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.p) return;
    const int64_t row = (int64_t)k * g.p + i;
    double* Mi = M + (size_t)i * g.p;
    if (row >= g.n) { Mi[i] = 1.0; return; }
    for (int s = 0; s < e.W; ++s) {
        int64_t col; double val;
        ell_get(e, row, s, &col, &val);
        if (val == 0.0) continue;
        const int64_t cb = col / g.p;
        if (cb == k) Mi[col - (int64_t)k * g.p] += val;
        else if (cb != k - 1 && cb != k + 1) atomicOr(status, 1);
    }
}

// Couplings of block k: Lc = entries of its rows into block k-1, Uc = into block k+1, as fixed-width
// lists (column local to the neighbouring block, value; unused slots have value 0).
__global__ void bt_extract_coupling(EllView e, BtGeom g, int k, int* __restrict__ lcol, double* __restrict__ lval,
                                    int* __restrict__ ucol, double* __restrict__ uval) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.p) return;
    const int64_t row = (int64_t)k * g.p + i;
    int nl = 0, nu = 0;
    const size_t o = (size_t)i * g.W;
    if (row < g.n) {
        for (int s = 0; s < e.W; ++s) {
            int64_t col; double val;
            ell_get(e, row, s, &col, &val);
            if (val == 0.0) continue;
            const int64_t cb = col / g.p;
            if (cb == k - 1) { lcol[o + nl] = (int)(col - (int64_t)(k - 1) * g.p); lval[o + nl] = val; ++nl; }
            else if (cb == k + 1) { ucol[o + nu] = (int)(col - (int64_t)(k + 1) * g.p); uval[o + nu] = val; ++nu; }
        }
    }
    for (; nl < g.W; ++nl) { lcol[o + nl] = 0; lval[o + nl] = 0.0; }
    for (; nu < g.W; ++nu) { ucol[o + nu] = 0; uval[o + nu] = 0.0; }
}

// M -= L_k T_{k-1} U_{k-1}; one thread per row i of block k, columns summed in ascending order of the
// U rows (deterministic).  L_k rows come from block k's couplings, U_{k-1} rows from block k-1's.
__global__ void bt_schur_update(BtGeom g, const int* __restrict__ lcol, const double* __restrict__ lval,
                                const int* __restrict__ ucol_prev, const double* __restrict__ uval_prev,
                                const double* __restrict__ Tprev, double* __restrict__ M) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.p) return;
    double* Mi = M + (size_t)i * g.p;
    for (int s = 0; s < g.W; ++s) {
        const double l = lval[(size_t)i * g.W + s];
        if (l == 0.0) continue;
        const double* Ta = Tprev + (size_t)lcol[(size_t)i * g.W + s] * g.p;
        for (int b = 0; b < g.p; ++b) {
            const double lt = l * Ta[b];
            for (int t = 0; t < g.W; ++t) {
                const double u = uval_prev[(size_t)b * g.W + t];
                if (u != 0.0) Mi[ucol_prev[(size_t)b * g.W + t]] -= lt * u;
            }
        }
    }
}

// ---- blocked in-place Gauss-Jordan inverse (no pivoting), GJ_NB pivots per round of 4 launches -----
// Round with pivot rows/columns K = [k0, k0+nb):
//   1. gj_pivot_block   P  = A[K,K]^-1                         (one workgroup, in LDS)
//   2. gj_row_panel     RP = P A[K,:]
//   3. gj_trailing      A[I,J] -= A[I,K] RP[:,J]               for I, J outside K
//   4. gj_finish        A[I,K] = -A[I,K] P (I outside K);  A[K,J] = RP[:,J] (J outside K);  A[K,K] = P
constexpr int GJ_NB = 32;

__global__ __launch_bounds__(256) void gj_pivot_block(int p, int k0, int nb, const double* __restrict__ A,
                                                        double* __restrict__ P) {
    __shared__ double s[GJ_NB][GJ_NB + 1];
    for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) s[t / nb][t % nb] = A[(size_t)(k0 + t / nb) * p + k0 + t % nb];
    __syncthreads();
    for (int piv = 0; piv < nb; ++piv) {
        const double pivot = 1.0 / s[piv][piv];
        __syncthreads();
        double upd[(GJ_NB * GJ_NB + 255) / 256];
        int cnt = 0;
        for (int t = threadIdx.x; t < nb * nb; t += blockDim.x, ++cnt) {
            const int i = t / nb, j = t % nb;
            double o;
            if (i == piv) o = (j == piv) ? pivot : s[i][j] * pivot;
            else {
                const double f = s[i][piv];
                o = (j == piv) ? -f * pivot : s[i][j] - f * (s[piv][j] * pivot);
            }
            upd[cnt] = o;
        }
        __syncthreads();
        cnt = 0;
        for (int t = threadIdx.x; t < nb * nb; t += blockDim.x, ++cnt) s[t / nb][t % nb] = upd[cnt];
        __syncthreads();
    }
    for (int t = threadIdx.x; t < nb * nb; t += blockDim.x) P[t] = s[t / nb][t % nb];
}

__global__ void gj_row_panel(int p, int k0, int nb, const double* __restrict__ A, const double* __restrict__ P,
                             double* __restrict__ RP) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= p) return;
    double acc = 0.0;
    for (int t = 0; t < nb; ++t) acc = fma(P[i * nb + t], A[(size_t)(k0 + t) * p + j], acc);
    RP[(size_t)i * p + j] = acc;
}

__global__ void gj_trailing(int p, int k0, int nb, double* __restrict__ A, const double* __restrict__ RP) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= p || (i >= k0 && i < k0 + nb) || (j >= k0 && j < k0 + nb)) return;
    const double* Ai = A + (size_t)i * p;
    double acc = Ai[j];
    for (int t = 0; t < nb; ++t) acc = fma(-Ai[k0 + t], RP[(size_t)t * p + j], acc);
    A[(size_t)i * p + j] = acc;
}

__global__ void gj_finish(int p, int k0, int nb, double* __restrict__ A, const double* __restrict__ P,
                          const double* __restrict__ RP) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p) return;
    double* Ai = A + (size_t)i * p;
    if (i >= k0 && i < k0 + nb) {
        const int r = i - k0;
        for (int j = 0; j < p; ++j) Ai[j] = (j >= k0 && j < k0 + nb) ? P[r * nb + (j - k0)] : RP[(size_t)r * p + j];
    } else {
        double old[GJ_NB];
        for (int t = 0; t < nb; ++t) old[t] = Ai[k0 + t];
        for (int j = 0; j < nb; ++j) {
            double acc = 0.0;
            for (int t = 0; t < nb; ++t) acc = fma(old[t], P[t * nb + j], acc);
            Ai[k0 + j] = -acc;
        }
    }
}

// The lane's share of the dot product of a dense row with a vector (entries lane, lane + 64, ...), accumulated in that order;
// eight pairs of loads are issued before the first fma (written as a plain loop the compiler waits for every pair: 34 memory
// latencies per row of 2178 entries -- 18 us per block step, now bound by the bytes of the dense block).
__device__ __forceinline__ double bt_lane_dot(const double* __restrict__ T, const double* __restrict__ v, int p, int lane, double d) {
    int c = lane;
    for (; c + 7 * WAVE < p; c += 8 * WAVE) {
        double t[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { t[u] = T[c + u * WAVE]; w[u] = v[c + u * WAVE]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) d = fma(t[u], w[u], d);
    }
    for (; c < p; c += WAVE) d = fma(T[c], v[c], d);
    return d;
}

// Forward step k >= 1: y_k = b_k - L_k (T_{k-1} y_{k-1}); one wave per row, rows without a coupling copy b.
__global__ __launch_bounds__(BLOCK) void bt_forward(BtGeom g, int k, const int* __restrict__ lcol,
                                                     const double* __restrict__ lval, const double* __restrict__ Tprev,
                                                     const double* __restrict__ b, double* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= g.p) return;
    const int64_t row = (int64_t)k * g.p + i;
    double acc = row < g.n ? b[row] : 0.0;
    const double* yp = y + (size_t)(k - 1) * g.p;
    for (int s = 0; s < g.W; ++s) {
        const double l = lval[(size_t)i * g.W + s];
        if (l == 0.0) continue;                                  // wave-uniform
        const double* Ta = Tprev + (size_t)lcol[(size_t)i * g.W + s] * g.p;
        const double d = bt_lane_dot(Ta, yp, g.p, lane, 0.0);
        acc -= l * wave_sum(d);
    }
    if (lane == 0) y[(size_t)k * g.p + i] = acc;
}

// The forward step in two launches, for rows with MANY couplings into the previous block (P2: ~20): z = T_{k-1} y_{k-1} once
// (bt_matvec, one wave per row, the same dot products bt_forward forms -- per coupling and row -- and therefore the same
// bits), then y_k = b_k - L_k z (bt_forward_sparse, one thread per row, couplings in stored order like bt_forward).
__global__ __launch_bounds__(BLOCK) void bt_matvec(int p, const double* __restrict__ T, const double* __restrict__ w, double* __restrict__ z) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= p) return;
    const double* Ti = T + (size_t)i * p;
    const double d = wave_sum(bt_lane_dot(Ti, w, p, lane, 0.0));
    if (lane == 0) z[i] = d;
}

__global__ void bt_forward_sparse(BtGeom g, int k, const int* __restrict__ lcol, const double* __restrict__ lval,
                                  const double* __restrict__ z, const double* __restrict__ b, double* __restrict__ y) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.p) return;
    const int64_t row = (int64_t)k * g.p + i;
    double acc = row < g.n ? b[row] : 0.0;
    for (int s = 0; s < g.W; ++s) {
        const double l = lval[(size_t)i * g.W + s];
        if (l != 0.0) acc -= l * z[lcol[(size_t)i * g.W + s]];
    }
    y[(size_t)k * g.p + i] = acc;
}

// Backward step: x_k = T_k (y_k - U_k x_{k+1}); one wave per row of block k.  The corrected right-hand
// side w = y_k - U_k x_{k+1} is formed first by `bt_backward_rhs` (rows with couplings only).
__global__ void bt_backward_rhs(BtGeom g, int k, const int* __restrict__ ucol, const double* __restrict__ uval,
                                const double* __restrict__ y, const double* __restrict__ x, double* __restrict__ w) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= g.p) return;
    double acc = y[(size_t)k * g.p + i];
    if (k + 1 < g.nb) {
        const double* xn = x + (size_t)(k + 1) * g.p;
        for (int s = 0; s < g.W; ++s) {
            const double u = uval[(size_t)i * g.W + s];
            if (u != 0.0) acc -= u * xn[ucol[(size_t)i * g.W + s]];
        }
    }
    w[i] = acc;
}

__global__ __launch_bounds__(BLOCK) void bt_backward(BtGeom g, int k, const double* __restrict__ T,
                                                      const double* __restrict__ w, double* __restrict__ x,
                                                      double* __restrict__ out_rows) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (i >= g.p) return;
    const double* Ti = T + (size_t)i * g.p;
    const double d = wave_sum(bt_lane_dot(Ti, w, g.p, lane, 0.0));
    if (lane == 0) {
        x[(size_t)k * g.p + i] = d;
        const int64_t row = (int64_t)k * g.p + i;
        if (row < g.n) out_rows[row] = d;
    }
}

}  // namespace mgk
