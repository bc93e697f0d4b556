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

// One Gauss-Jordan pivot step of the in-place inverse, read from `in`, written to `out`.
__global__ void bt_gauss_jordan_step(int p, int piv, const double* __restrict__ in, double* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= p) return;
    const double pivot = 1.0 / in[(size_t)piv * p + piv];
    const double mij = in[(size_t)i * p + j];
    double o;
    if (i == piv) o = (j == piv) ? pivot : mij * pivot;
    else {
        const double f = in[(size_t)i * p + piv];
        o = (j == piv) ? -f * pivot : mij - f * (in[(size_t)piv * p + j] * pivot);
    }
    out[(size_t)i * p + j] = o;
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
        double d = 0.0;
        for (int c = lane; c < g.p; c += WAVE) d = fma(Ta[c], yp[c], d);
        acc -= l * wave_sum(d);
    }
    if (lane == 0) y[(size_t)k * g.p + i] = acc;
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
    double d = 0.0;
    for (int c = lane; c < g.p; c += WAVE) d = fma(Ti[c], w[c], d);
    d = wave_sum(d);
    if (lane == 0) {
        x[(size_t)k * g.p + i] = d;
        const int64_t row = (int64_t)k * g.p + i;
        if (row < g.n) out_rows[row] = d;
    }
}

}  // namespace mgk
