// K weighted-Jacobi sweeps of a BLOCK of a 3-D level per launch, the block resident on the CU (round 3).
//
// Levels of 10^5 .. 10^7 rows (65^3, 129^3: the middle of every hierarchy) are too small for the plane marches -- a march
// step costs microseconds whatever the plane's size, and a few dozen tiles cannot fill 256 CUs -- and were smoothed by one
// launch per sweep (multigrid.py:225-227: fifty sweeps in a row), each of them about 10 us of launch, ramp and drain around
// a few microseconds of work.  Here one 1024-thread workgroup takes a block of 32 x 32 x EZ cells -- the (32 - 2K) x (32 - 2K)
// x (EZ - 2K) cells it owns and a halo of K cells around them --, reads x, f and the class bytes of all of them ONCE, relaxes
// the block K times, each sweep on a region one cell smaller on every side (what the neighbours' halos recompute), and
// stores the cells it owns: no workgroup ever needs another one's results, so there is nothing to wait for between sweeps
// but the workgroup's own barrier.
//
// A thread owns a COLUMN of the block along z: its EZ values of x, f and its class bytes stay in registers for the whole
// launch, so the +-P neighbours are the thread's own registers; the +-1 neighbours come from the lanes next door (DPP, the
// 32 cells of a grid line are half a wave: the lanes where the rotation wraps are cells of the block's rim, which are never
// relaxed); only the +-nx neighbours go through LDS (one image of the block: every thread writes its column, barrier,
// reads the columns of the lines below and above).  Cells outside the grid are zeros of class 0 (the zero row) and stay
// zeros.  The sum of a row is formed in the order of the one-sweep kernels (-P, -nx, -1, 0, +1, +nx, +P) with the same
// fma chain and epilogue: bit-identical to single sweeps.
//
// Blocks are cut in GRID coordinates (i, j, k), not in row space: the kernel therefore needs what every grid matrix has --
// no entry that couples the last cell of a grid line to the first of the next one (or the last line of a plane to the next
// plane's first) -- and the host checks exactly that on the level's class table before it lets a level take this pass
// (sdia_grid_decoupled).
//
// MEASURED (profiles/r03_block_pass.txt, r03_block_cycles.txt): alone the pass is no faster per sweep than one launch per
// sweep -- 129^3 rows 13.2 us per sweep at best (three sweeps per launch, blocks of 11 planes) against 14.3, 65^3 3.4 - 3.9
// against 4.0, 257^3 90 against 52 for the K-sweep march: the halos a block recomputes (2.2 cells relaxed per cell kept at
// K = 3) and the per-cell class test make it VALU-bound, one 16-wave workgroup per CU cannot overlap its load phase with its
// sweeps, and blocks of 19 planes spill (100-200 registers).  Inside a cycle it still pays on the 129^3 and 65^3 levels,
// where a one-sweep launch costs 14.8 / 5.8 us and the pass has a third of the launches: BASELINE config 3 132.8 -> 137.5
// cycles/s with three sweeps per launch on blocks of 11 planes (the default; the cost model's choice of 19 planes: 124.4).
// Tried for the 19-plane blocks and dropped: f read again in every sweep instead of kept in registers, and the sweeps as a
// run-time loop (a third of the code) -- still 60 to 160 spilled registers (the spills sit in the load phase: 19 x loads, 19
// class bytes and their offsets in flight), 24.7 us per sweep on 129^3 rows, and the 11-plane shapes got slower too.
#pragma once
#include "mg_jacobik3d.hip.h"

namespace mgk {

struct JBArgs {
    const double* x;            // row-based
    const double* f;
    double* out;                // != x
    const unsigned char* cls;   // row-based
    const double* ctab;
    int ncls, cmain;
    double cm[8];
    double omega;
    int nx, ny, nz;
    int64_t P;
    int nbx, nby;               // blocks per dimension (x, y; the grid's z extent gives the third)
};

constexpr int JB_E = 32;        // cells of a block along x and along y

template <int EZ> constexpr size_t jb_lds_bytes(int ncls) { return sizeof(double) * ((size_t)EZ * JB_E * JB_E + (size_t)ncls * CLS_W); }

template <int K, int EZ>
__global__ __launch_bounds__(1024) void sdia_jacobi_block(JBArgs a) {
    constexpr int E = JB_E, OW = E - 2 * K, OZ = EZ - 2 * K;
    static_assert(OW > 0 && OZ > 0, "block too small for K sweeps");
    extern __shared__ double j2_smem[];
    double* const sX = j2_smem;                     // [EZ][E][E]: the block's current iterate
    double* const sT = sX + EZ * E * E;             // ncls x 8: entries of the row classes, [7] = omega / diagonal
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int bi = (int)(blockIdx.x % (unsigned)a.nbx), bj = (int)((blockIdx.x / (unsigned)a.nbx) % (unsigned)a.nby);
    const int bk = (int)(blockIdx.x / (unsigned)(a.nbx * a.nby));
    const int gx = bi * OW - K + tx, gy = bj * OW - K + ty, gz0 = bk * OZ - K;
    const bool col_in = gx >= 0 && gx < a.nx && gy >= 0 && gy < a.ny;
    for (int i = tid; i < a.ncls * CLS_W; i += 1024) {
        double v = a.ctab[i];
        if ((i & (CLS_W - 1)) == CLS_W - 1) {
            const double d = a.ctab[i - 4];
            v = a.omega * (1.0 / (d != 0.0 ? d : 1.0));
        }
        sT[i] = v;
    }
    const double m0 = a.cm[0], m1 = a.cm[1], m2 = a.cm[2], m3 = a.cm[3], m4 = a.cm[4], m5 = a.cm[5], m6 = a.cm[6];
    const double mcf = a.omega * (1.0 / (m3 != 0.0 ? m3 : 1.0));
    const int cmain = a.cmain;

    // the thread's column: x, f and the class bytes (four to a register) of its EZ cells
    double xr[EZ], fr[EZ];
    unsigned cl[(EZ + 3) / 4];
#pragma unroll
    for (int q = 0; q < (EZ + 3) / 4; ++q) cl[q] = 0u;
    const int64_t col = (int64_t)gy * a.nx + gx;
    {
        // `scalar base + 32-bit byte offset` (the level has fewer than 2^28 rows: block_sweeps_ok).  Cells outside the grid read
        // the element in front of row 0 -- the vectors' zero slack, class 0 in front of the class bytes -- so nothing is selected
        // behind the loads (a select per value keeps every loaded register alive beside its result: hundreds of spills)
        auto sbase = [](const void* p) -> gcptr_t {
            const unsigned long long u = (unsigned long long)p;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
            return (gcptr_t)(((unsigned long long)hi << 32) | lo);
        };
        const gcptr_t xb = sbase(a.x - 1), fb = sbase(a.f - 1), cb = sbase(a.cls - 1);
        const unsigned pstep = (unsigned)a.P;
        unsigned r = (unsigned)((int64_t)gz0 * a.P + col) + 1u;  // (wraps for cells below the grid: not used there)
#pragma unroll
        for (int z = 0; z < EZ; ++z) {
            const int gz = gz0 + z;
            unsigned e = col_in && gz >= 0 && gz < a.nz ? r : 0u;
            asm volatile("" : "+v"(e));
            xr[z] = *(const __attribute__((address_space(1))) double*)(xb + 8u * e);
            fr[z] = *(const __attribute__((address_space(1))) double*)(fb + 8u * e);
            cl[z >> 2] |= (unsigned)*(const __attribute__((address_space(1))) unsigned char*)(cb + e) << (8 * (z & 3));
            r += pstep;
        }
    }
    // bit z: cell z of every lane of this wave has the most frequent class (entries in scalar registers) -- decided once, the
    // classes do not change from sweep to sweep
    unsigned fastz = 0u;
#pragma unroll
    for (int z = 0; z < EZ; ++z) {
        const int c = (int)((cl[z >> 2] >> (8 * (z & 3))) & 255u);
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(c != cmain) == 0ull))) fastz |= 1u << z;
    }
    fastz = (unsigned)__builtin_amdgcn_readfirstlane((int)fastz);
    // LDS addresses as `register + 16-bit immediate`: one register per eight planes (64 KB) for the thread's own cell and for
    // the cells of the lines below / above (element offsets, pinned: left alone the compiler keeps one per plane)
    constexpr int NG = (EZ + 7) / 8;
    int bw[NG], bs[NG], bn[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        bw[g] = (g * 8 * E + ty) * E + tx;
        bs[g] = (g * 8 * E + max(ty - 1, 0)) * E + tx;
        bn[g] = (g * 8 * E + min(ty + 1, E - 1)) * E + tx;
        asm volatile("" : "+v"(bw[g]));
        asm volatile("" : "+v"(bs[g]));
        asm volatile("" : "+v"(bn[g]));
    }
#pragma unroll
    for (int s = 1; s <= K; ++s) {
#pragma unroll
        for (int z = 0; z < EZ; ++z) sX[bw[z >> 3] + (z & 7) * E * E] = xr[z];
        __syncthreads();
        // cells at least s cells inside the block have all their neighbours' values of sweep s - 1
        const bool act = tx >= s && tx < E - s && ty >= s && ty < E - s;
        double below = xr[s - 1];
#pragma unroll
        for (int z = s; z < EZ - s; ++z) {
            const double xc = xr[z], up = xr[z + 1];
            const double xw = jk3_from_west(xc), xe = jk3_from_east(xc);
            const double xs = sX[bs[z >> 3] + (z & 7) * E * E], xn = sX[bn[z >> 3] + (z & 7) * E * E];
            double acc, cf;
            if (fastz >> z & 1u) {
                acc = fma(m0, below, 0.0);
                acc = fma(m1, xs, acc);
                acc = fma(m2, xw, acc);
                acc = fma(m3, xc, acc);
                acc = fma(m4, xe, acc);
                acc = fma(m5, xn, acc);
                acc = fma(m6, up, acc);
                cf = mcf;
            } else {
                const int c = (int)((cl[z >> 2] >> (8 * (z & 3))) & 255u);
                const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * c);
                const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                acc = fma(t01.x, below, 0.0);
                acc = fma(t01.y, xs, acc);
                acc = fma(t23.x, xw, acc);
                acc = fma(t23.y, xc, acc);
                acc = fma(t45.x, xe, acc);
                acc = fma(t45.y, xn, acc);
                acc = fma(t67.x, up, acc);
                cf = t67.y;
            }
            const double o = xc + cf * (fr[z] - acc);
            below = xc;
            xr[z] = act ? o : xc;
            // (a few cells in flight at a time: scheduled freely, the reads of all EZ cells are hoisted and their registers spill)
            if ((z & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (s < K) __syncthreads();
    }
    if (col_in && tx >= K && tx < E - K && ty >= K && ty < E - K) {
#pragma unroll
        for (int z = K; z < EZ - K; ++z) {
            const int gz = gz0 + z;
            if (gz >= 0 && gz < a.nz) a.out[(int64_t)gz * a.P + col] = xr[z];
        }
    }
}

// Does any row couple cells that are no grid neighbours?  (see the header; one flag, set when it does)
struct JBCheckArgs {
    const unsigned char* cls;   // row-based
    const double* ctab;
    int nx, ny, nz;
    int64_t P, n;
    int* flag;
};

__global__ void sdia_grid_decoupled(JBCheckArgs a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.n) return;
    const int i = (int)(row % a.nx), j = (int)((row / a.nx) % a.ny), k = (int)(row / a.P);
    const double* const t = a.ctab + (size_t)CLS_W * a.cls[row];
    bool bad = false;
    if (i == 0 && t[2] != 0.0) bad = true;
    if (i == a.nx - 1 && t[4] != 0.0) bad = true;
    if (j == 0 && t[1] != 0.0) bad = true;
    if (j == a.ny - 1 && t[5] != 0.0) bad = true;
    if (k == 0 && t[0] != 0.0) bad = true;
    if (k == a.nz - 1 && t[6] != 0.0) bad = true;
    if (bad) atomicExch(a.flag, 1);
}

}  // namespace mgk
