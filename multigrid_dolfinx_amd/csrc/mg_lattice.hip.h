// Plane march for wide lattice stencils (P2 levels of BASELINE config 5) read through stencil classes.
//
// ell_cls_apply (mg_kernels.hip.h) gathers every neighbour of a row from global memory: a P2 row has 23 entries on
// average (51 at most), so each x value is fetched ~23 times through L1 / L2 and the kernel runs at the rate of those
// gathers (4.5 ms per sweep of the 513^3 lattice, 9 % of what its 25 bytes per row need).  Here a workgroup marches a
// 64 x 16 tile of the grid along z with the five planes k-2 .. k+2 of x (tile + two cells of rim) in LDS: x comes from
// global memory once per tile (x 1.33 for the rim) and the 23 gathers per row are LDS reads.
//
// Lanes are dealt by lattice PARITY: the (i, j) parities of a wave's 64 cells are equal, the plane's k parity is equal
// anyway, so away from the boundary all lanes of a wave have the same stencil class and the class's (offset, value) pairs
// are wave-uniform: they are read from an LDS copy of the level's most frequent classes (the eight interior parity types)
// as broadcast reads.  Waves whose lanes differ in class -- tiles on the boundary -- run the pairs per lane from the
// global table like ell_cls_apply does.  Either way a row's entries are applied in stored order with the same fma chain
// and the same epilogue as ell_cls_apply: results are bit-identical.
//
//   Jacobi / residual: every cell of the plane, four per lane; a wave's four cells have the four (i, j) parities, so the 51-entry
//     vertex rows (the other types have 19 or 21 entries) are spread evenly over the waves.
//   Gauss-Seidel colour (in place): only the planes and cells of that colour; all four waves share its cells, one per lane.
//     A colour never reads its own colour, so the stale LDS copies of the cells a launch updates are never used.
//
// LDS (64 x 16 tile): 6 x 68 x 20 doubles (65 KB: planes k-2 .. k+2 are read while k+3 is written, one barrier per plane) +
// the class tables (8 classes x 56 entries x 32 B: 14 KB): two workgroups per CU.
#pragma once

namespace mgk {

// Tile shapes: cells with results TI x TJ (+ a rim of two), NT threads.  64 x 16 x 256, two workgroups per CU, is what runs.
// 128 x 16 x 512 (one per CU) moves fewer bytes -- a tile row of 132 cells is 9.25 cache lines for 8.25 lines of data where
// one of 68 cells is 5.25 for 4.25, and the rim is 29 % instead of 33 %; measured at the L2 / fabric boundary the narrow
// tile reads 2.77 GB per colour launch of the 513^3 lattice, 2.6 x the vector -- but is slower (8.0 against 6.9 ms per
// Gauss-Seidel sweep): "lattice_tile" 2, kept for experiments.
constexpr int LM_NS = 6;                                    // plane slots: k-2 .. k+2 are read while k+3 arrives
constexpr int LM_K = 8;                                     // classes with an LDS copy
constexpr int LM_NST = 2;                                   // register stages of the plane loads: plane k+3+LM_NST-1 is requested at step k

struct LatArgs {
    const double* x;            // row-based source (GS: also the destination)
    const double* f;            // row-based
    double* out;                // row-based
    const unsigned char* cls;   // row-based class of every owned row
    const int* s_pack;          // 256 x W: (dk + 2) << 16 | (dj + 2) << 8 | (di + 2) of every entry
    const double* s_val;        // 256 x W
    const int* s_cnt;           // 256
    int W, WP, ntop;            // WP = table pitch in LDS: W rounded up to a multiple of 4, + 4 (the loop reads one group ahead)
    int top[LM_K];              // classes copied to LDS (most frequent first)
    int64_t nloc, xlo, xhi, P;  // owned rows; rows [xlo, xhi) of x exist (halo planes of a slab); plane size
    int nx, ny, nz;             // owned planes nz
    int kg0;                    // global index of local plane 0 (colours)
    int color;
    double omega;
    int ntx, nty, seglen;
    unsigned nitems, xcd_chunk;
};

inline size_t lm_lds_bytes(int W, int TI, int TJ) {
    const size_t WP = (size_t)(W + 3) / 4 * 4 + 4, PS = (size_t)(TI + 4) * (TJ + 4);
    return sizeof(double) * (LM_NS * PS + LM_K * WP + LM_K) + sizeof(int) * (LM_NS * LM_K * WP + LM_K + 256 / 4);
}

// device histogram of the class bytes (set-up: which classes get the LDS copy)
__global__ void lm_class_histogram(const unsigned char* __restrict__ cls, int64_t n, int* __restrict__ hist) {
    __shared__ int h[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) atomicAdd(&h[cls[i]], 1);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += blockDim.x)
        if (h[i]) atomicAdd(&hist[i], h[i]);
}

template <int MODE, int LM_TI, int LM_TJ, int LM_THREADS>
__global__ __launch_bounds__(LM_THREADS) void lat_march(LatArgs a) {
    constexpr int LM_PX = LM_TI + 4, LM_PY = LM_TJ + 4, LM_PS = LM_PX * LM_PY;      // tile + rim; cells per LDS plane
    constexpr int LM_LOADS = (LM_PS + LM_THREADS - 1) / LM_THREADS;
    constexpr int LM_CENTER = 2 * LM_PX + 2;                    // in-plane offset of the cell itself (rim of two)
    constexpr int NW = LM_THREADS / 64, HX = LM_TI / 2;
    constexpr int NCP = (LM_TI / 2) * (LM_TJ / 2) / 64;        // wave-sized groups of cells per (i, j) parity
    constexpr int NC = MODE == MODE_GS ? 1 : 4 * NCP / NW;
    static_assert(NW % 4 == 0 && NCP == (NW / 4) * (4 * NCP / NW) && NCP <= NW, "tile shape / thread count");
    extern __shared__ double lm_smem[];
    const int WP = a.WP;
    double* const xs = lm_smem;                                     // LM_NS planes, plane p in slot p mod LM_NS
    double* const tval = xs + LM_NS * LM_PS;                        // [slot][t], padded with zeros to a multiple of 4
    double* const tdval = tval + LM_K * WP;                         // [slot]: the diagonal entry (1 if none / zero)
    int* const toff = reinterpret_cast<int*>(tdval + LM_K);         // [k mod LM_NS][slot][t]: LDS offset of the entry's cell
    int* const tcnt = toff + LM_NS * LM_K * WP;                     // [slot]: entries, rounded up to a multiple of 4
    unsigned char* const tslot = reinterpret_cast<unsigned char*>(tcnt + LM_K);     // [class] -> slot, 255 = no LDS copy
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3, ch = a.xcd_chunk;
        id = ((j / ch) * 8u + xcd) * ch + (j % ch);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int seg = (int)(id / ntile);
    const unsigned t_ = id % ntile;
    const int i0 = (int)(t_ % (unsigned)a.ntx) * LM_TI, j0 = (int)(t_ / (unsigned)a.ntx) * LM_TJ;
    const int z0 = seg * a.seglen, z1 = min(a.nz, z0 + a.seglen);
    if (z1 <= z0) return;

    // ---- class tables of the LDS-resident classes ----
    constexpr int DIAG = (2 << 16) | (2 << 8) | 2;
    auto inplane = [](int pk) -> int { return ((pk >> 8) & 255) * LM_PX + (pk & 255); };
    for (int e = tid; e < a.ntop * WP; e += LM_THREADS) {
        const int s = e / WP, t = e - s * WP, c = a.top[s];
        const bool real = t < a.s_cnt[c];
        // (padding: 0 * x of the cell itself, which leaves the sum as it is)
        tval[e] = real ? a.s_val[(size_t)c * a.W + t] : 0.0;
        const int pk = real ? a.s_pack[(size_t)c * a.W + t] : DIAG;
#pragma unroll
        for (int m = 0; m < LM_NS; ++m) toff[(m * LM_K + s) * WP + t] = ((m + (pk >> 16) + LM_NS - 2) % LM_NS) * LM_PS + inplane(pk);
    }
    if (tid < 256) tslot[tid] = 255;
    __syncthreads();
    if (tid < a.ntop) {
        const int c = a.top[tid], n = a.s_cnt[c];
        double d = 1.0;
        for (int t = 0; t < n; ++t)
            if (a.s_pack[(size_t)c * a.W + t] == DIAG && a.s_val[(size_t)c * a.W + t] != 0.0) d = a.s_val[(size_t)c * a.W + t];
        tdval[tid] = d;
        tcnt[tid] = (n + 3) & ~3;
        tslot[c] = (unsigned char)tid;
    }

    // ---- this thread's cells ----
    int pi, pj, pk_ = 0;
    if (MODE == MODE_GS) {
        const int par = a.color == 8 ? 0 : a.color;
        pi = par & 1; pj = (par >> 1) & 1; pk_ = (par >> 2) & 1;
    } else {
        pi = pj = 0;            // per cell, below
    }
    int cbase[NC];              // LDS offset of the cell inside a plane (the entries' offsets carry the rim)
    int64_t crow[NC];           // row of the cell in plane 0
    bool cok[NC];
    int chalf[NC];              // (i >> 1) + (j >> 1): splits the vertex type into the colours 0 and 8
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        // The cells of one (i, j) parity form NCP groups of 64 (lane -> half-column, group -> half-rows).  Jacobi / residual:
        // cell c of wave w has the parity (w + c) & 3 -- every wave gets each parity equally often per plane, so the
        // 51-entry vertex rows are spread over the waves -- and the group (w >> 2) * NC + c.  A colour launch: wave w has
        // group w of the colour's parity.
        const int par2 = MODE == MODE_GS ? (pi | (pj << 1)) : ((wave + c) & 3);
        const int grp = MODE == MODE_GS ? wave : (wave >> 2) * NC + c;
        const int q = grp * 64 + lane;
        const int ci = 2 * (q % HX) + (par2 & 1);
        const int cj = 2 * (q / HX) + (par2 >> 1);
        cbase[c] = cj * LM_PX + ci;
        crow[c] = (int64_t)(j0 + cj) * a.nx + (i0 + ci);
        cok[c] = i0 + ci < a.nx && j0 + cj < a.ny && (MODE != MODE_GS || wave < NCP);
        chalf[c] = ((i0 + ci) >> 1) + ((j0 + cj) >> 1);
    }
    // ---- this thread's share of a plane's loads (tile + rim) ----
    // One scalar base per plane + a 32-bit element offset per load, fixed for the march; cells outside the grid read the
    // nearest cell inside and planes outside the vector the nearest plane inside (whole planes exist or not: the halo of a
    // slab), zeros are selected afterwards: no load sits under a branch.
    unsigned goff[LM_LOADS];
    bool gok[LM_LOADS];
#pragma unroll
    for (int q = 0; q < LM_LOADS; ++q) {
        const int e = tid + q * LM_THREADS;
        const int lj = e / LM_PX, li = e - lj * LM_PX;
        const int gi = i0 - 2 + li, gj = j0 - 2 + lj;
        gok[q] = e < LM_PS && gi >= 0 && gi < a.nx && gj >= 0 && gj < a.ny;
        goff[q] = (unsigned)(min(max(gj, 0), a.ny - 1) * a.nx + min(max(gi, 0), a.nx - 1));
    }
    const int plo = (int)(a.xlo / a.P), phi = (int)(a.xhi / a.P);       // planes [plo, phi) of x exist
    // (the zeros are selected when a plane is STORED, steps after its loads were issued: a select right behind a load is an
    //  instruction that waits for it, and the compiler then drains the loads at the end of the step that issued them)
    auto load_plane = [&](int p, double (&r)[LM_LOADS]) {
        const double* const xb = a.x + (int64_t)min(max(p, plo), phi - 1) * a.P;
#pragma unroll
        for (int q = 0; q < LM_LOADS; ++q) r[q] = xb[goff[q]];
    };
    auto store_plane = [&](int p, const double (&r)[LM_LOADS]) {
        const bool pok = p >= plo && p < phi;
        double* const dst = xs + ((p + LM_NS) % LM_NS) * LM_PS;            // (p >= -2)
#pragma unroll
        for (int q = 0; q < LM_LOADS; ++q) {
            const int e = tid + q * LM_THREADS;
            if ((q + 1) * LM_THREADS <= LM_PS || e < LM_PS) dst[e] = (gok[q] && pok) ? r[q] : 0.0;
        }
    };
    // is this thread's cell c a row this launch relaxes in plane k?
    auto active = [&](int c, int k) -> bool {
        bool act = cok[c] && crow[c] + (int64_t)k * a.P < a.nloc;
        if (MODE == MODE_GS) {
            const int kg = a.kg0 + k;
            act = act && (kg & 1) == pk_;
            if (a.color == 0 || a.color == 8) act = act && (((chalf[c] + (kg >> 1)) & 1) ? 8 : 0) == a.color;
        }
        return act;
    };
    // class and right-hand side of the cells, one plane ahead of their use
    int cl_n[NC];
    double f_n[NC];
    auto load_cf = [&](int k) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            // (unconditional: a row outside the level reads the last row; such cells are never relaxed)
            const int64_t row = min(crow[c] + (int64_t)min(k, a.nz - 1) * a.P, a.nloc - 1);
            cl_n[c] = (int)a.cls[row];
            f_n[c] = a.f[row];
        }
    };

    // One plane of the march.  The plane loaded into `rs` LM_NST-1 steps ago (k+3) is stored at the end; the loads of plane
    // k+3+LM_NST-1 into `rl` are issued first and have LM_NST-1 steps to arrive (the march is unrolled over the register
    // stages, so nothing is copied behind a load).
    auto step = [&](const int k, const double (&rs)[LM_LOADS], double (&rl)[LM_LOADS]) {
        int cl[NC];
        double fr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { cl[c] = cl_n[c]; fr[c] = f_n[c]; }
        // (classes and f first: loads return in order, and the next plane wants these while `rl` may stay in flight)
        load_cf(k + 1);
        load_plane(k + 3 + LM_NST - 1, rl);
        const int m = (k % LM_NS + LM_NS) % LM_NS;
        if (MODE != MODE_GS || ((a.kg0 + k) & 1) == pk_) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!active(c, k)) continue;
                const int64_t row = crow[c] + (int64_t)k * a.P;
                const int c0 = __builtin_amdgcn_readfirstlane(cl[c]);
                const int slot = tslot[c0];
                const bool uniform = __ballot(cl[c] != c0) == 0ull && slot != 255;
                const double* const xc = xs + cbase[c];
                const double xr = xc[m * LM_PS + LM_CENTER];
                double s_ = 0.0, diag;
                if (uniform) {
                    const int n = __builtin_amdgcn_readfirstlane(tcnt[slot]);
                    const double* const pv = tval + slot * WP;
                    const int* const po = toff + (m * LM_K + slot) * WP;
                    diag = tdval[slot];
                    // four entries per turn; the (value, offset) pairs of the next turn are read while this turn's x values
                    // arrive (the tables have four entries of slack behind the last group)
                    double v0 = pv[0], v1 = pv[1], v2 = pv[2], v3 = pv[3];
                    int o0 = __builtin_amdgcn_readfirstlane(po[0]), o1 = __builtin_amdgcn_readfirstlane(po[1]);
                    int o2 = __builtin_amdgcn_readfirstlane(po[2]), o3 = __builtin_amdgcn_readfirstlane(po[3]);
                    for (int t = 0; t < n; t += 4) {
                        const double x0 = xc[o0], x1 = xc[o1], x2 = xc[o2], x3 = xc[o3];
                        const double w0 = pv[t + 4], w1 = pv[t + 5], w2 = pv[t + 6], w3 = pv[t + 7];
                        const int q0 = po[t + 4], q1 = po[t + 5], q2 = po[t + 6], q3 = po[t + 7];
                        s_ = fma(v0, x0, s_);
                        s_ = fma(v1, x1, s_);
                        s_ = fma(v2, x2, s_);
                        s_ = fma(v3, x3, s_);
                        v0 = w0; v1 = w1; v2 = w2; v3 = w3;
                        o0 = __builtin_amdgcn_readfirstlane(q0); o1 = __builtin_amdgcn_readfirstlane(q1);
                        o2 = __builtin_amdgcn_readfirstlane(q2); o3 = __builtin_amdgcn_readfirstlane(q3);
                    }
                } else {
                    const int n = a.s_cnt[cl[c]];
                    const int* const pp = a.s_pack + (size_t)cl[c] * a.W;
                    const double* const pv = a.s_val + (size_t)cl[c] * a.W;
                    diag = 1.0;
                    for (int t = 0; t < n; ++t) {
                        const int pk = pp[t];
                        const double v = pv[t];
                        const double xv = xc[((m + (pk >> 16) + LM_NS - 2) % LM_NS) * LM_PS + inplane(pk)];
                        if (pk == DIAG && v != 0.0) diag = v;
                        s_ = fma(v, xv, s_);
                    }
                }
                a.out[row] = MODE == MODE_RESIDUAL ? fr[c] - s_ : xr + (a.omega * (1.0 / diag)) * (fr[c] - s_);
            }
        }
        store_plane(k + 3, rs);                                     // slot (k-3) mod 6: last read a step ago
        __syncthreads();
    };

    double rr[LM_NST][LM_LOADS];
    for (int p = z0 - 2; p <= z0 + 2; ++p) {
        load_plane(p, rr[0]);
        store_plane(p, rr[0]);
    }
#pragma unroll
    for (int st = 0; st < LM_NST - 1; ++st) load_plane(z0 + 3 + st, rr[st]);
    load_cf(z0);
    __syncthreads();
    for (int k = z0; k < z1;) {
#pragma unroll
        for (int st = 0; st < LM_NST; ++st) {
            if (k < z1) {
                step(k, rr[st], rr[(st + LM_NST - 1) % LM_NST]);
                ++k;
            }
        }
    }
}


// ---- two colours of the nine-colour Gauss-Seidel sweep per launch ---------------------------------------------------------
// A colour launch above is a pass over the whole vector for one eighth of its rows.  Two colours c1 < c2 share a pass here:
//   * OUT OF PLACE: the sweep reads `xold` for the colours it has not relaxed yet and `xnew` for those it has (c < c1: done by
//     earlier launches of the sweep), and every colour is written once, into `xnew` -- no workgroup ever reads what another one
//     writes during the launch, so tiles need not agree on their progress;
//   * c1 is relaxed on the tile PLUS a rim of two cells and, along z, two planes beyond the segment's ends (the neighbours relax
//     those cells too; only the owner stores them) and its new values replace the old ones in the LDS planes;
//   * c2 is relaxed on the tile three planes behind, when every c1 value within its reach is in LDS;
//   * planes k-5 .. k+2 are read while plane k+3 arrives: nine slots of a (TI + 8) x (TJ + 8) plane, one barrier per plane.
// MEASURED SLOWER than the nine colour launches (9.1 against 6.8 ms per sweep of the 513^3 lattice; "lattice_gs2", off by
// default): nine slots of a plane only fit 32-wide tiles beside a second workgroup on the CU, twice as many tile columns march
// twice as many plane-steps, and the march is bound by its steps, not by the bytes it saves (19 instead of 27 GB per sweep).
// c2 < 0: one colour (the ninth).  The caller swaps the two vectors after the sweep.  Whole levels only: relaxing c1 on the
// halo planes of a slab would need the neighbour's rows.  Same entries, same order, same epilogue as the colour launches:
// bit-identical to the in-place sweep.
constexpr int G2_TI = 32, G2_TJ = 16, G2_R = 4;             // tile, rim
constexpr int G2_PX = G2_TI + 2 * G2_R, G2_PY = G2_TJ + 2 * G2_R, G2_PS = G2_PX * G2_PY;
constexpr int G2_NS = 9, G2_THREADS = 256, G2_LOADS = (G2_PS + G2_THREADS - 1) / G2_THREADS;
constexpr int G2_CENTER = G2_R * G2_PX + G2_R;

struct Gs2Args {
    const double* xold;         // row-based: the iterate before the sweep
    double* xnew;               // row-based: the iterate after the sweep (colours < c1 already in it)
    const double* f;
    const unsigned char* cls;
    const int* s_pack;
    const double* s_val;
    const int* s_cnt;
    int W, WP, ntop;
    int top[LM_K];
    int64_t nloc, P;
    int nx, ny, nz;
    int c1, c2;
    double omega;
    int ntx, nty, seglen;
    unsigned nitems, xcd_chunk;
};

inline size_t g2_lds_bytes(int W) {
    const size_t WP = (size_t)(W + 3) / 4 * 4 + 4;
    return sizeof(double) * (G2_NS * (size_t)G2_PS + LM_K * WP + LM_K) + sizeof(int) * (LM_K * WP + LM_K + 256 / 4);
}

__device__ __forceinline__ int g2_color(int i, int j, int k) {
    const int par = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2);
    if (par) return par;
    return (((i >> 1) + (j >> 1) + (k >> 1)) & 1) ? 8 : 0;
}

__global__ __launch_bounds__(G2_THREADS) void lat_gs2(Gs2Args a) {
    extern __shared__ double lm_smem[];
    const int WP = a.WP;
    double* const xs = lm_smem;                                     // G2_NS planes, plane p in slot p mod 9
    double* const tval = xs + G2_NS * G2_PS;                        // [slot][t]
    double* const tdval = tval + LM_K * WP;                         // [slot]
    int* const tpk = reinterpret_cast<int*>(tdval + LM_K);          // [slot][t]: (dk + 2) << 16 | in-plane offset (rim included)
    int* const tcnt = tpk + LM_K * WP;
    unsigned char* const tslot = reinterpret_cast<unsigned char*>(tcnt + LM_K);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3, ch = a.xcd_chunk;
        id = ((j / ch) * 8u + xcd) * ch + (j % ch);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int seg = (int)(id / ntile);
    const unsigned t_ = id % ntile;
    const int i0 = (int)(t_ % (unsigned)a.ntx) * G2_TI, j0 = (int)(t_ / (unsigned)a.ntx) * G2_TJ;
    const int z0 = seg * a.seglen, z1 = min(a.nz, z0 + a.seglen);
    if (z1 <= z0) return;

    constexpr int DIAG = (2 << 16) | (2 << 8) | 2;
    // in-plane LDS offset of an entry relative to a cell's own position (the cell positions below carry the rim)
    auto inplane = [](int pk) -> int { return (((pk >> 8) & 255) - 2) * G2_PX + ((pk & 255) - 2); };
    for (int e = tid; e < a.ntop * WP; e += G2_THREADS) {
        const int s = e / WP, t = e - s * WP, c = a.top[s];
        const bool real = t < a.s_cnt[c];
        tval[e] = real ? a.s_val[(size_t)c * a.W + t] : 0.0;
        const int pk = real ? a.s_pack[(size_t)c * a.W + t] : DIAG;
        tpk[e] = (pk & 0x70000) | (inplane(pk) & 0xffff);
    }
    tslot[tid] = 255;
    __syncthreads();
    if (tid < a.ntop) {
        const int c = a.top[tid], n = a.s_cnt[c];
        double d = 1.0;
        for (int t = 0; t < n; ++t)
            if (a.s_pack[(size_t)c * a.W + t] == DIAG && a.s_val[(size_t)c * a.W + t] != 0.0) d = a.s_val[(size_t)c * a.W + t];
        tdval[tid] = d;
        tcnt[tid] = (n + 3) & ~3;
        tslot[c] = (unsigned char)tid;
    }

    // ---- cells: colour c1 on the tile + rim of two (18 x 10 cells of its parity: waves 0 .. 2), colour c2 on the tile (16 x 8:
    //      waves 2, 3 -- wave 2 has the short end of c1) ----
    const int par1 = a.c1 == 8 ? 0 : a.c1, par2 = a.c2 == 8 ? 0 : max(a.c2, 0);
    constexpr int E1X = (G2_TI + 4) / 2, E1N = E1X * ((G2_TJ + 4) / 2);      // 18, 180
    constexpr int E2X = G2_TI / 2, E2N = E2X * (G2_TJ / 2);                  // 16, 128
    // c1: lane q of 180 -> extended-region cell (2a + pi - 2 [+2 if pi..], ...): positions relative to the tile origin in [-2, TI+2)
    int c1pos = -1, c1i = 0, c1j = 0;
    {
        const int q = wave * 64 + lane;
        if (wave < 3 && q < E1N) {
            const int aa = q % E1X, bb = q / E1X;
            c1i = 2 * aa - 2 + (par1 & 1); c1j = 2 * bb - 2 + ((par1 >> 1) & 1);           // [-2, TI+2) x [-2, TJ+2)
            c1pos = (c1j + G2_R) * G2_PX + (c1i + G2_R);
        }
    }
    int c2pos = -1, c2i = 0, c2j = 0;
    if (a.c2 >= 0 && wave >= 2) {
        const int q = (wave - 2) * 64 + lane;
        if (q < E2N) {
            const int aa = q % E2X, bb = q / E2X;
            c2i = 2 * aa + (par2 & 1); c2j = 2 * bb + ((par2 >> 1) & 1);
            c2pos = (c2j + G2_R) * G2_PX + (c2i + G2_R);
        }
    }
    const bool c1grid = c1pos >= 0 && i0 + c1i >= 0 && i0 + c1i < a.nx && j0 + c1j >= 0 && j0 + c1j < a.ny;
    const bool c1own = c1grid && c1i >= 0 && c1i < G2_TI && c1j >= 0 && c1j < G2_TJ;
    const bool c2grid = c2pos >= 0 && i0 + c2i < a.nx && j0 + c2j < a.ny;
    const int64_t c1row = (int64_t)(j0 + c1j) * a.nx + (i0 + c1i), c2row = (int64_t)(j0 + c2j) * a.nx + (i0 + c2i);
    const int c1half = ((i0 + c1i) >> 1) + ((j0 + c1j) >> 1), c2half = ((i0 + c2i) >> 1) + ((j0 + c2j) >> 1);

    // ---- plane loads (tile + rim of four): the source of a cell is the vector that holds its colour's current value ----
    unsigned goff[G2_LOADS];
    bool gok[G2_LOADS];
    int gi_[G2_LOADS], gj_[G2_LOADS];
#pragma unroll
    for (int q = 0; q < G2_LOADS; ++q) {
        const int e = tid + q * G2_THREADS;
        const int lj = e / G2_PX, li = e - lj * G2_PX;
        const int gi = i0 - G2_R + li, gj = j0 - G2_R + lj;
        gok[q] = e < G2_PS && gi >= 0 && gi < a.nx && gj >= 0 && gj < a.ny;
        gi_[q] = gi; gj_[q] = gj;
        goff[q] = (unsigned)(min(max(gj, 0), a.ny - 1) * a.nx + min(max(gi, 0), a.nx - 1));
    }
    auto load_plane = [&](int p, double (&r)[G2_LOADS]) {
        const int64_t po = (int64_t)min(max(p, 0), a.nz - 1) * a.P;
#pragma unroll
        for (int q = 0; q < G2_LOADS; ++q) {
            const bool done = g2_color(gi_[q], gj_[q], p) < a.c1;           // relaxed by an earlier launch of this sweep
            const double* const src = done ? a.xnew : a.xold;
            r[q] = src[po + goff[q]];
        }
    };
    auto store_plane = [&](int p, const double (&r)[G2_LOADS]) {         // (zeros selected here, not behind the loads)
        const bool pok = p >= 0 && p < a.nz;
        double* const dst = xs + ((p + 2 * G2_NS) % G2_NS) * G2_PS;        // (p >= -9)
#pragma unroll
        for (int q = 0; q < G2_LOADS; ++q) {
            const int e = tid + q * G2_THREADS;
            if ((q + 1) * G2_THREADS <= G2_PS || e < G2_PS) dst[e] = (gok[q] && pok) ? r[q] : 0.0;
        }
    };
    // one row: the class's entries in stored order from LDS (wave-uniform class with an LDS copy) or from the global table
    auto relax = [&](const int cl, const double fr, const int pos, const int k, double* out) {
        const int c0 = __builtin_amdgcn_readfirstlane(cl);
        const int slot = tslot[c0];
        const bool uniform = __ballot(cl != c0) == 0ull && slot != 255;
        const int m = (k + 2 * G2_NS) % G2_NS;
        const double* const xc = xs + pos;
        const double xr = xc[m * G2_PS];
        double s_ = 0.0, diag;
        if (uniform) {
            const int n = __builtin_amdgcn_readfirstlane(tcnt[slot]);
            const double* const pv = tval + slot * WP;
            const int* const po = tpk + slot * WP;
            diag = tdval[slot];
            auto at = [&](int pk) -> int {
                int sl = m + (pk >> 16) - 2;
                sl = sl < 0 ? sl + G2_NS : (sl >= G2_NS ? sl - G2_NS : sl);
                return sl * G2_PS + (int)(short)(pk & 0xffff);
            };
            // four entries per turn, the next four (value, offset) pairs in flight (four entries of slack in the tables)
            double v0 = pv[0], v1 = pv[1], v2 = pv[2], v3 = pv[3];
            int o0 = at(__builtin_amdgcn_readfirstlane(po[0])), o1 = at(__builtin_amdgcn_readfirstlane(po[1]));
            int o2 = at(__builtin_amdgcn_readfirstlane(po[2])), o3 = at(__builtin_amdgcn_readfirstlane(po[3]));
            for (int t = 0; t < n; t += 4) {
                const double x0 = xc[o0], x1 = xc[o1], x2 = xc[o2], x3 = xc[o3];
                const double w0 = pv[t + 4], w1 = pv[t + 5], w2 = pv[t + 6], w3 = pv[t + 7];
                const int q0 = po[t + 4], q1 = po[t + 5], q2 = po[t + 6], q3 = po[t + 7];
                s_ = fma(v0, x0, s_);
                s_ = fma(v1, x1, s_);
                s_ = fma(v2, x2, s_);
                s_ = fma(v3, x3, s_);
                v0 = w0; v1 = w1; v2 = w2; v3 = w3;
                o0 = at(__builtin_amdgcn_readfirstlane(q0)); o1 = at(__builtin_amdgcn_readfirstlane(q1));
                o2 = at(__builtin_amdgcn_readfirstlane(q2)); o3 = at(__builtin_amdgcn_readfirstlane(q3));
            }
        } else {
            const int n = a.s_cnt[cl];
            const int* const pp = a.s_pack + (size_t)cl * a.W;
            const double* const pv = a.s_val + (size_t)cl * a.W;
            diag = 1.0;
            for (int t = 0; t < n; ++t) {
                const int pk = pp[t];
                const double v = pv[t];
                int sl = m + (pk >> 16) - 2;
                sl = sl < 0 ? sl + G2_NS : (sl >= G2_NS ? sl - G2_NS : sl);
                const double xv = xc[sl * G2_PS + inplane(pk)];
                if (pk == DIAG && v != 0.0) diag = v;
                s_ = fma(v, xv, s_);
            }
        }
        *out = xr + (a.omega * (1.0 / diag)) * (fr - s_);
    };
    auto is_color = [&](int col, int par, int half, int kg) -> bool {
        if (((par >> 2) & 1) != (kg & 1)) return false;
        if (col == 0 || col == 8) return ((((half + (kg >> 1)) & 1) ? 8 : 0) == col);
        return true;
    };

    // planes: c1 on [z0-2, z1+2) (clipped to the level), c2 on [z0, z1), three steps behind; loads two steps ahead
    const int ka = max(z0 - 2, 0), kb = min(z1 + 2, a.nz);
    double ra[G2_LOADS], rb[G2_LOADS];
    for (int p = ka - 2; p <= ka + 2; ++p) {
        load_plane(p, ra);
        store_plane(p, ra);
    }
    load_plane(ka + 3, ra);
    __syncthreads();
    // class and right-hand side of this thread's two cells, one step ahead of their use (unconditional loads: rows outside the
    // level read the nearest row inside, such cells are never relaxed)
    int cl1n = 0, cl2n = 0;
    double f1n = 0.0, f2n = 0.0;
    auto load_cf = [&](int k) {                                     // for the step that relaxes c1 on plane k and c2 on plane k-3
        const int64_t r1 = min(max(c1row, (int64_t)0) + (int64_t)min(max(k, 0), a.nz - 1) * a.P, a.nloc - 1);
        const int64_t r2 = min(max(c2row, (int64_t)0) + (int64_t)min(max(k - 3, 0), a.nz - 1) * a.P, a.nloc - 1);
        cl1n = (int)a.cls[r1]; f1n = a.f[r1];
        cl2n = (int)a.cls[r2]; f2n = a.f[r2];
    };
    auto step = [&](const int k, const double (&rs)[G2_LOADS], double (&rl)[G2_LOADS]) {
        const int cl1 = cl1n, cl2 = cl2n;
        const double f1 = f1n, f2 = f2n;
        load_cf(k + 1);
        load_plane(k + 4, rl);
        // ---- c1 on plane k ----
        if (k < kb && c1grid && is_color(a.c1, par1, c1half, k)) {
            const int64_t row = c1row + (int64_t)k * a.P;
            double o;
            relax(cl1, f1, c1pos, k, &o);
            xs[((k + 2 * G2_NS) % G2_NS) * G2_PS + c1pos] = o;
            if (c1own && k >= z0 && k < z1) a.xnew[row] = o;
        }
        // ---- c2 on plane k-3: every c1 value within its reach was replaced in an earlier step ----
        const int k2 = k - 3;
        if (a.c2 >= 0 && k2 >= z0 && k2 < z1 && c2grid && is_color(a.c2, par2, c2half, k2)) {
            const int64_t row = c2row + (int64_t)k2 * a.P;
            double o;
            relax(cl2, f2, c2pos, k2, &o);
            a.xnew[row] = o;
        }
        store_plane(k + 3, rs);                                     // slot of plane k-6: last read a step ago
        __syncthreads();
    };
    load_cf(ka);
    int k = ka;
    const int kend = kb + 3;                                        // c2 trails by three planes
    for (; k + 1 < kend; k += 2) {
        step(k, ra, rb);
        step(k + 1, rb, ra);
    }
    if (k < kend) step(k, ra, rb);
}

}  // namespace mgk
