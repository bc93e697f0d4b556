// K weighted-Jacobi sweeps in one pass over HBM (K = 3, 4, 5) on class-coded 3-D seven-point levels.
//
// multigrid.py:223-228 runs mu = 50 identical sweeps in a row.  The two-sweep pass (mg_jacobi2.hip.h) already streams
// its 25 bytes per row at the rate the memory system gives a plane march, so the only lever left is bytes per SWEEP:
// here a workgroup carries K sweeps through the planes at once and a pass costs one read of x, f and the class byte
// and one write of the result per K sweeps (plus the tile rims, which grow with K: a tile loads EX x (EY + 2) cells of
// a plane and stores (EX - 2K) x (EY - 2K + 2)).
//
// The march.  "Level" t is the iterate after t sweeps (level 0 = x).  A thread owns NC cells of the tile; at step k
// the workgroup FINISHES plane k+K-t of level t for t = 1 .. K (level K of plane k is the result) and STARTS the plane
// above it.  A row's sum is accumulated in the order of the one-sweep kernels, -P, -nx, -1, 0, +1, +nx, +P:
//   phase A  (registers only)  the level-(t-1) value of plane k+K-t+1 has just been finished (or, level 0, arrived
//            from memory): it is the missing +P term of level t in plane k+K-t, whose result is thereby complete --
//            and in turn the missing term of level t+1 one plane below: the K finishes chain through registers.  The
//            value a cell relaxes (its own previous-level value, the diagonal operand) is read back from the cell's own
//            slot of the level's LDS image, which then takes the new plane.  The next plane's sum starts with its -P
//            term, that same read-back value.
//   barrier
//   phase B  the in-plane terms (-nx, -1, 0, +1, +nx) of the planes started in phase A.  A thread reads its own cells'
//            values back from the images and the line below / above its lines; the -1 / +1 neighbours are its own
//            values of the neighbouring lanes (DPP wave rotates; the lane at a 64-cell seam takes the rotated value of
//            the cell next door), the +-nx neighbours between a thread's own lines are its own registers.
//   barrier
// so per cell and level there are K partial sums in registers and ONE plane image in LDS -- K (EY+2) x EX images in
// all --, not three planes of every level, and per cell, level and step the LDS sees one write and three reads.  f is
// needed by every level of a plane: a ring of K values per cell, the newest arriving from memory a step before its first use; class bytes are packed four cells to a register.  The
// loads of plane k+K+1 are issued right after phase A has consumed plane k+K's registers and are in flight through
// phase B and both barriers.
// A wave whose cells of the planes k .. k+K all carry the level's most frequent class (the interior stencil), in a
// step whose planes all exist, runs straight-line code with the entries in scalar registers; other waves (next to the
// boundary, first / last planes) take the general path: per cell group either the scalar entries or the class table.
// Everything is done in ROW space like the other passes (cell <-> row whether or not it wraps around a grid line; rows
// outside the level are zeros at every level), with the same fma chain and the same IEEE division as sdia_cls_body:
// bit-identical to K single sweeps.  Out-of-grid cells that the tile clamps (lines beyond ny + 1 or below -2, cells
// past the end of a grid line) hold finite, wrong values; they reach grid rows only through couplings that do not
// exist (stored zeros), so no result depends on them.
//
// Slabs: with K halo planes of x (and K - 1 of f and the classes) on either side a rank relaxes level t on the planes
// [-(K-t), nz + (K-t)) of its neighbours too (plo / phi), so K sweeps need ONE exchange and no boundary chain.
#pragma once
#include "mg_jacobi2.hip.h"
#include <type_traits>

namespace mgk {

struct JK3Args {
    const double* x;            // row-based source iterate, zero slack (vec_reach) on both sides
    const double* f;            // row-based
    double* out;                // row-based, != x
    const unsigned char* cls;   // class of row r is cls[r + clead]
    const double* ctab;         // ctab[class][CLS_W]: the row's seven entries in column order
    int64_t clead, P;
    int ncls, cmain;
    double cm[8];               // entries of the most frequent class (scalar registers)
    double omega;
    int nx, ny, nz;             // nz: owned planes
    int plo, phi;               // halo planes below / above whose rows exist on a neighbour (0: the level ends here)
    // work items: tiles x plane segments of `seglen` planes; the launch covers the planes [za0, za1) and then [zb0, zb1)
    // (slabs: the planes the neighbours wait for in one launch, the rest in another; whole levels: [0, nz) and nothing)
    int ntx, nty, seglen, za0, za1, zb0, zb1;
    // ... and so that the last round of workgroups is not a nearly empty one, the tiles from `ta` on (fewer than there are
    // CUs; plane range a only) are cut into shorter segments of `seglen_b` planes and come last
    int ta, seglen_b;
    unsigned nitems, xcd_chunk;
    int nt_store;               // 1: non-temporal stores of the result
    // rows of class CLS_ESCAPE (mg_jacobi2.hip.h): read from the symmetric diagonal storage, slices of 1 << sshift rows
    int escape;                 // 1: the level has such rows
    const double* dvals;
    int64_t mlead;
    int sshift;
    int force_form;             // -1; timing experiments (mg_time_kernel only, results are wrong): every step in form 0 / 1 / 2
};

// Rows of class CLS_ESCAPE (levels with more than 255 distinct rows, mg_jacobi2.hip.h): when a plane arrives, every lane
// that holds such a row fetches its seven entries from the symmetric diagonal storage into a row of a POOL behind the
// class table -- rows handed out in turn, workgroup-wide, through an LDS counter -- and goes on with the pool row's number
// as the cell's class: the general form of the step then finds the entries where it finds those of any class.  A pool row
// is needed for the K + 1 steps its plane stays in the ring; the host checks (jk3_escape_window) that no tile ever takes
// more than jk3_pool(K) rows within K + 2 consecutive planes before it lets a level use the march.
// (a power of two that fits next to the K plane images of the 64 x 32 tiles: 64 KB, 32 KB with five images)
constexpr int jk3_pool(int K) { return K >= 5 ? 512 : 1024; }

// (TR: rows of the class table kept in LDS -- 256, or 64 for the shapes that run two workgroups per CU)
template <int K, int NW, int LPW, int M, int TR> constexpr size_t jk3_lds_doubles() {
    return TR * CLS_W + (size_t)K * (NW * LPW + 2) * (64 * M) + 2 * (64 * M + 2);
}
template <int K, int NW, int LPW, int M, int TR, bool ESC = false> constexpr size_t jk3_lds_bytes() {
    constexpr size_t base = jk3_lds_doubles<K, NW, LPW, M, TR>();
    return sizeof(double) * (ESC ? (base + CLS_W - 1) / CLS_W * CLS_W + (size_t)(jk3_pool(K) + 1) * CLS_W : base);
}

// (jk3_from_west / jk3_from_east -- the value of the lane next door through DPP -- live in mg_jacobi2.hip.h)

template <int K, int NW, int LPW, int M, bool DPP, int PF, int TR, bool ESC = false>
__device__ __forceinline__ void jk3_body(const JK3Args& a) {
    constexpr int EX = 64 * M, EY = NW * LPW, NC = M * LPW, IMG = (EY + 2) * EX;
    // classes in the ring: a byte per cell, four cells to a register -- sixteen bits where pool rows are classes too
    constexpr int CBITS = ESC ? 16 : 8, CPR = 32 / CBITS, CW = (NC + CPR - 1) / CPR;
    constexpr unsigned CMASK = (1u << CBITS) - 1u;
    constexpr int DYN0 = (int)((jk3_lds_doubles<K, NW, LPW, M, TR>() + CLS_W - 1) / CLS_W);   // the pool's first row, counted from sT
    static_assert(CLS_W == 8, "table rows of eight doubles");
    constexpr int WI = EX - 2 * K, HY = EY - 2 * K + 2;       // cells per line / lines of a tile that get all K sweeps
    static_assert(NC * (K + 1) <= 64, "the fast-path flags live in one scalar register pair");
    static_assert(WI > 0 && HY > 0, "tile too small for K sweeps");
    static_assert(PF == 1 || PF == 2, "planes of x in flight");
    extern __shared__ double j2_smem[];
    double* const sT = j2_smem;                               // 256 x 8   entries of the row classes, [7] = omega / diagonal
    double* const sI = sT + TR * CLS_W + (EX + 2);           // K x (EY+2) x EX   one plane of level t, origin (0,-1)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3, ch = a.xcd_chunk;
        id = ((j / ch) * 8u + xcd) * ch + (j % ch);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int nsa = (a.za1 - a.za0 + a.seglen - 1) / a.seglen;
    const int nsb = (a.zb1 - a.zb0 + a.seglen - 1) / a.seglen;
    const unsigned items_a = (unsigned)a.ta * (unsigned)(nsa + max(nsb, 0));
    unsigned tt;
    int z0, z1;
    if (id < items_a) {
        const int seg = (int)(id / (unsigned)a.ta);
        tt = id % (unsigned)a.ta;
        z0 = seg < nsa ? a.za0 + seg * a.seglen : a.zb0 + (seg - nsa) * a.seglen;
        z1 = min(seg < nsa ? a.za1 : a.zb1, z0 + a.seglen);
    } else {
        const unsigned j = id - items_a, tb = ntile - (unsigned)a.ta;
        tt = (unsigned)a.ta + j % tb;
        z0 = a.za0 + (int)(j / tb) * a.seglen_b;
        z1 = min(a.za1, z0 + a.seglen_b);
    }
    const int tiy = (int)(tt / (unsigned)a.ntx), tix = (int)(tt % (unsigned)a.ntx);
    if (z1 <= z0) return;
    const int tx0 = tix * WI - K, ty0 = tiy * HY - (K - 1);   // grid position of cell (0, 0)

    {
        const int nt = a.ncls * CLS_W;
        for (int i = threadIdx.x; i < nt; i += NW * WAVE) {
            double v = a.ctab[i];
            if ((i & (CLS_W - 1)) == CLS_W - 1) {
                const double d = a.ctab[i - 4];
                v = a.omega * (1.0 / (d != 0.0 ? d : 1.0));
            }
            sT[i] = v;
        }
        for (int i = threadIdx.x; i < K * IMG + 2 * (EX + 2); i += NW * WAVE) sT[TR * CLS_W + i] = 0.0;
    }
    unsigned* const esc_count = reinterpret_cast<unsigned*>(sT + (size_t)CLS_W * (DYN0 + jk3_pool(K)));   // (ESC) pool rows handed out so far
    if constexpr (ESC) {
        if (threadIdx.x == 0) *esc_count = 0u;
    }
    const double m0 = a.cm[0], m1 = a.cm[1], m2 = a.cm[2], m3 = a.cm[3], m4 = a.cm[4], m5 = a.cm[5], m6 = a.cm[6];
    const double mcf = a.omega * (1.0 / (m3 != 0.0 ? m3 : 1.0));
    const int cmain = a.cmain;

    // cell c = l*M + r of this thread: ex = lane + 64 r, ey = wave*LPW + l
    const int ey0 = wave * LPW;
    const int lw0 = (ey0 + 1) * EX + lane;
    auto lwof = [&](int c) -> int { return (c / M) * EX + 64 * (c % M); };      // cell c relative to cell 0
    // cell 0 in the images 0, 1 / 2, 3 / 4: one address register per pair, so that every LDS access of the march is
    // `register + 16-bit immediate` (the images span more than 64 KB; left alone the compiler keeps an address per
    // cell and image)
    int ibase[(K + 1) / 2];     // (element offsets from sI; the pin keeps the compiler from folding them back into one per access)
#pragma unroll
    for (int q = 0; q < (K + 1) / 2; ++q) {
        ibase[q] = 2 * q * IMG + lw0;
        asm volatile("" : "+v"(ibase[q]));
    }
    auto image = [&](int t) -> double* { return sI + ibase[t >> 1] + (t & 1) * IMG; };  // level t's plane, at cell 0
    unsigned inT = 0;           // bit c: this lane's cell c gets all K sweeps and lies on the grid
    const int64_t rb0 = (int64_t)(ty0 + ey0) * a.nx + (tx0 + lane);
    auto rowof = [&](int c) -> int64_t { return rb0 + (int64_t)(c / M) * a.nx + 64 * (c % M); };
    // the row of cell c in plane p exists  <=>  the level's plane range holds p + sh(c), sh = -1, 0, +1 (two bits per cell)
    unsigned shw = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ex = lane + 64 * (c % M), ey = ey0 + c / M;
        if (ex >= K && ex < EX - K && ey >= K - 1 && ey <= EY - K && tx0 + ex < a.nx && ty0 + ey < a.ny) inT |= 1u << c;
        const int64_t r = rowof(c);
        shw |= (r < 0 ? 0u : (r >= a.P ? 2u : 1u)) << (2 * c);
    }
    auto shof = [&](int c) -> int { return (int)((shw >> (2 * c)) & 3u) - 1; };
    // (ESC) bit c: cell c is a stand-in -- its line or its place in the line is clamped (see `eo`), it reads another cell's
    // row and no result depends on it: if that row is an escape row, the cell goes on as class 0 and takes no pool row
    unsigned standin = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int line = ty0 + ey0 + c / M;
        if (lane + 64 * (c % M) > a.nx + 1 - tx0 || line < -2 || line > a.ny + 1) standin |= 1u << c;
    }
    const bool wlo = wave == 0, whi = wave == NW - 1;

    // Addresses: `uniform base of the plane + a byte offset fixed for the whole march` (see j2c_body).  Lines are
    // clamped to [-2, ny + 1], cells past the end of their grid line to the cell behind it, planes to the slack planes.
    const int bias = 2 * a.nx + K + 2;
    const int xcl = a.nx + 1 - tx0;
    unsigned eo[NC], eor[M];    // byte offsets of the cells' doubles (>> 3: of their class bytes)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        eo[c] = 8u * (unsigned)(min(max(ty0 + ey0 + c / M, -2), a.ny + 1) * a.nx + tx0 + min(lane + 64 * (c % M), xcl) + bias);
#pragma unroll
    for (int r = 0; r < M; ++r)
        eor[r] = 8u * (unsigned)((wlo ? max(ty0 - 1, -2) : min(ty0 + EY, a.ny + 1)) * a.nx + tx0 + min(lane + 64 * r, xcl) + bias);
    const unsigned char* const clsb = a.cls + a.clead - bias;
    const double* const xb0 = a.x - bias;
    const double* const fb0 = a.f - bias;
    auto sbase = [](const void* p) -> gcptr_t {
        const unsigned long long u = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
        return (gcptr_t)(((unsigned long long)hi << 32) | lo);
    };
    // (the pins keep the 32-bit offsets from being widened once and for all outside the loop: `scalar base + 32-bit
    //  register offset` is an addressing mode, a 64-bit register pair per cell is not)
    auto ldd = [](gcptr_t b, unsigned e8) -> double {
        asm volatile("" : "+v"(e8));
        return *(const __attribute__((address_space(1))) double*)(b + e8);
    };
    // (the class byte arrives as the low byte of an unaligned 32-bit load and is masked where it is first used: a byte
    //  load is masked by the compiler right behind the load -- which makes every wave wait for the loads it has just issued)
    typedef unsigned int __attribute__((aligned(1))) u32_unaligned_t;
    auto ldc = [](gcptr_t b, unsigned e8) -> int {
        unsigned e = e8 >> 3;
        asm volatile("" : "+v"(e));
        return (int)*(const __attribute__((address_space(1))) u32_unaligned_t*)(b + e);
    };
    const int pmin = -a.plo - 1, pmax = a.nz + a.phi;

    // registers per cell: K partial sums; f of the planes k .. k+K-1; x, f and the class of the plane that arrives
    double acc[K][NC], fr[K][NC];
    // x, the classes and the y ring of x of the planes that arrive: PF sets, taken in turn by successive steps
    double XA[NC], XB[PF == 2 ? NC : 1], hyA[M], hyB[PF == 2 ? M : 1];
    int CA[NC], CB[PF == 2 ? NC : 1];
    double FN[PF == 2 ? NC : 1];            // (PF = 2) f of the plane after the newest of the ring, staged like x
    unsigned cw[K + 1][CW];     // class bytes of the planes k .. k+K, four cells to a register
    unsigned long long fast = 0;        // wave-uniform; bit j*NC + c: all 64 cells c of plane k+j are of class cmain
    constexpr unsigned long long ALLFAST = (NC * (K + 1) == 64) ? ~0ull : ((1ull << (NC * (K + 1))) - 1ull);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int t = 0; t < K; ++t) acc[t][c] = fr[t][c] = 0.0;
        XA[c] = 0.0;
        CA[c] = 0;
    }
#pragma unroll
    for (int j = 0; j <= K; ++j)
#pragma unroll
        for (int q = 0; q < CW; ++q) cw[j][q] = 0u;
#pragma unroll
    for (int r = 0; r < M; ++r) hyA[r] = 0.0;
    XB[0] = 0.0; hyB[0] = 0.0; CB[0] = 0; FN[0] = 0.0;

    // Loads are issued in slices, slice i of n between the pieces of phase B: issued in one burst (14 per wave, all twelve
    // waves at the same point of the step) they keep every wave at the vector-memory issue port in front of the barrier
    // while nothing computes -- measured: the burst added its whole issue time to the step.
    // Slice i: x and the classes of its cells of `plane` into a register set (the y ring with the last slice), f of the
    // plane `fplane` into the slot of the ring that has just become free (a step needs it one step after x of that plane).
    auto load_slice = [&](const int i, const int n, const int plane, const int fplane, double (&X)[NC], int (&C)[NC], double (&hy)[M])
        __attribute__((always_inline)) {
        const int64_t o = (int64_t)min(max(plane, pmin), pmax) * a.P;
        const gcptr_t cb = sbase(clsb + o);
        const gcptr_t xb = sbase(xb0 + o);
        const gcptr_t fb = sbase(fb0 + (int64_t)min(max(fplane, pmin), pmax) * a.P);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if ((c * n) / NC != i) continue;
            X[c] = ldd(xb, eo[c]);
            C[c] = ldc(cb, eo[c]);
            if constexpr (PF == 2) FN[c] = ldd(fb, eo[c]);
            else fr[K - 1][c] = ldd(fb, eo[c]);
        }
        if (i == n - 1 && (wlo || whi)) {
#pragma unroll
            for (int r = 0; r < M; ++r) hy[r] = ldd(xb, eor[r]);
        }
    };
    auto cls_of = [&](int j, int c) -> int { return (int)((cw[j][c / CPR] >> (CBITS * (c % CPR))) & CMASK); };
    // (general form: extracted anew at every use -- kept, the table addresses of all cells and ring planes cost 20 registers)
    auto cls_now = [&](int j, int c) -> int {
        unsigned w = cw[j][c / CPR];
        asm volatile("" : "+v"(w));
        return (int)((w >> (CBITS * (c % CPR))) & CMASK);
    };
    int k_plane = 0;            // the step: plane whose result it stores, whether it stores, the plane's base in `out`
    bool k_store = false;
    __attribute__((address_space(1))) char* k_out = nullptr;
    // (a cell that stores is not clamped: its byte offset is its row's)
    auto store = [&](int c, double v) {
        unsigned e8 = eo[c];
        asm volatile("" : "+v"(e8));
        // (the result is next read a whole pass later: stored past the caches, it leaves the L2 to the tile rims)
        if (a.nt_store) __builtin_nontemporal_store(v, (__attribute__((address_space(1))) double*)(k_out + e8));
        else *(__attribute__((address_space(1))) double*)(k_out + e8) = v;
    };
    // plane range of level t (rows outside it are zeros)
    int lo_t[K + 1], n_t[K + 1];
#pragma unroll
    for (int t = 1; t <= K; ++t) {
        lo_t[t] = -min(a.plo, K - t);
        n_t[t] = a.nz + min(a.phi, K - t) - lo_t[t];
    }

    // ---- the three forms of a step (chosen per wave and step; all of them read and write the images alike) ----
    // Straight-line form.  COL = false: every cell has the interior stencil, entries in scalar registers.  COL = true:
    // a cell's class is the same in all planes of the ring -- it depends on the lane (tile columns along the x boundary) and / or
    // on the cell's grid line (tile rows along the y boundary), away from the first and last planes: the entries of each cell's
    // row come from the class table once per cell and phase.  (Until the y boundary took this form too its tiles ran the
    // general one, 1.6 x slower -- and on a level with one round of workgroups, 257^3, the slowest tiles ARE the launch.)
    auto phase_a_fast = [&](auto col_tag, const double (&X)[NC]) __attribute__((always_inline)) {
        constexpr bool COL = decltype(col_tag)::value;
#pragma unroll
        for (int r = 0; r < M; ++r) {
#pragma unroll
            for (int l = 0; l < LPW; ++l) {
                const int c = l * M + r, iw = lwof(c);
                double k0 = m0, k6 = m6, kcf = mcf;
                if constexpr (COL) {
                    const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * cls_of(0, c));
                    const dvec2_t t01 = tr[0], t67 = tr[3];
                    k0 = t01.x; k6 = t67.x; kcf = t67.y;
                }
                double nw = X[c];
#pragma unroll
                for (int t = 1; t <= K; ++t) {
                    double* const img = image(t - 1);
                    const double wd = img[iw];
                    const double s = fma(k6, nw, acc[t - 1][c]);      // +P
                    const double o = wd + kcf * (fr[K - t][c] - s);
                    acc[t - 1][c] = fma(k0, wd, 0.0);                 // -P of the plane above
                    img[iw] = nw;
                    nw = o;
                }
                if (k_store && (inT >> c & 1u)) store(c, nw);
            }
        }
    };
    // (phase B comes in K pieces, one per level, so that the loads can go out in COMMON code between them: a load issued
    //  inside the branches of the three forms makes the compiler spill hundreds of registers)
    auto phase_b_fast = [&](auto col_tag, const int t) __attribute__((always_inline)) {
        constexpr bool COL = decltype(col_tag)::value;
        const double* const img = image(t - 1);
        double v[LPW][M], ys[M], yn[M];
#pragma unroll
        for (int r = 0; r < M; ++r) {
            ys[r] = img[-EX + 64 * r];
#pragma unroll
            for (int l = 0; l < LPW; ++l) v[l][r] = img[l * EX + 64 * r];
            yn[r] = img[LPW * EX + 64 * r];
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
            double yw[M], ye[M];
            if constexpr (DPP) {
                double fw[M], fe[M];
#pragma unroll
                for (int r = 0; r < M; ++r) { fw[r] = jk3_from_west(v[l][r]); fe[r] = jk3_from_east(v[l][r]); }
#pragma unroll
                for (int r = 0; r < M; ++r) {
                    yw[r] = (r > 0 && lane == 0) ? fw[r > 0 ? r - 1 : 0] : fw[r];
                    ye[r] = (r + 1 < M && lane == 63) ? fe[r + 1 < M ? r + 1 : r] : fe[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < M; ++r) {
                    yw[r] = img[l * EX + 64 * r - 1];
                    ye[r] = img[l * EX + 64 * r + 1];
                }
            }
#pragma unroll
            for (int r = 0; r < M; ++r) {
                const int c = l * M + r;
                double k1 = m1, k2 = m2, k3 = m3, k4 = m4, k5 = m5;
                if constexpr (COL) {
                    const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * cls_of(0, c));
                    const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2];
                    k1 = t01.y; k2 = t23.x; k3 = t23.y; k4 = t45.x; k5 = t45.y;
                }
                double s = acc[t - 1][c];
                s = fma(k1, l > 0 ? v[l > 0 ? l - 1 : 0][r] : ys[r], s);
                s = fma(k2, yw[r], s);
                s = fma(k3, v[l][r], s);
                s = fma(k4, ye[r], s);
                s = fma(k5, l + 1 < LPW ? v[l + 1 < LPW ? l + 1 : l][r] : yn[r], s);
                acc[t - 1][c] = s;
            }
        }
    };
    // General form: every cell's entries from the class table, rows outside the level masked to zero.
    auto phase_a_general = [&](const double (&X)[NC]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = lwof(c);
            double nw = X[c];
            int pc = k_plane + shof(c);
            asm volatile("" : "+v"(pc));            // (one register per cell and step, not one per cell and level for the whole march)
#pragma unroll
            for (int t = 1; t <= K; ++t) {
                const int j = K - t;                                  // ring index of the plane being finished
                double* const img = image(t - 1);
                const double wd = img[iw];
                const dvec2_t t67 = reinterpret_cast<const dvec2_t*>(sT + CLS_W * cls_now(j, c))[3];
                const dvec2_t t01 = reinterpret_cast<const dvec2_t*>(sT + CLS_W * cls_now(j + 1, c))[0];
                const double s = fma(t67.x, nw, acc[t - 1][c]);
                double o = wd + t67.y * (fr[j][c] - s);
                o = (unsigned)(pc + j - lo_t[t]) < (unsigned)n_t[t] ? o : 0.0;
                acc[t - 1][c] = fma(t01.x, wd, 0.0);
                img[iw] = nw;
                nw = o;
                __builtin_amdgcn_sched_barrier(0);      // (rare path: one cell and level at a time keeps it out of the
                                                        //  register budget of the straight-line forms)
            }
            if (k_store && (inT >> c & 1u)) store(c, nw);
        }
    };
    auto phase_b_general = [&](const int t) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = lwof(c);
            const double* const img = image(t - 1);
            const double ys = img[iw - EX], yw = img[iw - 1], yd = img[iw], ye = img[iw + 1], yn = img[iw + EX];
            const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * cls_now(K - t, c));
            const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2];
            double s = acc[t - 1][c];
            s = fma(t01.y, ys, s);
            s = fma(t23.x, yw, s);
            s = fma(t23.y, yd, s);
            s = fma(t45.x, ye, s);
            s = fma(t45.y, yn, s);
            acc[t - 1][c] = s;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // One step: `X, C, hy` hold plane k+K (arrived).  PF = 1: when phase A has consumed them, the loads of plane k+K+1 go out
    // into the same registers and are in flight through phase B.  PF = 2: a second set `XN, CN, hyN` holds plane k+K+1, in
    // flight since the step before (with f of plane k+K in `FN`); behind phase A the first set takes the second one over (a
    // whole step after those loads went out) and the loads of plane k+K+2 go out into the second set: a plane is in
    // flight at all times.
    auto step = [&](const int k, double (&X)[NC], int (&C)[NC], double (&hy)[M], double (&XN)[PF == 2 ? NC : 1],
                    int (&CN)[PF == 2 ? NC : 1], double (&hyN)[PF == 2 ? M : 1]) __attribute__((always_inline)) {
        k_plane = k; k_store = k >= z0;
        {
            const unsigned long long u = (unsigned long long)(a.out - bias + (int64_t)k * a.P);
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
            const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
            k_out = (__attribute__((address_space(1))) char*)(((unsigned long long)hi << 32) | lo);
        }
        // ---- the plane k+K has arrived: its classes join the ring ----
        // (the loaded values are first touched HERE: left alone the compiler masks the class bytes right behind their
        //  loads, a step early, and every wave then waits for its loads before the barrier instead of computing)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            asm volatile("" : "+v"(C[c]));
            asm volatile("" : "+v"(X[c]));
            C[c] &= 255;
        }
        if constexpr (ESC) {
            // rows of class CLS_ESCAPE: their entries into pool rows, the pool row as the cell's class (see jk3_pool)
            if (a.escape) {
                const int64_t mp = (int64_t)min(max(k + K, pmin), pmax) * a.P - bias + a.mlead;
                const int sh = a.sshift;
                const double* const dv = a.dvals;
                auto at = [&](int64_t m, int slot) -> double { return dv[((((m >> sh) << 2) + slot) << sh) + (m & (((int64_t)1 << sh) - 1))]; };
                // (one counter update per wave and plane; the loads of all its cells in flight together)
                unsigned long long em[NC];
                unsigned total = 0u;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (C[c] == CLS_ESCAPE && (standin >> c & 1u)) C[c] = 0;
                    em[c] = __ballot(C[c] == CLS_ESCAPE);
                    total += (unsigned)__popcll(em[c]);
                }
                if (total != 0u) {
                    unsigned b0 = 0u;
                    if (lane == 0) b0 = atomicAdd(esc_count, total);
                    b0 = (unsigned)__builtin_amdgcn_readfirstlane((int)b0);
                    double e[NC][7];
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (C[c] == CLS_ESCAPE) {
                            const int64_t m = mp + (int64_t)(eo[c] >> 3);
                            e[c][0] = at(m - a.P, 3); e[c][1] = at(m - a.nx, 2); e[c][2] = at(m - 1, 1);
                            e[c][3] = at(m, 0); e[c][4] = at(m, 1); e[c][5] = at(m, 2); e[c][6] = at(m, 3);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        if (C[c] == CLS_ESCAPE) {
                            const unsigned below = __builtin_amdgcn_mbcnt_hi((unsigned)(em[c] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em[c], 0u));
                            const int row = DYN0 + (int)((b0 + below) & (unsigned)(jk3_pool(K) - 1));
                            double* const tr = sT + CLS_W * row;
#pragma unroll
                            for (int i = 0; i < 7; ++i) tr[i] = e[c][i];
                            tr[7] = a.omega * (1.0 / (e[c][3] != 0.0 ? e[c][3] : 1.0));
                            C[c] = row;
                        }
                        b0 += (unsigned)__popcll(em[c]);
                    }
                }
            }
        }
        {
            unsigned m = 0;
#pragma unroll
            for (int q = 0; q < CW; ++q) cw[K][q] = 0u;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                cw[K][c / CPR] |= (unsigned)C[c] << (CBITS * (c % CPR));
                if (__builtin_amdgcn_readfirstlane((int)(__ballot(C[c] != cmain) == 0ull))) m |= 1u << c;
            }
            fast |= (unsigned long long)m << (K * NC);
        }
        // which form: 0 = interior stencil everywhere in this wave's cells of the planes k .. k+K, 1 = every cell's class the
        // same in all those planes -- both only where every row the step touches exists (planes k-1 .. k+K+1 through sh) or lies on a
        // neighbouring slab (there a level's values beyond its plane range are wrong instead of zero, and reach no
        // result) --, 2 = general
        int form = 2;
        if ((a.plo > 0 || k >= 1) && (a.phi > 0 || k + K + 1 < a.nz)) {
            if (fast == ALLFAST) {
                form = 0;
            } else {
                bool same = true;
#pragma unroll
                for (int j = 0; j <= K; ++j)
#pragma unroll
                    for (int c = 0; c < NC; ++c) same = same && cls_of(j, c) == cls_of(0, c);
                if (__builtin_amdgcn_readfirstlane((int)(__ballot(!same) == 0ull))) form = 1;
            }
        }
        if (a.force_form >= 0) form = a.force_form;
        // ---- phase A: finish plane k+K-t of level t, t = 1 .. K; start the plane above it ----
        if (form == 0) phase_a_fast(std::false_type{}, X);
        else if (form == 1) phase_a_fast(std::true_type{}, X);
        else phase_a_general(X);
        if (wlo || whi) {
            double* const ring = image(0) + (wlo ? -EX : LPW * EX);
#pragma unroll
            for (int r = 0; r < M; ++r) ring[64 * r] = hy[r];
        }
        // ---- the rings move on; the next loads go out ----
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int j = 0; j + 1 < K; ++j) fr[j][c] = fr[j + 1][c];
            asm volatile("" : "+v"(fr[K - 2][c]));
        }
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int q = 0; q < CW; ++q) cw[j][q] = cw[j + 1][q];
        fast >>= NC;
        __syncthreads();
        // ---- phase B: the in-plane terms of the planes started above (ring index K - t of the moved ring); the next
        //      loads go out between its pieces ----
        if constexpr (PF == 2) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                X[c] = XN[c]; C[c] = CN[c]; fr[K - 1][c] = FN[c];
                asm volatile("" : "+v"(X[c]));
                asm volatile("" : "+v"(C[c]));
                asm volatile("" : "+v"(fr[K - 1][c]));
            }
            if (wlo || whi) {
#pragma unroll
                for (int r = 0; r < M; ++r) { hy[r] = hyN[r]; asm volatile("" : "+v"(hy[r])); }
            }
        }
        auto issue = [&](const int i, const int n) __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF == 2) load_slice(i, n, k + K + 2, k + K + 1, XN, CN, hyN);
            else load_slice(i, n, k + K + 1, k + K, X, C, hy);
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int t = 1; t <= K; ++t) {
            if (form == 0) phase_b_fast(std::false_type{}, t);
            else if (form == 1) phase_b_fast(std::true_type{}, t);
            else phase_b_general(t);
            issue(t - 1, K);
        }
        __syncthreads();
    };

    load_slice(0, 1, z0 - K, z0 - K - 1, XA, CA, hyA);
    if constexpr (PF == 2) {
#pragma unroll
        for (int c = 0; c < NC; ++c) fr[K - 1][c] = FN[c];      // (the first slice's f went to the staging set)
        load_slice(0, 1, z0 - K + 1, z0 - K, XB, CB, hyB);
    }
    __syncthreads();

    for (int k = z0 - 2 * K; k < z1; ++k) step(k, XA, CA, hyA, XB, CB, hyB);
}

// WPE: waves per SIMD the register budget is set for (12-wave workgroups: 3, one per CU; 6- and 8-wave workgroups: 3 / 4,
// two per CU, so that one workgroup's waves run while the other's wait at a barrier)
template <int K, int NW, int LPW, int M, bool DPP, int PF, int WPE, int TR>
__global__ __attribute__((amdgpu_flat_work_group_size(NW * WAVE, NW * WAVE), amdgpu_waves_per_eu(WPE, WPE)))
void sdia_jacobikc(JK3Args a) {
    jk3_body<K, NW, LPW, M, DPP, PF, TR>(a);
}

// (its own symbol for the finest level, so that profiler summaries list the dominant launches apart)
template <int K, int NW, int LPW, int M, bool DPP, int PF, int WPE, int TR>
__global__ __attribute__((amdgpu_flat_work_group_size(NW * WAVE, NW * WAVE), amdgpu_waves_per_eu(WPE, WPE)))
void sdia_jacobikc_finest(JK3Args a) {
    jk3_body<K, NW, LPW, M, DPP, PF, TR>(a);
}

// (levels with rows of class CLS_ESCAPE)
template <int K, int NW, int LPW, int M, bool DPP, int PF, int WPE, int TR>
__global__ __attribute__((amdgpu_flat_work_group_size(NW * WAVE, NW * WAVE), amdgpu_waves_per_eu(WPE, WPE)))
void sdia_jacobikc_escape(JK3Args a) {
    jk3_body<K, NW, LPW, M, DPP, PF, TR, true>(a);
}

// The most pool rows any tile of the march <K, NW, LPW, M> takes within K + 2 consecutive planes: one workgroup per tile
// counts the cells of class CLS_ESCAPE plane by plane -- the tile's cells at the very rows the march reads their classes
// from, but for the clamped ones (`standin` there) -- and keeps the largest sum over a window of planes.
struct JK3WindowArgs {
    const unsigned char* cls;
    int64_t clead, P;
    int nx, ny, nz, ntx, nty;
    unsigned* most;
};

template <int K, int NW, int LPW, int M>
__global__ __launch_bounds__(256) void jk3_escape_window(JK3WindowArgs a) {
    constexpr int EX = 64 * M, EY = NW * LPW, WI = EX - 2 * K, HY = EY - 2 * K + 2, W = K + 2;
    __shared__ unsigned s_ring[W], s_plane;
    const int tiy = (int)(blockIdx.x / (unsigned)a.ntx), tix = (int)(blockIdx.x % (unsigned)a.ntx);
    const int tx0 = tix * WI - K, ty0 = tiy * HY - (K - 1), xcl = a.nx + 1 - tx0;
    unsigned win = 0u, best = 0u;
    for (int p = 0; p < a.nz; ++p) {
        if (threadIdx.x == 0) s_plane = 0u;
        __syncthreads();
        unsigned n = 0u;
        for (int i = threadIdx.x; i < EX * EY; i += 256) {
            const int ex = i % EX, line = ty0 + i / EX;
            if (ex > xcl || line < -2 || line > a.ny + 1) continue;
            const int64_t row = (int64_t)p * a.P + (int64_t)line * a.nx + tx0 + ex;
            n += a.cls[a.clead + row] == CLS_ESCAPE ? 1u : 0u;
        }
        if (n) atomicAdd(&s_plane, n);
        __syncthreads();
        if (threadIdx.x == 0) {
            win += s_plane - (p >= W ? s_ring[p % W] : 0u);
            s_ring[p % W] = s_plane;
            best = max(best, win);
        }
    }
    if (threadIdx.x == 0 && best) atomicMax(a.most, best);
}

}  // namespace mgk
