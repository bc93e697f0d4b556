// Two weighted-Jacobi sweeps in one pass over HBM (3-D seven-point levels in symmetric diagonal storage).
//
// One sweep of sdia_jacobi_finest moves 56 B per row (32 matrix + 8 x + 8 f + 8 out) and runs at the HBM
// rate, so the only way to make V(mu1, mu2) faster is to move fewer bytes: here the matrix, f and x are
// read ONCE for two sweeps and the intermediate iterate never leaves the CU.
//
// A workgroup owns a tile of EX x EY grid lines (x, y) and marches through the planes (z).  In row space the
// seven offsets are {0, +-1, +-nx, +-P}: the +-P neighbours of a cell are the SAME thread's cell one step
// earlier / later (registers), the +-1 and +-nx neighbours are other threads' cells of the same plane (LDS).
// At step k the workgroup
//     A   relaxes plane k+1 once (x, matrix: registers + the LDS image of plane k+1) -> v1[k+1], and completes
//         the second relaxation of plane k with its last term, a(+P) * v1[k+1]                       -> out
//     B   starts the second relaxation of plane k+1: every term but the last (v1[k+1] neighbours: LDS)
//     C   parks the operands of plane k+2, whose loads were issued at the top of the step, in LDS
// so every global load is issued one full step (~100 KB per CU) before its first use.  Sweep 1 is evaluated
// on the whole tile, sweep 2 on its interior (TX x TY = (EX-2) x (EY-2)); the one-cell ring of x / matrix
// entries sweep 1 needs around the tile is loaded by the edge lanes / edge waves.  Everything is done in ROW
// space (cell (ex, ey, k) <-> row k*P + (ty0+ey)*nx + tx0+ex, whether or not that wraps around a grid line),
// which is exactly what the one-sweep kernel computes, with the same fma order and the same IEEE division:
// results are bit-identical to two sdia_jacobi launches.  Tiles are independent (out != x), the plane range
// is cut into segments to have >> 256 work items, each paying one round of warm-up loads and two extra steps.
#pragma once
#include "mg_kernels.hip.h"

namespace mgk {

struct J2Args {
    const double* vals;     // symmetric diagonal storage (WU = 4: diagonal, +1, +nx, +P)
    const double* x;        // source iterate, row-based (x[row], zero slack on both sides)
    const double* f;        // row-based
    double* out;            // row-based, != x
    int64_t nloc, mlead, P;
    // slabs (one rank's planes of a distributed level): x also holds the neighbours' planes in rows [xlo, 0) and
    // [nloc, xhi), the matrix lead rows [slo, 0) hold the +P entries of the plane below; the second sweep of the
    // rows outside [st_lo, st_hi) needs the neighbours' once-relaxed planes and is left to the caller, for whom
    // the once-relaxed iterate of the rows below k1_lo / from k1_hi on is written to v1out.
    int64_t xlo, xhi, slo, st_lo, st_hi, k1_lo, k1_hi;
    double* v1out;          // row-based, may be null (whole levels)
    const double* zero;     // >= 3*S+1 stored zeros (the slack in front of a vector)
    // row classes (sdia_jacobi2c): cls[row + clead] indexes ctab[class][CLS_W] = the row's seven entries in column order
    const unsigned char* cls;
    const double* ctab;
    int64_t clead;          // padding in front of cls[]: class of row r is cls[r + clead]
    int ncls, cmain;        // classes in use (0 .. ncls-1); the most frequent one, whose entries are
    double cm[8];           // passed by value (scalar registers)
    int wi;                 // class-coded pass: cells per tile line that get their second sweep (<= 124)
    unsigned xcd_chunk;     // consecutive work items per XCD at a time (the grid is a multiple of 8 * xcd_chunk)
    // one-sweep march (sdia_sweep1c), MODE_GS: colour relaxed by this launch, its kind, global index of local row 0
    int color, color_kind;
    int64_t grow0;
    int nx, ny, nz;
    // plane segments: 0 = [0, zb), nseg-1 = [nz-zb, nz), the others cut [zb, nz-zb) into pieces of seglen planes;
    // this launch covers the segments seg0, seg0 + seg_stride, ... (nitems / (ntx*nty) of them)
    int ntx, nty, nseg, zb, seglen, seg0, seg_stride;
    unsigned nitems;
    double omega;
};

constexpr int J2_EX = 128;

template <int NW, int LPW> constexpr size_t j2_lds_bytes() {
    constexpr int EY = NW * LPW, PV = J2_EX + 2;
    return sizeof(double) * ((size_t)(EY + 2) * PV + 2 * (size_t)EY * PV + 2 * (size_t)(EY + 1) * J2_EX + (size_t)EY * J2_EX);
}

// Loads never branch: a row outside the level reads a stored zero instead (`zero`: the slack in front of a
// vector / the lead rows of the matrix), so all loads of a step are issued back to back.
template <bool NT> __device__ __forceinline__ double j2_ld(const double* p, bool ok, const double* zero) {
    const double* q = ok ? p : zero;
    if constexpr (NT) return __builtin_nontemporal_load(q);
    else return *q;
}

template <int R, int NW, int LPW, bool NT>
__device__ __forceinline__ void j2_body(const J2Args& a) {
    constexpr int S = WAVE * R, EX = J2_EX, EY = NW * LPW, NC = 2 * LPW, PV = EX + 2;
    extern __shared__ double j2_smem[];
    double* const sV0 = j2_smem;                          // (EY+2) x PV       x of one plane, origin (-1,-1)
    double* const sU1 = sV0 + (EY + 2) * PV;              // 2 x EY x PV       +1 diagonal, origin (-1, 0)
    double* const sU2 = sU1 + 2 * EY * PV;                // 2 x (EY+1) x EX   +nx diagonal, origin (0,-1)
    double* const sV1 = sU2 + 2 * (EY + 1) * EX;          // EY x EX           once-relaxed iterate of one plane
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    // work item: 32 consecutive items per XCD at a time (blocks are dealt round-robin over the 8 XCDs)
    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3;
        id = ((j >> 5) * 8u + xcd) * 32u + (j & 31u);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int seg = a.seg0 + (int)(id / ntile) * a.seg_stride;
    const unsigned t = id % ntile;
    const int tiy = (int)(t / (unsigned)a.ntx), tix = (int)(t % (unsigned)a.ntx);
    int z0, z1;
    if (seg == 0) { z0 = 0; z1 = min(a.zb, a.nz); }
    else if (seg == a.nseg - 1) { z0 = max(a.nz - a.zb, a.zb); z1 = a.nz; }
    else { z0 = a.zb + (seg - 1) * a.seglen; z1 = min(a.nz - a.zb, z0 + a.seglen); }
    if (z1 <= z0) return;
    const int tx0 = tix * (EX - 2) - 1, ty0 = tiy * (EY - 2) - 1;      // grid position of cell (0, 0)

    // cell c = 2*l + r of this thread: ex = lane + 64 r, ey = wave*LPW + l; everything else is an offset from cell 0
    const int ey0 = wave * LPW;
    const int64_t rb0 = (int64_t)(ty0 + ey0) * a.nx + (tx0 + lane);     // row of cell 0 in plane 0
    const int lv0 = (ey0 + 1) * PV + lane + 1;                            // cell 0 in sV0; in sU1 it is lv0 - PV
    const int lw0 = ey0 * EX + lane;                                      // cell 0 in sV1; in sU2 it is lw0 + EX
    auto rowof = [&](int c) -> int64_t { return rb0 + (int64_t)(c >> 1) * a.nx + 64 * (c & 1); };
    auto lvof = [&](int c) -> int { return lv0 + (c >> 1) * PV + 64 * (c & 1); };
    auto lwof = [&](int c) -> int { return lw0 + (c >> 1) * EX + 64 * (c & 1); };
    unsigned inT = 0;       // bit c: interior cell whose second sweep this tile stores
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ex = lane + 64 * (c & 1), ey = ey0 + (c >> 1);
        if (ex >= 1 && ex < EX - 1 && ey >= 1 && ey < EY - 1 && tx0 + ex < a.nx && ty0 + ey < a.ny) inT |= 1u << c;
    }
    const bool hl = lane == 0, hr = lane == 63;
    const bool wlo = wave == 0, whi = wave == NW - 1;

    auto mat = [&](int64_t row) -> const double* {        // address of the row's diagonal slot
        const uint64_t m = (uint64_t)(row + a.mlead);
        return a.vals + (size_t)(m / S) * (4 * S) + (size_t)(m % S);
    };
    const double* const zx = a.zero;
    const double* const zf = a.zero;
    const double* const zm = a.zero;                      // stands for all four slots of a row (m + c*S)

    // registers.  plane k: +P diagonal, f, omega/diag, the second sweep's sum up to the +nx term, v1;
    // planes k+1 and k+2 (in flight): the matrix row and f; x of planes k .. k+3
    double s0[NC], f0[NC], cf0[NC], ap0[NC], w0[NC];
    double d1[NC], p1[NC], q1[NC], s1[NC], f1[NC];
    double d2[NC], p2[NC], q2[NC], s2[NC], f2[NC];
    double va[NC], vb[NC], vc[NC], vd[NC], w1[NC];
    double hxv[LPW], hxu[LPW], hyv[2], hyu[2];
#pragma unroll
    for (int c = 0; c < NC; ++c) f0[c] = cf0[c] = ap0[c] = w0[c] = w1[c] = 0.0;

    // the matrix row, f and the ring (x and the diagonals its sweep needs) of one plane
    // (`on` false: a plane past the segment that no step uses -- every lane reads the stored zero)
    auto load_plane = [&](const int plane, const bool on, double (&d)[NC], double (&p)[NC], double (&q)[NC],
                          double (&sd)[NC], double (&fr)[NC]) {
        const int64_t o = (int64_t)plane * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t r = rowof(c) + o;
            const bool ok = on && r >= a.slo && r < a.nloc;         // lead rows: upper entries only, diagonal 0
            const double* m = ok ? mat(r) : zm;
            d[c] = j2_ld<NT>(m, true, zm);
            p[c] = j2_ld<NT>(m + S, true, zm);
            q[c] = j2_ld<NT>(m + 2 * S, true, zm);
            sd[c] = j2_ld<NT>(m + 3 * S, true, zm);
            fr[c] = j2_ld<NT>(a.f + r, ok && r >= 0, zf);
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
            // x ring: the cell left of ex = 0 (lane 0) and right of ex = EX-1 (lane 63)
            const int64_t hrow = (hl ? rowof(2 * l) - 1 : rowof(2 * l + 1) + 1) + o;
            const bool okx = on && (hl || hr) && hrow >= a.xlo && hrow < a.xhi;
            const bool okh = okx && hl && hrow >= a.slo && hrow < a.nloc;
            hxv[l] = j2_ld<false>(a.x + hrow, okx, zx);
            hxu[l] = j2_ld<false>((okh ? mat(hrow) : zm) + S, true, zm);
        }
        // y ring: the line below ey = 0 (first wave) and above ey = EY-1 (last wave)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t hrow = (wlo ? rowof(r) - a.nx : rowof(2 * (LPW - 1) + r) + a.nx) + o;
            const bool okx = on && (wlo || whi) && hrow >= a.xlo && hrow < a.xhi;
            const bool okh = okx && wlo && hrow >= a.slo && hrow < a.nloc;
            hyv[r] = j2_ld<false>(a.x + hrow, okx, zx);
            hyu[r] = j2_ld<false>((okh ? mat(hrow) : zm) + 2 * S, true, zm);
        }
    };
    auto load_x = [&](const int plane, const bool on, double (&v)[NC]) {
        const int64_t o = (int64_t)plane * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t r = rowof(c) + o;
            v[c] = j2_ld<false>(a.x + r, on && r >= a.xlo && r < a.xhi, zx);
        }
    };
    // LDS image of one plane: x with its ring, the +1 / +nx diagonals (slot = plane parity) with theirs
    auto park = [&](const int slot, const double (&v)[NC], const double (&p)[NC], const double (&q)[NC]) {
        double* const u1s = sU1 + slot * (EY * PV) - PV;
        double* const u2s = sU2 + slot * ((EY + 1) * EX) + EX;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iv = lvof(c), iw = lwof(c);
            sV0[iv] = v[c];
            u1s[iv] = p[c];
            u2s[iw] = q[c];
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
            const int rowv = (ey0 + l + 1) * PV;
            if (hl) { sV0[rowv] = hxv[l]; u1s[rowv] = hxu[l]; }
            if (hr) sV0[rowv + EX + 1] = hxv[l];
        }
        if (wlo) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { sV0[lane + 64 * r + 1] = hyv[r]; u2s[lane + 64 * r - EX] = hyu[r]; }
        } else if (whi) {
#pragma unroll
            for (int r = 0; r < 2; ++r) sV0[(EY + 1) * PV + lane + 64 * r + 1] = hyv[r];
        }
    };

    // ---- warm-up: everything the first step (k = z0-2) finds in place, in one round of loads ----
    load_plane(z0 - 1, true, d1, p1, q1, s1, f1);
    load_x(z0 - 2, true, va);
    load_x(z0 - 1, true, vb);
    load_x(z0, true, vc);
    {
        const int64_t o = (int64_t)(z0 - 2) * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int64_t r = rowof(c) + o;
            s0[c] = j2_ld<NT>((r >= a.slo && r < a.nloc ? mat(r) : zm) + 3 * S, true, zm);
        }
    }
    park((z0 - 1) & 1, vb, p1, q1);
    __syncthreads();

    for (int k = z0 - 2; k < z1; ++k) {
        // ---- issue the loads of plane k+2 (matrix, f, ring) and of plane k+3 (x) ----
        load_plane(k + 2, k + 2 <= z1, d2, p2, q2, s2, f2);
        load_x(k + 3, k + 3 <= z1 + 1, vd);

        const int sp = (k + 1) & 1;
        const double* const u1p = sU1 + sp * (EY * PV) - PV;          // indexed like sV0
        const double* const u2p = sU2 + sp * ((EY + 1) * EX) + EX;    // indexed like sV1
        // ---- A: first sweep on plane k+1; the second sweep of plane k gets its last (+P) term ----
        {
            const int64_t o1 = (int64_t)(k + 1) * a.P;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const int iv = lvof(c), iw = lwof(c);
                double acc = 0.0;
                acc = fma(s0[c], va[c], acc);                           // -P
                acc = fma(u2p[iw - EX], sV0[iv - PV], acc);             // -nx
                acc = fma(u1p[iv - 1], sV0[iv - 1], acc);               // -1
                const double diag = d1[c] != 0.0 ? d1[c] : 1.0;
                acc = fma(d1[c], vb[c], acc);
                acc = fma(p1[c], sV0[iv + 1], acc);                     // +1
                acc = fma(q1[c], sV0[iv + PV], acc);                    // +nx
                acc = fma(s1[c], vc[c], acc);                           // +P
                const double o = vb[c] + (a.omega * (1.0 / diag)) * (f1[c] - acc);
                const int64_t r1 = rowof(c) + o1;
                w1[c] = (r1 >= 0 && r1 < a.nloc) ? o : 0.0;
                sV1[iw] = w1[c];
                if (inT >> c & 1u) {
                    const int64_t r0 = r1 - a.P;
                    if (k >= z0 && r0 >= a.st_lo && r0 < a.st_hi)
                        a.out[r0] = w0[c] + cf0[c] * (f0[c] - fma(s0[c], w1[c], ap0[c]));
                    if (a.v1out && k + 1 >= z0 && k + 1 < z1 && (r1 < a.k1_lo || r1 >= a.k1_hi)) a.v1out[r1] = w1[c];
                }
            }
        }
        __syncthreads();
        // ---- B: second sweep of plane k+1 up to its +nx term (the neighbours of v1[k+1] are in LDS now) ----
        if (k >= z0 - 1) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                double acc = 0.0;
                if (inT >> c & 1u) {
                    const int iv = lvof(c), iw = lwof(c);
                    acc = fma(s0[c], w0[c], acc);
                    acc = fma(u2p[iw - EX], sV1[iw - EX], acc);
                    acc = fma(u1p[iv - 1], sV1[iw - 1], acc);
                    acc = fma(d1[c], w1[c], acc);
                    acc = fma(p1[c], sV1[iw + 1], acc);
                    acc = fma(q1[c], sV1[iw + EX], acc);
                }
                ap0[c] = acc;
            }
        }
        // ---- C: park plane k+2 (x arrived a step ago, matrix and ring just now); rotate ----
        park(k & 1, vc, p2, q2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double diag = d1[c] != 0.0 ? d1[c] : 1.0;
            cf0[c] = a.omega * (1.0 / diag);
            s0[c] = s1[c]; f0[c] = f1[c]; w0[c] = w1[c];
            d1[c] = d2[c]; p1[c] = p2[c]; q1[c] = q2[c]; s1[c] = s2[c]; f1[c] = f2[c];
            va[c] = vb[c]; vb[c] = vc[c]; vc[c] = vd[c];
        }
        __syncthreads();
    }
}

// ---- row classes ------------------------------------------------------------------------------
// On the meshes this path is built for (uniform grids, constant coefficients) the rows of a level's matrix take
// only a handful of distinct values: the interior stencil, its variants next to a Dirichlet boundary, the
// identity rows.  When the FULL rows -- the seven entries a(-P) a(-nx) a(-1) a(0) a(+1) a(+nx) a(+P) exactly as
// sdia_body reads them: the upper ones from the row's own slots, the lower ones from the slots of the rows
// below -- take at most 255 distinct non-zero values BIT FOR BIT, every kernel that applies the matrix reads one
// byte per row -- its class -- instead of 32 bytes of matrix and looks the entries up in a table of 8 doubles per
// class (CLS_W; slot 7 is spare: the kernels keep omega/diagonal there): 25 instead of 56 bytes per row, with exactly
// the same arithmetic on exactly the same numbers.  A class names the whole row, so no kernel needs a neighbour's
// class.  The dictionary is built on the device (hash insert, compaction in slot order, encode + bitwise
// verification of every row against its entry, histogram); a level with more distinct rows simply has no classes.
// Five-point rows of 2-D levels (stored diagonals {0, +1, +nx}) use the same table with a(-P) = a(+P) = 0.
constexpr int CLS_SLOTS = 4096;         // (CLS_W = 8 doubles per table row: mg_kernels.hip.h)

struct ClsArgs {
    const double* dvals;            // symmetric diagonal storage
    int wu;                         // slots per stored row (3: {0,+1,+nx}; 4: {0,+1,+nx,+P})
    int64_t nloc, mlead;
    int64_t up[4];                  // positive offsets of the slots (up[0] = 0)
    unsigned long long* tags;       // CLS_SLOTS hash tags, 0 = free
    double* svals;                  // CLS_SLOTS x CLS_W
    int* count;                     // distinct non-zero rows seen
    int* slot_class;                // CLS_SLOTS
    double* ctab;                   // 256 x CLS_W
    unsigned char* cls;             // crows: cls[i] is the class of row i - clead (class 0 outside the level)
    int64_t crows, clead;
    int* flag;                      // set when a row does not match its dictionary entry
    unsigned* hist;                 // 256: rows per class
    int qbits;                      // "storage_ulps": entries are compared after rounding away this many low mantissa bits
                                    // (0: bit for bit); a class then stands for rows that agree to about 2^qbits ulps and
                                    // the table holds the entries of the first such row seen
};

// an entry's bits with the lowest q mantissa bits rounded away (q = 0: the bits themselves)
__device__ __forceinline__ unsigned long long cls_quant(unsigned long long b, int q) {
    if (q <= 0) return b;
    const unsigned long long half = 1ull << (q - 1);
    return ((b + half) >> q) << q;
}

__device__ __forceinline__ unsigned long long cls_mix(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}

// the seven entries of row `row` (0 <= row < nloc) in ascending column order
template <int S> __device__ __forceinline__ void cls_row(const ClsArgs& a, int64_t row, unsigned long long (&b)[7]) {
    const int64_t m = row + a.mlead;
    auto at = [&](int64_t mm, int c) -> unsigned long long {
        return (unsigned long long)__double_as_longlong(a.dvals[((size_t)(mm / S) * a.wu + c) * S + (size_t)(mm % S)]);
    };
    b[3] = at(m, 0);
    b[4] = at(m, 1); b[2] = at(m - a.up[1], 1);
    b[5] = at(m, 2); b[1] = at(m - a.up[2], 2);
    b[6] = a.wu > 3 ? at(m, 3) : 0ull;
    b[0] = a.wu > 3 ? at(m - a.up[3], 3) : 0ull;
}

__device__ __forceinline__ bool cls_zero(const unsigned long long (&b)[7]) {
    unsigned long long o = 0ull;
#pragma unroll
    for (int c = 0; c < 7; ++c) o |= b[c];                    // bit patterns: a -0.0 entry makes its own class
    return o == 0ull;
}

__device__ __forceinline__ unsigned long long cls_hash(const unsigned long long (&b)[7], int q) {
    unsigned long long h = 0x9e3779b97f4a7c15ull;
    // (with a tolerance the hash looks at far fewer bits than the comparison allows to differ, so that rows a few ulps
    //  apart almost never fall into different buckets; the comparison in cls_encode then applies the tolerance itself)
    const int qh = q > 0 ? q + 16 : 0;
#pragma unroll
    for (int c = 0; c < 7; ++c) h = cls_mix(h ^ (cls_quant(b[c], qh) + 0x3c6ef372fe94f82bull * (unsigned long long)(c + 1)));
    return h ? h : 1ull;
}

template <int S>
__global__ void cls_insert(ClsArgs a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    unsigned long long b[7];
    cls_row<S>(a, row, b);
    if (cls_zero(b)) return;                                  // class 0: the all-zero row
    const unsigned long long h = cls_hash(b, a.qbits);
    unsigned s = (unsigned)h & (CLS_SLOTS - 1);
    for (int probe = 0; probe < CLS_SLOTS; ++probe) {
        // almost every row finds its class already there: look before touching the slot with an atomic
        const unsigned long long seen = *(volatile unsigned long long*)(a.tags + s);
        if (seen == h) return;
        if (seen != 0ull) { s = (s + 1) & (CLS_SLOTS - 1); continue; }
        if (*(volatile int*)a.count > 255) return;             // too many distinct rows: the caller gives up
        const unsigned long long old = atomicCAS(a.tags + s, 0ull, h);
        if (old == 0ull) {
#pragma unroll
            for (int c = 0; c < 7; ++c) a.svals[CLS_W * s + c] = __longlong_as_double((long long)b[c]);
            atomicAdd(a.count, 1);
            return;
        }
        if (old == h) return;
        s = (s + 1) & (CLS_SLOTS - 1);
    }
}

// one thread: classes 1, 2, ... in slot order (deterministic for a given matrix)
__global__ void cls_assign(ClsArgs a) {
    if (blockIdx.x || threadIdx.x) return;
    for (int c = 0; c < CLS_W; ++c) a.ctab[c] = 0.0;
    int id = 1;
    for (int s = 0; s < CLS_SLOTS; ++s) {
        a.slot_class[s] = 0;
        if (a.tags[s] && id < 256) {
            a.slot_class[s] = id;
            for (int c = 0; c < 7; ++c) a.ctab[CLS_W * id + c] = a.svals[CLS_W * s + c];
            a.ctab[CLS_W * id + 7] = 0.0;
            ++id;
        }
    }
    for (; id < 256; ++id)
        for (int c = 0; c < CLS_W; ++c) a.ctab[CLS_W * id + c] = 0.0;
}

// ---- more than 255 distinct rows: the frequent rows as classes, the others "escape" to their stored row -----------------
// (class CLS_ESCAPE: the K-sweep march reads such a row from the symmetric diagonal storage; the other class kernels leave
// a level with escapes alone.)  Which rows are frequent is found on a SAMPLE spread over the level -- a row that occurs with
// frequency p is in the table after ~1/p samples --, with an occurrence count per slot.  A row enters the table when it is
// seen for the SECOND time (a bit per hash value remembers the first): rows that occur once in the sample -- the odd ones,
// of which a level may have millions -- never get there, so they cannot crowd the frequent rows out.
constexpr int CLS_ESCAPE = 255;
constexpr int CLS_SEEN_BITS = 1 << 22;

template <int S>
__global__ void cls_sample_insert(ClsArgs a, int64_t nsample, unsigned* slot_count, unsigned* seen_bits) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nsample) return;
    const int64_t row = (int64_t)(((unsigned long long)t * 0x9e3779b97f4a7c15ull >> 11) % (unsigned long long)a.nloc);
    unsigned long long b[7];
    cls_row<S>(a, row, b);
    if (cls_zero(b)) return;
    const unsigned long long h = cls_hash(b, a.qbits);
    unsigned s = (unsigned)h & (CLS_SLOTS - 1);
    bool marked = false;        // this row's first sighting has been dealt with
    for (int probe = 0; probe < CLS_SLOTS; ++probe) {
        const unsigned long long tag = *(volatile unsigned long long*)(a.tags + s);
        if (tag == h) { atomicAdd(slot_count + s, 1u); return; }
        if (tag != 0ull) { s = (s + 1) & (CLS_SLOTS - 1); continue; }
        // not in the table (yet)
        if (!marked) {
            const unsigned bit = (unsigned)(h >> 32) & (CLS_SEEN_BITS - 1), mask = 1u << (bit & 31);
            if (!(atomicOr(seen_bits + (bit >> 5), mask) & mask)) return;       // the first sighting only leaves its mark
            marked = true;
        }
        if (*(volatile int*)a.count >= (CLS_SLOTS * 3) / 4) return;           // table as full as it gets
        const unsigned long long old = atomicCAS(a.tags + s, 0ull, h);
        if (old == 0ull) {
#pragma unroll
            for (int c = 0; c < 7; ++c) a.svals[CLS_W * s + c] = __longlong_as_double((long long)b[c]);
            atomicAdd(a.count, 1);
            atomicAdd(slot_count + s, 1u);
            return;
        }
        if (old == h) { atomicAdd(slot_count + s, 1u); return; }
        s = (s + 1) & (CLS_SLOTS - 1);
    }
}

// every row: its class if its row is one of the chosen ones (slot_class, bit for bit or within the tolerance), else CLS_ESCAPE
template <int S>
__global__ __launch_bounds__(256) void cls_encode_escape(ClsArgs a) {
    __shared__ unsigned s_hist[256];
    s_hist[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.crows) {
        const int64_t row = i - a.clead;
        int id = 0;
        if (row >= 0 && row < a.nloc) {
            unsigned long long b[7];
            cls_row<S>(a, row, b);
            if (!cls_zero(b)) {
                id = CLS_ESCAPE;
                const unsigned long long h = cls_hash(b, a.qbits);
                unsigned s = (unsigned)h & (CLS_SLOTS - 1);
                for (int probe = 0; probe < CLS_SLOTS; ++probe) {
                    const unsigned long long t = a.tags[s];
                    if (t == h) {
                        bool same = true;
#pragma unroll
                        for (int c = 0; c < 7; ++c) {
                            const long long sv = __double_as_longlong(a.svals[CLS_W * s + c]), bv = (long long)b[c];
                            const long long d = sv - bv;
                            same = same && (sv == bv || (a.qbits > 0 && (sv < 0) == (bv < 0) && (d < 0 ? -d : d) <= (1ll << a.qbits)));
                        }
                        if (same && a.slot_class[s] != 0) id = a.slot_class[s];
                        break;
                    }
                    if (t == 0ull) break;
                    s = (s + 1) & (CLS_SLOTS - 1);
                }
            }
            atomicAdd(&s_hist[id], 1u);
        }
        a.cls[i] = (unsigned char)id;
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(a.hist + threadIdx.x, s_hist[threadIdx.x]);
}

template <int S>
__global__ __launch_bounds__(256) void cls_encode(ClsArgs a) {
    // (more than 255 distinct rows: the insert pass may have filled the table to the brim before its threads saw the count, and a
    //  row that is not in a full table walks all of it -- seconds on a level of 10^9 rows; the caller discards the encoding anyway)
    if (*a.count > 255) {
        if (blockIdx.x == 0 && threadIdx.x == 0) *a.flag = 1;
        return;
    }
    __shared__ unsigned s_hist[256];
    s_hist[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.crows) {
        const int64_t row = i - a.clead;
        int id = 0;
        if (row >= 0 && row < a.nloc) {
            unsigned long long b[7];
            cls_row<S>(a, row, b);
            if (!cls_zero(b)) {
                const unsigned long long h = cls_hash(b, a.qbits);
                unsigned s = (unsigned)h & (CLS_SLOTS - 1);
                bool found = false;
                for (int probe = 0; probe < CLS_SLOTS; ++probe) {
                    const unsigned long long t = a.tags[s];
                    if (t == h) {
                        bool same = true;
#pragma unroll
                        for (int c = 0; c < 7; ++c)
                        {
                            const long long sv = __double_as_longlong(a.svals[CLS_W * s + c]), bv = (long long)b[c];
                            const long long d = sv - bv;
                            same = same && (sv == bv || (a.qbits > 0 && (sv < 0) == (bv < 0) && (d < 0 ? -d : d) <= (1ll << a.qbits)));
                        }
                        id = a.slot_class[s];
                        found = same && id != 0;                    // hash collision / overflow: no classes
                        break;
                    }
                    if (t == 0ull) break;
                    s = (s + 1) & (CLS_SLOTS - 1);
                }
                if (!found) { atomicExch(a.flag, 1); id = 0; }
            }
            atomicAdd(&s_hist[id], 1u);
        }
        a.cls[i] = (unsigned char)id;
    }
    __syncthreads();
    if (s_hist[threadIdx.x]) atomicAdd(a.hist + threadIdx.x, s_hist[threadIdx.x]);
}

// ---- the two-sweep pass on class-coded rows ----------------------------------------------------
// Same march, same arithmetic as j2_body; per cell and step it loads one class byte, f and x (17 bytes instead of
// 48).  A class names the whole row (seven entries + omega/diagonal, one IEEE division per class and workgroup
// instead of two per cell and step), so there is nothing to look up about the neighbours:
//   * waves whose 64 cells of a line all carry the level's most frequent class (`cmain`: the interior stencil,
//     > 99 % of the rows of a large level) take the entries from kernel arguments (SGPRs): no table access at all;
//   * other cells read their row with four 16-byte LDS loads from the table.
// The plane images are double-buffered (x and the once-relaxed iterate: plane mod 2) and the WHOLE second sweep of
// plane k is evaluated in the step that relaxes plane k+1 once, from the image of v1[k] written a step earlier --
// a step only reads what earlier steps wrote, so there is one barrier per plane.
//
// Tile geometry along x.  All tiles of a level are alike (tiles that run at different speeds fall out of step with
// their neighbours and the cells they share are then fetched twice; measured +10 % with one narrower column): a
// tile's 128 cells per line are  [ring | W1 first-sweep cells | ring | idle],  W1 = WI + 2 <= 126, of which the
// inner WI = ceil(nx / ntx) get their second sweep, so that ntx columns cover the nx grid columns without a nearly
// empty last one (nx = 1025: 9 x 114 instead of 9 x 126).  The two ring cells are ordinary cells whose first-sweep
// value nobody uses; idle cells are read at the address of the right ring cell (the same memory request: no traffic).
//
// What the compiler must not do to the software pipeline (both were measured, both cost > 20 %):
//   * plane loads through FLAT pointers -- a pointer rebuilt from integers is flat unless its address space is spelled
//     out, and flat loads count as LDS operations too (lgkmcnt, out of order): every wait for an LDS read then waits
//     for every global load in flight;
//   * taking over the loaded values at the END of the loop body (where phi elimination puts the copies), behind the
//     next loads: the copies wait for the loads just issued.  Hence unconditional loads (nothing under an `if`) and the
//     empty asm statements that pin the take-over before the loads.
template <int NW, int LPW> constexpr size_t j2c_lds_bytes() {
    constexpr int EY = NW * LPW;
    return sizeof(double) * (256 * CLS_W + 2 * (size_t)(EY + 2) * J2_EX + 2 * (size_t)EY * J2_EX + 2 * (J2_EX + 2));
}

typedef const __attribute__((address_space(1))) char* gcptr_t;

template <int NW, int LPW>
__device__ __forceinline__ void j2c_body(const J2Args& a) {
    constexpr int EX = J2_EX, EY = NW * LPW, NC = 2 * LPW;
    extern __shared__ double j2_smem[];
    constexpr int V0S = (EY + 2) * EX, V1S = EY * EX;
    double* const sT = j2_smem;                           // 256 x 8   entries of the row classes, [7] = omega / diagonal
    double* const sV0 = sT + 256 * CLS_W + (EX + 2);      // 2 x (EY+2) x EX   x of a plane, origin (0,-1)
    double* const sV1 = sV0 + 2 * V0S;                    // 2 x EY x EX       once-relaxed iterate of a plane
    // (cell 0 of the first line reads one element in front of an image, cells of the last line one line behind it:
    //  EX + 2 elements of slack on both ends of the pair of arrays; such cells store nothing)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // scalar: branches on it are s_cbranch

    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3, ch = a.xcd_chunk;
        id = ((j / ch) * 8u + xcd) * ch + (j % ch);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int seg = a.seg0 + (int)(id / ntile) * a.seg_stride;
    const unsigned t = id % ntile;
    const int tiy = (int)(t / (unsigned)a.ntx), tix = (int)(t % (unsigned)a.ntx);
    int z0, z1;
    if (seg == 0) { z0 = 0; z1 = min(a.zb, a.nz); }
    else if (seg == a.nseg - 1) { z0 = max(a.nz - a.zb, a.zb); z1 = a.nz; }
    else { z0 = a.zb + (seg - 1) * a.seglen; z1 = min(a.nz - a.zb, z0 + a.seglen); }
    if (z1 <= z0) return;
    const int wi = a.wi, w1 = wi + 2;                     // cells with a second / a first sweep per line
    const int tx0 = tix * wi - 2, ty0 = tiy * (EY - 2) - 1;           // grid position of cell (0, 0)

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
    }
    // the most frequent class: entries straight from the kernel arguments
    const double m0 = a.cm[0], m1 = a.cm[1], m2 = a.cm[2], m3 = a.cm[3], m4 = a.cm[4], m5 = a.cm[5], m6 = a.cm[6];
    const double mcf = a.omega * (1.0 / (m3 != 0.0 ? m3 : 1.0));
    const int cmain = a.cmain;

    // cell c = 2*l + r of this thread: ex = lane + 64 r, ey = wave*LPW + l
    const int ey0 = wave * LPW;
    const int lw0 = ey0 * EX + lane;                      // cell 0 in sV1; in sV0 it is lw0 + EX
    auto lwof = [&](int c) -> int { return lw0 + (c >> 1) * EX + 64 * (c & 1); };
    unsigned inT = 0;           // bit c: this lane's cell c is an interior cell whose second sweep this tile stores
    unsigned lineT = 0;         // bit c (wave-uniform): the 64 cells c of this wave hold interior cells at all
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ex = lane + 64 * (c & 1), ey = ey0 + (c >> 1);
        const bool line = ey >= 1 && ey < EY - 1 && ty0 + ey < a.ny && 64 * (c & 1) < w1;
        if (line) lineT |= 1u << c;
        if (line && ex >= 2 && ex < w1 && tx0 + ex < a.nx) inT |= 1u << c;
    }
    const bool wlo = wave == 0, whi = wave == NW - 1;

    // Addresses.  Every load of a plane is `uniform base of the plane + an element offset fixed for the whole march`:
    // no per-lane predicates, no address arithmetic in the loop.  That needs every address to be readable whether or
    // not its row exists: the vectors carry zero slack of a plane + 2 lines on both sides (vec_reach), the class array
    // is padded alike with class 0 (the all-zero row), planes -1 and nz are read like any other and planes beyond
    // them are read at the nearest of these.  Grid lines from ny+2 on (tiles that stick out of the grid; their cells
    // feed no result) are read at line ny+1 so that the slack suffices.  x, f and the class of a cell share one offset.
    const unsigned bias = 2u * (unsigned)a.nx + 2u;
    unsigned eo[NC], eor[2];                                 // the cells; the cells of the y ring's line
#pragma unroll
    for (int c = 0; c < NC; ++c)
        eo[c] = (unsigned)((int64_t)min(ty0 + ey0 + (c >> 1), a.ny + 1) * a.nx + tx0 + min(lane + 64 * (c & 1), w1 + 1) + (int64_t)bias);
#pragma unroll
    for (int r = 0; r < 2; ++r)
        eor[r] = (unsigned)((int64_t)(wlo ? ty0 - 1 : min(ty0 + EY, a.ny + 1)) * a.nx + tx0 + min(lane + 64 * r, w1 + 1) + (int64_t)bias);
    const unsigned char* const clsb = a.cls + a.clead - bias;       // + plane*P: class of element offset 0
    const double* const xb0 = a.x - bias;
    const double* const fb0 = a.f - bias;
    // the plane bases are made opaque scalars (readfirstlane) so that they live in SGPRs
    auto sbase = [](const void* p) -> gcptr_t {
        const unsigned long long u = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
        return (gcptr_t)(((unsigned long long)hi << 32) | lo);
    };
    auto ldd = [](gcptr_t b, unsigned e) -> double {
        return *(const __attribute__((address_space(1))) double*)(b + ((unsigned long long)e << 3));
    };
    auto ldc = [](gcptr_t b, unsigned e) -> int {
        return *(const __attribute__((address_space(1))) unsigned char*)(b + (unsigned long long)e);
    };

    // registers per cell: x of planes k, k+1, k+2 (va vb vc) and k+3 (vd, in flight); f and the class of planes k, k+1
    // and k+2 (in flight, with the y ring of x in plane k+2); the once-relaxed iterate of planes k-1, k, k+1 (wm w0 w1)
    int c0[NC], c1[NC], c2[NC];
    double f0[NC], f1[NC], f2[NC], wm[NC], w0[NC], wn[NC], va[NC], vb[NC], vc[NC], vd[NC];
    double hy[2];
    unsigned fast = 0;          // wave-uniform; bit c: all 64 cells c of plane k are of class cmain; bit NC + c: plane k+1
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        f0[c] = f1[c] = f2[c] = wm[c] = w0[c] = wn[c] = va[c] = vb[c] = vc[c] = vd[c] = 0.0;
        c0[c] = c1[c] = c2[c] = 0;
    }
    hy[0] = hy[1] = 0.0;

    // Loads never branch (a load under a condition makes the compiler carry its result through copies, and a copy
    // waits for the load): a plane outside [-1, nz] -- beyond the zero slack -- is read at the nearest plane inside;
    // what such a plane contributes only reaches rows outside the level, whose results are discarded.
    // classes + f of a plane, and the y ring of x in that plane (the line below ey = 0 / above ey = EY-1: only the first
    // and the last wave's exist; the branch is scalar)
    auto load_cf = [&](const int plane, int (&cc)[NC], double (&fr)[NC]) {
        const int64_t o = (int64_t)min(max(plane, -1), a.nz) * a.P;
        const gcptr_t cb = sbase(clsb + o);
        const gcptr_t fb = sbase(fb0 + o);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cc[c] = ldc(cb, eo[c]);
            fr[c] = ldd(fb, eo[c]);
        }
        if (wlo || whi) {
            const gcptr_t xb = sbase(xb0 + o);
#pragma unroll
            for (int r = 0; r < 2; ++r) hy[r] = ldd(xb, eor[r]);
        }
    };
    auto load_x = [&](const int plane, double (&v)[NC]) {
        const gcptr_t xb = sbase(xb0 + (int64_t)min(max(plane, -1), a.nz) * a.P);
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = ldd(xb, eo[c]);
    };
    // LDS image of plane `plane` of x (cells + y ring)
    auto park = [&](const int plane, const double (&v)[NC]) {
        double* const xs = sV0 + (plane & 1) * V0S;
#pragma unroll
        for (int c = 0; c < NC; ++c) xs[lwof(c) + EX] = v[c];
        if (wlo || whi) {
            const int rowv = wlo ? 0 : (EY + 1) * EX;
#pragma unroll
            for (int r = 0; r < 2; ++r) xs[rowv + lane + 64 * r] = hy[r];
        }
    };
    auto all_main = [&](const int (&cc)[NC]) -> unsigned {
        unsigned m = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (__builtin_amdgcn_readfirstlane((int)(__ballot(cc[c] != cmain) == 0ull))) m |= 1u << c;
        return m;
    };
    const int64_t rb0 = (int64_t)(ty0 + ey0) * a.nx + (tx0 + lane);
    auto rowof = [&](int c) -> int64_t { return rb0 + (int64_t)(c >> 1) * a.nx + 64 * (c & 1); };

    // ---- warm-up: what step k = z0-2 finds in place ----
    load_x(z0 - 2, va);
    load_x(z0 - 1, vb);
    load_x(z0, vc);
    load_cf(z0 - 2, c0, f0);
    load_cf(z0 - 1, c1, f1);                // (with the y ring of plane z0-1)
    park(z0 - 1, vb);
    fast = all_main(c0) | (all_main(c1) << NC);
    // the loads of a step are issued BEFORE the barrier that ends the step before it: the time the waves spend at
    // the barrier is flight time too
    load_cf(z0, c2, f2);
    load_x(z0 + 1, vd);
    __syncthreads();

    for (int k = z0 - 2; k < z1; ++k) {
        const double* const x1 = sV0 + ((k + 1) & 1) * V0S + EX;      // x of plane k+1, indexed like sV1
        double* const v1w = sV1 + ((k + 1) & 1) * V1S;                // v1 of plane k+1 (written here)
        const double* const v1r = sV1 + (k & 1) * V1S;                // v1 of plane k (written a step ago)
        const int64_t o1 = (int64_t)(k + 1) * a.P;
        const bool second = k >= z0;
        const bool keep1 = a.v1out != nullptr && k + 1 >= z0 && k + 1 < z1;
        // ---- first sweep on plane k+1 ----
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = lwof(c);
            const double xs = x1[iw - EX], xw = x1[iw - 1], xe = x1[iw + 1], xn = x1[iw + EX];
            double o;
            if (fast >> (NC + c) & 1u) {
                double acc = 0.0;
                acc = fma(m0, va[c], acc);                           // -P
                acc = fma(m1, xs, acc);                              // -nx
                acc = fma(m2, xw, acc);                              // -1
                acc = fma(m3, vb[c], acc);
                acc = fma(m4, xe, acc);                              // +1
                acc = fma(m5, xn, acc);                              // +nx
                acc = fma(m6, vc[c], acc);                           // +P
                o = vb[c] + mcf * (f1[c] - acc);
            } else {
                const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * c1[c]);
                const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                double acc = 0.0;
                acc = fma(t01.x, va[c], acc);
                acc = fma(t01.y, xs, acc);
                acc = fma(t23.x, xw, acc);
                acc = fma(t23.y, vb[c], acc);
                acc = fma(t45.x, xe, acc);
                acc = fma(t45.y, xn, acc);
                acc = fma(t67.x, vc[c], acc);
                o = vb[c] + t67.y * (f1[c] - acc);
            }
            const int64_t r1 = rowof(c) + o1;
            wn[c] = (r1 >= 0 && r1 < a.nloc) ? o : 0.0;
            v1w[iw] = wn[c];
            if (keep1 && (inT >> c & 1u) && (r1 < a.k1_lo || r1 >= a.k1_hi)) a.v1out[r1] = wn[c];
        }
        // ---- second sweep on plane k: its in-plane neighbours were written a step ago ----
        if (second) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!(lineT >> c & 1u)) continue;
                const int iw = lwof(c);
                const double ys = v1r[iw - EX], yw = v1r[iw - 1], ye = v1r[iw + 1], yn = v1r[iw + EX];
                double o;
                if (fast >> c & 1u) {
                    double acc = 0.0;
                    acc = fma(m0, wm[c], acc);
                    acc = fma(m1, ys, acc);
                    acc = fma(m2, yw, acc);
                    acc = fma(m3, w0[c], acc);
                    acc = fma(m4, ye, acc);
                    acc = fma(m5, yn, acc);
                    acc = fma(m6, wn[c], acc);
                    o = w0[c] + mcf * (f0[c] - acc);
                } else {
                    const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * c0[c]);
                    const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                    double acc = 0.0;
                    acc = fma(t01.x, wm[c], acc);
                    acc = fma(t01.y, ys, acc);
                    acc = fma(t23.x, yw, acc);
                    acc = fma(t23.y, w0[c], acc);
                    acc = fma(t45.x, ye, acc);
                    acc = fma(t45.y, yn, acc);
                    acc = fma(t67.x, wn[c], acc);
                    o = w0[c] + t67.y * (f0[c] - acc);
                }
                const int64_t r0 = rowof(c) + o1 - a.P;
                if ((inT >> c & 1u) && r0 >= a.st_lo && r0 < a.st_hi) a.out[r0] = o;
            }
        }
        // ---- park plane k+2, rotate, issue the loads of the step after the next ----
        park(k + 2, vc);
        fast = (fast >> NC) | (all_main(c2) << NC);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            c0[c] = c1[c]; c1[c] = c2[c];
            f0[c] = f1[c]; f1[c] = f2[c]; wm[c] = w0[c]; w0[c] = wn[c];
            va[c] = vb[c]; vb[c] = vc[c]; vc[c] = vd[c];
            // The values that arrived (classes, f, x of the planes ahead) are taken over HERE, before the registers they
            // arrived in are handed to the next loads (see the comment above the kernel).
            asm volatile("" : "+v"(c1[c]));
            asm volatile("" : "+v"(f1[c]));
            asm volatile("" : "+v"(vc[c]));
        }
        load_cf(k + 3, c2, f2);
        load_x(k + 4, vd);
        __syncthreads();
    }
}

// ---- the two-sweep pass on the stored rows, round-2 structure (levels without row classes) ---------------------
// j2_body above is round 1's pass; its address selects (`ok ? p : zero`) under lane conditions were compiled into
// exec-masked branches with a full vmcnt(0) inside, and the take-over copies sat behind the next loads.  This is the
// same march in the structure of j2c_body: the x ring is part of the tile (126 first-sweep cells, 124 with a second
// sweep per 128-cell line), every load is unconditional (rows outside the level read the stored zeros at `zero`: an
// address select, no branch), loads are issued a step ahead and taken over before the next ones are issued.  The
// lower entries of a row are its neighbours' upper entries: -P from the cell's own previous plane (register), -1 / -nx
// through LDS images of the +1 / +nx diagonals of the plane being relaxed once; the two values a cell reads there are
// kept in registers for its second sweep one step later, so the images need two slots only.
// Same entries, same fma order, same IEEE division as sdia_body: bit-identical to two single sweeps.
template <int NW, int LPW> constexpr size_t j2p_lds_bytes() {
    constexpr int EY = NW * LPW;
    return sizeof(double) * (2 * (size_t)(EY + 2) * J2_EX + 2 * (size_t)EY * J2_EX + 2 * (size_t)EY * J2_EX +
                             2 * (size_t)(EY + 1) * J2_EX + 4 * (J2_EX + 2));
}

template <int R, int NW, int LPW>
__device__ __forceinline__ void j2p_body(const J2Args& a) {
    constexpr int S = WAVE * R, EX = J2_EX, EY = NW * LPW, NC = 2 * LPW;
    extern __shared__ double j2_smem[];
    constexpr int V0S = (EY + 2) * EX, V1S = EY * EX, QS = (EY + 1) * EX;
    double* const sV0 = j2_smem + (EX + 2);               // 2 x (EY+2) x EX   x of a plane, origin (0,-1)
    double* const sV1 = sV0 + 2 * V0S;                    // 2 x EY x EX       once-relaxed iterate of a plane
    double* const sP = sV1 + 2 * V1S + (EX + 2);          // 2 x EY x EX       +1 diagonal of a plane
    double* const sQ = sP + 2 * V1S + (EX + 2);           // 2 x (EY+1) x EX   +nx diagonal of a plane, origin (0,-1)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3, ch = a.xcd_chunk;
        id = ((j / ch) * 8u + xcd) * ch + (j % ch);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int seg = a.seg0 + (int)(id / ntile) * a.seg_stride;
    const unsigned t = id % ntile;
    const int tiy = (int)(t / (unsigned)a.ntx), tix = (int)(t % (unsigned)a.ntx);
    int z0, z1;
    if (seg == 0) { z0 = 0; z1 = min(a.zb, a.nz); }
    else if (seg == a.nseg - 1) { z0 = max(a.nz - a.zb, a.zb); z1 = a.nz; }
    else { z0 = a.zb + (seg - 1) * a.seglen; z1 = min(a.nz - a.zb, z0 + a.seglen); }
    if (z1 <= z0) return;
    const int wi = a.wi, w1 = wi + 2;
    const int tx0 = tix * wi - 2, ty0 = tiy * (EY - 2) - 1;

    const int ey0 = wave * LPW;
    const int lw0 = ey0 * EX + lane;
    auto lwof = [&](int c) -> int { return lw0 + (c >> 1) * EX + 64 * (c & 1); };
    unsigned inT = 0, lineT = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ex = lane + 64 * (c & 1), ey = ey0 + (c >> 1);
        const bool line = ey >= 1 && ey < EY - 1 && ty0 + ey < a.ny && 64 * (c & 1) < w1;
        if (line) lineT |= 1u << c;
        if (line && ex >= 2 && ex < w1 && tx0 + ex < a.nx) inT |= 1u << c;
    }
    const bool wlo = wave == 0, whi = wave == NW - 1;

    // rows of the cells in plane 0 (grid lines from ny+2 on are read at line ny+1, see j2c_body); the y ring's line
    int rw[NC], rwr[2];                     // (a level has fewer than 2^31 rows)
#pragma unroll
    for (int c = 0; c < NC; ++c)
        rw[c] = (int)((int64_t)min(ty0 + ey0 + (c >> 1), a.ny + 1) * a.nx + tx0 + min(lane + 64 * (c & 1), w1 + 1));
#pragma unroll
    for (int r = 0; r < 2; ++r)
        rwr[r] = (int)((int64_t)(wlo ? ty0 - 1 : min(ty0 + EY, a.ny + 1)) * a.nx + tx0 + min(lane + 64 * r, w1 + 1));
    const int64_t rb0 = (int64_t)(ty0 + ey0) * a.nx + (tx0 + lane);
    auto rowof = [&](int c) -> int64_t { return rb0 + (int64_t)(c >> 1) * a.nx + 64 * (c & 1); };
    // the diagonal slot of a stored row, or the stored zeros for rows the level does not have
    auto mat = [&](int64_t row) -> const double* {
        const uint64_t m = (uint64_t)(row + a.mlead);
        const double* p = a.vals + (size_t)(m / S) * (4 * S) + (size_t)(m % S);
        return (row >= a.slo && row < a.nloc) ? p : a.zero;
    };
    auto xat = [&](int64_t row) -> const double* { return (row >= a.xlo && row < a.xhi) ? a.x + row : a.zero; };
    auto fat = [&](int64_t row) -> const double* { return (row >= 0 && row < a.nloc) ? a.f + row : a.zero; };

    // plane k: matrix row, f, the two lower entries read at its first sweep, the +P entry of plane k-1;
    // plane k+1; plane k+2 (in flight); x of planes k .. k+3; the once-relaxed iterate of planes k-1 .. k+1
    double d0[NC], p0[NC], q0[NC], s0[NC], f0[NC], pw0[NC], qs0[NC], tm[NC];      // tm = (+P entry of plane k-1) * v1[k-1]
    double d1[NC], p1[NC], q1[NC], s1[NC], f1[NC];
    double d2[NC], p2[NC], q2[NC], s2[NC], f2[NC];
    double va[NC], vb[NC], vc[NC], vd[NC], w0[NC], wn[NC];
    double hy[2], hq[2];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        d0[c] = p0[c] = q0[c] = s0[c] = f0[c] = pw0[c] = qs0[c] = tm[c] = w0[c] = wn[c] = 0.0;
    }
    hy[0] = hy[1] = hq[0] = hq[1] = 0.0;

    // matrix rows + f of a plane, and the y ring of that plane (x above / below the tile; the +nx entries of the line below)
    auto load_m = [&](const int plane, double (&d)[NC], double (&p)[NC], double (&q)[NC], double (&sd)[NC], double (&fr)[NC]) {
        const int64_t o = (int64_t)min(max(plane, -1), a.nz) * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double* m = mat(rw[c] + o);
            d[c] = m[0]; p[c] = m[S]; q[c] = m[2 * S]; sd[c] = m[3 * S];
            fr[c] = *fat(rw[c] + o);
        }
        if (wlo || whi) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                hy[r] = *xat(rwr[r] + o);
                hq[r] = mat(rwr[r] + o)[2 * S];
            }
        }
    };
    auto load_x = [&](const int plane, double (&v)[NC]) {
        const int64_t o = (int64_t)min(max(plane, -1), a.nz) * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = *xat(rw[c] + o);
    };
    // LDS images of a plane: x (with the y ring), the +1 and +nx diagonals (the latter with the line below the tile)
    auto park = [&](const int plane, const double (&v)[NC], const double (&p)[NC], const double (&q)[NC]) {
        const int slot = plane & 1;
        double* const xs = sV0 + slot * V0S;
        double* const ps = sP + slot * V1S;
        double* const qs = sQ + slot * QS;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = lwof(c);
            xs[iw + EX] = v[c];
            ps[iw] = p[c];
            qs[iw + EX] = q[c];
        }
        if (wlo) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { xs[lane + 64 * r] = hy[r]; qs[lane + 64 * r] = hq[r]; }
        } else if (whi) {
#pragma unroll
            for (int r = 0; r < 2; ++r) xs[(EY + 1) * EX + lane + 64 * r] = hy[r];
        }
    };

    // ---- warm-up: what step k = z0-2 finds in place ----
    load_x(z0 - 2, va);
    load_x(z0 - 1, vb);
    load_x(z0, vc);
    {
        const int64_t o = (int64_t)min(max(z0 - 2, -1), a.nz) * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) s0[c] = mat(rw[c] + o)[3 * S];        // +P entries of plane z0-2
    }
    load_m(z0 - 1, d1, p1, q1, s1, f1);
    park(z0 - 1, vb, p1, q1);
    load_m(z0, d2, p2, q2, s2, f2);
    load_x(z0 + 1, vd);
    __syncthreads();

    for (int k = z0 - 2; k < z1; ++k) {
        const int s1i = (k + 1) & 1;
        const double* const x1 = sV0 + s1i * V0S + EX;
        const double* const pim = sP + s1i * V1S;
        const double* const qim = sQ + s1i * QS + EX;
        double* const v1w = sV1 + s1i * V1S;
        const double* const v1r = sV1 + (k & 1) * V1S;
        const int64_t o1 = (int64_t)(k + 1) * a.P;
        const bool second = k >= z0;
        const bool keep1 = a.v1out != nullptr && k + 1 >= z0 && k + 1 < z1;
        double pw1[NC], qs1[NC];
        // ---- first sweep on plane k+1 ----
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = lwof(c);
            const double xs = x1[iw - EX], xw = x1[iw - 1], xe = x1[iw + 1], xn = x1[iw + EX];
            pw1[c] = pim[iw - 1];
            qs1[c] = qim[iw - EX];
            double acc = 0.0;
            acc = fma(s0[c], va[c], acc);                               // -P
            acc = fma(qs1[c], xs, acc);                                 // -nx
            acc = fma(pw1[c], xw, acc);                                 // -1
            const double diag = d1[c] != 0.0 ? d1[c] : 1.0;
            acc = fma(d1[c], vb[c], acc);
            acc = fma(p1[c], xe, acc);                                  // +1
            acc = fma(q1[c], xn, acc);                                  // +nx
            acc = fma(s1[c], vc[c], acc);                               // +P
            const double o = vb[c] + (a.omega * (1.0 / diag)) * (f1[c] - acc);
            const int64_t r1 = rowof(c) + o1;
            wn[c] = (r1 >= 0 && r1 < a.nloc) ? o : 0.0;
            v1w[iw] = wn[c];
            if (keep1 && (inT >> c & 1u) && (r1 < a.k1_lo || r1 >= a.k1_hi)) a.v1out[r1] = wn[c];
        }
        // ---- second sweep on plane k ----
        if (second) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (!(lineT >> c & 1u)) continue;
                const int iw = lwof(c);
                const double ys = v1r[iw - EX], yw = v1r[iw - 1], ye = v1r[iw + 1], yn = v1r[iw + EX];
                double acc = tm[c];                                         // fma(s[k-1], v1[k-1], 0), taken a step ago
                acc = fma(qs0[c], ys, acc);
                acc = fma(pw0[c], yw, acc);
                const double diag = d0[c] != 0.0 ? d0[c] : 1.0;
                acc = fma(d0[c], w0[c], acc);
                acc = fma(p0[c], ye, acc);
                acc = fma(q0[c], yn, acc);
                acc = fma(s0[c], wn[c], acc);
                const double o = w0[c] + (a.omega * (1.0 / diag)) * (f0[c] - acc);
                const int64_t r0 = rowof(c) + o1 - a.P;
                if ((inT >> c & 1u) && r0 >= a.st_lo && r0 < a.st_hi) a.out[r0] = o;
            }
        }
        // ---- park plane k+2, rotate, issue the loads of the step after the next ----
        park(k + 2, vc, p2, q2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            tm[c] = fma(s0[c], w0[c], 0.0);
            d0[c] = d1[c]; p0[c] = p1[c]; q0[c] = q1[c]; s0[c] = s1[c]; f0[c] = f1[c]; pw0[c] = pw1[c]; qs0[c] = qs1[c];
            d1[c] = d2[c]; p1[c] = p2[c]; q1[c] = q2[c]; s1[c] = s2[c]; f1[c] = f2[c];
            w0[c] = wn[c];
            va[c] = vb[c]; vb[c] = vc[c]; vc[c] = vd[c];
            asm volatile("" : "+v"(d1[c]));
            asm volatile("" : "+v"(p1[c]));
            asm volatile("" : "+v"(q1[c]));
            asm volatile("" : "+v"(s1[c]));
            asm volatile("" : "+v"(f1[c]));
            asm volatile("" : "+v"(vc[c]));
        }
        load_m(k + 3, d2, p2, q2, s2, f2);
        load_x(k + 4, vd);
        __syncthreads();
    }
}

template <int R, int NW, int LPW>
__global__ __launch_bounds__(NW * WAVE) void sdia_jacobi2p(J2Args a) {
    j2p_body<R, NW, LPW>(a);
}

template <int R, int NW, int LPW>
__global__ __launch_bounds__(NW * WAVE) void sdia_jacobi2p_finest(J2Args a) {
    j2p_body<R, NW, LPW>(a);
}

// ---- ONE sweep as a plane march on class-coded rows (residual, single Jacobi sweeps, Gauss-Seidel colours) -------
// The slice kernels (sdia_cls_body) read a row's six x neighbours as unaligned 16-byte loads through L1; here the
// pipeline of the pair pass is used for one sweep: a tile of 126 x EY result cells (+ the x ring inside the tile's
// 128 cells per line, the y ring from the edge waves) marches through the planes, x of plane k in an LDS image,
// x of planes k-1 / k+1 in registers, loads a step ahead.  Same entries, same order as sdia_cls_body: bit-identical.
// MODE_GS relaxes the rows of one colour in place (out == x): a row only reads unknowns of other colours, which no
// workgroup writes during the launch.
template <int NW, int LPW> constexpr size_t j1c_lds_bytes() {
    return sizeof(double) * (256 * CLS_W + 2 * (size_t)(NW * LPW + 2) * J2_EX + 2 * (J2_EX + 2));
}

template <int NW, int LPW, int MODE>
__device__ __forceinline__ void j1c_body(const J2Args& a) {
    constexpr int EX = J2_EX, EY = NW * LPW, NC = 2 * LPW;
    extern __shared__ double j2_smem[];
    constexpr int V0S = (EY + 2) * EX;
    double* const sT = j2_smem;                           // 256 x 8   entries of the row classes, [7] = omega / diagonal
    double* const sV0 = sT + 256 * CLS_W + (EX + 2);      // 2 x (EY+2) x EX   x of a plane, origin (0,-1)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));

    unsigned id;
    {
        const unsigned b = blockIdx.x, xcd = b & 7u, j = b >> 3, ch = a.xcd_chunk;
        id = ((j / ch) * 8u + xcd) * ch + (j % ch);
    }
    if (id >= a.nitems) return;
    const unsigned ntile = (unsigned)(a.ntx * a.nty);
    const int seg = (int)(id / ntile);
    const unsigned t = id % ntile;
    const int tiy = (int)(t / (unsigned)a.ntx), tix = (int)(t % (unsigned)a.ntx);
    const int z0 = seg * a.seglen, z1 = min(a.nz, z0 + a.seglen);
    if (z1 <= z0) return;
    constexpr int w1 = EX - 2;                            // result cells per line: ex = 1 .. w1
    const int tx0 = tix * w1 - 1, ty0 = tiy * EY;         // grid position of cell (0, 0)

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
    }
    const double m0 = a.cm[0], m1 = a.cm[1], m2 = a.cm[2], m3 = a.cm[3], m4 = a.cm[4], m5 = a.cm[5], m6 = a.cm[6];
    const double mcf = a.omega * (1.0 / (m3 != 0.0 ? m3 : 1.0));
    const int cmain = a.cmain;

    const int ey0 = wave * LPW;
    const int lw0 = ey0 * EX + lane;
    auto lwof = [&](int c) -> int { return lw0 + (c >> 1) * EX + 64 * (c & 1); };
    unsigned inT = 0;           // bit c: this lane's cell c is a result cell on the grid
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ex = lane + 64 * (c & 1), ey = ey0 + (c >> 1);
        if (ex >= 1 && ex <= w1 && tx0 + ex < a.nx && ty0 + ey < a.ny) inT |= 1u << c;
    }
    const bool wlo = wave == 0, whi = wave == NW - 1;

    const unsigned bias = 2u * (unsigned)a.nx + 2u;
    unsigned eo[NC], eor[2];
#pragma unroll
    for (int c = 0; c < NC; ++c)
        eo[c] = (unsigned)((int64_t)min(ty0 + ey0 + (c >> 1), a.ny + 1) * a.nx + tx0 + lane + 64 * (c & 1) + (int64_t)bias);
#pragma unroll
    for (int r = 0; r < 2; ++r)
        eor[r] = (unsigned)((int64_t)(wlo ? ty0 - 1 : min(ty0 + EY, a.ny + 1)) * a.nx + tx0 + lane + 64 * r + (int64_t)bias);
    const unsigned char* const clsb = a.cls + a.clead - bias;
    const double* const xb0 = a.x - bias;
    const double* const fb0 = a.f - bias;
    auto sbase = [](const void* p) -> gcptr_t {
        const unsigned long long u = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
        return (gcptr_t)(((unsigned long long)hi << 32) | lo);
    };
    auto ldd = [](gcptr_t b, unsigned e) -> double {
        return *(const __attribute__((address_space(1))) double*)(b + ((unsigned long long)e << 3));
    };
    auto ldc = [](gcptr_t b, unsigned e) -> int {
        return *(const __attribute__((address_space(1))) unsigned char*)(b + (unsigned long long)e);
    };

    // registers per cell: x of planes k-1, k, k+1 (va vb vc) and k+2 (vd, in flight); f and the class of planes k and
    // k+1 (in flight, with the y ring of x in plane k+1)
    int c1[NC], c2[NC];
    double f1[NC], f2[NC], va[NC], vb[NC], vc[NC], vd[NC];
    double hy[2];
    unsigned fast = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) { f1[c] = f2[c] = va[c] = vb[c] = vc[c] = vd[c] = 0.0; c1[c] = c2[c] = 0; }
    hy[0] = hy[1] = 0.0;

    auto load_cf = [&](const int plane, int (&cc)[NC], double (&fr)[NC]) {
        const int64_t o = (int64_t)min(max(plane, -1), a.nz) * a.P;
        const gcptr_t cb = sbase(clsb + o);
        const gcptr_t fb = sbase(fb0 + o);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cc[c] = ldc(cb, eo[c]);
            fr[c] = ldd(fb, eo[c]);
        }
        if (wlo || whi) {
            const gcptr_t xb = sbase(xb0 + o);
#pragma unroll
            for (int r = 0; r < 2; ++r) hy[r] = ldd(xb, eor[r]);
        }
    };
    auto load_x = [&](const int plane, double (&v)[NC]) {
        const gcptr_t xb = sbase(xb0 + (int64_t)min(max(plane, -1), a.nz) * a.P);
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = ldd(xb, eo[c]);
    };
    auto park = [&](const int plane, const double (&v)[NC]) {
        double* const xs = sV0 + (plane & 1) * V0S;
#pragma unroll
        for (int c = 0; c < NC; ++c) xs[lwof(c) + EX] = v[c];
        if (wlo || whi) {
            const int rowv = wlo ? 0 : (EY + 1) * EX;
#pragma unroll
            for (int r = 0; r < 2; ++r) xs[rowv + lane + 64 * r] = hy[r];
        }
    };
    auto all_main = [&](const int (&cc)[NC]) -> unsigned {
        unsigned m = 0;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            if (__builtin_amdgcn_readfirstlane((int)(__ballot(cc[c] != cmain) == 0ull))) m |= 1u << c;
        return m;
    };
    const int64_t rb0 = (int64_t)(ty0 + ey0) * a.nx + (tx0 + lane);
    auto rowof = [&](int c) -> int64_t { return rb0 + (int64_t)(c >> 1) * a.nx + 64 * (c & 1); };

    // ---- warm-up: what step k = z0 finds in place ----
    load_x(z0 - 1, va);
    load_x(z0, vb);
    load_x(z0 + 1, vc);
    load_cf(z0, c1, f1);                    // (with the y ring of plane z0)
    park(z0, vb);
    fast = all_main(c1);
    load_cf(z0 + 1, c2, f2);
    load_x(z0 + 2, vd);
    __syncthreads();

    for (int k = z0; k < z1; ++k) {
        const double* const x1 = sV0 + (k & 1) * V0S + EX;            // x of plane k, indexed by lwof
        const int64_t o1 = (int64_t)k * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iw = lwof(c);
            const double xs = x1[iw - EX], xw = x1[iw - 1], xe = x1[iw + 1], xn = x1[iw + EX];
            double acc = 0.0, cf;
            if (fast >> c & 1u) {
                acc = fma(m0, va[c], acc);
                acc = fma(m1, xs, acc);
                acc = fma(m2, xw, acc);
                acc = fma(m3, vb[c], acc);
                acc = fma(m4, xe, acc);
                acc = fma(m5, xn, acc);
                acc = fma(m6, vc[c], acc);
                cf = mcf;
            } else {
                const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * c1[c]);
                const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                acc = fma(t01.x, va[c], acc);
                acc = fma(t01.y, xs, acc);
                acc = fma(t23.x, xw, acc);
                acc = fma(t23.y, vb[c], acc);
                acc = fma(t45.x, xe, acc);
                acc = fma(t45.y, xn, acc);
                acc = fma(t67.x, vc[c], acc);
                cf = t67.y;
            }
            const int64_t r = rowof(c) + o1;
            bool store = (inT >> c & 1u) != 0 && r >= 0 && r < a.nloc;
            if (MODE == MODE_GS) store = store && lattice_color(a.color_kind, r + a.grow0, a.nx, a.ny) == a.color;
            if (store) a.out[r] = MODE == MODE_RESIDUAL ? f1[c] - acc : vb[c] + cf * (f1[c] - acc);
        }
        // ---- park plane k+1, rotate, issue the loads of the step after the next ----
        park(k + 1, vc);
        fast = all_main(c2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            c1[c] = c2[c]; f1[c] = f2[c];
            va[c] = vb[c]; vb[c] = vc[c]; vc[c] = vd[c];
            asm volatile("" : "+v"(c1[c]));
            asm volatile("" : "+v"(f1[c]));
            asm volatile("" : "+v"(vc[c]));
        }
        load_cf(k + 2, c2, f2);
        load_x(k + 3, vd);
        __syncthreads();
    }
}

template <int NW, int LPW, int MODE>
__global__ __launch_bounds__(NW * WAVE) void sdia_sweep1c(J2Args a) {
    j1c_body<NW, LPW, MODE>(a);
}

template <int NW, int LPW>
__global__ __launch_bounds__(NW * WAVE) void sdia_jacobi2c(J2Args a) {
    j2c_body<NW, LPW>(a);
}

// (its own symbol for the finest level, like sdia_jacobi2_finest)
template <int NW, int LPW>
__global__ __launch_bounds__(NW * WAVE) void sdia_jacobi2c_finest(J2Args a) {
    j2c_body<NW, LPW>(a);
}

template <int R, int NW, int LPW, bool NT>
__global__ __launch_bounds__(NW * WAVE) void sdia_jacobi2(J2Args a) {
    j2_body<R, NW, LPW, NT>(a);
}

// The same pass under its own symbol for the FINEST level, so that profiler summaries (rocprofv3 --stats) list
// the dominant launches apart from the shorter ones of the coarser levels.
template <int R, int NW, int LPW, bool NT>
__global__ __launch_bounds__(NW * WAVE) void sdia_jacobi2_finest(J2Args a) {
    j2_body<R, NW, LPW, NT>(a);
}

}  // namespace mgk

namespace mgk {

// the value of lane - 1 (lane 0: lane 63) / lane + 1 (lane 63: lane 0) of the wave
__device__ __forceinline__ double jk3_from_west(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x13C, 0xF, 0xF, true);      // wave_ror:1
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x13C, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double jk3_from_east(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x134, 0xF, 0xF, true);      // wave_rol:1
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x134, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// ---- K sweeps in one launch on 2-D levels (five-point rows through row classes) --------------------------------
// The reference's own configurations are 2-D (Multigrid_prototype.py:35-46: 64 x 64, V(50,50)); a 2-D level is a few
// MB at most, so its sweeps are launch- and latency-bound, not bandwidth-bound.  Here a 1024-thread workgroup loads
// a region of 128 x H cells (x, f and the class bytes: 17 bytes per cell) into LDS, relaxes it K times there -- the
// region that is still exact shrinks by one ring per sweep -- and stores the inner (128 - 2K) x (H - 2K) cells:
// one launch and one pass over memory for K sweeps.  Everything is done in ROW space like the 3-D pass (cell
// (ex, ey) <-> row (ty0+ey)*nx + tx0+ex whether or not that wraps around a grid line; rows outside the level are
// zeros), with the arithmetic of sdia_cls_body<3, ...>: bit-identical to K single sweeps.
struct JKArgs {
    const double* x;        // row-based
    const double* f;
    double* out;            // != x
    const unsigned char* cls;       // row-based
    const double* ctab;
    int ncls, cmain;
    double cm[8];
    int64_t n;              // rows of the level
    int nx, nlines;         // grid: nx columns, nlines lines (row = line * nx + column)
    int ntx, nty;
    double omega;
};

constexpr int JK_W = 128;

template <int H> constexpr size_t jk_lds_bytes() { return (size_t)2 * H * JK_W * sizeof(double) + 256 * CLS_W * sizeof(double); }

// (round 3: a thread keeps x, f and the class of its H / 8 cells -- the same column of the region, eight lines apart -- in
//  registers; LDS holds the two copies of the iterate only, the +-1 neighbours come from the lanes next door (DPP; a wave is 64
//  consecutive cells of a line, its first and last lane read the LDS copy) and a cell costs three LDS accesses per sweep
//  instead of eight.  H = 32 or 16: two workgroups per CU, one computing while the other waits at its barrier -- faster than one
//  workgroup on 64 lines although a sweep keeps fewer of the lines, see launch_jacobik_t.)
template <int K, int H>
__global__ __launch_bounds__(1024) void sdia_jacobik2d(JKArgs a) {
    constexpr int W = JK_W, NT = 1024, CELLS = W * H, PER = CELLS / NT;
    static_assert(CELLS % NT == 0, "whole rounds of the workgroup");
    extern __shared__ double j2_smem[];
    double* const sT = j2_smem;                               // 256 x 8 (classes in use), [7] = omega / diagonal
    double* const sX = sT + 256 * CLS_W;                      // 2 x CELLS
    const int tid = threadIdx.x, lane = tid & 63;
    const int tix = (int)(blockIdx.x % (unsigned)a.ntx), tiy = (int)(blockIdx.x / (unsigned)a.ntx);
    const int tx0 = tix * (W - 2 * K) - K, ty0 = tiy * (H - 2 * K) - K;       // grid position of cell (0, 0)
    const int64_t r00 = (int64_t)ty0 * a.nx + tx0;
    const int ex = tid % W, ey0 = tid / W;                   // the thread's cells: (ex, ey0 + (NT / W) i)

    // ---- region -> registers and the first LDS copy (rows outside the level: zeros, class 0) ----
    double xr[PER], fr[PER];
    int cr[PER];
    unsigned okmask = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int idx = tid + i * NT;
        const int64_t row = r00 + (int64_t)(idx / W) * a.nx + (idx % W);
        const bool ok = row >= 0 && row < a.n;
        xr[i] = ok ? a.x[row] : 0.0;
        fr[i] = ok ? a.f[row] : 0.0;
        cr[i] = ok ? (int)a.cls[row] : 0;
        okmask |= (ok ? 1u : 0u) << i;
        sX[idx] = xr[i];
    }
    for (int i = tid; i < a.ncls * CLS_W; i += NT) {
        double v = a.ctab[i];
        if ((i & (CLS_W - 1)) == CLS_W - 1) {
            const double d = a.ctab[i - 4];
            v = a.omega * (1.0 / (d != 0.0 ? d : 1.0));
        }
        sT[i] = v;
    }
    const double m1 = a.cm[1], m2 = a.cm[2], m3 = a.cm[3], m4 = a.cm[4], m5 = a.cm[5];
    const double mcf = a.omega * (1.0 / (m3 != 0.0 ? m3 : 1.0));
    // bit i: the i-th cell of every lane of this wave has the most frequent class (the classes do not change between sweeps)
    unsigned fastmask = 0u;
#pragma unroll
    for (int i = 0; i < PER; ++i)
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(cr[i] != a.cmain) == 0ull))) fastmask |= 1u << i;
    fastmask = (unsigned)__builtin_amdgcn_readfirstlane((int)fastmask);
    __syncthreads();

    // ---- K sweeps: sweep s is exact on [s, W-s) x [s, H-s) ----
#pragma unroll
    for (int s = 1; s <= K; ++s) {
        const double* const src = sX + ((s - 1) & 1) * CELLS;
        double* const dst = sX + (s & 1) * CELLS;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int idx = tid + i * NT;
            const int ey = ey0 + (NT / W) * i;                          // a wave: 64 consecutive cells of one line
            if (ey < s || ey >= H - s) continue;                        // (uniform within the wave)
            const double xc = xr[i];
            const double dw = jk3_from_west(xc), de = jk3_from_east(xc);
            const double xw = lane == 0 ? src[idx - 1] : dw;            // (the neighbour belongs to another wave)
            const double xe = lane == 63 ? src[idx + 1] : de;
            const double xs = src[idx - W], xn = src[idx + W];
            double o;
            if (fastmask >> i & 1u) {
                double acc = 0.0;
                acc = fma(m1, xs, acc);
                acc = fma(m2, xw, acc);
                acc = fma(m3, xc, acc);
                acc = fma(m4, xe, acc);
                acc = fma(m5, xn, acc);
                o = xc + mcf * (fr[i] - acc);
            } else {
                const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * cr[i]);
                const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                double acc = 0.0;
                acc = fma(t01.y, xs, acc);
                acc = fma(t23.x, xw, acc);
                acc = fma(t23.y, xc, acc);
                acc = fma(t45.x, xe, acc);
                acc = fma(t45.y, xn, acc);
                o = xc + t67.y * (fr[i] - acc);
            }
            if (ex >= s && ex < W - s) {
                o = (okmask >> i & 1u) ? o : 0.0;
                dst[idx] = o;
                xr[i] = o;
            }
        }
        __syncthreads();
    }

    // ---- the inner cells that lie on the grid ----
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int ey = ey0 + (NT / W) * i;
        const int gx = tx0 + ex, gy = ty0 + ey;
        if (ex >= K && ex < W - K && ey >= K && ey < H - K && gx < a.nx && gy < a.nlines)
            a.out[(int64_t)gy * a.nx + gx] = xr[i];
    }
}

}  // namespace mgk

namespace mgk {

// ---- all sweeps of a small level in one launch ------------------------------------------------------------------
// The reference's shipped configuration (Multigrid_prototype.py:35-46) smooths levels of 4225 and 1089 unknowns fifty
// times per leg: such a level fits one CU, so ONE 1024-thread workgroup runs all nw sweeps with a barrier between them
// -- one launch instead of nw (or nw / 5).  Rows are handled in linear row space; thread t owns the rows t, t + 1024, ...
// (RPT of them) and keeps their x, f and class in REGISTERS for the whole launch: a sweep reads only the +-nx (+-plane)
// neighbours from the LDS image of the previous iterate (two copies, zero padding of the largest offset on both sides) --
// the +-1 neighbours are the rows of the lanes next door (DPP; the first and last lane of a wave read the image) -- and
// writes its new value into the other copy: three LDS accesses per row and sweep instead of eight (round 2 kept f, the
// classes and x in LDS only), and whether a wave's rows all have the most frequent class is decided once, not per sweep:
// 1.77 -> 1.37 us per sweep on 4225 rows, 0.68 -> 0.60 on 1089 (profiles/r03_small_kernel.txt; reading all of a thread's
// neighbours ahead of the arithmetic, CH > 1 below, made it slower: 1.61).  What is left is instruction issue on ONE CU.
// Arithmetic of sdia_cls_body<WU, ...> in the same order: bit-identical to single sweeps.  Five- and seven-point levels
// with row classes.
struct JSArgs {
    const double* x;        // row-based
    const double* f;
    double* out;
    const unsigned char* cls;       // row-based
    const double* ctab;
    int ncls, cmain;
    double cm[8];
    int n, nw;
    int up1, up2, up3;      // the positive offsets (+1, +nx, +plane; up3 = 0 for five-point rows)
    double omega;
};

inline size_t js_lds_bytes(int n, int pad) { return (size_t)256 * CLS_W * 8 + (size_t)2 * (n + 2 * pad) * 8; }

template <int WU, int RPT, bool DPP = true>
__global__ __launch_bounds__(1024) void sdia_jacobi_small(JSArgs a) {
    extern __shared__ double j2_smem[];
    const int n = a.n, pad = WU == 4 ? a.up3 : a.up2, stride = n + 2 * pad;
    double* const sT = j2_smem;
    double* const sX = sT + 256 * CLS_W;                      // 2 x (pad | n | pad)
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * stride; i += 1024) sX[i] = 0.0;
    for (int i = tid; i < a.ncls * CLS_W; i += 1024) {
        double v = a.ctab[i];
        if ((i & (CLS_W - 1)) == CLS_W - 1) {
            const double d = a.ctab[i - 4];
            v = a.omega * (1.0 / (d != 0.0 ? d : 1.0));
        }
        sT[i] = v;
    }
    __syncthreads();
    double xr[RPT], fr[RPT];
    int cr[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = tid + 1024 * i;
        const bool in = r < n;
        xr[i] = in ? a.x[r] : 0.0;
        fr[i] = in ? a.f[r] : 0.0;
        cr[i] = in ? (int)a.cls[r] : 0;
        if (in) sX[pad + r] = xr[i];
    }
    const double m0 = a.cm[0], m1 = a.cm[1], m2 = a.cm[2], m3 = a.cm[3], m4 = a.cm[4], m5 = a.cm[5], m6 = a.cm[6];
    const double mcf = a.omega * (1.0 / (m3 != 0.0 ? m3 : 1.0));
    // bit i: the i-th row of every lane of this wave has the most frequent class (entries in scalar registers) -- the classes
    // do not change from sweep to sweep
    unsigned fastmask = 0u;
#pragma unroll
    for (int i = 0; i < RPT; ++i)
        if (__builtin_amdgcn_readfirstlane((int)(__ballot(cr[i] != a.cmain) == 0ull))) fastmask |= 1u << i;
    fastmask = (unsigned)__builtin_amdgcn_readfirstlane((int)fastmask);
    __syncthreads();
    for (int s = 0; s < a.nw; ++s) {
        const double* const src = sX + (s & 1) * stride + pad;
        double* const dst = sX + ((s + 1) & 1) * stride + pad;
        // the neighbours of several of the thread's rows first, in straight-line code (read row by row inside the branches below,
        // each row waited for its own reads: a sweep was a chain of RPT LDS latencies), CH rows at a time
        constexpr int CH = 1;       // rows whose neighbours are read ahead of the arithmetic (more: slower, see above)
#pragma unroll
        for (int i0 = 0; i0 < RPT; i0 += CH) {
            double xw[CH], xe[CH], xs[CH], xn[CH], xd[WU == 4 ? CH : 1], xu[WU == 4 ? CH : 1];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int i = i0 + j < RPT ? i0 + j : RPT - 1;
                const int r = tid + 1024 * i;
                const int rr = r < n ? r : n - 1;
                if constexpr (DPP) {
                    // (rows past the end hold zeros, like the image's padding; the wave's first / last lane: the row belongs to another wave)
                    const double dw = jk3_from_west(xr[i]), de = jk3_from_east(xr[i]);
                    xw[j] = lane == 0 ? src[rr - 1] : dw;
                    xe[j] = lane == 63 ? src[rr + 1] : de;
                } else {
                    xw[j] = src[rr - 1];
                    xe[j] = src[rr + 1];
                }
                xs[j] = src[rr - a.up2];
                xn[j] = src[rr + a.up2];
                if constexpr (WU == 4) {
                    xd[j] = src[rr - a.up3];
                    xu[j] = src[rr + a.up3];
                }
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                if (i0 + j >= RPT) continue;
                const int i = i0 + j;
                const int r = tid + 1024 * i;
                const bool in = r < n;
                const int c = cr[i];
                const double xc = xr[i];
                double acc = 0.0, cf;
                if (fastmask >> i & 1u) {
                    if (WU == 4) acc = fma(m0, xd[WU == 4 ? j : 0], acc);
                    acc = fma(m1, xs[j], acc);
                    acc = fma(m2, xw[j], acc);
                    acc = fma(m3, xc, acc);
                    acc = fma(m4, xe[j], acc);
                    acc = fma(m5, xn[j], acc);
                    if (WU == 4) acc = fma(m6, xu[WU == 4 ? j : 0], acc);
                    cf = mcf;
                } else {
                    const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * c);
                    const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                    if (WU == 4) acc = fma(t01.x, xd[WU == 4 ? j : 0], acc);
                    acc = fma(t01.y, xs[j], acc);
                    acc = fma(t23.x, xw[j], acc);
                    acc = fma(t23.y, xc, acc);
                    acc = fma(t45.x, xe[j], acc);
                    acc = fma(t45.y, xn[j], acc);
                    if (WU == 4) acc = fma(t67.x, xu[WU == 4 ? j : 0], acc);
                    cf = t67.y;
                }
                if (in) {
                    const double o = xc + cf * (fr[i] - acc);
                    dst[r] = o;
                    xr[i] = o;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int r = tid + 1024 * i;
        if (r < n) a.out[r] = xr[i];
    }
}

}  // namespace mgk
