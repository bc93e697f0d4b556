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
    // row classes (sdia_jacobi2c): cls[row + mlead] indexes ctab[class][4] = (diagonal, +1, +nx, +P entries)
    const unsigned char* cls;
    const double* ctab;
    int64_t clead;          // padding in front of cls[]: class of row r is cls[r + clead]
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
// identity rows.  When the stored rows (diagonal, +1, +nx, +P entry; the lower entries are those of other rows)
// take at most 255 distinct non-zero values BIT FOR BIT, the two-sweep pass reads one byte per row -- its class --
// instead of 32 bytes of matrix, and looks the entries up in a 256 x 4 table kept in LDS: 25 instead of 56 bytes
// per row and pair of sweeps, with exactly the same arithmetic on exactly the same numbers.  The dictionary is
// built on the device (hash insert, compaction, encode + bitwise verification); a level with more distinct rows
// simply keeps the plain pass.
constexpr int CLS_SLOTS = 4096;

struct ClsArgs {
    const double* dvals;            // symmetric diagonal storage, WU = 4
    int64_t mrows;                  // rows stored (lead rows included)
    unsigned long long* tags;       // CLS_SLOTS hash tags, 0 = free
    double* svals;                  // CLS_SLOTS x 4
    int* count;                     // distinct non-zero rows seen
    int* slot_class;                // CLS_SLOTS
    double* ctab;                   // 256 x 4
    unsigned char* cls;             // crows: cls[i] is the class of stored row i - cshift (class 0 where there is none)
    int64_t crows, cshift;
    int* flag;                      // set when a row does not match its dictionary entry
};

__device__ __forceinline__ unsigned long long cls_mix(unsigned long long x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    return x;
}

template <int S> __device__ __forceinline__ void cls_row(const ClsArgs& a, int64_t m, unsigned long long (&b)[4]) {
    const size_t base = (size_t)(m / S) * (4 * S) + (size_t)(m % S);
#pragma unroll
    for (int c = 0; c < 4; ++c) b[c] = (unsigned long long)__double_as_longlong(a.dvals[base + (size_t)c * S]);
}

__device__ __forceinline__ unsigned long long cls_hash(const unsigned long long (&b)[4]) {
    unsigned long long h = cls_mix(b[0] + 0x9e3779b97f4a7c15ull);
    h = cls_mix(h ^ (b[1] + 0x3c6ef372fe94f82bull));
    h = cls_mix(h ^ (b[2] + 0xdaa66d2c7ddf743full));
    h = cls_mix(h ^ (b[3] + 0x78dde6e5fd29f054ull));
    return h ? h : 1ull;
}

template <int S>
__global__ void cls_insert(ClsArgs a) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= a.mrows) return;
    unsigned long long b[4];
    cls_row<S>(a, m, b);
    if ((b[0] | b[1] | b[2] | b[3]) == 0ull) return;          // class 0: the all-zero row
    const unsigned long long h = cls_hash(b);
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
            for (int c = 0; c < 4; ++c) a.svals[4 * s + c] = __longlong_as_double((long long)b[c]);
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
    for (int c = 0; c < 4; ++c) a.ctab[c] = 0.0;
    int id = 1;
    for (int s = 0; s < CLS_SLOTS; ++s) {
        a.slot_class[s] = 0;
        if (a.tags[s] && id < 256) {
            a.slot_class[s] = id;
            for (int c = 0; c < 4; ++c) a.ctab[4 * id + c] = a.svals[4 * s + c];
            ++id;
        }
    }
    for (; id < 256; ++id)
        for (int c = 0; c < 4; ++c) a.ctab[4 * id + c] = 0.0;
}

template <int S>
__global__ void cls_encode(ClsArgs a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.crows) return;
    const int64_t m = i - a.cshift;
    if (m < 0 || m >= a.mrows) { a.cls[i] = 0; return; }
    unsigned long long b[4];
    cls_row<S>(a, m, b);
    if ((b[0] | b[1] | b[2] | b[3]) == 0ull) { a.cls[i] = 0; return; }
    const unsigned long long h = cls_hash(b);
    unsigned s = (unsigned)h & (CLS_SLOTS - 1);
    for (int probe = 0; probe < CLS_SLOTS; ++probe) {
        const unsigned long long t = a.tags[s];
        if (t == h) {
            bool same = true;
#pragma unroll
            for (int c = 0; c < 4; ++c) same = same && (unsigned long long)__double_as_longlong(a.svals[4 * s + c]) == b[c];
            if (!same || a.slot_class[s] == 0) atomicExch(a.flag, 1);       // hash collision / overflow: no classes
            a.cls[i] = (unsigned char)a.slot_class[s];
            return;
        }
        if (t == 0ull) break;
        s = (s + 1) & (CLS_SLOTS - 1);
    }
    atomicExch(a.flag, 1);
    a.cls[i] = 0;
}

// ---- the two-sweep pass on class-coded rows ----------------------------------------------------
// Same march, same arithmetic as j2_body; per cell and step it loads one class byte, f and x (17 bytes instead of
// 48) and keeps classes where j2_body keeps matrix entries: the row's own entries and omega/diag come from the LDS
// table (sT, sCF: one IEEE division per class and workgroup instead of two per cell and step), the neighbours'
// +1 / +nx entries through the plane image of the classes (sC).  With so few registers per cell the step is
// arranged differently: the plane images are multi-buffered and the WHOLE second sweep of plane k is evaluated
// in the step that relaxes plane k+1 once, from the image of v1[k] written a step earlier -- a step only reads
// what earlier steps wrote, so there is one barrier per plane instead of two.
template <int NW, int LPW> constexpr size_t j2c_lds_bytes() {
    constexpr int EY = NW * LPW, PV = J2_EX + 2;
    constexpr size_t v0 = (size_t)(EY + 2) * PV, v1 = (size_t)EY * J2_EX;
    return sizeof(double) * (256 * 4 + 256 + 2 * v0 + 2 * v1) + 3 * v0;
}

template <int NW, int LPW>
__device__ __forceinline__ void j2c_body(const J2Args& a) {
    constexpr int EX = J2_EX, EY = NW * LPW, NC = 2 * LPW, PV = EX + 2;
    extern __shared__ double j2_smem[];
    double* const sT = j2_smem;                           // 256 x 4    entries of the row classes
    double* const sCF = sT + 256 * 4;                     // 256        omega / diagonal
    // ONE barrier per plane: the images are multi-buffered (slot = plane mod 2 / mod 3), so that a step only reads
    // what earlier steps wrote and only writes what no wave can still be reading
    constexpr int V0S = (EY + 2) * PV, V1S = EY * EX, CS = (EY + 2) * PV;
    double* const sV0 = sCF + 256;                        // 2 x (EY+2) x PV   x of a plane, origin (-1,-1)
    double* const sV1 = sV0 + 2 * V0S;                    // 2 x EY x EX       once-relaxed iterate of a plane
    unsigned char* const sC = reinterpret_cast<unsigned char*>(sV1 + 2 * V1S);   // 3 x (EY+2) x PV classes, like sV0
    auto slot3 = [](int plane) -> int { return ((plane % 3) + 3) % 3; };
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

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
    const int tx0 = tix * (EX - 2) - 1, ty0 = tiy * (EY - 2) - 1;

    for (int i = threadIdx.x; i < 256 * 4; i += NW * WAVE) sT[i] = a.ctab[i];
    for (int i = threadIdx.x; i < 256; i += NW * WAVE) {
        const double d = a.ctab[4 * i];
        sCF[i] = a.omega * (1.0 / (d != 0.0 ? d : 1.0));
    }

    const int ey0 = wave * LPW;
    const int64_t rb0 = (int64_t)(ty0 + ey0) * a.nx + (tx0 + lane);
    const int lv0 = (ey0 + 1) * PV + lane + 1;
    const int lw0 = ey0 * EX + lane;
    auto rowof = [&](int c) -> int64_t { return rb0 + (int64_t)(c >> 1) * a.nx + 64 * (c & 1); };
    auto lvof = [&](int c) -> int { return lv0 + (c >> 1) * PV + 64 * (c & 1); };
    auto lwof = [&](int c) -> int { return lw0 + (c >> 1) * EX + 64 * (c & 1); };
    unsigned inT = 0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ex = lane + 64 * (c & 1), ey = ey0 + (c >> 1);
        if (ex >= 1 && ex < EX - 1 && ey >= 1 && ey < EY - 1 && tx0 + ex < a.nx && ty0 + ey < a.ny) inT |= 1u << c;
    }
    const bool hl = lane == 0;
    const bool wlo = wave == 0, whi = wave == NW - 1;

    // Addresses.  Every load of a plane is `uniform base of the plane + a 32-bit element offset fixed for the whole
    // march`: no per-lane predicates, no 64-bit address arithmetic in the loop.  That needs every address to be
    // readable whether or not its row exists: the vectors carry zero slack of a plane + 2 lines on both sides
    // (vec_reach), the class array is padded alike with class 0 (the all-zero row), planes -1 and nz are read like
    // any other and planes beyond them are skipped by a uniform test.  Grid lines from ny+2 on (tiles that stick
    // out of the grid; their cells feed no result) are read at line ny+1 so that the slack suffices.
    const unsigned bias = 2u * (unsigned)a.nx + 2u;
    unsigned eo[LPW], eor;                                   // cell (line l, r = 0); the line of the y ring
#pragma unroll
    for (int l = 0; l < LPW; ++l)
        eo[l] = (unsigned)((int64_t)min(ty0 + ey0 + l, a.ny + 1) * a.nx + tx0 + lane + (int64_t)bias);
    eor = (unsigned)((int64_t)(wlo ? ty0 - 1 : min(ty0 + EY, a.ny + 1)) * a.nx + tx0 + lane + (int64_t)bias);
    const unsigned char* const clsb = a.cls + a.clead - bias;       // + plane*P: class of element offset 0
    const double* const xb0 = a.x - bias;
    const double* const fb0 = a.f - bias;
    // the plane bases are made opaque scalars (readfirstlane) so that the compiler keeps them in SGPRs and emits
    // `global_load v, v_offset, s[base:base+1] offset:imm` instead of carrying a 64-bit VGPR address per stream
    auto sbase = [](const void* p) -> const char* {
        const unsigned long long u = (unsigned long long)p;
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u);
        const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
        return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
    };
    auto ldd = [](const char* b, unsigned e, int d) -> double { return *reinterpret_cast<const double*>(b + (e << 3) + d * 8); };
    auto ldc = [](const char* b, unsigned e, int d) -> int { return *reinterpret_cast<const unsigned char*>(b + e + d); };

    // classes of the cell in planes k-1, k, k+1: one byte each of cpk (k-1 lowest); c2: plane k+2, in flight
    unsigned cpk[NC];
    int c1[NC], c2[NC];
    auto cls_m = [&](int c) -> int { return (int)(cpk[c] & 255u); };
    auto cls_0 = [&](int c) -> int { return (int)((cpk[c] >> 8) & 255u); };
    auto cls_1 = [&](int c) -> int { return (int)((cpk[c] >> 16) & 255u); };
    double f0[NC], f1[NC], f2[NC], wm[NC], w0[NC], w1[NC], va[NC], vb[NC], vc[NC], vd[NC];
    double hxl[LPW], hyv[2];
    int hxc[LPW], hyc[2];
#pragma unroll
    for (int c = 0; c < NC; ++c) { f0[c] = wm[c] = w0[c] = w1[c] = 0.0; }
#pragma unroll
    for (int l = 0; l < LPW; ++l) { hxl[l] = 0.0; hxc[l] = 0; }
    hyv[0] = hyv[1] = 0.0; hyc[0] = hyc[1] = 0;

    auto load_plane = [&](const int plane, const bool on, int (&cc)[NC], double (&fr)[NC]) {
        if (on && plane >= -1 && plane <= a.nz) {
            const int64_t o = (int64_t)plane * a.P;
            const char* const cb = sbase(clsb + o);
            const char* const fb = sbase(fb0 + o);
            const char* const xb = sbase(xb0 + o);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                cc[c] = ldc(cb, eo[c >> 1], 64 * (c & 1));
                fr[c] = ldd(fb, eo[c >> 1], 64 * (c & 1));
            }
            if (hl || lane == 63) {                 // x ring: the cell left of ex = 0 / right of ex = EX-1
#pragma unroll
                for (int l = 0; l < LPW; ++l) {
                    hxl[l] = ldd(xb, hl ? eo[l] - 1u : eo[l] + 65u, 0);
                    hxc[l] = ldc(cb, eo[l], -1);
                }
            }
            if (wlo || whi) {                       // y ring: the line below ey = 0 / above ey = EY-1
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    hyv[r] = ldd(xb, eor, 64 * r);
                    hyc[r] = ldc(cb, eor, 64 * r);
                }
            }
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) { cc[c] = 0; fr[c] = 0.0; }
#pragma unroll
            for (int l = 0; l < LPW; ++l) { hxl[l] = 0.0; hxc[l] = 0; }
            hyv[0] = hyv[1] = 0.0; hyc[0] = hyc[1] = 0;
        }
    };
    auto load_x = [&](const int plane, const bool on, double (&v)[NC]) {
        if (on && plane >= -1 && plane <= a.nz) {
            const char* const xb = sbase(xb0 + (int64_t)plane * a.P);
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = ldd(xb, eo[c >> 1], 64 * (c & 1));
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = 0.0;
        }
    };
    auto load_c = [&](const int plane, int (&cc)[NC]) {
        if (plane >= -1 && plane <= a.nz) {
            const char* const cb = sbase(clsb + (int64_t)plane * a.P);
#pragma unroll
            for (int c = 0; c < NC; ++c) cc[c] = ldc(cb, eo[c >> 1], 64 * (c & 1));
        } else {
#pragma unroll
            for (int c = 0; c < NC; ++c) cc[c] = 0;
        }
    };
    auto park = [&](const int plane, const double (&v)[NC], const int (&cc)[NC]) {
        unsigned char* const cs = sC + slot3(plane) * CS;
        double* const xs = sV0 + (plane & 1) * V0S;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iv = lvof(c);
            xs[iv] = v[c];
            cs[iv] = (unsigned char)cc[c];
        }
#pragma unroll
        for (int l = 0; l < LPW; ++l) {
            const int rowv = (ey0 + l + 1) * PV;
            if (hl) { xs[rowv] = hxl[l]; cs[rowv] = (unsigned char)hxc[l]; }
            if (lane == 63) xs[rowv + EX + 1] = hxl[l];
        }
        if (wlo) {
#pragma unroll
            for (int r = 0; r < 2; ++r) { xs[lane + 64 * r + 1] = hyv[r]; cs[lane + 64 * r + 1] = (unsigned char)hyc[r]; }
        } else if (whi) {
#pragma unroll
            for (int r = 0; r < 2; ++r) xs[(EY + 1) * PV + lane + 64 * r + 1] = hyv[r];
        }
    };

    load_plane(z0 - 1, true, c1, f1);
    load_x(z0 - 2, true, va);
    load_x(z0 - 1, true, vb);
    load_x(z0, true, vc);
    load_c(z0 - 2, c2);
    park(z0 - 1, vb, c1);
#pragma unroll
    for (int c = 0; c < NC; ++c) cpk[c] = ((unsigned)c2[c] << 8) | ((unsigned)c1[c] << 16);
    // the loads of a step are issued BEFORE the barrier that ends the step before it: the time the waves spend at
    // the barrier is flight time too
    load_plane(z0, z0 <= z1, c2, f2);
    load_x(z0 + 1, z0 + 1 <= z1 + 1, vd);
    __syncthreads();

    for (int k = z0 - 2; k < z1; ++k) {

        const unsigned char* const cp1 = sC + slot3(k + 1) * CS;      // classes of plane k+1
        const unsigned char* const cp0 = sC + slot3(k) * CS;          // classes of plane k
        const double* const x1 = sV0 + ((k + 1) & 1) * V0S;           // x of plane k+1
        double* const v1w = sV1 + ((k + 1) & 1) * V1S;                // v1 of plane k+1 (written here)
        const double* const v1r = sV1 + (k & 1) * V1S;                // v1 of plane k (written a step ago)
        const int64_t o1 = (int64_t)(k + 1) * a.P;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int iv = lvof(c), iw = lwof(c);
            const double s0 = sT[4 * cls_0(c) + 3];
            const double* const t1 = sT + 4 * cls_1(c);
            // first sweep on plane k+1
            double acc = 0.0;
            acc = fma(s0, va[c], acc);                                          // -P
            acc = fma(sT[4 * cp1[iv - PV] + 2], x1[iv - PV], acc);              // -nx
            acc = fma(sT[4 * cp1[iv - 1] + 1], x1[iv - 1], acc);                // -1
            acc = fma(t1[0], vb[c], acc);
            acc = fma(t1[1], x1[iv + 1], acc);                                  // +1
            acc = fma(t1[2], x1[iv + PV], acc);                                 // +nx
            acc = fma(t1[3], vc[c], acc);                                       // +P
            const double o = vb[c] + sCF[cls_1(c)] * (f1[c] - acc);
            const int64_t r1 = rowof(c) + o1;
            w1[c] = (r1 >= 0 && r1 < a.nloc) ? o : 0.0;
            v1w[iw] = w1[c];
            if (inT >> c & 1u) {
                const int64_t r0 = r1 - a.P;
                if (k >= z0 && r0 >= a.st_lo && r0 < a.st_hi) {
                    // second sweep on plane k: its in-plane neighbours were written a step ago
                    const double* const t0 = sT + 4 * cls_0(c);
                    double ac2 = 0.0;
                    ac2 = fma(sT[4 * cls_m(c) + 3], wm[c], ac2);                // -P
                    ac2 = fma(sT[4 * cp0[iv - PV] + 2], v1r[iw - EX], ac2);     // -nx
                    ac2 = fma(sT[4 * cp0[iv - 1] + 1], v1r[iw - 1], ac2);       // -1
                    ac2 = fma(t0[0], w0[c], ac2);
                    ac2 = fma(t0[1], v1r[iw + 1], ac2);                         // +1
                    ac2 = fma(t0[2], v1r[iw + EX], ac2);                        // +nx
                    ac2 = fma(s0, w1[c], ac2);                                  // +P
                    a.out[r0] = w0[c] + sCF[cls_0(c)] * (f0[c] - ac2);
                }
                if (a.v1out && k + 1 >= z0 && k + 1 < z1 && (r1 < a.k1_lo || r1 >= a.k1_hi)) a.v1out[r1] = w1[c];
            }
        }
        park(k + 2, vc, c2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            cpk[c] = (cpk[c] >> 8) | ((unsigned)c2[c] << 16);
            f0[c] = f1[c]; f1[c] = f2[c]; wm[c] = w0[c]; w0[c] = w1[c];
            va[c] = vb[c]; vb[c] = vc[c]; vc[c] = vd[c];
        }
        if (k + 1 < z1) {
            load_plane(k + 3, k + 3 <= z1, c2, f2);
            load_x(k + 4, k + 4 <= z1 + 1, vd);
        }
        __syncthreads();
    }
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
