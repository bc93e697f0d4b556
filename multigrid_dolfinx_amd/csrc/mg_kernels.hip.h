// Device kernels of the V-cycle hot path for gfx950 (CDNA4, wave64).
//
// All of these are HBM-bound sparse applies / gathers at <= 0.17 flop/byte, so there is
// no MFMA here: the levers are coalesced 8-16 B/lane tile loads, enough bytes in flight
// per CU, x-vector reuse through L1/L2/MALL and an XCD-aware block -> tile map.
//
// Data layout ("sliced ELL", one slice = 64*R consecutive lexicographic rows, one wave per
// slice, lane l owns rows l*R .. l*R+R-1 of the slice):
//     vals[((slice*W + k)*64 + lane)*R + r]     fp64   k-th stored entry of that row
//     cols[  same index                   ]     int32  its column, as an index into the
//                                                       local vector (halo planes included)
// so a wave reads W contiguous runs of 64*R*8 B (values) and 64*R*4 B (columns): every
// load instruction is a fully coalesced 8*R / 4*R bytes per lane.  Padding entries carry
// value 0.0 and the row's own column.
//
// Offset-coded columns (default whenever a level has <= 255 distinct values of col - row, which is
// every lexicographically renumbered grid matrix): the int32 column array is replaced by
//     codes[((slice*CW + q)*64 + lane)*R + r]   uint64   eight 1-byte codes of entries 8q .. 8q+7
//     offsets[code]                              int32    col - row for that code (<= 255 per level)
// i.e. 8 bytes per row instead of 4*W, and D^-1 is recomputed from the row's own diagonal entry
// instead of being streamed: 88 instead of 116 bytes per row and sweep for 7-point rows.  The
// arithmetic (entry order, fma chain, IEEE division) is unchanged, so results are bit-identical
// to the int32-column kernels.
//
// Symmetric diagonal storage (the shipped default for bit-for-bit symmetric grid matrices; see the
// comment above sdia_body): diagonal + upper diagonals only, 32 B of matrix per 7-point row, lower entries
// re-read as shifted loads of the upper ones: 56 bytes per row and sweep.
//
// A level picks the most compact format it qualifies for at set-up (mg_capi.hip: encode_level,
// repack_sdia); every format has the same four modes (residual, Jacobi, SpMV[+dot], Gauss-Seidel colour).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgk {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;          // 4 waves
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

template <int R> struct alignas(8 * R) DVec { double d[R]; };
template <int R> struct alignas(4 * R) IVec { int d[R]; };
template <int R> struct alignas(8 * R) UVec { unsigned long long d[R]; };
// R consecutive doubles at an address that is only 8-byte aligned (shifted stencil accesses): one
// global_load_dwordx4 for R = 2 instead of two dwordx2 loads.
template <int R> struct alignas(8) DVecU { double d[R]; };

// MODE_GS: in-place relaxation of the rows of one colour (red-black Gauss-Seidel half sweep);
// the colour of a row is the parity of its global lexicographic index, a valid two-colouring of
// the pruned P1 stencils on grids with an odd number of nodes per axis.
enum { MODE_RESIDUAL = 0, MODE_JACOBI = 1, MODE_SPMV = 2, MODE_GS = 3 };
enum { COLOR_PARITY = 0, COLOR_LATTICE9 = 1 };

// Colour of a lexicographic lattice index.  COLOR_LATTICE9 (P2 rows, two lattice planes of reach): the seven parity
// classes of the mid-points keep their parity bits (1..7), the vertices (all coordinates even) are split red / black
// by (i/2 + j/2 + k/2) mod 2 into 0 and 8: no two coupled unknowns of a pruned P2 or P1 Poisson matrix share a colour
// (checked per level before use, ell_check_coloring).  Colours are relaxed in ascending order.
__device__ __forceinline__ int lattice_color(int kind, int64_t g, int nx, int ny) {
    if (kind == COLOR_PARITY) return (int)(g & 1);
    const int64_t line = g / nx;
    const int i = (int)(g - line * nx);
    const int j = (int)(line % ny), k = (int)(line / ny);
    const int par = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2);
    if (par) return par;
    return (((i >> 1) + (j >> 1) + (k >> 1)) & 1) ? 8 : 0;
}

// Geometry of one level's (slab of the) grid.  2-D grids are stored as (nx, 1, nz).
struct Grid {
    int nx, ny, nz;          // global nodes per axis (ny == 1 in 2-D)
    int k0, nk;              // owned planes [k0, k0+nk) along z (the slab axis)
    int64_t plane;           // nx*ny
    int64_t lead;            // elements in front of the first owned row (lower halo plane or 0)
    int refine_y;            // 1 if the y axis takes part in coarsening (3-D)
};

// Transfer / generator kernels run one thread per node of a grid plane: blockIdx.x * blockDim.x +
// threadIdx.x enumerates the plane (flattened, so odd line lengths leave no ragged tail per line),
// blockIdx.y the owned planes.
__device__ __forceinline__ bool plane_node(const Grid& g, int* i, int* j) {
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (unsigned)g.plane) return false;
    const unsigned jj = t / (unsigned)g.nx;
    *j = (int)jj;
    *i = (int)(t - jj * (unsigned)g.nx);
    return true;
}

// ---- XCD-aware block -> tile map ------------------------------------------------------
// Blocks are dealt round-robin over the 8 XCDs (b % 8 says which blocks share an L2).
// With chunk c > 1, each XCD works on c consecutive tiles at a time, so lines of the
// source vector fetched for one grid row are re-used from the same L2 by the rows next
// to it, while all XCDs still advance through the vector together (which keeps the
// plane-distance re-reads inside the 256 MiB memory-side cache).  Speed only.
__device__ __forceinline__ unsigned swizzle_block(unsigned b, unsigned nb, unsigned chunk) {
    if (chunk <= 1) return b;
    const unsigned group = 8u * chunk;
    const unsigned full = (nb / group) * group;
    if (b >= full) return b;
    const unsigned g = b / group, in = b % group;
    const unsigned xcd = in % 8u, idx = in / 8u;
    return g * group + xcd * chunk + idx;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Block-wide sum, result valid in thread 0.
__device__ __forceinline__ double block_sum(double v) {
    __shared__ double s_part[WAVES_PER_BLOCK];
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_part[wave] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < WAVES_PER_BLOCK; ++w) t += s_part[w];
    }
    return t;
}

struct EllArgs {
    const double* vals;
    const int* cols;
    const double* x;        // source vector, base of storage (cols index into it)
    const double* f;        // right-hand side, row-based (already offset by lead)
    const double* dinv;     // 1/diag, row-based
    double* out;            // row-based
    double* partials;       // per-block partial dot (MODE_SPMV with DOT)
    const int* done_flag;   // optional early-exit flag (coarse solver)
    int64_t nloc;           // owned rows
    int64_t lead;           // x[lead + row] is the row's own entry
    int64_t slice0, nslices;  // slices [slice0, slice0 + nslices) are processed
    // ... in two ranges when gap > 0: the slices from the split-th on lie `gap` slices further up (a slab's first and
    // last slices in one launch)
    int64_t split, gap;
    double omega;
    int W;                  // runtime width (used when the template width is 0)
    unsigned chunk;
    // offset-coded columns
    const unsigned long long* codes;
    const int* offsets;
    int ntable;
    int dcode;              // code of offset 0 (the diagonal / padding)
    // XCD strip traversal (strip_ns == 0: chunked map): see strip_block()
    int color;              // MODE_GS: colour relaxed by this launch
    int color_kind;         // MODE_GS: 0 = parity of the global lexicographic index (red-black), 1 = the nine lattice
                            //          colours of P2 rows (row_color)
    int64_t grow0;          // MODE_GS: global lexicographic index of local row 0
    int gnx, gny;           // MODE_GS: lattice points per line / lines per plane (colour kind 1)
    // symmetric-diagonal storage (sdia_apply): up[0] = 0, up[1..] = the positive offsets, ascending
    int up[8];
    int64_t mlead;          // matrix rows stored in front of row 0 (multiple of the slice height)
    unsigned strip_ns;      // strips per pseudo-plane, a multiple of 8 (one share per XCD)
    unsigned strip_bmax;    // blocks (4 slices each) in the widest strip
    unsigned ps4;           // blocks per pseudo-plane
    unsigned kp;            // pseudo-planes
    // row classes (sdia_cls_apply): class of row r = cls[r] (row-based, zero-padded), the row's seven entries in
    // ctab[class][CLS_W]; ncls classes in use; cmain = the most frequent one, entries by value in cm[]
    const unsigned char* cls;
    const double* ctab;
    int ncls, cmain;
    double cm[8];
    unsigned nvirt;         // groups of four slices to process (persistent blocks stride over them)
};

// ---- XCD strip traversal ----------------------------------------------------------------------
// For 3-D grids the k+-1 neighbours of a row are a whole plane (8.4 MB at 1025^2) away, far beyond a
// 4 MiB XCD L2, so with a plane-by-plane sweep every x line is pulled across the fabric ~4 times.
// Here each XCD instead walks a STRIP of ~8 grid lines through all planes: the slices are viewed as
// pseudo-planes of `ps` slices (the plane size rounded to whole slices; the ~1-row drift per plane
// only blurs locality); every pseudo-plane is cut into `strip_ns` = 8m strips of (almost) equal width,
// strip s belongs to XCD s % 8 -- every XCD gets exactly m strips, so there is no tail -- and the blocks
// of one XCD (blockIdx % 8, dealt round-robin by the dispatcher) enumerate (strip, plane,
// block-in-strip) with the plane index fastest.  The three x planes a strip touches at step k
// (~64 KB each) are then still in that XCD's L2 at steps k+1 and k+2.  It is a bijection on slices
// whatever the dispatcher does: placement changes speed only.
// Returns the first slice of the block's 4-slice group, or -1 for a padding block.
__device__ __forceinline__ int64_t strip_block(const EllArgs& a, unsigned b) {
    const unsigned x = b & 7u, q = b >> 3;
    const unsigned bi = q % a.strip_bmax, t = q / a.strip_bmax;
    const unsigned k = t % a.kp, s = (t / a.kp) * 8u + x;
    const unsigned lo = (unsigned)(((unsigned long long)s * a.ps4) / a.strip_ns);
    const unsigned hi = (unsigned)(((unsigned long long)(s + 1) * a.ps4) / a.strip_ns);
    if (bi >= hi - lo) return -1;
    return ((int64_t)k * a.ps4 + lo + bi) * 4;
}

typedef double dvec2_t __attribute__((ext_vector_type(2)));
typedef double dvec4_t __attribute__((ext_vector_type(4)));
typedef unsigned long long uvec2_t __attribute__((ext_vector_type(2)));
typedef unsigned long long uvec4_t __attribute__((ext_vector_type(4)));

// Streaming (non-temporal) loads/stores for data touched once per sweep, so that they do not evict
// the re-used x lines from L2.
template <int R, bool NT> __device__ __forceinline__ DVec<R> load_d(const double* p) {
    DVec<R> o;
    if constexpr (!NT) return *reinterpret_cast<const DVec<R>*>(p);
    else if constexpr (R == 1) { o.d[0] = __builtin_nontemporal_load(p); }
    else if constexpr (R == 2) { const dvec2_t t = __builtin_nontemporal_load(reinterpret_cast<const dvec2_t*>(p)); o.d[0] = t.x; o.d[1] = t.y; }
    else { const dvec4_t t = __builtin_nontemporal_load(reinterpret_cast<const dvec4_t*>(p)); o.d[0] = t.x; o.d[1] = t.y; o.d[2 % R] = t.z; o.d[3 % R] = t.w; }
    return o;
}
template <int R, bool NT> __device__ __forceinline__ UVec<R> load_u(const unsigned long long* p) {
    UVec<R> o;
    if constexpr (!NT) return *reinterpret_cast<const UVec<R>*>(p);
    else if constexpr (R == 1) { o.d[0] = __builtin_nontemporal_load(p); }
    else if constexpr (R == 2) { const uvec2_t t = __builtin_nontemporal_load(reinterpret_cast<const uvec2_t*>(p)); o.d[0] = t.x; o.d[1] = t.y; }
    else { const uvec4_t t = __builtin_nontemporal_load(reinterpret_cast<const uvec4_t*>(p)); o.d[0] = t.x; o.d[1] = t.y; o.d[2 % R] = t.z; o.d[3 % R] = t.w; }
    return o;
}
template <int R, bool NT> __device__ __forceinline__ void store_d(double* p, const DVec<R>& v) {
    if constexpr (!NT) { *reinterpret_cast<DVec<R>*>(p) = v; }
    else if constexpr (R == 1) { __builtin_nontemporal_store(v.d[0], p); }
    else if constexpr (R == 2) { dvec2_t t; t.x = v.d[0]; t.y = v.d[1]; __builtin_nontemporal_store(t, reinterpret_cast<dvec2_t*>(p)); }
    else { dvec4_t t; t.x = v.d[0]; t.y = v.d[1]; t.z = v.d[2 % R]; t.w = v.d[3 % R]; __builtin_nontemporal_store(t, reinterpret_cast<dvec4_t*>(p)); }
}

// Shared output stage of the offset-coded and symmetric-diagonal kernels: acc = (A x)_row, diag = a_ii,
// xr = x_row for the R rows of this lane.
template <int R, int MODE, bool DOT, bool NT>
__device__ __forceinline__ void tile_epilogue(const EllArgs& a, int64_t row, const double* acc, const double* diag,
                                              const double* xr, double& dot) {
    DVec<R> o;
        if (MODE != MODE_GS && row + R <= a.nloc) {
            if (MODE == MODE_SPMV) {
#pragma unroll
                for (int r = 0; r < R; ++r) o.d[r] = acc[r];
                if (DOT) {
#pragma unroll
                    for (int r = 0; r < R; ++r) dot += xr[r] * acc[r];
                }
            } else {
                const DVec<R> fr = load_d<R, NT>(a.f + row);
                if (MODE == MODE_RESIDUAL) {
#pragma unroll
                    for (int r = 0; r < R; ++r) o.d[r] = fr.d[r] - acc[r];
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) o.d[r] = xr[r] + (a.omega * (1.0 / diag[r])) * (fr.d[r] - acc[r]);
                }
            }
            // the Jacobi / SpMV output is the next sweep's gathered source: keep it cacheable
            store_d<R, NT && MODE == MODE_RESIDUAL>(a.out + row, o);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t rr = row + r;
                if (MODE == MODE_GS && lattice_color(a.color_kind, rr + a.grow0, a.gnx, a.gny) != a.color) continue;
                if (rr < a.nloc) {
                    double val;
                    if (MODE == MODE_SPMV) {
                        val = acc[r];
                        if (DOT) dot += xr[r] * acc[r];
                    } else if (MODE == MODE_RESIDUAL) {
                        val = a.f[rr] - acc[r];
                    } else {
                        val = xr[r] + (a.omega * (1.0 / diag[r])) * (a.f[rr] - acc[r]);
                    }
                    a.out[rr] = val;
                }
            }
        }
    }

// Offset-coded variant of ell_apply (same modes, same arithmetic, 8 B of column data per row).
template <int WT, int R, int MODE, bool DOT, bool NT>
__global__ __launch_bounds__(BLOCK) void ell_apply_coded(EllArgs a) {
    __shared__ int s_off[256];
    if (a.done_flag && *a.done_flag) return;
    constexpr int CW = (WT + 7) / 8;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t sl;
    if (a.strip_ns) {
        const int64_t first = strip_block(a, blockIdx.x);
        if (!DOT && first < 0) return;
        sl = first < 0 ? a.nslices : first + wave;
    } else {
        sl = (int64_t)swizzle_block(blockIdx.x, gridDim.x, a.chunk) * WAVES_PER_BLOCK + wave;
    }
    for (int t = threadIdx.x; t < a.ntable; t += BLOCK) s_off[t] = a.offsets[t];
    __syncthreads();
    double dot = 0.0;
    if (MODE == MODE_GS && a.color_kind == COLOR_LATTICE9 && sl < a.nslices) {
        // a colour fixes the parity of the line (j, k) except for the two vertex colours, which fix (even, even): a slice
        // (64 R consecutive rows: part of one or two lines when a line is longer) none of whose lines can hold the colour
        // is skipped before it streams its part of the matrix
        const int64_t g0 = (a.slice0 + sl + (sl >= a.split ? a.gap : 0)) * (WAVE * R) + a.grow0, g1 = g0 + WAVE * R - 1;
        const int64_t l0 = g0 / a.gnx, l1 = g1 / a.gnx;
        const int want = (a.color == 8 ? 0 : a.color) >> 1;              // bits (j & 1) | (k & 1) << 1
        bool any = false;
        for (int64_t l = l0; l <= l1 && l < l0 + 4; ++l) any = any || ((int)((l % a.gny) & 1) | (int)(((l / a.gny) & 1) << 1)) == want;
        if (l1 - l0 >= 4) any = true;
        if (!any) sl = a.nslices;
    }
    if (sl < a.nslices) {
        const int64_t slice = a.slice0 + sl + (sl >= a.split ? a.gap : 0);
        const int W = WT > 0 ? WT : a.W;                       // WT == 0: run-time width (wide stencils)
        const int ncw = WT > 0 ? CW : (a.W + 7) / 8;
        const size_t base = (size_t)slice * (size_t)W * (WAVE * R) + (size_t)lane * R;
        const size_t cbase = (size_t)slice * (size_t)ncw * (WAVE * R) + (size_t)lane * R;
        const int64_t row = slice * (WAVE * R) + (int64_t)lane * R;
        const double* xrow = a.x + a.lead + row;
        double acc[R], diag[R], xr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r] = 0.0; diag[r] = 1.0; xr[r] = 0.0; }
        auto entry = [&](const UVec<R>& word, int kk, const DVec<R>& val) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int code = (int)((word.d[r] >> (8 * kk)) & 0xffull);
                const double xv = xrow[r + s_off[code]];
                if (MODE != MODE_RESIDUAL) {
                    const bool on_diag = code == a.dcode;
                    if (on_diag) xr[r] = xv;
                    if (on_diag && val.d[r] != 0.0) diag[r] = val.d[r];
                }
                acc[r] = fma(val.d[r], xv, acc[r]);
            }
        };
        if constexpr (WT > 0) {
            DVec<R> v[WT > 0 ? WT : 1];
            UVec<R> cw[CW > 0 ? CW : 1];
#pragma unroll
            for (int q = 0; q < CW; ++q) cw[q] = load_u<R, NT>(a.codes + cbase + (size_t)q * (WAVE * R));
#pragma unroll
            for (int k = 0; k < WT; ++k) v[k] = load_d<R, NT>(a.vals + base + (size_t)k * (WAVE * R));
#pragma unroll
            for (int k = 0; k < WT; ++k) entry(cw[k / 8], k % 8, v[k]);
        } else {
            for (int q = 0; q < ncw; ++q) {
                const UVec<R> word = load_u<R, NT>(a.codes + cbase + (size_t)q * (WAVE * R));
                const int kend = min(8, W - 8 * q);
                DVec<R> v[8];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    if (kk < kend) v[kk] = load_d<R, NT>(a.vals + base + (size_t)(8 * q + kk) * (WAVE * R));
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
                    if (kk < kend) entry(word, kk, v[kk]);
            }
        }
        tile_epilogue<R, MODE, DOT, NT>(a, row, acc, diag, xr, dot);
    }
    if (DOT) {
        const double t = block_sum(dot);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
    }
}

// ---- offset-coded rows read through stencil classes -------------------------------------------------
// Wide stencils on structured meshes repeat as well: a P2 level has one row per parity class of the lattice point and
// position next to the boundary (217 at most in 3-D).  Where an offset-coded level has at most 255 distinct rows --
// the W (offset, value) pairs in stored order, bit for bit -- a row is one class byte and the pairs come from a table
// in global memory that stays in L2 (classes x W x 12 bytes): 25 instead of 8 W + 8 ceil(W/8) + 24 bytes per row and
// sweep (488 for 3-D P2).  Entries are applied in stored order like ell_apply_coded does (trailing padding skipped:
// it adds 0 * x[row]), so results are the same.  All four modes; lanes of one wave run as many entries as their class has.
struct SclsArgs {
    const double* vals;                 // offset-coded ELL
    const unsigned long long* codes;
    const int* offsets;                 // code -> offset
    int W, dcode;
    int64_t nloc, nslices;
    unsigned long long* tags;           // SCLS_SLOTS hash tags, 0 = free
    int64_t* slot_row;                  // SCLS_SLOTS: a row that has the slot's stencil
    int* count;                         // distinct rows seen
    int* slot_class;                    // SCLS_SLOTS
    unsigned char* cls;                 // nloc (+ padding): class of every row
    int* s_off;                         // 256 x W
    double* s_val;                      // 256 x W
    int* s_cnt;                         // 256: entries in use per class (up to the last non-zero value)
    int* flag;
};
constexpr int SCLS_SLOTS = 4096;

template <int R> __device__ __forceinline__ unsigned long long scls_hash(const SclsArgs& a, int64_t row) {
    constexpr int S = WAVE * R;
    const int64_t slice = row / S, within = row % S;
    const int CW = (a.W + 7) / 8;
    unsigned long long h = 0x9e3779b97f4a7c15ull;
    for (int k = 0; k < a.W; ++k) {
        const unsigned long long v = (unsigned long long)__double_as_longlong(a.vals[((size_t)slice * a.W + k) * S + within]);
        const unsigned long long w = a.codes[((size_t)slice * CW + k / 8) * S + within];
        const unsigned long long code = (w >> (8 * (k % 8))) & 0xffull;
        unsigned long long x = h ^ (v + 0x3c6ef372fe94f82bull * (unsigned long long)(k + 1)) ^ (code << 56);
        x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
        h = x;
    }
    return h ? h : 1ull;
}

template <int R> __device__ __forceinline__ bool scls_same(const SclsArgs& a, int64_t r0, int64_t r1) {
    constexpr int S = WAVE * R;
    const int CW = (a.W + 7) / 8;
    const int64_t s0 = r0 / S, w0 = r0 % S, s1 = r1 / S, w1 = r1 % S;
    for (int k = 0; k < a.W; ++k) {
        if (__double_as_longlong(a.vals[((size_t)s0 * a.W + k) * S + w0]) != __double_as_longlong(a.vals[((size_t)s1 * a.W + k) * S + w1])) return false;
        const unsigned long long c0 = (a.codes[((size_t)s0 * CW + k / 8) * S + w0] >> (8 * (k % 8))) & 0xffull;
        const unsigned long long c1 = (a.codes[((size_t)s1 * CW + k / 8) * S + w1] >> (8 * (k % 8))) & 0xffull;
        if (c0 != c1) return false;
    }
    return true;
}

template <int R>
__global__ void scls_insert(SclsArgs a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    const unsigned long long h = scls_hash<R>(a, row);
    unsigned s = (unsigned)h & (SCLS_SLOTS - 1);
    for (int probe = 0; probe < SCLS_SLOTS; ++probe) {
        const unsigned long long seen = *(volatile unsigned long long*)(a.tags + s);
        if (seen == h) return;
        if (seen != 0ull) { s = (s + 1) & (SCLS_SLOTS - 1); continue; }
        if (*(volatile int*)a.count > 255) return;
        const unsigned long long old = atomicCAS(a.tags + s, 0ull, h);
        if (old == 0ull) { a.slot_row[s] = row; atomicAdd(a.count, 1); return; }
        if (old == h) return;
        s = (s + 1) & (SCLS_SLOTS - 1);
    }
}

// one block: classes 0, 1, ... in slot order; the table rows are copied from each slot's representative row
template <int R>
__global__ void scls_assign(SclsArgs a) {
    constexpr int S = WAVE * R;
    __shared__ int s_id[1];
    if (threadIdx.x == 0) {
        int id = 0;
        for (int s = 0; s < SCLS_SLOTS; ++s) {
            a.slot_class[s] = -1;
            if (a.tags[s] && id < 256) a.slot_class[s] = id++;
        }
        s_id[0] = id;
    }
    __syncthreads();
    const int CW = (a.W + 7) / 8;
    for (int s = threadIdx.x; s < SCLS_SLOTS; s += blockDim.x) {
        const int id = a.slot_class[s];
        if (id < 0) continue;
        const int64_t row = a.slot_row[s], slice = row / S, within = row % S;
        int cnt = 0;
        for (int k = 0; k < a.W; ++k) {
            const double v = a.vals[((size_t)slice * a.W + k) * S + within];
            const unsigned long long w = a.codes[((size_t)slice * CW + k / 8) * S + within];
            a.s_val[(size_t)id * a.W + k] = v;
            a.s_off[(size_t)id * a.W + k] = a.offsets[(int)((w >> (8 * (k % 8))) & 0xffull)];
            if (v != 0.0) cnt = k + 1;
        }
        a.s_cnt[id] = cnt;
    }
}

template <int R>
__global__ void scls_encode(SclsArgs a) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    const unsigned long long h = scls_hash<R>(a, row);
    unsigned s = (unsigned)h & (SCLS_SLOTS - 1);
    for (int probe = 0; probe < SCLS_SLOTS; ++probe) {
        const unsigned long long t = a.tags[s];
        if (t == h) {
            const int id = a.slot_class[s];
            if (id < 0 || !scls_same<R>(a, row, a.slot_row[s])) atomicExch(a.flag, 1);     // overflow / hash collision
            a.cls[row] = (unsigned char)(id < 0 ? 0 : id);
            return;
        }
        if (t == 0ull) break;
        s = (s + 1) & (SCLS_SLOTS - 1);
    }
    atomicExch(a.flag, 1);
    a.cls[row] = 0;
}

// the four modes on class-coded wide rows (one wave = one slice, like ell_apply_coded)
template <int R, int MODE, bool DOT, bool NT>
__global__ __launch_bounds__(BLOCK) void ell_cls_apply(EllArgs a, const unsigned char* __restrict__ cls, const int* __restrict__ s_off,
                                                        const double* __restrict__ s_val, const int* __restrict__ s_cnt) {
    if (a.done_flag && *a.done_flag) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t sl = (int64_t)swizzle_block(blockIdx.x, gridDim.x, a.chunk) * WAVES_PER_BLOCK + wave;
    double dot = 0.0;
    if (MODE == MODE_GS && a.color_kind == COLOR_LATTICE9 && sl < a.nslices) {
        const int64_t g0 = (a.slice0 + sl + (sl >= a.split ? a.gap : 0)) * (WAVE * R) + a.grow0, g1 = g0 + WAVE * R - 1;
        const int64_t l0 = g0 / a.gnx, l1 = g1 / a.gnx;
        const int want = (a.color == 8 ? 0 : a.color) >> 1;
        bool any = l1 - l0 >= 4;
        for (int64_t l = l0; l <= l1 && l < l0 + 4; ++l) any = any || ((int)((l % a.gny) & 1) | (int)(((l / a.gny) & 1) << 1)) == want;
        if (!any) sl = a.nslices;
    }
    if (sl < a.nslices) {
        const int64_t slice = a.slice0 + sl + (sl >= a.split ? a.gap : 0);
        const int64_t row = slice * (WAVE * R) + (int64_t)lane * R;
        const double* xrow = a.x + a.lead + row;
        double acc[R], diag[R], xr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = 0.0; diag[r] = 1.0; xr[r] = 0.0;
            if (row + r >= a.nloc) continue;
            if (MODE == MODE_GS && lattice_color(a.color_kind, row + r + a.grow0, a.gnx, a.gny) != a.color) continue;
            const int c = cls[row + r];
            const int n = s_cnt[c];
            const int* po = s_off + (size_t)c * a.W;
            const double* pv = s_val + (size_t)c * a.W;
            double s = 0.0;
            for (int t = 0; t < n; ++t) {
                const int off = po[t];
                const double v = pv[t];
                const double xv = xrow[r + off];
                if (MODE != MODE_RESIDUAL && off == 0) { xr[r] = xv; if (v != 0.0) diag[r] = v; }
                s = fma(v, xv, s);
            }
            acc[r] = s;
        }
        tile_epilogue<R, MODE, DOT, NT>(a, row, acc, diag, xr, dot);
    }
    if (DOT) {
        const double t = block_sum(dot);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
    }
}

// ---- symmetric diagonal storage ---------------------------------------------------------------
// When a level's offsets are symmetric (o in the table <=> -o in the table, which every grid stencil
// satisfies) and the matrix is bit-for-bit symmetric (a_ij == a_ji: checked at set-up), entry k of every
// row can sit in the slot of its offset, which makes the per-row codes redundant, and the lower entry at
// offset -d of row i is the upper entry at +d of row i-d, which makes the lower half redundant:
//     dvals[(((row + mlead) / S) * WU + c) * S + (row + mlead) % S]    c = 0 diagonal, c >= 1 offset +up[c]
// with S = 64*R and WU = (W+1)/2 (4 for 7-point rows).  The kernel streams 8*WU bytes per row (32 B
// instead of 56 + 8), re-reads the lower entries as shifted, coalesced loads that hit L2 / Infinity
// Cache like the x neighbours do, and applies them in ascending column order -- the same fma chain as the
// other kernels, so results are bit-identical.  `mlead` rows in front of row 0 hold the lower entries
// of the first rows whose partner rows live on another slab (or do not exist: zeros).
// Per row and sweep: 32 (matrix) + 8 (x) + 8 (f) + 8 (out) = 56 B.
template <int WU, int R, int MODE, bool DOT, bool NT>
__device__ __forceinline__ void sdia_body(const EllArgs& a) {
    if (a.done_flag && *a.done_flag) return;
    constexpr int S = WAVE * R;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int64_t sl;
    if (a.strip_ns) {
        const int64_t first = strip_block(a, blockIdx.x);
        if (!DOT && first < 0) return;
        sl = first < 0 ? a.nslices : first + wave;
    } else {
        sl = (int64_t)swizzle_block(blockIdx.x, gridDim.x, a.chunk) * WAVES_PER_BLOCK + wave;
    }
    double dot = 0.0;
    if (sl < a.nslices) {
        const int64_t slice = a.slice0 + sl + (sl >= a.split ? a.gap : 0);
        const int64_t row = slice * S + (int64_t)lane * R;
        const int64_t m = row + a.mlead;                               // matrix row of r = 0 (m % S == lane*R)
        const size_t base = (size_t)(m / S) * WU * S + (size_t)lane * R;
        const double* xrow = a.x + a.lead + row;
        DVec<R> up[WU];
        // the diagonal is read once (streamed); the upper slots are read again as the lower entries of
        // later rows and must stay cacheable
        up[0] = load_d<R, NT>(a.vals + base);
#pragma unroll
        for (int c = 1; c < WU; ++c) up[c] = load_d<R, false>(a.vals + base + (size_t)c * S);
        // the lower entries and the x neighbours of a lane's R rows are R consecutive doubles at shifted
        // (8-byte aligned) addresses: one unaligned vector load each, unless the run crosses a slice
        DVecU<R> lo[WU], xl[WU], xu[WU];
#pragma unroll
        for (int c = 1; c < WU; ++c) {
            const int64_t mm = m - a.up[c];
            const int64_t in = mm % S;
            if (in + R <= S) {
                lo[c] = *reinterpret_cast<const DVecU<R>*>(a.vals + ((size_t)(mm / S) * WU + c) * S + (size_t)in);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int64_t m2 = mm + r;
                    lo[c].d[r] = a.vals[((size_t)(m2 / S) * WU + c) * S + (size_t)(m2 % S)];
                }
            }
            xl[c] = *reinterpret_cast<const DVecU<R>*>(xrow - a.up[c]);
            xu[c] = *reinterpret_cast<const DVecU<R>*>(xrow + a.up[c]);
        }
        const DVec<R> x0 = *reinterpret_cast<const DVec<R>*>(xrow);
        double acc[R], diag[R], xr[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            acc[r] = 0.0;
#pragma unroll
            for (int c = WU - 1; c >= 1; --c) acc[r] = fma(lo[c].d[r], xl[c].d[r], acc[r]);
            xr[r] = x0.d[r];
            diag[r] = up[0].d[r] != 0.0 ? up[0].d[r] : 1.0;
            acc[r] = fma(up[0].d[r], xr[r], acc[r]);
#pragma unroll
            for (int c = 1; c < WU; ++c) acc[r] = fma(up[c].d[r], xu[c].d[r], acc[r]);
        }
        tile_epilogue<R, MODE, DOT, NT>(a, row, acc, diag, xr, dot);
    }
    if (DOT) {
        const double t = block_sum(dot);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
    }
}

template <int WU, int R, int MODE, bool DOT, bool NT>
__global__ __launch_bounds__(BLOCK) void sdia_apply(EllArgs a) {
    sdia_body<WU, R, MODE, DOT, NT>(a);
}

// The same Jacobi sweep under its own symbol for the FINEST level, so that profiler summaries
// (rocprofv3 --stats) list the dominant launches apart from the short coarse-level ones.
template <int WU, int R, bool NT>
__global__ __launch_bounds__(BLOCK) void sdia_jacobi_finest(EllArgs a) {
    sdia_body<WU, R, MODE_JACOBI, false, NT>(a);
}

// ---- symmetric diagonal storage read through row classes ------------------------------------------
// Where a level has row classes (mg_jacobi2.hip.h: at most 255 distinct FULL rows, bit for bit), the one-sweep
// kernels need not stream the 32-byte rows either: one class byte per row names the row's seven entries
// (ctab[class][CLS_W], ascending column order).  Same entries, same order, same results as sdia_body; per row and
// sweep 1 (class) + 8 (x) + 8 (f) + 8 (out) = 25 B instead of 56 B.  All four modes (residual, Jacobi, SpMV[+dot],
// Gauss-Seidel colour); WU = 4 seven-point rows {0, +-1, +-nx, +-P}, WU = 3 the five-point rows of 2-D levels.
//   * blocks are persistent (grid-stride over groups of four slices), so the table -- only the classes in use,
//     64 B each -- is copied into LDS once per block, behind the first group's global loads;
//   * a wave whose 64*R rows all carry the level's most frequent class (the interior stencil) takes the entries
//     from the kernel arguments (scalar registers) and touches no table at all.
constexpr int CLS_W = 8;

template <int WU, int R, int MODE, bool DOT, bool NT>
__device__ __forceinline__ void sdia_cls_body(const EllArgs& a) {
    __shared__ double sT[256 * CLS_W];
    static_assert(WU == 3 || WU == 4, "row classes: five- and seven-point rows");
    if (a.done_flag && *a.done_flag) return;
    constexpr int S = WAVE * R;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    double dot = 0.0;
    bool filled = false;
    for (unsigned vb = blockIdx.x; vb < a.nvirt; vb += gridDim.x) {
        int64_t sl;
        if (a.strip_ns) {
            const int64_t first = strip_block(a, vb);
            sl = first < 0 ? a.nslices : first + wave;
        } else {
            sl = (int64_t)swizzle_block(vb, a.nvirt, a.chunk) * WAVES_PER_BLOCK + wave;
        }
        const bool active = sl < a.nslices;
        const int64_t row = (a.slice0 + (active ? sl + (sl >= a.split ? a.gap : 0) : 0)) * S + (int64_t)lane * R;
        const double* xrow = a.x + a.lead + row;
        int own[R];
        DVecU<R> xl[WU], xu[WU];
        DVec<R> x0;
        if (active) {
            const unsigned char* crow = a.cls + row;
            if constexpr (R == 1) own[0] = crow[0];
            else if constexpr (R == 2) {
                const unsigned w = *reinterpret_cast<const unsigned short*>(crow);
                own[0] = (int)(w & 255u); own[1] = (int)(w >> 8);
            } else {
                const unsigned w = *reinterpret_cast<const unsigned*>(crow);
#pragma unroll
                for (int r = 0; r < R; ++r) own[r] = (int)((w >> (8 * r)) & 255u);
            }
#pragma unroll
            for (int c = 1; c < WU; ++c) {
                xl[c] = *reinterpret_cast<const DVecU<R>*>(xrow - a.up[c]);
                xu[c] = *reinterpret_cast<const DVecU<R>*>(xrow + a.up[c]);
            }
            x0 = *reinterpret_cast<const DVec<R>*>(xrow);
        }
        if (!filled) {
            const int nt = a.ncls * CLS_W;
            for (int i = threadIdx.x; i < nt; i += BLOCK) sT[i] = a.ctab[i];
            __syncthreads();
            filled = true;
        }
        if (!active) continue;
        bool other = false;
#pragma unroll
        for (int r = 0; r < R; ++r) other = other || own[r] != a.cmain;
        const bool fast = __builtin_amdgcn_readfirstlane((int)(__ballot(other) == 0ull)) != 0;
        double acc[R], diag[R], xr[R];
        auto apply = [&](int r, double e0, double e1, double e2, double e3, double e4, double e5, double e6) {
            double s = 0.0;
            if constexpr (WU == 4) s = fma(e0, xl[WU - 1].d[r], s);
            s = fma(e1, xl[2].d[r], s);
            s = fma(e2, xl[1].d[r], s);
            xr[r] = x0.d[r];
            diag[r] = e3 != 0.0 ? e3 : 1.0;
            s = fma(e3, xr[r], s);
            s = fma(e4, xu[1].d[r], s);
            s = fma(e5, xu[2].d[r], s);
            if constexpr (WU == 4) s = fma(e6, xu[WU - 1].d[r], s);
            acc[r] = s;
        };
        if (fast) {
#pragma unroll
            for (int r = 0; r < R; ++r) apply(r, a.cm[0], a.cm[1], a.cm[2], a.cm[3], a.cm[4], a.cm[5], a.cm[6]);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const dvec2_t* const tr = reinterpret_cast<const dvec2_t*>(sT + CLS_W * own[r]);
                const dvec2_t t01 = tr[0], t23 = tr[1], t45 = tr[2], t67 = tr[3];
                apply(r, t01.x, t01.y, t23.x, t23.y, t45.x, t45.y, t67.x);
            }
        }
        tile_epilogue<R, MODE, DOT, NT>(a, row, acc, diag, xr, dot);
    }
    if (DOT) {
        const double t = block_sum(dot);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
    }
}

template <int WU, int R, int MODE, bool DOT, bool NT>
__global__ __launch_bounds__(BLOCK) void sdia_cls_apply(EllArgs a) {
    sdia_cls_body<WU, R, MODE, DOT, NT>(a);
}

// (its own symbol for the finest level's Jacobi sweep, like sdia_jacobi_finest)
template <int WU, int R, bool NT>
__global__ __launch_bounds__(BLOCK) void sdia_cls_jacobi_finest(EllArgs a) {
    sdia_cls_body<WU, R, MODE_JACOBI, false, NT>(a);
}

struct SdiaArgs {
    const double* vals;                 // offset-coded ELL being converted
    const unsigned long long* codes;
    const int* offsets;                 // code -> offset
    int W, WU, NU, dcode;               // WU = slots laid out, NU <= WU = slots in use
    int up[8];                          // slot -> positive offset (up[0] = 0; unused slots 0)
    int64_t nloc, mlead;
    double* dvals;
};

__device__ __forceinline__ int sdia_slot(const SdiaArgs& a, int off) {
    for (int c = 0; c < a.NU; ++c)
        if (a.up[c] == off) return c;
    return 0;
}

// Pass 1: diagonal and upper entries of every row; lower entries whose partner row is not stored
// locally go to the partner's slot in the lead region.
template <int R>
__global__ void sdia_fill(SdiaArgs a) {
    constexpr int S = WAVE * R;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    const int CW = (a.W + 7) / 8;
    const int64_t slice = row / S, within = row % S;
    const int64_t m = row + a.mlead;
    for (int k = 0; k < a.W; ++k) {
        const double v = a.vals[((size_t)slice * a.W + k) * S + within];
        if (v == 0.0) continue;
        const unsigned long long w = a.codes[((size_t)slice * CW + k / 8) * S + within];
        const int off = a.offsets[(int)((w >> (8 * (k % 8))) & 0xffull)];
        if (off >= 0) {
            const int c = sdia_slot(a, off);
            a.dvals[((size_t)(m / S) * a.WU + c) * S + (size_t)(m % S)] = v;
        } else if (row + off < 0) {
            const int c = sdia_slot(a, -off);
            const int64_t mm = m + off;
            a.dvals[((size_t)(mm / S) * a.WU + c) * S + (size_t)(mm % S)] = v;
        }
    }
}

// Pass 2: every row's lower entries (absent = 0) must equal, bit for bit, what the symmetric kernel will
// read for them; flag[0] |= 1 otherwise.
// report (set-up diagnostics, mg_level_storage): report[0] = lowest row (+ 1) with a pair that is not bit-for-bit
// symmetric, report[1] = largest distance between the halves of a pair in units in the last place (2^62: a pair with
// one half absent or of the other sign)
template <int R>
__global__ void sdia_check(SdiaArgs a, int* flag, int qbits, unsigned long long* report) {
    constexpr int S = WAVE * R;
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    const int CW = (a.W + 7) / 8;
    const int64_t slice = row / S, within = row % S;
    const int64_t m = row + a.mlead;
    double lower[8];
    for (int c = 0; c < 8; ++c) lower[c] = 0.0;
    for (int k = 0; k < a.W; ++k) {
        const double v = a.vals[((size_t)slice * a.W + k) * S + within];
        if (v == 0.0) continue;
        const unsigned long long w = a.codes[((size_t)slice * CW + k / 8) * S + within];
        const int off = a.offsets[(int)((w >> (8 * (k % 8))) & 0xffull)];
        if (off < 0) lower[sdia_slot(a, -off)] = v;
    }
    bool bad = false;
    for (int c = 1; c < a.NU; ++c) {
        const int64_t mm = m - a.up[c];
        const double want = a.dvals[((size_t)(mm / S) * a.WU + c) * S + (size_t)(mm % S)];
        // ("storage_ulps": the two halves of a pair may differ in their lowest qbits mantissa bits; the upper one is kept)
        const long long dw = __double_as_longlong(want) - __double_as_longlong(lower[c]);
        const bool close = qbits > 0 && (want < 0.0) == (lower[c] < 0.0) && (dw < 0 ? -dw : dw) <= (1ll << qbits);
        const bool differs = __double_as_longlong(want) != __double_as_longlong(lower[c]) && !(want == 0.0 && lower[c] == 0.0);
        if (differs && !close) bad = true;
        if (differs) {
            const bool comparable = want != 0.0 && lower[c] != 0.0 && (want < 0.0) == (lower[c] < 0.0);
            atomicMin(report, (unsigned long long)(row + 1));
            atomicMax(report + 1, comparable ? (unsigned long long)(dw < 0 ? -dw : dw) : (1ull << 62));
        }
    }
    if (bad) atomicOr(flag, 1);
}

// ---- building the offset code book (set-up) ---------------------------------------------------
constexpr int kDeltaSlots = 1024;
constexpr int kDeltaEmpty = INT32_MIN;

__device__ __forceinline__ unsigned delta_hash(int d) { return ((unsigned)d * 2654435761u) >> 22; }

// Distinct values of col - row over all stored entries -> open-addressing set
// (count[0] = size, count[1] = 1 if the set overflowed).
template <int R>
__global__ void ell_collect_deltas(const int* __restrict__ cols, int64_t nslices, int W, int64_t nloc, int64_t lead,
                                   int* table, int* count) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nslices * W * (WAVE * R);
    if (t >= total) return;
    const int64_t per_slice = (int64_t)W * (WAVE * R);
    const int64_t slice = t / per_slice;
    const int64_t within = (t % per_slice) % (WAVE * R);
    const int64_t row = slice * (WAVE * R) + within;
    const int delta = row < nloc ? cols[t] - (int)(lead + row) : 0;
    unsigned h = delta_hash(delta);
    for (int probe = 0; probe < kDeltaSlots; ++probe, h = (h + 1) & (kDeltaSlots - 1)) {
        const int cur = table[h];
        if (cur == delta) return;
        if (cur == kDeltaEmpty) {
            const int old = atomicCAS(&table[h], kDeltaEmpty, delta);
            if (old == kDeltaEmpty) { atomicAdd(count, 1); return; }
            if (old == delta) return;
        }
    }
    atomicOr(count + 1, 1);         // table full: far more than 255 distinct offsets
}

// One thread per row: pack the row's W column offsets into ceil(W/8) code words.
template <int R>
__global__ void ell_encode(const int* __restrict__ cols, unsigned long long* __restrict__ codes, int64_t nslices, int W,
                           int64_t nloc, int64_t lead, const int* __restrict__ offsets, int ntable, int dcode) {
    __shared__ int s_off[256];
    for (int t = threadIdx.x; t < ntable; t += blockDim.x) s_off[t] = offsets[t];
    __syncthreads();
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= nslices * (WAVE * R)) return;
    const int64_t slice = row / (WAVE * R), within = row % (WAVE * R);
    const int CW = (W + 7) / 8;
    const size_t base = (size_t)slice * W * (WAVE * R) + within;
    const size_t cbase = (size_t)slice * CW * (WAVE * R) + within;
    for (int q = 0; q < CW; ++q) {
        unsigned long long word = 0;
        for (int kk = 0; kk < 8; ++kk) {
            const int k = 8 * q + kk;
            int code = dcode;
            if (k < W && row < nloc) {
                const int delta = cols[base + (size_t)k * (WAVE * R)] - (int)(lead + row);
                for (int c = 0; c < ntable; ++c)
                    if (s_off[c] == delta) { code = c; break; }
            }
            word |= (unsigned long long)code << (8 * kk);
        }
        codes[cbase + (size_t)q * (WAVE * R)] = word;
    }
}

// flag[0] |= 1 when a stored off-diagonal entry couples two rows of the same colour (the colouring is then not a
// Gauss-Seidel ordering of this matrix).  Entries below 1e-12 of the row's diagonal do not count: couplings that vanish
// analytically (P2 vertex-vertex pairs across a face or body diagonal) come out of an assembly as round-off of that size
// when h is not a power of two; a colour launch treats them Jacobi-style, like the oracle does.  One thread per row;
// offset-coded or int32 columns.
template <int R>
__global__ void ell_check_coloring(EllArgs a, int64_t nslices, int* flag) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    constexpr int S = WAVE * R;
    const int64_t slice = row / S, within = row % S;
    const int mine = lattice_color(a.color_kind, row + a.grow0, a.gnx, a.gny);
    bool bad = false;
    for (int k = 0; k < a.W; ++k) {
        const double v = a.vals[((size_t)slice * a.W + k) * S + within];
        if (fabs(v * a.dinv[row]) <= 1e-12) continue;
        int64_t off;
        if (a.codes) {
            const unsigned long long w = a.codes[((size_t)slice * ((a.W + 7) / 8) + k / 8) * S + within];
            off = a.offsets[(int)((w >> (8 * (k % 8))) & 0xffull)];
        } else {
            off = (int64_t)a.cols[((size_t)slice * a.W + k) * S + within] - (a.lead + row);
        }
        if (off != 0 && lattice_color(a.color_kind, row + off + a.grow0, a.gnx, a.gny) == mine) bad = true;
    }
    if (bad) atomicOr(flag, 1);
}

// (the same for symmetric diagonal storage: the stored upper entries name every coupled pair)
template <int R>
__global__ void sdia_check_coloring(EllArgs a, int wu, int* flag) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= a.nloc) return;
    constexpr int S = WAVE * R;
    const int64_t m = row + a.mlead;
    const size_t base = (size_t)(m / S) * wu * S + (size_t)(m % S);
    const double diag = a.vals[base];
    const int mine = lattice_color(a.color_kind, row + a.grow0, a.gnx, a.gny);
    bool bad = false;
    for (int c = 1; c < wu; ++c) {
        const double v = a.vals[base + (size_t)c * S];
        if (a.up[c] == 0 || fabs(v) <= 1e-12 * fabs(diag)) continue;
        if (lattice_color(a.color_kind, row + a.up[c] + a.grow0, a.gnx, a.gny) == mine) bad = true;
    }
    if (bad) atomicOr(flag, 1);
}

// One wave = one slice.  MODE_RESIDUAL: out = f - A x.  MODE_JACOBI: out = x + w D^-1 (f - A x)
// (jacobiRelaxation, multigrid.py:226, in the algebraically identical one-matrix form).
// MODE_SPMV: out = A x, with DOT also partial sums of x . (A x).
template <int WT, int R, int MODE, bool DOT>
__global__ __launch_bounds__(BLOCK) void ell_apply(EllArgs a) {
    if (a.done_flag && *a.done_flag) return;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned b = swizzle_block(blockIdx.x, gridDim.x, a.chunk);
    const int64_t sl = (int64_t)b * WAVES_PER_BLOCK + wave;
    double dot = 0.0;
    if (sl < a.nslices) {
        const int64_t slice = a.slice0 + sl + (sl >= a.split ? a.gap : 0);
        const int W = WT > 0 ? WT : a.W;
        const size_t base = (size_t)slice * (size_t)W * (WAVE * R) + (size_t)lane * R;
        const int64_t row = slice * (WAVE * R) + (int64_t)lane * R;
        double acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0;
        if (WT > 0) {
            DVec<R> v[WT > 0 ? WT : 1];
            IVec<R> c[WT > 0 ? WT : 1];
#pragma unroll
            for (int k = 0; k < WT; ++k) {
                c[k] = *reinterpret_cast<const IVec<R>*>(a.cols + base + (size_t)k * (WAVE * R));
                v[k] = *reinterpret_cast<const DVec<R>*>(a.vals + base + (size_t)k * (WAVE * R));
            }
#pragma unroll
            for (int k = 0; k < WT; ++k) {
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fma(v[k].d[r], a.x[c[k].d[r]], acc[r]);
            }
        } else {
            for (int k = 0; k < W; ++k) {
                const IVec<R> c = *reinterpret_cast<const IVec<R>*>(a.cols + base + (size_t)k * (WAVE * R));
                const DVec<R> v = *reinterpret_cast<const DVec<R>*>(a.vals + base + (size_t)k * (WAVE * R));
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = fma(v.d[r], a.x[c.d[r]], acc[r]);
            }
        }
        DVec<R> o;
        if (MODE != MODE_GS && row + R <= a.nloc) {
            if (MODE == MODE_SPMV) {
#pragma unroll
                for (int r = 0; r < R; ++r) o.d[r] = acc[r];
                if (DOT) {
                    const DVec<R> xr = *reinterpret_cast<const DVec<R>*>(a.x + a.lead + row);
#pragma unroll
                    for (int r = 0; r < R; ++r) dot += xr.d[r] * acc[r];
                }
            } else {
                const DVec<R> fr = *reinterpret_cast<const DVec<R>*>(a.f + row);
                if (MODE == MODE_RESIDUAL) {
#pragma unroll
                    for (int r = 0; r < R; ++r) o.d[r] = fr.d[r] - acc[r];
                } else {
                    const DVec<R> xr = *reinterpret_cast<const DVec<R>*>(a.x + a.lead + row);
                    const DVec<R> di = *reinterpret_cast<const DVec<R>*>(a.dinv + row);
#pragma unroll
                    for (int r = 0; r < R; ++r) o.d[r] = xr.d[r] + (a.omega * di.d[r]) * (fr.d[r] - acc[r]);
                }
            }
            *reinterpret_cast<DVec<R>*>(a.out + row) = o;
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t rr = row + r;
                if (MODE == MODE_GS && lattice_color(a.color_kind, rr + a.grow0, a.gnx, a.gny) != a.color) continue;
                if (rr < a.nloc) {
                    double val;
                    if (MODE == MODE_SPMV) {
                        val = acc[r];
                        if (DOT) dot += a.x[a.lead + rr] * acc[r];
                    } else if (MODE == MODE_RESIDUAL) {
                        val = a.f[rr] - acc[r];
                    } else {
                        val = a.x[a.lead + rr] + (a.omega * a.dinv[rr]) * (a.f[rr] - acc[r]);
                    }
                    a.out[rr] = val;
                }
            }
        }
    }
    if (DOT) {
        const double t = block_sum(dot);
        if (threadIdx.x == 0) a.partials[blockIdx.x] = t;
    }
}

// ---- CSR hand-off -> sliced ELL ----------------------------------------------------------
struct CsrArgs {
    const void* indptr;      // int32 or int64
    int indptr64;
    const int* indices;
    const double* data;
    const int* perm;         // dof -> global lexicographic node, or null
    int64_t n;               // global rows
    int64_t row0, nloc;      // owned global rows [row0, row0+nloc)
    int64_t lead, xlen;      // local vector = [lead | nloc | upper halo], xlen total
    int prune;
};

__device__ __forceinline__ int64_t csr_ptr(const CsrArgs& a, int64_t i) {
    return a.indptr64 ? reinterpret_cast<const int64_t*>(a.indptr)[i]
                      : (int64_t) reinterpret_cast<const int*>(a.indptr)[i];
}

// Pass 1: widest kept row among owned rows, number of kept entries, sanity flags.
// stats[0] = max width, stats[1] = kept entries, stats[2] = error bits
// (1: column outside the slab's reach, 2: zero/missing diagonal).
__global__ void csr_scan(CsrArgs a, unsigned long long* stats) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.n) return;
    const int64_t p = a.perm ? a.perm[d] : d;
    const int64_t lr = p - a.row0;
    if (lr < 0 || lr >= a.nloc) return;
    const int64_t b = csr_ptr(a, d), e = csr_ptr(a, d + 1);
    unsigned kept = 0, err = 0;
    bool diag = false;
    for (int64_t q = b; q < e; ++q) {
        const double v = a.data[q];
        const int64_t cp = a.perm ? a.perm[a.indices[q]] : a.indices[q];
        if (cp == p && v != 0.0) diag = true;
        if (a.prune && v == 0.0) continue;
        const int64_t lc = cp - a.row0 + a.lead;
        if (lc < 0 || lc >= a.xlen) err |= 1u;
        ++kept;
    }
    if (!diag) err |= 2u;
    atomicMax(&stats[0], (unsigned long long)kept);
    atomicAdd(&stats[1], (unsigned long long)kept);
    if (err) atomicOr(&stats[2], (unsigned long long)err);
}

// Pass 0: every slot = (0.0, own column); rows past nloc point at a valid element.
template <int R>
__global__ void ell_fill_padding(double* vals, int* cols, int64_t nslices, int W, int64_t nloc, int64_t lead) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nslices * W * (WAVE * R);
    if (t >= total) return;
    const int64_t per_slice = (int64_t)W * (WAVE * R);
    const int64_t slice = t / per_slice;
    const int64_t in = t % per_slice;
    const int64_t within = in % (WAVE * R);          // lane*R + r
    int64_t row = slice * (WAVE * R) + within;
    if (row >= nloc) row = nloc - 1;
    vals[t] = 0.0;
    cols[t] = (int)(lead + row);
}

// Pass 2: scatter kept entries in stored order; D^-1.
template <int R>
__global__ void csr_to_ell(CsrArgs a, double* vals, int* cols, double* dinv, int W) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= a.n) return;
    const int64_t p = a.perm ? a.perm[d] : d;
    const int64_t lr = p - a.row0;
    if (lr < 0 || lr >= a.nloc) return;
    const int64_t slice = lr / (WAVE * R);
    const int64_t within = lr % (WAVE * R);
    const size_t base = (size_t)slice * W * (WAVE * R) + within;
    const int64_t b = csr_ptr(a, d), e = csr_ptr(a, d + 1);
    int k = 0;
    double diag = 1.0;
    for (int64_t q = b; q < e; ++q) {
        const double v = a.data[q];
        const int64_t cp = a.perm ? a.perm[a.indices[q]] : a.indices[q];
        if (cp == p && v != 0.0) diag = v;
        if (a.prune && v == 0.0) continue;
        if (k < W) {
            vals[base + (size_t)k * (WAVE * R)] = v;
            cols[base + (size_t)k * (WAVE * R)] = (int)(cp - a.row0 + a.lead);
        }
        ++k;
    }
    dinv[lr] = 1.0 / diag;
}

// getJacobiMatrices (multigrid.py:48-56): D^-1 and the entries of D^-1 (A - D), in place of
// SciPy's diagonal(), CSR - DIA and DIA . CSR.  One row per thread (set-up, once per level).
__global__ void jacobi_split(CsrArgs a, double* dinv, double* scaled, unsigned char* keep) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const int64_t b = csr_ptr(a, i), e = csr_ptr(a, i + 1);
    double diag = 0.0;
    for (int64_t q = b; q < e; ++q)
        if (a.indices[q] == i) diag += a.data[q];          // duplicates sum, as A.diagonal() does
    const double di = 1.0 / diag;
    dinv[i] = di;
    for (int64_t q = b; q < e; ++q) {
        const double v = a.data[q];
        const bool on_diag = a.indices[q] == i;
        scaled[q] = di * v;
        keep[q] = (!on_diag && v != 0.0) ? 1 : 0;
    }
}

// ---- synthetic P1 Poisson level, written straight into tiles --------------------------------
// Same entries, in the same order, as poisson.lexicographic_level + csr_to_ell give.
struct GenArgs {
    Grid g;
    int N;              // elements per dimension
    int dim;
    int prune;
    int W;
    double h, w, diag, fh;   // h = 1/N, axis coupling magnitude, interior diagonal, f*h^dim
    int noff;
    int off[15][3];     // sorted pattern offsets (di, dj, dk) in unified (x, y, z) axes
    int odd;            // "gen_odd_rows": of 10000 interior rows, picked by a hash of their grid index, this many get a
                        // reaction term of their own on the diagonal (rows unlike any other: what the escape rows of the
                        // row dictionary are measured on)
};

// the diagonal of interior row (i, j, k): a.diag, or a.diag * (1 + r), 0 <= r < 1 from the row's hash, for the odd rows
__device__ __forceinline__ double gen_diag(const GenArgs& a, int i, int j, int k) {
    if (a.odd <= 0) return a.diag;
    unsigned long long x = ((unsigned long long)k * (unsigned long long)a.g.ny + (unsigned long long)j) * (unsigned long long)a.g.nx + (unsigned long long)i;
    x += 0x9e3779b97f4a7c15ull;
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31;
    if ((int)(x % 10000ull) >= a.odd) return a.diag;
    return a.diag * (1.0 + (double)((x >> 32) & 0xfffffull) / 1048576.0);
}

__device__ __forceinline__ double gen_g(const GenArgs& a, int i, int j, int k) {
    const double x = (double)i / (double)a.N;
    if (a.dim == 2) {
        const double y = (double)k / (double)a.N;      // 2-D grids use the z axis for y
        return 1.0 + x * x + 2.0 * (y * y);
    }
    const double y = (double)j / (double)a.N;
    const double z = (double)k / (double)a.N;
    return 1.0 + x * x + 2.0 * (y * y) + 3.0 * (z * z);
}

__device__ __forceinline__ bool gen_on_boundary(const GenArgs& a, int i, int j, int k) {
    bool bnd = (i == 0) || (i == a.g.nx - 1) || (k == 0) || (k == a.g.nz - 1);
    if (a.dim == 3) bnd = bnd || (j == 0) || (j == a.g.ny - 1);
    return bnd;
}

template <int R>
__global__ void gen_poisson(GenArgs a, double* vals, int* cols, double* dinv, double* f,
                            unsigned long long* counts) {
    int i = 0, j = 0;
    const bool active = plane_node(a.g, &i, &j);
    const int kl = blockIdx.y;              // local plane
    unsigned nz = 0, kept = 0;
    if (active) {
        const int k = a.g.k0 + kl;
        const int64_t lr = (int64_t)kl * a.g.plane + (int64_t)j * a.g.nx + i;
        const int64_t slice = lr / (WAVE * R);
        const int64_t within = lr % (WAVE * R);
        const size_t base = (size_t)slice * a.W * (WAVE * R) + within;
        const bool bnd = gen_on_boundary(a, i, j, k);
        double b = bnd ? gen_g(a, i, j, k) : a.fh;
        for (int t = 0; t < a.noff; ++t) {
            const int di = a.off[t][0], dj = a.off[t][1], dk = a.off[t][2];
            const int ii = i + di, jj = j + dj, kk = k + dk;
            if (ii < 0 || ii >= a.g.nx || jj < 0 || jj >= a.g.ny || kk < 0 || kk >= a.g.nz) continue;
            const int naxis = (di != 0) + (dj != 0) + (dk != 0);
            double v;
            if (naxis == 0) {
                v = bnd ? 1.0 : gen_diag(a, i, j, k);
            } else if (naxis == 1) {
                const bool nb = gen_on_boundary(a, ii, jj, kk);
                v = (!bnd && !nb) ? -a.w : 0.0;
                if (!bnd && nb) b = b + a.w * gen_g(a, ii, jj, kk);
            } else {
                v = 0.0;
            }
            if (v != 0.0) ++nz;
            if (a.prune && v == 0.0) continue;
            const int64_t lc = a.g.lead + lr + (int64_t)dk * a.g.plane + (int64_t)dj * a.g.nx + di;
            if ((int)kept < a.W) {
                vals[base + (size_t)kept * (WAVE * R)] = v;
                cols[base + (size_t)kept * (WAVE * R)] = (int)lc;
            }
            ++kept;
        }
        dinv[lr] = 1.0 / (bnd ? 1.0 : gen_diag(a, i, j, k));
        f[lr] = b;
    }
    // one atomic pair per wave; counts[0] = stored entries, counts[1] = non-zero entries
    const unsigned ksum = (unsigned)wave_sum((double)kept);
    const unsigned zsum = (unsigned)wave_sum((double)nz);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], (unsigned long long)ksum);
        atomicAdd(&counts[1], (unsigned long long)zsum);
    }
}

// ---- synthetic level from per-parity-class lattice stencils (P2 elements, BASELINE config 5) ---------------------
// On the structured simplicial mesh every interior lattice node of one parity class (i, j, k mod 2) has the same row:
// the caller hands over the eight interior stencils (offsets in ascending column order + values for this level's h)
// and the load of a constant source per class; the kernel applies the reference's boundary treatment
// (Multigrid_prototype.py:77-108: identity rows, zeroed columns, lifted right-hand side) exactly as gen_poisson does.
constexpr int LAT_MAX = 64;
struct LatticeArgs {
    Grid g;
    int N, dim, W;
    const int* count;           // [8]
    const int* off;             // [8][LAT_MAX][3]  (di, dj, dk), ascending column order
    const double* val;          // [8][LAT_MAX]
    const double* load;         // [8]
};

template <int R>
__global__ void gen_lattice(LatticeArgs a, double* vals, int* cols, double* dinv, double* f, unsigned long long* counts) {
    int i = 0, j = 0;
    const bool active = plane_node(a.g, &i, &j);
    const int kl = blockIdx.y;
    unsigned nz = 0, kept = 0;
    if (active) {
        const int k = a.g.k0 + kl;
        const int64_t lr = (int64_t)kl * a.g.plane + (int64_t)j * a.g.nx + i;
        const size_t base = (size_t)(lr / (WAVE * R)) * a.W * (WAVE * R) + (size_t)(lr % (WAVE * R));
        GenArgs ga{};                       // boundary data / boundary test shared with gen_poisson
        ga.g = a.g; ga.N = a.N; ga.dim = a.dim;
        const bool bnd = gen_on_boundary(ga, i, j, k);
        const int cls = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2);
        double b = bnd ? gen_g(ga, i, j, k) : a.load[cls];
        double diag = 1.0;
        if (bnd) {
            vals[base] = 1.0;
            cols[base] = (int)(a.g.lead + lr);
            kept = nz = 1;
        } else {
            const int n = a.count[cls];
            for (int t = 0; t < n; ++t) {
                const int* o = a.off + ((size_t)cls * LAT_MAX + t) * 3;
                const int ii = i + o[0], jj = j + o[1], kk = k + o[2];
                double v = a.val[(size_t)cls * LAT_MAX + t];
                if (gen_on_boundary(ga, ii, jj, kk)) {
                    b = b - v * gen_g(ga, ii, jj, kk);
                    continue;                               // zeroed column, pruned
                }
                if (o[0] == 0 && o[1] == 0 && o[2] == 0) diag = v;
                if ((int)kept < a.W) {
                    vals[base + (size_t)kept * (WAVE * R)] = v;
                    cols[base + (size_t)kept * (WAVE * R)] = (int)(a.g.lead + lr + (int64_t)o[2] * a.g.plane + (int64_t)o[1] * a.g.nx + o[0]);
                }
                ++kept; ++nz;
            }
        }
        dinv[lr] = 1.0 / diag;
        f[lr] = b;
    }
    const unsigned ksum = (unsigned)wave_sum((double)kept);
    const unsigned zsum = (unsigned)wave_sum((double)nz);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&counts[0], (unsigned long long)ksum);
        atomicAdd(&counts[1], (unsigned long long)zsum);
    }
}

// ---- grid transfers (lexicographic index arithmetic; no coordinate hashing) ---------------------
// Injection (Restriction2D_direct, multigrid.py:123-132): coarse (I,J,K) <- fine (2I,2J,2K).
__global__ void restrict_inject(Grid gc, Grid gf, const double* __restrict__ rf, double* __restrict__ fc) {
    int i, j;
    if (!plane_node(gc, &i, &j)) return;
    const int kl = blockIdx.y;
    const int kf = 2 * (gc.k0 + kl) - gf.k0;                 // local fine plane
    const int jf = gf.refine_y ? 2 * j : j;
    const int64_t src = gf.lead + (int64_t)kf * gf.plane + (int64_t)jf * gf.nx + 2 * i;
    fc[gc.lead + (int64_t)kl * gc.plane + (int64_t)j * gc.nx + i] = rf[src];
}

// Residual + injection in one pass (multigrid.py:244 followed by :251-252): with injection only the
// residual at the coarse nodes survives, so only those rows (every other row of every other line of
// every other plane) are evaluated -- a quarter of the matrix bytes of a full residual sweep.  Same
// per-row arithmetic as ell_apply<..., MODE_RESIDUAL>, hence bit-identical coarse right-hand sides.
struct FusedRestrictArgs {
    const double* vals; const int* cols; const unsigned long long* codes; const int* offsets;
    const double* x;        // fine iterate, base of storage
    const double* f;        // fine right-hand side, row-based
    double* fc;             // coarse right-hand side, base of storage
    int W, R, coded;        // coded: 0 int32 columns, 1 offset codes, 2 symmetric diagonals (W = WU), 3 the same
                            // through row classes (cls row-based, ctab[class][8]: the full row)
    int up[8];
    int64_t mlead;
    Grid gc, gf;
    const unsigned char* cls;
    const double* ctab;
    // coded == 4: offset-coded rows through stencil classes (cls row-based; s_off / s_val / s_cnt as ell_cls_apply)
    const int* s_off; const double* s_val; const int* s_cnt;
};

__global__ void residual_inject(FusedRestrictArgs a) {
    __shared__ int s_off[256];
    if (a.coded == 1) {
        for (int t = threadIdx.x; t < 256; t += blockDim.x) s_off[t] = a.offsets[t];
        __syncthreads();
    }
    int i, j;
    if (!plane_node(a.gc, &i, &j)) return;
    const int kl = blockIdx.y;
    const int kf = 2 * (a.gc.k0 + kl) - a.gf.k0;
    const int jf = a.gf.refine_y ? 2 * j : j;
    const int64_t row = (int64_t)kf * a.gf.plane + (int64_t)jf * a.gf.nx + 2 * i;       // local fine row
    const int64_t S = (int64_t)WAVE * a.R;
    const int64_t slice = row / S, within = row % S;
    const size_t base = (size_t)slice * a.W * S + within;
    double acc = 0.0;
    if (a.coded == 4) {
        const int c = a.cls[row];
        const int n = a.s_cnt[c];
        const double* xrow = a.x + a.gf.lead + row;
        for (int t = 0; t < n; ++t) acc = fma(a.s_val[(size_t)c * a.W + t], xrow[a.s_off[(size_t)c * a.W + t]], acc);
    } else if (a.coded == 3) {
        const unsigned char* crow = a.cls + row;
        const double* xrow = a.x + a.gf.lead + row;
        const double* t = a.ctab + 8 * crow[0];             // the whole row: a(-P) a(-nx) a(-1) a(0) a(+1) a(+nx) a(+P)
        for (int c = a.W - 1; c >= 1; --c) acc = fma(t[3 - c], xrow[-a.up[c]], acc);
        acc = fma(t[3], xrow[0], acc);
        for (int c = 1; c < a.W; ++c) acc = fma(t[3 + c], xrow[a.up[c]], acc);
    } else if (a.coded == 2) {
        const int64_t m = row + a.mlead;
        const double* xrow = a.x + a.gf.lead + row;
        for (int c = a.W - 1; c >= 1; --c) {
            const int64_t mm = m - a.up[c];
            acc = fma(a.vals[((size_t)(mm / S) * a.W + c) * S + (size_t)(mm % S)], xrow[-a.up[c]], acc);
        }
        const size_t mb = (size_t)(m / S) * a.W * S + (size_t)(m % S);
        acc = fma(a.vals[mb], xrow[0], acc);
        for (int c = 1; c < a.W; ++c) acc = fma(a.vals[mb + (size_t)c * S], xrow[a.up[c]], acc);
    } else if (a.coded) {
        const int CW = (a.W + 7) / 8;
        const size_t cbase = (size_t)slice * CW * S + within;
        const double* xrow = a.x + a.gf.lead + row;
        for (int q = 0; q < CW; ++q) {
            const unsigned long long w = a.codes[cbase + (size_t)q * S];
            const int kend = min(8, a.W - 8 * q);
            for (int kk = 0; kk < kend; ++kk) {
                const int code = (int)((w >> (8 * kk)) & 0xffull);
                acc = fma(a.vals[base + (size_t)(8 * q + kk) * S], xrow[s_off[code]], acc);
            }
        }
    } else {
        for (int k = 0; k < a.W; ++k)
            acc = fma(a.vals[base + (size_t)k * S], a.x[a.cols[base + (size_t)k * S]], acc);
    }
    a.fc[a.gc.lead + (int64_t)kl * a.gc.plane + (int64_t)j * a.gc.nx + i] = a.f[row] - acc;
}

// Full weighting (Restriction2D, multigrid.py:135-198); neighbours outside the grid are skipped
// (:172-194).  2-D: (1/16)(corners + 2 edges + 4 centre) in the reference's summation order;
// 3-D (no reference): (1/64)(corners + 2 edges + 4 faces + 8 centre).  Needs valid fine halos.
__global__ void restrict_full_weighting(Grid gc, Grid gf, const double* __restrict__ rf, double* __restrict__ fc) {
    int i, j;
    if (!plane_node(gc, &i, &j)) return;
    const int kl = blockIdx.y;
    const int K = gc.k0 + kl;
    const int fi = 2 * i, fk = 2 * K;
    auto at = [&](int di, int dj, int dk, double& acc) {
        const int ii = fi + di, kk = fk + dk;
        const int jj = (gf.refine_y ? 2 * j : j) + dj;
        if (ii < 0 || ii >= gf.nx || jj < 0 || jj >= gf.ny || kk < 0 || kk >= gf.nz) return;
        acc = acc + rf[gf.lead + (int64_t)(kk - gf.k0) * gf.plane + (int64_t)jj * gf.nx + ii];
    };
    double out;
    if (!gf.refine_y) {
        // reference tuples are (x, y) = (i, k here)
        double s1 = 0.0, s2 = 0.0, c = 0.0;
        at(-1, 0, -1, s1); at(-1, 0, 1, s1); at(1, 0, -1, s1); at(1, 0, 1, s1);
        at(0, 0, -1, s2); at(0, 0, 1, s2); at(-1, 0, 0, s2); at(1, 0, 0, s2);
        at(0, 0, 0, c);
        out = (1.0 / 16.0) * (s1 + 2.0 * s2 + 4.0 * c);
    } else {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int dk = -1; dk <= 1; ++dk)
            for (int dj = -1; dj <= 1; ++dj)
                for (int di = -1; di <= 1; ++di) {
                    const int m = (di != 0) + (dj != 0) + (dk != 0);
                    at(di, dj, dk, s[m]);
                }
        out = (1.0 / 64.0) * (s[3] + 2.0 * s[2] + 4.0 * s[1] + 8.0 * s[0]);
    }
    fc[gc.lead + (int64_t)kl * gc.plane + (int64_t)j * gc.nx + i] = out;
}

// Q1 prolongation + correction (Interpolation2D, multigrid.py:59-120, and `v_h + err_h`, :260):
// coincident nodes copy, edge nodes 0.5*(a+b), face/cell centres 0.25*(..)/0.125*(..), summed
// x-neighbour first as the reference does.  err (optional) receives the interpolated values.
// One thread handles the fine nodes (2p, 2p+1) of a line: they share their coarse neighbours, and the
// parity along x -- the only one that differs between adjacent lanes -- no longer diverges.
template <bool ADD, bool KEEP>
__global__ void prolong_correct(Grid gc, Grid gf, const double* __restrict__ vc, double* __restrict__ vf,
                                double* __restrict__ err) {
    const unsigned ppl = (unsigned)(gf.nx + 1) / 2u;                 // pairs per line
    const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ppl * (unsigned)gf.ny) return;
    const int j = (int)(t / ppl);
    const int p = (int)(t - (unsigned)j * ppl);
    const int kl = blockIdx.y;
    const int k = gf.k0 + kl;
    const int pk = k & 1;
    const int pj = gf.refine_y ? (j & 1) : 0;
    const int kc = (k >> 1) - gc.k0;
    const int jc = gf.refine_y ? (j >> 1) : j;
    const bool has_odd = 2 * p + 1 < gf.nx;                          // the last pair of an odd line is a single node
    const double* base = vc + gc.lead + (int64_t)kc * gc.plane + (int64_t)jc * gc.nx + p;
    double s0 = 0.0, s1 = 0.0;
    bool first = true;
    for (int dk = 0; dk <= pk; ++dk)
        for (int dj = 0; dj <= pj; ++dj) {
            const double* q = base + (int64_t)dk * gc.plane + (int64_t)dj * gc.nx;
            const double a = q[0];
            const double b = has_odd ? q[1] : 0.0;
            s0 = first ? a : s0 + a;                                 // even node: (dk, dj) corners
            s1 = first ? a : s1 + a;                                 // odd node: x-neighbour first, as the reference
            s1 = s1 + b;
            first = false;
        }
    const int c0 = 1 << (pj + pk);
    const double e0 = c0 == 1 ? s0 : (1.0 / (double)c0) * s0;
    const double e1 = (1.0 / (double)(2 * c0)) * s1;
    const int64_t o = gf.lead + (int64_t)kl * gf.plane + (int64_t)j * gf.nx + 2 * p;
    if (KEEP) { err[o] = e0; if (has_odd) err[o + 1] = e1; }
    if (ADD) { vf[o] = vf[o] + e0; if (has_odd) vf[o + 1] = vf[o + 1] + e1; }
}

// Prolongation from a table (mg_set_prolongation_table): a fine lattice point combines the coarse lattice points
// 2 * floor((i, j, k) / 4) + offset[residue][t] with weight[residue][t], residue = (i mod 4) + 4 (j mod 4) + 16 (k mod 4)
// -- the natural embedding of the coarse P2 space on nested simplicial meshes (poisson.p2_prolongation_table; no
// reference counterpart: Interpolation2D, multigrid.py:59-120, is the bilinear table).  Summed in table order,
// multiply then add (no fma), as the oracle does.  Needs the coarse halo planes the offsets reach (up to two).
struct ProlongTable {
    const int* count;       // [64]
    const int* off;         // [64][10][3]
    const double* w;        // [64][10]
};

template <bool ADD, bool KEEP>
__global__ void prolong_table(Grid gc, Grid gf, ProlongTable t, const double* __restrict__ vc, double* __restrict__ vf,
                              double* __restrict__ err) {
    // the table in LDS, offsets already as element distances on the coarse grid (lanes of a wave differ in residue:
    // from global memory every entry cost four more loads per lane)
    __shared__ int s_cnt[64];
    __shared__ int64_t s_lin[640];
    __shared__ double s_w[640];
    for (int e = threadIdx.x; e < 640; e += blockDim.x) {
        const int* o = t.off + (size_t)e * 3;
        s_lin[e] = (int64_t)o[2] * gc.plane + (int64_t)o[1] * gc.nx + o[0];
        s_w[e] = t.w[e];
        if (e < 64) s_cnt[e] = t.count[e];
    }
    __syncthreads();
    int i, j;
    if (!plane_node(gf, &i, &j)) return;
    const int kl = blockIdx.y;
    const int k = gf.k0 + kl;
    const int res = (i & 3) | (gf.refine_y ? (j & 3) << 2 : 0) | (k & 3) << 4;
    const int bi = 2 * (i >> 2), bj = gf.refine_y ? 2 * (j >> 2) : j, bk = 2 * (k >> 2);
    const int n = s_cnt[res];
    const double* const base = vc + gc.lead + (int64_t)(bk - gc.k0) * gc.plane + (int64_t)bj * gc.nx + bi;
    // all (up to ten) coarse values first, then the sum in table order (a loop of n dependent iterations waits for every
    // load by itself: n memory latencies per fine point)
    double v[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) v[e] = e < n ? base[s_lin[res * 10 + e]] : 0.0;
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < 10; ++e) {
        if (e < n) {
            const double term = s_w[res * 10 + e] * v[e];
            s = e == 0 ? term : s + term;
        }
    }
    const int64_t o = gf.lead + (int64_t)kl * gf.plane + (int64_t)j * gf.nx + i;
    if (KEEP) err[o] = s;
    if (ADD) vf[o] = vf[o] + s;
}

// Restriction from a table (mg_set_restriction_table): the transpose of the table prolongation, gathered per coarse
// lattice point: an interior coarse point of type (a & 1) + 2 (b & 1) + 4 (c & 1) sums weight[type][t] * r[2 A + offset[type][t]]
// over the INTERIOR fine points (the lifted system decouples the Dirichlet rows: a boundary coarse point takes the
// coincident fine value, as injection does).  Summed in table order, multiply then add.  Whole levels only (reach +-3).
struct RestrictTable {
    const int* count;       // [8]
    const int* off;         // [8][M][3]
    const double* w;        // [8][M]
    int M;
};

__global__ void restrict_table(Grid gc, Grid gf, RestrictTable t, const double* __restrict__ rf, double* __restrict__ fc) {
    // the table in LDS (up to RT_MAX entries per type; longer tables are read from global memory as before)
    constexpr int RT_MAX = 128;
    __shared__ int s_cnt[8];
    __shared__ int s_off[8 * RT_MAX];          // (o0 + 8) | (o1 + 8) << 8 | (o2 + 8) << 16
    __shared__ double s_w[8 * RT_MAX];
    const bool lds = t.M <= RT_MAX;
    if (lds) {
        for (int e = threadIdx.x; e < 8 * t.M; e += blockDim.x) {
            const int typ = e / t.M, k = e - typ * t.M;
            const int* o = t.off + (size_t)e * 3;
            s_off[typ * RT_MAX + k] = (o[0] + 8) | (o[1] + 8) << 8 | (o[2] + 8) << 16;
            s_w[typ * RT_MAX + k] = t.w[e];
        }
        if (threadIdx.x < 8) s_cnt[threadIdx.x] = t.count[threadIdx.x];
    }
    __syncthreads();
    int i, j;
    if (!plane_node(gc, &i, &j)) return;
    const int kl = blockIdx.y;
    const int K = gc.k0 + kl;
    const int fi = 2 * i, fj = gf.refine_y ? 2 * j : j, fk = 2 * K;
    const bool bnd = i == 0 || i == gc.nx - 1 || K == 0 || K == gc.nz - 1 || (gf.refine_y && (j == 0 || j == gc.ny - 1));
    double s = 0.0;
    if (bnd) {
        s = rf[gf.lead + (int64_t)(fk - gf.k0) * gf.plane + (int64_t)fj * gf.nx + fi];
    } else {
        const int typ = (i & 1) | (gf.refine_y ? (j & 1) << 1 : 0) | (K & 1) << 2;
        const int n = lds ? s_cnt[typ] : t.count[typ];
        bool first = true;
        // eight entries at a time: their fine values first (entries that fall on the boundary or past the table's end read
        // nothing), then the sum in table order -- one entry per iteration waited for every load by itself
        constexpr int NB = 8;
        for (int e0 = 0; e0 < n; e0 += NB) {
            double w[NB], v[NB];
            bool use[NB];
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const int e = e0 + u;
                int o0 = 0, o1 = 0, o2 = 0;
                w[u] = 0.0;
                if (e < n) {
                    if (lds) {
                        const int pk = s_off[typ * RT_MAX + e];
                        o0 = (pk & 255) - 8; o1 = (pk >> 8 & 255) - 8; o2 = (pk >> 16 & 255) - 8;
                        w[u] = s_w[typ * RT_MAX + e];
                    } else {
                        const int* o = t.off + ((size_t)typ * t.M + e) * 3;
                        o0 = o[0]; o1 = o[1]; o2 = o[2];
                        w[u] = t.w[(size_t)typ * t.M + e];
                    }
                }
                const int ii = fi + o0, jj = fj + o1, kk = fk + o2;
                use[u] = e < n && !(ii <= 0 || ii >= gf.nx - 1 || kk <= 0 || kk >= gf.nz - 1) &&
                         !(gf.refine_y && (jj <= 0 || jj >= gf.ny - 1));
                v[u] = use[u] ? rf[gf.lead + (int64_t)(kk - gf.k0) * gf.plane + (int64_t)jj * gf.nx + ii] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                if (use[u]) {
                    const double term = w[u] * v[u];
                    s = first ? term : s + term;
                    first = false;
                }
            }
        }
    }
    fc[gc.lead + (int64_t)kl * gc.plane + (int64_t)j * gc.nx + i] = s;
}

// ---- vector utilities ---------------------------------------------------------------------------
__global__ void fill_zero(double* x, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        x[t] = 0.0;
}

// caller numbering -> local lexicographic storage (owned rows and halo planes)
__global__ void scatter_in(const double* __restrict__ in, const int* __restrict__ perm, int64_t n,
                           int64_t row0, int64_t lead, int64_t xlen, double* __restrict__ x) {
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = perm ? perm[d] : d;
        const int64_t l = p - row0 + lead;
        if (l >= 0 && l < xlen) x[l] = in[d];
    }
}

// local lexicographic storage (owned rows) -> caller numbering
__global__ void gather_out(const double* __restrict__ x, const int* __restrict__ perm, int64_t n,
                           int64_t row0, int64_t nloc, int64_t lead, double* __restrict__ out) {
    for (int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; d < n; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = perm ? perm[d] : d;
        const int64_t l = p - row0;
        if (l >= 0 && l < nloc) out[d] = x[lead + l];
    }
}

// jacobiRelaxation in the reference's own split form (multigrid.py:226):
//     out = (1 - w) v + w (D^-1 f) - w q        with q = (D^-1 R) v from ell_apply<..., MODE_SPMV>,
// evaluated left to right like the NumPy expression.  q and out may alias.
__global__ void jacobi_split_combine(const double* __restrict__ v, const double* __restrict__ f,
                                     const double* __restrict__ dinv, const double* q, double* out, int64_t n,
                                     double omega) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = ((1.0 - omega) * v[t] + omega * (dinv[t] * f[t])) - omega * q[t];
}

// partial sums of x.y over n entries -> partials[blockIdx]; deterministic (no atomics)
__global__ __launch_bounds__(BLOCK) void dot_partial(const double* __restrict__ x, const double* __restrict__ y,
                                                      int64_t n, double* __restrict__ partials) {
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // two doubles per lane per step where alignment allows
    for (; t * 2 + 1 < n; t += stride) {
        const double2 a = reinterpret_cast<const double2*>(x)[t];
        const double2 b = reinterpret_cast<const double2*>(y)[t];
        s = fma(a.x, b.x, s);
        s = fma(a.y, b.y, s);
    }
    if (t * 2 < n && t * 2 + 1 >= n) s = fma(x[t * 2], y[t * 2], s);
    const double r = block_sum(s);
    if (threadIdx.x == 0) partials[blockIdx.x] = r;
}

// out[0] = sum(partials[0..np))
__global__ __launch_bounds__(BLOCK) void reduce_partials(const double* __restrict__ partials, int np, double* out) {
    double s = 0.0;
    for (int t = threadIdx.x; t < np; t += blockDim.x) s += partials[t];
    const double r = block_sum(s);
    if (threadIdx.x == 0) out[0] = r;
}

// ---- coarsest-level solver: Jacobi-preconditioned CG, scalars kept on the device -------------------
// Scalars: sc[0], sc[1] = r.z of even / odd iterations (double-buffered so that the kernel which
// computes the new value can still hand the old one to all of its blocks), sc[2]=rr, sc[3]=bb,
// sc[5]=iterations done.
struct PcgArgs {
    double* x; double* r; double* z; double* p; double* q;   // row-based (p is stored with halo lead)
    const double* b; const double* dinv;
    double* part_a; double* part_b; int nparts;               // per-block partial sums
    double* sc; int* done; int64_t n; double rtol2;
};

__device__ __forceinline__ double sum_partials(const double* part, int np) {
    __shared__ double s_tot;
    double s = 0.0;
    for (int t = threadIdx.x; t < np; t += blockDim.x) s += part[t];
    const double r = block_sum(s);
    if (threadIdx.x == 0) s_tot = r;
    __syncthreads();
    const double out = s_tot;
    __syncthreads();
    return out;
}

// x = 0, r = b, z = D^-1 r, p = z; partial r.z -> part_a, b.b -> part_b
__global__ __launch_bounds__(BLOCK) void pcg_init(PcgArgs a) {
    double rz = 0.0, bb = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.n; t += (int64_t)gridDim.x * blockDim.x) {
        const double b = a.b[t];
        const double z = a.dinv[t] * b;
        a.x[t] = 0.0; a.r[t] = b; a.z[t] = z; a.p[t] = z;
        rz = fma(b, z, rz); bb = fma(b, b, bb);
    }
    const double s1 = block_sum(rz);
    const double s2 = block_sum(bb);
    if (threadIdx.x == 0) { a.part_a[blockIdx.x] = s1; a.part_b[blockIdx.x] = s2; }
}

__global__ __launch_bounds__(BLOCK) void pcg_init_finish(PcgArgs a) {
    const double rz = sum_partials(a.part_a, a.nparts);
    const double bb = sum_partials(a.part_b, a.nparts);
    if (threadIdx.x == 0) {
        a.sc[0] = rz; a.sc[1] = rz; a.sc[3] = bb; a.sc[2] = bb; a.sc[5] = 0.0;
        *a.done = (bb == 0.0) ? 1 : 0;
    }
}

// after q = A p (partials of p.q in part_a, np_spmv of them), iteration `it`:
// alpha = rz/pq; x += alpha p; r -= alpha q; z = D^-1 r; partials r.z -> part_b[0..), r.r -> part_b[nparts..)
__global__ __launch_bounds__(BLOCK) void pcg_update(PcgArgs a, int np_spmv, int it) {
    if (*a.done) return;
    const double pq = sum_partials(a.part_a, np_spmv);
    const double alpha = a.sc[it & 1] / pq;
    double rz = 0.0, rr = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.n; t += (int64_t)gridDim.x * blockDim.x) {
        a.x[t] = fma(alpha, a.p[t], a.x[t]);
        const double r = fma(-alpha, a.q[t], a.r[t]);
        const double z = a.dinv[t] * r;
        a.r[t] = r; a.z[t] = z;
        rz = fma(r, z, rz); rr = fma(r, r, rr);
    }
    const double s1 = block_sum(rz);
    const double s2 = block_sum(rr);
    if (threadIdx.x == 0) { a.part_b[blockIdx.x] = s1; a.part_b[a.nparts + blockIdx.x] = s2; }
}

// beta = rz_new/rz_old; p = z + beta p.  Every block reduces the partials itself; block 0 also publishes
// rz_new into the OTHER scalar slot (nobody reads that slot in this launch), the iteration count and the
// convergence flag.  A block that starts after the flag was raised skips its part of p, which is
// harmless: x is final once the flag is up.
__global__ __launch_bounds__(BLOCK) void pcg_direction(PcgArgs a, int it) {
    if (*a.done) return;
    const double rz_new = sum_partials(a.part_b, a.nparts);
    const double beta = rz_new / a.sc[it & 1];
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < a.n; t += (int64_t)gridDim.x * blockDim.x)
        a.p[t] = fma(beta, a.p[t], a.z[t]);
    if (blockIdx.x == 0) {
        const double rr = sum_partials(a.part_b + a.nparts, a.nparts);
        if (threadIdx.x == 0) {
            a.sc[(it + 1) & 1] = rz_new; a.sc[2] = rr; a.sc[5] += 1.0;
            if (rr <= a.rtol2 * a.sc[3]) *a.done = 1;
        }
    }
}

}  // namespace mgk
