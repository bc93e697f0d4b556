// C ABI of libmg_hip.so (see include/mg_hip.h): hierarchy handle, level loop of the V-cycle
// kept on the device, RCCL slab exchange.  Host code only orchestrates launches on the
// handle's stream; all arithmetic of the path runs in the kernels of mg_kernels.hip.h.
#include "../../include/mg_hip.h"
#include "mg_kernels.hip.h"
#include "mg_direct.hip.h"
#include "mg_jacobi2.hip.h"
#include "mg_jacobik3d.hip.h"
#include "mg_jacobiblk.hip.h"
#include "mg_lattice.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace mgk;

namespace {

thread_local std::string g_err;

int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + \
                        ":" + std::to_string(__LINE__) + ")");                                \
    } while (0)

#define MG_TRY(expr)            \
    do {                        \
        int r_ = (expr);        \
        if (r_ != 0) return r_; \
    } while (0)

// A device vector in local lexicographic storage: [pad | lower halo | owned rows | upper halo],
// padded so that the first owned row is 32-byte aligned (the tile kernels load R rows per lane).
struct DVector {
    double* raw = nullptr;
    double* base = nullptr;     // start of [lower halo | owned | upper halo]
    double* rows = nullptr;     // first owned row = base + lead
};

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    // MG_RCCL_LIBRARY: load another implementation of the same ten entry points (tests use an in-process
    // stand-in to exercise the asynchronous stream / event ordering of the slab transport on one GPU)
    void* lib = nullptr;
    if (const char* override_path = getenv("MG_RCCL_LIBRARY")) {
        lib = dlopen(override_path, RTLD_NOW | RTLD_LOCAL);
        if (!lib) return fail(std::string("cannot load MG_RCCL_LIBRARY: ") + dlerror());
    }
    if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return fail(std::string("cannot load librccl: ") + dlerror());
#define SYM(field, name)                                                      \
    *reinterpret_cast<void**>(&g_rccl.field) = dlsym(lib, name);              \
    if (!g_rccl.field) return fail(std::string("librccl lacks ") + name)
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(AllReduce, "ncclAllReduce");
    SYM(Broadcast, "ncclBroadcast");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.lib = lib;
    return 0;
}

#define NCCL_TRY(expr)                                                                   \
    do {                                                                                 \
        ncclResult_t r_ = (expr);                                                        \
        if (r_ != ncclSuccess)                                                           \
            return fail(std::string(#expr) + ": " + g_rccl.GetErrorString(r_));          \
    } while (0)

struct Comm {
    int rank = 0, world = 1;
    int64_t replicate_below = 0;
    ncclComm_t nccl = nullptr;
    mg_exchange_fn ex = nullptr;
    mg_allreduce_fn ar = nullptr;
    mg_allgatherv_fn ag = nullptr;
    void* user = nullptr;
    // pinned host staging for the callback transport
    double* h_stage = nullptr;
    size_t h_stage_elems = 0;
    bool active() const { return world > 1; }
};

struct Level {
    bool set = false;
    bool has_matrix = false;
    int N = 0;
    Grid g{};
    int64_t n_global = 0, row0 = 0, nloc = 0, xlen = 0, halo_lo = 0, halo_hi = 0;
    bool replicated = true;      // false: this rank holds one slab and exchanges halos
    int hd = 0;                  // halo planes each vector of this level has room for (slabs)
    // how the matrix got its storage (mg_level_storage): bit-for-bit symmetric / within 2^sym_qbits ulps / not; first
    // asymmetric row (local lexicographic, -1: none) and the largest ulp distance of a pair; distinct rows seen by the class
    // dictionary (-1: not built) and the tolerance it was built with
    int rep_sym = -1, rep_sym_qbits = 0, rep_cls_qbits = 0, rep_distinct = -1;
    int64_t rep_first_asym = -1, rep_max_ulps = 0;
    bool cls_escape = false;     // rows of class CLS_ESCAPE exist: read from the stored row (K-sweep march only; other class kernels off)
    bool grid_decoupled = false; // no entry couples cells that are no grid neighbours (sdia_grid_decoupled; levels with row classes)
    int esc_kmax = 0;            // ... and the most sweeps per pass whose tiles' escape rows fit the march's pool (jk3_escape_window)
    int64_t rep_escape = 0;      // ... how many
    int cls_halo = 0;            // row classes of the neighbours' planes next to this slab: 0 not built yet, 1 in place, -1 unavailable
    bool flat = false;           // no grid structure (stand-alone smoother on any matrix)
    int W = 0, R = 1;
    int64_t nslices = 0;
    double* vals = nullptr;
    int* cols = nullptr;                    // freed once the offset codes exist
    unsigned long long* codes = nullptr;    // offset-coded columns (see mg_kernels.hip.h)
    int* offsets = nullptr;
    int ntable = 0, dcode = 0;
    bool coded = false;
    bool rb_ok = false;                     // index parity is a valid red-black colouring of the matrix
    int mc_ok = -1;                         // the nine lattice colours are a valid colouring of the matrix (-1: not checked yet)
    // symmetric diagonal storage (replaces vals + codes when the matrix is bit-for-bit symmetric)
    double* dvals = nullptr;
    bool sdia = false;
    int wu = 0;
    int up[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t mlead = 0, mslices = 0;
    // row classes of the symmetric diagonal storage (mg_jacobi2.hip.h): one byte per row + a 256 x 8 table of full rows
    unsigned char* cls = nullptr;
    double* ctab = nullptr;
    int ncls = 0;
    // stencil classes of an offset-coded level (mg_kernels.hip.h, ell_cls_apply): one byte per row + (offset, value) lists
    unsigned char* scls = nullptr;
    int* s_off = nullptr;
    double* s_val = nullptr;
    int* s_cnt = nullptr;
    int nscls = 0;
    // ... and what the plane march of wide lattice stencils needs (mg_lattice.hip.h): entries as (di, dj, dk), the
    // most frequent classes
    int* s_pack = nullptr;
    int lm_ntop = 0;
    int lm_top[LM_K] = {0, 0, 0, 0, 0, 0, 0, 0};
    int cmain = 0;                          // the most frequent class and its entries (passed to the kernels by value)
    double cm[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t cls_lead = 0, cls_rows = 0;     // cls[row + cls_lead], zero padding of cls_lead entries on both sides
    double* dinv = nullptr;
    DVector v, v2, f, err, ftrue;
    DVector sw;                             // once-relaxed boundary planes of a slab (paired sweeps, world > 1)
    int* perm = nullptr;
    unsigned long long nnz_stored = 0, nnz_nonzero = 0;
    // per-rank plane ownership (for gathers): k-plane boundaries s[0..world]
    std::vector<int> splits;
};

}  // namespace

// Block-tridiagonal LU of the coarsest level (mg_direct.hip.h)
struct DirectSolver {
    bool tried = false, ok = false;
    BtGeom g{};
    double* T = nullptr;                    // nb dense p x p inverses of the Schur complements
    int *lcol = nullptr, *ucol = nullptr;   // couplings, nb * p * W each
    double *lval = nullptr, *uval = nullptr;
    double *y = nullptr, *x = nullptr, *w = nullptr;
};

// One captured V-cycle (hipGraph): valid for a level, a parameter epoch and a ping-pong state.
struct CycleGraph {
    int level = -1;
    uint64_t epoch = 0;
    std::vector<double*> pre, post;     // raw pointer of MG_VEC_V on levels 0..level before / after the cycle
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

struct mg_context {
    int dim = 2, nlev = 0, device = 0;
    hipStream_t stream = nullptr;
    hipStream_t comm_stream = nullptr;      // halo exchange overlapped with interior sweeps (world > 1)
    hipEvent_t ev_boundary = nullptr, ev_halo = nullptr;
    int overlap = 1;
    int halo_planes = 1;                    // grid planes exchanged with each slab neighbour (2: stencils that reach two planes, P2)
    int halo_depth = 0;                     // halo planes ALLOCATED per vector (>= halo_planes; 0: just those): the K-sweep march on slabs
                                            // exchanges K planes once per K sweeps (mg_jacobik3d.hip.h)
    int64_t overlap_min_rows = (int64_t)1 << 22;
    std::vector<Level> L;
    int mu1 = 50, mu2 = 50;
    double omega = 2.0 / 3.0;
    int restriction = MG_RESTRICT_INJECTION, smoother = MG_SMOOTH_JACOBI;
    double coarse_rtol = 1e-14;
    int coarse_maxit = 20000;
    int keep_err = 0;
    // tuning
    int use_codes = 1;          // offset-coded columns where a level allows it
    int use_sdia = 1;           // symmetric diagonal storage where a level is bit-for-bit symmetric
    int strip_slices = 64;      // XCD strip traversal for 3-D levels (0 = chunked map only)
    int nontemporal = 1;        // streaming loads for matrix / rhs data
    // Dynamic LDS bytes requested per block by the symmetric-diagonal launches on large levels.  The kernels do not
    // use it; it caps the resident blocks per CU at 160 KiB / pad = 5 (20 waves instead of the 28 the register
    // budget allows), which measured 2.4 % faster on the 1025^3 sweep in an interleaved A/B (tools/ab_lds_pad.py:
    // 11.64 vs 11.93 ms; 3 blocks per CU: 14.2 ms) -- fewer concurrent streams per HBM page.
    int lds_pad = 32768;
    int rows_per_lane = 2;      // measured best on MI355X (profiles/): 16 B value loads per lane
    unsigned chunk = 8;         // XCD chunk of the block -> tile map
    int pcg_chunk = 16;
    // scratch
    double* partials = nullptr;     // 2 * kMaxParts doubles
    double* scalars = nullptr;      // 8 doubles
    int* done = nullptr;
    double* h_scalars = nullptr;    // pinned, 8 doubles + flag
    double *pcg_r = nullptr, *pcg_z = nullptr, *pcg_q = nullptr, *pcg_part_a = nullptr, *pcg_part_b = nullptr;
    DVector pcg_p;
    int pcg_parts = 0, pcg_parts_a = 0;
    int pcg_predict = 0;
    int use_graph = 1;              // replay whole V-cycles as hipGraphs (single GPU, direct coarsest solve)
    int comm_priority = 1;          // communication stream created with the highest priority (MG_COMM_PRIORITY=0: lowest)
    int lattice_march = 1;          // wide lattice stencils (P2 levels) as a plane march with x in LDS (mg_lattice.hip.h)
    int64_t lattice_march_min_rows = 1 << 22;       // (measured on C5: the 129^3 and 65^3 lattices are faster with the gathering kernel)
    int lattice_gs2 = 0;            // nine-colour Gauss-Seidel on whole lattice levels: two colours per launch, out of place (lat_gs2;
                                    // measured slower than nine colour launches: 9.1 against 6.8 ms per sweep of the 513^3 lattice)
    int lattice_tile = 0;           // tile of that march: 0 / 1 = 64 x 16 cells (256 threads), 2 = 128 x 16 (512 threads)
    int lattice_segments = 0;       // plane segments per tile of that march, 0 = chosen from the tile count
    int slab_pair_form = 0;         // overlapped pair sweeps on slabs: 0 = chain beside one launch of the pass, 1 = boundary segments first
    int graph_comm = 0;             // ... on slabs too: the RCCL exchanges are captured with the kernels (opt-in)
    uint64_t epoch = 1;             // bumped by every call that changes what a V-cycle launches
    std::vector<CycleGraph> graphs;
    int use_direct = 1;             // exact block-tridiagonal coarsest solve where the level allows it
    int require_diagonal = 1;       // 0: operators without a diagonal (D^-1 R of the split smoother)
    int fuse_restrict = 1;          // residual evaluated at the coarse nodes only when restricting by injection
    // K sweeps per launch on blocks resident on the CU (mg_jacobiblk.hip.h): whole 3-D levels with row classes of
    // fuse_block_min_rows <= rows < fuse_block_max_rows (below the plane marches' sizes); 2: whatever the size (tests).
    // Three sweeps per launch on blocks of 11 planes: alone no faster per sweep than one launch per sweep (129^3: 13.2 us
    // against 14.3; 65^3: 3.9 against 4.0; profiles/r03_block_pass.txt), but a third of the launches inside a cycle, where a
    // one-sweep launch costs 14.8 / 5.8 us: BASELINE config 3 132.8 -> 137.5 cycles/s (profiles/r03_block_cycles.txt).  Blocks
    // of 19 planes spill and are slower (124.4).
    int fuse_block = 1;
    int64_t fuse_block_min_rows = (int64_t)1 << 15, fuse_block_max_rows = (int64_t)1 << 23;
    int fuse_block_k = 0, fuse_block_ez = 11;   // sweeps per launch (0: three, four on levels whose blocks then all run at once) / planes per block (11 or 19)
    int direct_block_rows = 2048;   // "direct_block_rows": rows per block of the coarsest level's block-tridiagonal LU, at least
    int fuse_2d_lines = 0;          // "fuse_2d_lines": lines per region of the 2-D K-sweep kernel (64, 32 or 16; 0: chosen per level)
    int gen_odd_rows = 0;           // "gen_odd_rows": mg_gen_poisson_level perturbs the diagonal of this many interior rows in 10000
    int cls_escape = 1;             // "row_escape": more than 255 distinct rows -> the frequent ones as classes, the rest read from their stored row
    int storage_auto = 1;           // "storage_auto": a level whose exact symmetry test / row dictionary fails is tried once more with 4 ulps
    int storage_qbits = 0;          // "storage_ulps": low mantissa bits ignored by the symmetry test and the row dictionary
    int use_classes = 1;            // one class byte per row where a level has <= 255 distinct rows (two-sweep pass)
    int fuse_sweeps = 1;            // pairs of Jacobi sweeps in one pass (mg_jacobi2.hip.h) on large 3-D levels
    int64_t fuse_min_rows = (int64_t)1 << 24;
    int fuse_segments = 0;          // plane segments per tile (0: chosen from the item count)
    int fuse_nontemporal = 0;       // streaming loads in the two-sweep kernel (measured slower: tiles re-read their rims)
    int fuse_classes = 1;           // the two-sweep pass reads row classes where the level has them
    int fuse_plain_shape = 0;       // ... its launch shape: 0 = 12 waves x 1 line (no spills), 1 = 8 x 2, 2 = 16 x 1
    int fuse_plain = 2;             // the pass on the stored rows (no classes): 2 = round-2 structure (sdia_jacobi2p), 1 = round 1's
    int class_sweeps = 1;           // so do the one-sweep kernels (residual, single Jacobi / Gauss-Seidel sweeps, SpMV)
    int fuse_k = 5;                 // sweeps per pass of the class-coded K-sweep march (mg_jacobik3d.hip.h): 3..5; < 3: pairs only
    int fuse_k_shape = 7;           // ... its tile: 0 = 128 x 24 cells (12 waves x 2 lines), 1 = 64 x 48 (12 waves x 4 lines),
                                    // 2 = 128 x 24 (8 waves x 3 lines, 256 registers), 3 .. 5 = 64 x 24 by 6 / 8 / 4 waves, two workgroups
                                    // per CU, 6 = 64 x 48 by 16 waves, 7 = 64 x 32 by 16 waves (no spills up to five sweeps: measured best)
    int fuse_k_segments = 0;        // ... plane segments per tile (0: chosen from the item count)
    int timing_force_form = -1;     // mg_time_kernel("jacobik3:formN"): every step of the K-sweep pass in one form (wrong results; how fast
                                    // each form is by itself)
    int64_t fuse_k_slab_min_rows = (int64_t)1 << 20;   // ... on slabs: levels whose smallest slab has at least this many rows
    int fuse_k_slab_min_sweeps = 4; // ... on slabs: smoother calls of at least this many sweeps (fewer: pairs with the boundary chain)
    int64_t fuse_k_min_rows = (int64_t)1 << 24;        // ... on whole levels with at least this many rows (fewer: single sweeps.  129^3 rows:
                                                       //     12.2 us per sweep in the march against 14.3 alone, but whole C3 cycles do not
                                                       //     get faster with it, 127.4 against 127.9 per second; 65^3: 11.6 against 4.0)
    int64_t fuse_k_small_rows = (int64_t)1 << 22;      // ... five sweeps per pass again on levels with fewer rows than this (129^3: 12.2 us
                                                       //     per sweep with five, 15.5 with four: profiles/r03_small_levels.txt)
    int64_t fuse_k5_min_rows = (int64_t)1 << 26;       // ... five sweeps per pass on levels with at least this many rows (257^3: four measured best)
    int64_t fuse_k4_min_rows = 0;                      // ... more than three sweeps per pass on levels with at least this many rows
    int fuse_k_nt_store = 0;        // ... non-temporal stores of its result (experiment)
    int fuse_k_tail = 1;            // ... the tiles left over for a last, nearly empty round of workgroups get shorter plane segments
    int fuse_k_small_tiles = 0;     // ... 64 x 24 tiles (shape 4) on levels whose planes hold fewer 64 x 48 tiles than there are CUs
    int fuse_k_pf = 1;              // ... register sets for the planes of x that arrive (2: x staged a step longer, K = 3 only; measured no faster)
    int fuse_k_dpp = 1;             // ... -1 / +1 neighbours from the neighbouring lanes' registers (0: through LDS, tile 0 only)
    std::vector<const void*> large_lds_kernels;     // kernels whose dynamic-LDS limit has been raised (allow_large_lds)
    int fuse_shape = 1;             // launch shape of the class-coded pass (launch_jacobi2); 1 measured best
    int fuse_wi = 0;                // experiments: cells per tile line with a second sweep (0: chosen per level)
    int fuse_even = 0;              // all tile columns of the class-coded pass equally wide (measured slower: more full tiles, same bytes)
    int march_sweeps = 1;           // one-sweep class kernels as a plane march on large 3-D levels (sdia_sweep1c)
    int64_t march_min_rows = (int64_t)1 << 22;
    int march_shape = 0;            // 0 = 12 waves x 2 lines, 1 = 16 waves x 2 lines
    int64_t fuse_small_2d_rows = 2048;  // ... 2-D levels only up to this many rows (larger ones: the K-sweep 2-D kernel)
    int fuse_small = 1;             // all sweeps of a level that fits one CU's LDS in one launch (sdia_jacobi_small)
    int fuse_2d = 1;                // K sweeps per launch on 2-D levels with row classes (sdia_jacobik2d)
    int fuse_2d_k = 5;              // ... at most this many (2..5)
    int fuse_xcd_chunk = 32;        // consecutive tiles of that pass per XCD at a time
    int cls_blocks_per_cu = 4;      // persistent blocks of the one-sweep class kernels (72 VGPRs, 94 SGPRs admit 7)
    DirectSolver direct;
    // the reference's L2(Omega) norms (res_calculator / err_calculator, multigrid.py:203-218) on the device: a mass
    // matrix M of one level (mg_set_mass_csr), optionally the exact solution's nodal values (mg_set_exact)
    // table prolongation (mg_set_prolongation_table; null: the reference's bilinear / trilinear interpolation)
    int* ptab_count = nullptr;
    int* rtab_count = nullptr;              // table restriction (mg_set_restriction_table), MG_RESTRICT_TABLE
    int* rtab_off = nullptr;
    double* rtab_w = nullptr;
    int rtab_m = 0;
    int* ptab_off = nullptr;
    double* ptab_w = nullptr;
    Level mass;
    int mass_level = -1;
    DVector mass_out, diff, uexact;
    int uexact_level = -1;
    int64_t uploads = 0, downloads = 0, graph_replays = 0;   // whole-vector host <-> device copies; hipGraphLaunch calls
    double* stage = nullptr;        // device staging for host vectors (caller numbering)
    int64_t stage_elems = 0;
    int64_t bytes = 0;
    Comm comm;
    hipDeviceProp_t prop{};
};

namespace {

constexpr int kMaxParts = 1024;

template <class T>
int dev_alloc(mg_context* c, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    c->bytes += (int64_t)(count * sizeof(T));
    return 0;
}

template <class T>
void dev_free(mg_context* c, T*& p, size_t count) {
    if (p) {
        (void)hipFree(p);
        c->bytes -= (int64_t)((count ? count : 1) * sizeof(T));
        p = nullptr;
    }
}

// Elements past the end of every vector: rows of the last, partly filled slice read (and ignore)
// x[row] for up to 64*R - 1 rows beyond nloc in the offset-coded kernels.
// The symmetric-diagonal kernels apply every offset to every row (absent entries are stored zeros), so
// the first / last rows also read x up to the largest offset before / after the vector.  Grid levels
// get zero-filled slack for the widest P1 pattern (offset plane + nx + 1) on both sides; a level whose
// offsets reach further keeps the coded format (repack_sdia checks against vec_reach).
constexpr int64_t kVecSlack = 260;

// (the class-coded two-sweep pass reads whole tiles without per-lane predicates: a plane + 2 lines + 2 beyond either end)
inline int64_t vec_reach(const Level& L) { return L.flat ? 0 : L.g.plane + 2 * (int64_t)L.g.nx + 8; }
inline int64_t vec_front(const Level& L) {
    // the first owned row starts a 128-byte line (hipMalloc returns 256-byte aligned memory)
    const int64_t slack = ((vec_reach(L) + 15) / 16) * 16;
    return slack + (16 - (L.g.lead % 16)) % 16;
}
inline int64_t vec_total(const Level& L) { return vec_front(L) + L.xlen + kVecSlack + vec_reach(L); }

int vec_alloc(mg_context* c, const Level& L, DVector* v) {
    if (v->raw) return 0;
    MG_TRY(dev_alloc(c, &v->raw, (size_t)vec_total(L)));
    v->base = v->raw + vec_front(L);
    v->rows = v->base + L.g.lead;
    HIP_TRY(hipMemsetAsync(v->raw, 0, (size_t)vec_total(L) * sizeof(double), c->stream));
    return 0;
}

void vec_free(mg_context* c, const Level& L, DVector* v) {
    if (v->raw) {
        dev_free(c, v->raw, (size_t)vec_total(L));
        v->base = v->rows = nullptr;
    }
}

inline unsigned blocks_for(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// every row of the level has a class from the table (no escapes): what all class kernels but the K-sweep march need
inline bool cls_full(const Level& L) { return L.cls != nullptr && !L.cls_escape; }

// one thread per node of a plane (x), owned planes (y); block = kPlaneBlock threads
constexpr int kPlaneBlock = 256;
dim3 grid3(const Grid& g, int nk) { return dim3((unsigned)((g.plane + kPlaneBlock - 1) / kPlaneBlock), (unsigned)nk, 1u); }

int check_level(mg_context* c, int level, bool must_be_set = true) {
    if (!c) return fail("null handle");
    if (level < 0 || level >= c->nlev) return fail("level " + std::to_string(level) + " out of range");
    if (must_be_set && !c->L[level].set) return fail("level " + std::to_string(level) + " has not been set");
    return 0;
}

int need_matrix(mg_context* c, int level) {
    MG_TRY(check_level(c, level));
    if (!c->L[level].has_matrix) return fail("level " + std::to_string(level) + " has no matrix (grid-only level)");
    return 0;
}

int need_grid(mg_context* c, int level) {
    MG_TRY(check_level(c, level));
    if (c->L[level].flat) return fail("level " + std::to_string(level) + " is flat (no grid): transfers are undefined");
    return 0;
}

// ---- slab geometry ---------------------------------------------------------------------------
// Plane boundaries on the coarsest grid t_g = floor(g*N0/G) (t_G = N0+1); level l uses
// t_g * 2^l, so coarse plane K and fine plane 2K always live on the same rank.
int setup_geometry(mg_context* c, Level& L, int level, int N, int64_t flat_rows = 0) {
    const int dim = c->dim;
    const Comm& cm = c->comm;
    Grid& g = L.g;
    L.flat = N == 0;
    if (L.flat) {
        // any square matrix: one "plane" holding every row; smoother / residual only
        if (cm.active()) return fail("flat levels are single-GPU only");
        if (flat_rows <= 0 || flat_rows >= (int64_t)INT32_MAX) return fail("bad row count");
        L.N = 0;
        g.nx = (int)flat_rows; g.ny = 1; g.nz = 1; g.plane = flat_rows; g.refine_y = 0;
        g.k0 = 0; g.nk = 1; g.lead = 0;
        L.n_global = L.nloc = L.xlen = flat_rows;
        L.row0 = L.halo_lo = L.halo_hi = 0;
        L.replicated = true;
        L.splits.assign(2, 0);
        L.splits[1] = 1;
        return 0;
    }
    L.N = N;
    g.nx = N + 1;
    g.ny = dim == 3 ? N + 1 : 1;
    g.nz = N + 1;
    g.plane = (int64_t)g.nx * g.ny;
    g.refine_y = dim == 3 ? 1 : 0;
    L.n_global = g.plane * g.nz;
    if (L.n_global >= (int64_t)INT32_MAX) return fail("level has more than 2^31-1 unknowns");
    L.replicated = !(cm.active() && L.n_global >= cm.replicate_below);
    if (level == 0 && cm.active()) L.replicated = true;      // the coarsest solve is never distributed
    L.splits.assign(cm.world + 1, 0);
    if ((N >> level) << level != N) return fail("elements_per_dim is not N0 * 2^level");
    const int N0 = N >> level;
    for (int r = 0; r < cm.world; ++r) L.splits[r] = (int)(((int64_t)r * N0) / cm.world) << level;
    L.splits[cm.world] = N + 1;
    if (!L.replicated) {
        if (N0 < cm.world) return fail("coarsest grid has fewer planes than ranks");
        g.k0 = L.splits[cm.rank];
        g.nk = L.splits[cm.rank + 1] - g.k0;
        if ((N0 / cm.world) << level < c->halo_planes) return fail("a slab is thinner than the halo");
        // room for "halo_depth" planes where the thinnest slab of the level has as many (every rank alike)
        L.hd = std::max(c->halo_planes, std::min(c->halo_depth, (N0 / cm.world) << level));
        L.halo_lo = cm.rank > 0 ? L.hd * g.plane : 0;
        L.halo_hi = cm.rank + 1 < cm.world ? L.hd * g.plane : 0;
    } else {
        g.k0 = 0;
        g.nk = g.nz;
        L.halo_lo = L.halo_hi = 0;
    }
    g.lead = L.halo_lo;
    L.row0 = (int64_t)g.k0 * g.plane;
    L.nloc = (int64_t)g.nk * g.plane;
    L.xlen = L.halo_lo + L.nloc + L.halo_hi;
    return 0;
}

int alloc_level_vectors(mg_context* c, Level& L) {
    MG_TRY(vec_alloc(c, L, &L.v));
    MG_TRY(vec_alloc(c, L, &L.v2));
    MG_TRY(vec_alloc(c, L, &L.f));
    return 0;
}

void free_direct(mg_context* c);

void drop_graphs(mg_context* c);

void free_level(mg_context* c, Level& L) {
    ++c->epoch;
    drop_graphs(c);
    if (&L == &c->L[0]) free_direct(c);
    // what the handle keeps sized for this level's geometry goes with it: the mass matrix and its work vector, the
    // exact solution and the difference vector (a level that is set again may have another size)
    if (c->mass_level >= 0 && &L == &c->L[c->mass_level]) {
        vec_free(c, L, &c->mass_out);
        c->mass_level = -1;
        free_level(c, c->mass);
    }
    if (c->uexact_level >= 0 && &L == &c->L[c->uexact_level]) {
        vec_free(c, L, &c->uexact);
        vec_free(c, L, &c->diff);
        c->uexact_level = -1;
    }
    const size_t ell = (size_t)L.nslices * L.W * (WAVE * L.R);
    dev_free(c, L.vals, ell);
    dev_free(c, L.cols, ell);
    dev_free(c, L.codes, (size_t)L.nslices * ((L.W + 7) / 8) * (WAVE * L.R));
    dev_free(c, L.offsets, 256);
    dev_free(c, L.dvals, (size_t)L.mslices * (L.wu > 0 ? L.wu : 1) * (WAVE * L.R));
    dev_free(c, L.cls, (size_t)L.cls_rows);
    dev_free(c, L.ctab, 256 * CLS_W);
    L.ncls = 0;
    L.cls_escape = false; L.esc_kmax = 0; L.rep_escape = 0; L.grid_decoupled = false;
    L.cls_halo = 0;
    L.rep_sym = -1; L.rep_sym_qbits = L.rep_cls_qbits = 0; L.rep_distinct = -1; L.rep_first_asym = -1; L.rep_max_ulps = 0;
    dev_free(c, L.scls, (size_t)L.nslices * WAVE * L.R);
    dev_free(c, L.s_off, (size_t)256 * L.W);
    dev_free(c, L.s_val, (size_t)256 * L.W);
    dev_free(c, L.s_cnt, 256);
    dev_free(c, L.s_pack, (size_t)256 * L.W);
    L.nscls = 0; L.lm_ntop = 0;
    L.coded = false;
    L.rb_ok = false;
    L.mc_ok = -1;
    L.sdia = false;
    dev_free(c, L.dinv, (size_t)L.nslices * WAVE * L.R);
    dev_free(c, L.perm, (size_t)L.n_global);
    vec_free(c, L, &L.v);
    vec_free(c, L, &L.v2);
    vec_free(c, L, &L.f);
    vec_free(c, L, &L.err);
    vec_free(c, L, &L.ftrue);
    vec_free(c, L, &L.sw);
    L.set = false;
    L.has_matrix = false;
}

// ---- tile kernel dispatch --------------------------------------------------------------------
template <int WT, int R>
void launch_ell_wr(int mode, bool dot, const EllArgs& a, unsigned grid, hipStream_t s) {
    if (mode == MODE_RESIDUAL)
        hipLaunchKernelGGL((ell_apply<WT, R, MODE_RESIDUAL, false>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_JACOBI)
        hipLaunchKernelGGL((ell_apply<WT, R, MODE_JACOBI, false>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_GS)
        hipLaunchKernelGGL((ell_apply<WT, R, MODE_GS, false>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (dot)
        hipLaunchKernelGGL((ell_apply<WT, R, MODE_SPMV, true>), dim3(grid), dim3(BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((ell_apply<WT, R, MODE_SPMV, false>), dim3(grid), dim3(BLOCK), 0, s, a);
}

template <int WT, int R, bool NT>
void launch_ell_coded_wrn(int mode, bool dot, const EllArgs& a, unsigned grid, hipStream_t s) {
    if (mode == MODE_RESIDUAL)
        hipLaunchKernelGGL((ell_apply_coded<WT, R, MODE_RESIDUAL, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_JACOBI)
        hipLaunchKernelGGL((ell_apply_coded<WT, R, MODE_JACOBI, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_GS)
        hipLaunchKernelGGL((ell_apply_coded<WT, R, MODE_GS, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (dot)
        hipLaunchKernelGGL((ell_apply_coded<WT, R, MODE_SPMV, true, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((ell_apply_coded<WT, R, MODE_SPMV, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
}

template <int WT, int R>
void launch_ell_coded_wr(int mode, bool dot, bool nt, const EllArgs& a, unsigned grid, hipStream_t s) {
    if (nt) launch_ell_coded_wrn<WT, R, true>(mode, dot, a, grid, s);
    else launch_ell_coded_wrn<WT, R, false>(mode, dot, a, grid, s);
}

template <int WU, int R, bool NT>
void launch_sdia_wrn(int mode, bool dot, bool finest, const EllArgs& a, unsigned grid, hipStream_t s, int lds_pad) {
    const unsigned lds = a.nslices >= 100000 ? (unsigned)lds_pad : 0u;
    if (mode == MODE_JACOBI && finest)
        hipLaunchKernelGGL((sdia_jacobi_finest<WU, R, NT>), dim3(grid), dim3(BLOCK), lds, s, a);
    else if (mode == MODE_RESIDUAL)
        hipLaunchKernelGGL((sdia_apply<WU, R, MODE_RESIDUAL, false, NT>), dim3(grid), dim3(BLOCK), lds, s, a);
    else if (mode == MODE_JACOBI)
        hipLaunchKernelGGL((sdia_apply<WU, R, MODE_JACOBI, false, NT>), dim3(grid), dim3(BLOCK), lds, s, a);
    else if (mode == MODE_GS)
        hipLaunchKernelGGL((sdia_apply<WU, R, MODE_GS, false, NT>), dim3(grid), dim3(BLOCK), lds, s, a);
    else if (dot)
        hipLaunchKernelGGL((sdia_apply<WU, R, MODE_SPMV, true, NT>), dim3(grid), dim3(BLOCK), lds, s, a);
    else
        hipLaunchKernelGGL((sdia_apply<WU, R, MODE_SPMV, false, NT>), dim3(grid), dim3(BLOCK), lds, s, a);
}

template <int WU, int R, bool NT>
void launch_sdia_cls_wrn(int mode, bool dot, bool finest, const EllArgs& a, unsigned grid, hipStream_t s) {
    if (mode == MODE_JACOBI && finest)
        hipLaunchKernelGGL((sdia_cls_jacobi_finest<WU, R, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_RESIDUAL)
        hipLaunchKernelGGL((sdia_cls_apply<WU, R, MODE_RESIDUAL, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_JACOBI)
        hipLaunchKernelGGL((sdia_cls_apply<WU, R, MODE_JACOBI, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (mode == MODE_GS)
        hipLaunchKernelGGL((sdia_cls_apply<WU, R, MODE_GS, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else if (dot)
        hipLaunchKernelGGL((sdia_cls_apply<WU, R, MODE_SPMV, true, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((sdia_cls_apply<WU, R, MODE_SPMV, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a);
}

template <int R>
void launch_sdia_cls_r(int WU, int mode, bool dot, bool nt, bool finest, const EllArgs& a, unsigned grid, hipStream_t s) {
    if (WU == 3) {
        if (nt) launch_sdia_cls_wrn<3, R, true>(mode, dot, finest, a, grid, s);
        else launch_sdia_cls_wrn<3, R, false>(mode, dot, finest, a, grid, s);
    } else {
        if (nt) launch_sdia_cls_wrn<4, R, true>(mode, dot, finest, a, grid, s);
        else launch_sdia_cls_wrn<4, R, false>(mode, dot, finest, a, grid, s);
    }
}

template <int R>
void launch_sdia_r(int WU, int mode, bool dot, bool nt, bool finest, const EllArgs& a, unsigned grid, hipStream_t s, int pad) {
    switch (WU) {
        case 3: nt ? launch_sdia_wrn<3, R, true>(mode, dot, finest, a, grid, s, pad) : launch_sdia_wrn<3, R, false>(mode, dot, finest, a, grid, s, pad); break;
        case 4: nt ? launch_sdia_wrn<4, R, true>(mode, dot, finest, a, grid, s, pad) : launch_sdia_wrn<4, R, false>(mode, dot, finest, a, grid, s, pad); break;
        default: nt ? launch_sdia_wrn<8, R, true>(mode, dot, finest, a, grid, s, pad) : launch_sdia_wrn<8, R, false>(mode, dot, finest, a, grid, s, pad); break;
    }
}

template <int R, bool NT>
void launch_ell_cls_rn(int mode, bool dot, const EllArgs& a, unsigned grid, hipStream_t s, const Level& L) {
    if (mode == MODE_RESIDUAL)
        hipLaunchKernelGGL((ell_cls_apply<R, MODE_RESIDUAL, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a, L.scls, L.s_off, L.s_val, L.s_cnt);
    else if (mode == MODE_JACOBI)
        hipLaunchKernelGGL((ell_cls_apply<R, MODE_JACOBI, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a, L.scls, L.s_off, L.s_val, L.s_cnt);
    else if (mode == MODE_GS)
        hipLaunchKernelGGL((ell_cls_apply<R, MODE_GS, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a, L.scls, L.s_off, L.s_val, L.s_cnt);
    else if (dot)
        hipLaunchKernelGGL((ell_cls_apply<R, MODE_SPMV, true, NT>), dim3(grid), dim3(BLOCK), 0, s, a, L.scls, L.s_off, L.s_val, L.s_cnt);
    else
        hipLaunchKernelGGL((ell_cls_apply<R, MODE_SPMV, false, NT>), dim3(grid), dim3(BLOCK), 0, s, a, L.scls, L.s_off, L.s_val, L.s_cnt);
}

template <int R>
void launch_ell_cls_r(int mode, bool dot, bool nt, const EllArgs& a, unsigned grid, hipStream_t s, const Level& L) {
    if (nt) launch_ell_cls_rn<R, true>(mode, dot, a, grid, s, L);
    else launch_ell_cls_rn<R, false>(mode, dot, a, grid, s, L);
}

// offset codes are used for every width up to 64 entries per row (5, 7 and 15 have unrolled kernels)
inline bool coded_width(int W) { return W >= 1 && W <= 64; }

template <int R>
void launch_ell_coded_r(int W, int mode, bool dot, bool nt, const EllArgs& a, unsigned grid, hipStream_t s) {
    switch (W) {
        case 5: launch_ell_coded_wr<5, R>(mode, dot, nt, a, grid, s); break;
        case 7: launch_ell_coded_wr<7, R>(mode, dot, nt, a, grid, s); break;
        case 15: launch_ell_coded_wr<15, R>(mode, dot, nt, a, grid, s); break;
        default: launch_ell_coded_wr<0, R>(mode, dot, nt, a, grid, s); break;
    }
}

template <int R>
void launch_ell_r(int W, int mode, bool dot, const EllArgs& a, unsigned grid, hipStream_t s) {
    switch (W) {
        case 5: launch_ell_wr<5, R>(mode, dot, a, grid, s); break;
        case 7: launch_ell_wr<7, R>(mode, dot, a, grid, s); break;
        case 15: launch_ell_wr<15, R>(mode, dot, a, grid, s); break;
        default: launch_ell_wr<0, R>(mode, dot, a, grid, s); break;
    }
}

int launch_sweep1c(mg_context* c, const Level& L, int mode, const double* x_rows, const double* f_rows, double* out_rows,
                   int color);
bool sweep1c_ok(const mg_context* c, const Level& L);

int allow_large_lds(mg_context* c, const void* kernel, size_t bytes);

// Plane march of wide lattice stencils (mg_lattice.hip.h): 3-D grid levels with stencil classes whose entries reach at
// most two cells / lines / planes (prepare_lat_march found them decomposable).
bool lat_march_ok(const mg_context* c, const Level& L) {
    return c->lattice_march && L.scls && L.s_pack && L.lm_ntop > 0 && !L.flat && L.g.ny >= 16 && L.g.nx >= 32 &&
           L.nloc >= c->lattice_march_min_rows;
}

template <int TI, int TJ, int NT>
int launch_lat_march_t(mg_context* c, const Level& L, LatArgs a, int mode) {
    a.ntx = (L.g.nx + TI - 1) / TI; a.nty = (L.g.ny + TJ - 1) / TJ;
    const int64_t ntile = (int64_t)a.ntx * a.nty;
    const size_t lds = lm_lds_bytes(L.W, TI, TJ);
    const int per_cu = lds > (size_t)80 * 1024 ? 1 : 2;             // resident workgroups per CU
    const int64_t cus = std::max(1, c->prop.multiProcessorCount);
    // Segments per tile.  Many tiles (the 513^3 lattice: 297): about 6.5 rounds of the resident workgroups -- measured, 5 .. 12
    // segments per tile are within 4 %, the finer cuts ahead: with that many workgroups the rounds are not rigid.  Few tiles
    // (257^3: 85, 129^3: 27): the rounds are rigid, so the cut that minimises rounds x (planes per segment + 5 planes of
    // warm-up); segments of at least 16 planes.
    const int64_t resident = per_cu * cus;
    int nseg = c->lattice_segments;
    if (nseg <= 0 && 2 * ntile >= resident) nseg = (int)std::max<int64_t>(1, (13 * resident / 2 + ntile - 1) / ntile);
    if (nseg <= 0) {
        double best = 1e300;
        for (int n = 1; n <= std::max(1, L.g.nk / 16); ++n) {
            const double cost = (double)((ntile * n + resident - 1) / resident) * ((L.g.nk + n - 1) / n + 5.0);
            if (cost < best) { best = cost; nseg = n; }
        }
    }
    nseg = std::max(1, std::min(nseg, std::max(1, L.g.nk / 16)));
    a.seglen = (L.g.nk + nseg - 1) / nseg;
    nseg = (L.g.nk + a.seglen - 1) / a.seglen;
    const int64_t items = ntile * nseg;
    if (items >= ((int64_t)1 << 31) - 4096) return fail("too many tiles");
    a.nitems = (unsigned)items;
    a.xcd_chunk = 16;
    const int64_t group = 8 * (int64_t)a.xcd_chunk;
    const unsigned grid = (unsigned)(((items + group - 1) / group) * group);
    void (*kern)(LatArgs) = mode == MODE_RESIDUAL ? lat_march<MODE_RESIDUAL, TI, TJ, NT>
                          : mode == MODE_GS ? lat_march<MODE_GS, TI, TJ, NT> : lat_march<MODE_JACOBI, TI, TJ, NT>;
    if (lds > (size_t)150 * 1024) return fail("lattice march: class tables too wide");
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern), (size_t)150 * 1024));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NT), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_lat_march(mg_context* c, const Level& L, int mode, const double* x_rows, const double* f_rows, double* out_rows,
                     int color) {
    LatArgs a{};
    a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.cls = L.scls; a.s_pack = L.s_pack; a.s_val = L.s_val; a.s_cnt = L.s_cnt;
    a.W = L.W; a.WP = (L.W + 3) / 4 * 4 + 4; a.ntop = L.lm_ntop;
    for (int t = 0; t < LM_K; ++t) a.top[t] = L.lm_top[t];
    a.nloc = L.nloc; a.P = L.g.plane;
    a.xlo = -(L.halo_lo ? (int64_t)c->halo_planes * L.g.plane : 0); a.xhi = L.nloc + (L.halo_hi ? (int64_t)c->halo_planes * L.g.plane : 0);
    a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk; a.kg0 = (int)(L.row0 / L.g.plane);
    a.color = color; a.omega = c->omega;
    // ("lattice_tile" 2: the wide tile moves 23 % fewer bytes but its one 512-thread workgroup per CU is slower than two of
    //  256 -- 8.0 against 6.9 ms per Gauss-Seidel sweep of the 513^3 lattice: kept for experiments)
    if (c->lattice_tile == 2 && L.g.nx >= 128) return launch_lat_march_t<128, 16, 512>(c, L, a, mode);
    return launch_lat_march_t<64, 16, 256>(c, L, a, mode);
}

// out = op(A, x) over all owned slices of the level
int launch_ell(mg_context* c, const Level& L, int mode, bool dot, const double* x_base, const double* f_rows,
               double* out_rows, double* partials, const int* done, unsigned* grid_out = nullptr,
               int64_t slice0 = 0, int64_t slice_count = -1, int color = 0, int64_t split = 0, int64_t gap = 0) {
    // (gap > 0: the slice_count slices are [slice0, slice0 + split) and [slice0 + split + gap, ...) -- two ranges, one launch)
    if (slice_count < 0) slice_count = L.nslices - slice0;
    if (slice_count == 0) return 0;
    // whole large 3-D levels with row classes: one sweep as a plane march (mg_jacobi2.hip.h, sdia_sweep1c)
    if (!dot && !done && mode != MODE_SPMV && slice0 == 0 && slice_count == L.nslices && sweep1c_ok(c, L))
        return launch_sweep1c(c, L, mode, x_base + L.g.lead, f_rows, out_rows, color);
    EllArgs a{};
    a.vals = L.vals; a.cols = L.cols; a.x = x_base; a.f = f_rows; a.dinv = L.dinv; a.out = out_rows;
    a.partials = partials; a.done_flag = done; a.nloc = L.nloc; a.lead = L.g.lead;
    a.slice0 = slice0; a.nslices = slice_count; a.split = split; a.gap = gap; a.omega = c->omega; a.W = L.W; a.chunk = c->chunk;
    a.codes = L.codes; a.offsets = L.offsets; a.ntable = L.ntable; a.dcode = L.dcode;
    a.color = color; a.color_kind = c->smoother == MG_SMOOTH_MCGS ? COLOR_LATTICE9 : COLOR_PARITY;
    a.grow0 = L.row0; a.gnx = L.g.nx; a.gny = L.g.ny;
    unsigned grid = blocks_for(slice_count, WAVES_PER_BLOCK);
    // XCD strip traversal (optional): pays when a plane is much larger than a strip
    auto plan_strips = [&]() {
        const int64_t ps4 = (L.g.plane / (WAVE * L.R)) / 4;            // blocks per pseudo-plane
        const int64_t want = c->strip_slices / 4;                       // blocks per strip asked for
        // (sub-ranges of a slab -- interior slices while the halo travels -- are strip-walked too: the
        // pseudo-planes then start at slice0, which shifts them against the real planes by a constant)
        const int64_t nblocks = (slice_count + 3) / 4;
        if (!(want > 0 && !dot && L.g.nz > 1 && ps4 >= 16 * want && nblocks >= 4 * ps4)) return;
        const int64_t m = std::max<int64_t>(1, (ps4 + 4 * want) / (8 * want));     // round(ps4 / want / 8)
        const int64_t ns = 8 * m;
        const int64_t kp = (nblocks + ps4 - 1) / ps4;
        const int64_t bmax = (ps4 + ns - 1) / ns;
        const int64_t g = ns * kp * bmax;
        if (g < ((int64_t)1 << 31) && kp < ((int64_t)1 << 31)) {
            a.strip_ns = (unsigned)ns; a.strip_bmax = (unsigned)bmax; a.ps4 = (unsigned)ps4; a.kp = (unsigned)kp;
            grid = (unsigned)g;
        }
    };
    if (L.sdia) {
        a.vals = L.dvals; a.mlead = L.mlead;
        for (int t = 0; t < 8; ++t) a.up[t] = L.up[t];
        plan_strips();
        if (grid_out) *grid_out = grid;
        const bool nt = c->nontemporal != 0;
        const bool finest = c->nlev > 1 && &L == &c->L[c->nlev - 1];
        if (cls_full(L) && c->class_sweeps) {
            // the rows through their classes: 25 instead of 56 bytes per row (mg_kernels.hip.h, sdia_cls_body);
            // persistent blocks (8 per CU, a multiple of 8 so that a block's groups stay on one XCD's share)
            a.cls = L.cls + L.cls_lead; a.ctab = L.ctab; a.ncls = L.ncls; a.cmain = L.cmain;
            for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
            a.nvirt = grid;
            // (with the dot product one group per block, as the other formats have it: the partial sums and with them
            //  the rounding of the result do not depend on the format)
            const unsigned resident = (unsigned)c->cls_blocks_per_cu * (unsigned)std::max(1, c->prop.multiProcessorCount);
            if (!dot) grid = std::min(grid, resident);
            if (grid_out) *grid_out = grid;
            switch (L.R) {
                case 1: launch_sdia_cls_r<1>(L.wu, mode, dot, nt, finest, a, grid, c->stream); break;
                case 2: launch_sdia_cls_r<2>(L.wu, mode, dot, nt, finest, a, grid, c->stream); break;
                case 4: launch_sdia_cls_r<4>(L.wu, mode, dot, nt, finest, a, grid, c->stream); break;
                default: return fail("unsupported rows_per_lane");
            }
            HIP_TRY(hipGetLastError());
            return 0;
        }
        switch (L.R) {
            case 1: launch_sdia_r<1>(L.wu, mode, dot, nt, finest, a, grid, c->stream, c->lds_pad); break;
            case 2: launch_sdia_r<2>(L.wu, mode, dot, nt, finest, a, grid, c->stream, c->lds_pad); break;
            case 4: launch_sdia_r<4>(L.wu, mode, dot, nt, finest, a, grid, c->stream, c->lds_pad); break;
            default: return fail("unsupported rows_per_lane");
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (L.coded && L.scls && c->class_sweeps) {
        // whole 3-D lattice levels: plane march with x in LDS (mg_lattice.hip.h)
        if (!dot && !done && mode != MODE_SPMV && slice0 == 0 && slice_count == L.nslices && gap == 0 && lat_march_ok(c, L))
            return launch_lat_march(c, L, mode, x_base + L.g.lead, f_rows, out_rows, color);
        // wide rows through their stencil classes: 25 bytes per row instead of the stored row (ell_cls_apply)
        if (grid_out) *grid_out = grid;
        const bool nt = c->nontemporal != 0;
        switch (L.R) {
            case 1: launch_ell_cls_r<1>(mode, dot, nt, a, grid, c->stream, L); break;
            case 2: launch_ell_cls_r<2>(mode, dot, nt, a, grid, c->stream, L); break;
            case 4: launch_ell_cls_r<4>(mode, dot, nt, a, grid, c->stream, L); break;
            default: return fail("unsupported rows_per_lane");
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (L.coded) {
        plan_strips();
        if (grid_out) *grid_out = grid;
        const bool nt = c->nontemporal != 0;
        switch (L.R) {
            case 1: launch_ell_coded_r<1>(L.W, mode, dot, nt, a, grid, c->stream); break;
            case 2: launch_ell_coded_r<2>(L.W, mode, dot, nt, a, grid, c->stream); break;
            case 4: launch_ell_coded_r<4>(L.W, mode, dot, nt, a, grid, c->stream); break;
            default: return fail("unsupported rows_per_lane");
        }
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (grid_out) *grid_out = grid;
    switch (L.R) {
        case 1: launch_ell_r<1>(L.W, mode, dot, a, grid, c->stream); break;
        case 2: launch_ell_r<2>(L.W, mode, dot, a, grid, c->stream); break;
        case 4: launch_ell_r<4>(L.W, mode, dot, a, grid, c->stream); break;
        default: return fail("unsupported rows_per_lane");
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- communication -----------------------------------------------------------------------------
int ensure_host_stage(mg_context* c, size_t elems) {
    Comm& cm = c->comm;
    if (cm.h_stage_elems >= elems) return 0;
    if (cm.h_stage) (void)hipHostFree(cm.h_stage);
    cm.h_stage = nullptr;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&cm.h_stage), elems * sizeof(double), hipHostMallocDefault));
    cm.h_stage_elems = elems;
    return 0;
}

// `count` doubles to and from each slab neighbour (device buffers; a rank without that neighbour passes nothing)
int exchange_raw(mg_context* c, const double* send_lo, const double* send_hi, double* recv_lo, double* recv_hi, size_t count,
                 hipStream_t stream) {
    Comm& cm = c->comm;
    const bool lo = cm.rank > 0, hi = cm.rank + 1 < cm.world;
    if (cm.nccl) {
        NCCL_TRY(g_rccl.GroupStart());
        if (lo) {
            NCCL_TRY(g_rccl.Send(send_lo, count, ncclDouble, cm.rank - 1, cm.nccl, stream));
            NCCL_TRY(g_rccl.Recv(recv_lo, count, ncclDouble, cm.rank - 1, cm.nccl, stream));
        }
        if (hi) {
            NCCL_TRY(g_rccl.Send(send_hi, count, ncclDouble, cm.rank + 1, cm.nccl, stream));
            NCCL_TRY(g_rccl.Recv(recv_hi, count, ncclDouble, cm.rank + 1, cm.nccl, stream));
        }
        NCCL_TRY(g_rccl.GroupEnd());
        return 0;
    }
    if (!cm.ex) return fail("no transport configured");
    MG_TRY(ensure_host_stage(c, 4 * count));
    double* h = cm.h_stage;
    if (lo) HIP_TRY(hipMemcpyAsync(h, send_lo, count * 8, hipMemcpyDeviceToHost, stream));
    if (hi) HIP_TRY(hipMemcpyAsync(h + count, send_hi, count * 8, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    if (cm.ex(cm.user, lo ? h : nullptr, hi ? h + count : nullptr, lo ? h + 2 * count : nullptr,
              hi ? h + 3 * count : nullptr, (int64_t)count) != 0)
        return fail("exchange callback failed");
    if (lo) HIP_TRY(hipMemcpyAsync(recv_lo, h + 2 * count, count * 8, hipMemcpyHostToDevice, stream));
    if (hi) HIP_TRY(hipMemcpyAsync(recv_hi, h + 3 * count, count * 8, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return 0;
}

// Fill the `planes` halo planes of `v` NEXT TO its owned rows from the neighbouring slabs (the first owned planes go down,
// the last ones go up); planes < 0: as many as one application of the level's rows reaches ("halo_planes").
int exchange_halo(mg_context* c, const Level& L, DVector& v, hipStream_t stream = nullptr, int planes = -1) {
    if (!stream) stream = c->stream;
    Comm& cm = c->comm;
    if (L.replicated || !cm.active()) return 0;
    if (planes < 0) planes = c->halo_planes;
    if (planes > L.hd) return fail("the vectors of this level have no room for that many halo planes");
    const size_t count = (size_t)planes * (size_t)L.g.plane;       // elements per message
    return exchange_raw(c, v.rows, v.rows + L.nloc - count, v.rows - count, v.rows + L.nloc, count, stream);
}

// In-place sum of `count` device doubles over all ranks.
int allreduce_sum(mg_context* c, double* dev, int64_t count) {
    Comm& cm = c->comm;
    if (!cm.active()) return 0;
    if (cm.nccl) {
        NCCL_TRY(g_rccl.AllReduce(dev, dev, (size_t)count, ncclDouble, ncclSum, cm.nccl, c->stream));
        return 0;
    }
    if (!cm.ar) return fail("no transport configured");
    MG_TRY(ensure_host_stage(c, (size_t)count));
    HIP_TRY(hipMemcpyAsync(cm.h_stage, dev, (size_t)count * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (cm.ar(cm.user, cm.h_stage, count) != 0) return fail("allreduce callback failed");
    HIP_TRY(hipMemcpyAsync(dev, cm.h_stage, (size_t)count * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// `full` holds every plane of a level (global lexicographic order); each rank has written the
// planes [splits[rank], splits[rank+1]) and receives the others.
int allgather_planes(mg_context* c, const std::vector<int>& splits, int64_t plane, double* full) {
    Comm& cm = c->comm;
    if (!cm.active()) return 0;
    if (cm.nccl) {
        NCCL_TRY(g_rccl.GroupStart());
        for (int r = 0; r < cm.world; ++r) {
            double* seg = full + (int64_t)splits[r] * plane;
            const size_t cnt = (size_t)(splits[r + 1] - splits[r]) * (size_t)plane;
            NCCL_TRY(g_rccl.Broadcast(seg, seg, cnt, ncclDouble, r, cm.nccl, c->stream));
        }
        NCCL_TRY(g_rccl.GroupEnd());
        return 0;
    }
    if (!cm.ag) return fail("no transport configured");
    const int64_t total = (int64_t)splits[cm.world] * plane;
    const int64_t mine = (int64_t)(splits[cm.rank + 1] - splits[cm.rank]) * plane;
    MG_TRY(ensure_host_stage(c, (size_t)(total + mine)));
    double* h_send = cm.h_stage + total;
    HIP_TRY(hipMemcpyAsync(h_send, full + (int64_t)splits[cm.rank] * plane, (size_t)mine * 8, hipMemcpyDeviceToHost,
                           c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int64_t> counts(cm.world);
    for (int r = 0; r < cm.world; ++r) counts[r] = (int64_t)(splits[r + 1] - splits[r]) * plane;
    if (cm.ag(cm.user, h_send, mine, cm.h_stage, counts.data()) != 0) return fail("allgatherv callback failed");
    HIP_TRY(hipMemcpyAsync(full, cm.h_stage, (size_t)total * 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- the path ----------------------------------------------------------------------------------------
DVector* pick(Level& L, int which) {
    switch (which) {
        case MG_VEC_V: return &L.v;
        case MG_VEC_F: return &L.f;
        case MG_VEC_R: return &L.v2;
        case MG_VEC_ERR: return &L.err;
        default: return nullptr;
    }
}

// Two sweeps in one pass (mg_jacobi2.hip.h): whole, undistributed 3-D levels whose stored diagonals are exactly
// {0, +1, +nx, +plane}.
// rows of the smallest slab of a level: decisions that every rank must take alike are made on it, not on the rank's
// own row count (uneven splits differ by 2^level planes)
int64_t min_slab_rows(const Level& L) {
    if (L.replicated || L.splits.size() < 2) return L.nloc;
    int nk = INT32_MAX;
    for (size_t r = 0; r + 1 < L.splits.size(); ++r) nk = std::min(nk, L.splits[r + 1] - L.splits[r]);
    return (int64_t)nk * L.g.plane;
}

bool fused_sweeps_ok(const mg_context* c, const Level& L, bool ignore_size = false) {
    if (!c->fuse_sweeps || !L.sdia || L.wu != 4 || L.flat) return false;
    if (L.g.nx < 32 || L.g.ny < 32 || min_slab_rows(L) < 8 * L.g.plane) return false;    // zero slack >= 3 slices, slabs >= 8 planes
    if (L.up[1] != 1 || L.up[2] != L.g.nx || (int64_t)L.up[3] != L.g.plane) return false;
    return ignore_size || min_slab_rows(L) >= c->fuse_min_rows;
}

// Kernels that ask for more than 64 KiB of dynamic LDS need the attribute once per function (and device: one
// device per handle); remembered in the handle.
// One launch of the two-colour Gauss-Seidel pass (mg_lattice.hip.h, lat_gs2): colours c1 and c2 (c2 < 0: c1 only) of the
// sweep that takes the level's iterate from `xold` to `xnew`.
int launch_lat_gs2(mg_context* c, const Level& L, int c1, int c2, const double* xold_rows, double* xnew_rows) {
    Gs2Args a{};
    a.xold = xold_rows; a.xnew = xnew_rows; a.f = L.f.rows;
    a.cls = L.scls; a.s_pack = L.s_pack; a.s_val = L.s_val; a.s_cnt = L.s_cnt;
    a.W = L.W; a.WP = (L.W + 3) / 4 * 4 + 4; a.ntop = L.lm_ntop;
    for (int t = 0; t < LM_K; ++t) a.top[t] = L.lm_top[t];
    a.nloc = L.nloc; a.P = L.g.plane; a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk;
    a.c1 = c1; a.c2 = c2; a.omega = c->omega;
    a.ntx = (L.g.nx + G2_TI - 1) / G2_TI; a.nty = (L.g.ny + G2_TJ - 1) / G2_TJ;
    const int64_t ntile = (int64_t)a.ntx * a.nty;
    const int64_t resident = 2 * (int64_t)std::max(1, c->prop.multiProcessorCount);
    // (as launch_lat_march_t; a segment pays 5 + 4 planes of warm-up and 3 of trailing second colour)
    int nseg = c->lattice_segments;
    if (nseg <= 0 && 2 * ntile >= resident) nseg = (int)std::max<int64_t>(1, (13 * resident / 2 + ntile - 1) / ntile);
    if (nseg <= 0) {
        double best = 1e300;
        for (int n = 1; n <= std::max(1, L.g.nk / 16); ++n) {
            const double cost = (double)((ntile * n + resident - 1) / resident) * ((L.g.nk + n - 1) / n + 12.0);
            if (cost < best) { best = cost; nseg = n; }
        }
    }
    nseg = std::max(1, std::min(nseg, std::max(1, L.g.nk / 16)));
    a.seglen = (L.g.nk + nseg - 1) / nseg;
    nseg = (L.g.nk + a.seglen - 1) / a.seglen;
    const int64_t items = ntile * nseg;
    if (items >= ((int64_t)1 << 31) - 4096) return fail("too many tiles");
    a.nitems = (unsigned)items;
    a.xcd_chunk = 16;
    const int64_t group = 8 * (int64_t)a.xcd_chunk;
    const unsigned grid = (unsigned)(((items + group - 1) / group) * group);
    const size_t lds = g2_lds_bytes(L.W);
    if (lds > (size_t)80 * 1024) return fail("lattice march: class tables too wide");
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(lat_gs2), (size_t)80 * 1024));
    hipLaunchKernelGGL(lat_gs2, dim3(grid), dim3(G2_THREADS), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int allow_large_lds(mg_context* c, const void* kernel, size_t bytes) {
    if (std::find(c->large_lds_kernels.begin(), c->large_lds_kernels.end(), kernel) != c->large_lds_kernels.end()) return 0;
    HIP_TRY(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    c->large_lds_kernels.push_back(kernel);
    return 0;
}

constexpr int kJ2Lines = 16;        // grid lines per tile (8 waves x 2)

// How a level's owned planes are cut into segments (one work item = one tile x one segment).  Segment 0 is
// [0, zb), the last one [nk-zb, nk), the others cut the planes between into pieces of `seglen`.
//   whole levels: equal pieces, their number chosen by a cost model -- rounds of 256 resident workgroups times
//     (planes per piece + ~2.5 plane-times of warm-up) -- which is what measured best on 1025^3 (8 pieces);
//   slabs: zb short (the planes whose once-relaxed values travel to the neighbours, plus one), so that the
//     boundary work that the exchanges wait for is small, the interior cut by the same cost model.
struct J2Plan { int ntx, nty, nseg, zb, seglen, wi; };

// grid lines per tile: 16 for the plain pass; the class-coded pass has shapes with 16 and 32 ("fuse_shape")
int jacobi2_lines(const mg_context* c, const Level& L) {
    const int sh = c->fuse_shape;
    if (!(cls_full(L) && c->fuse_classes)) return (c->fuse_plain == 2 && c->fuse_plain_shape == 0) ? 12 : kJ2Lines;
    return (sh == 1 || sh == 3) ? 24 : kJ2Lines;
}

J2Plan jacobi2_plan(const mg_context* c, const Level& L, bool slab, int64_t boundary_rows) {
    J2Plan p{};
    const int lines = jacobi2_lines(c, L);
    p.ntx = (L.g.nx + J2_EX - 3) / (J2_EX - 2);
    p.nty = (L.g.ny + lines - 3) / (lines - 2);
    if ((cls_full(L) && c->fuse_classes) || c->fuse_plain == 2) {
        // class-coded pass / round-2 plain pass: the x ring is part of the tile (124 cells with a second sweep at most), and all tile
        // columns have the same width, the smallest that covers the grid ("fuse_even" 0: always the widest)
        p.ntx = (L.g.nx + J2_EX - 5) / (J2_EX - 4);
        p.wi = c->fuse_even ? (L.g.nx + p.ntx - 1) / p.ntx : J2_EX - 4;
        if (c->fuse_wi > 0) {                                       // experiments: a given width
            p.wi = std::min(J2_EX - 4, std::max(8, c->fuse_wi));
            p.ntx = (L.g.nx + p.wi - 1) / p.wi;
        }
    }
    const int64_t ntile = (int64_t)p.ntx * p.nty;
    const int nk = L.g.nk;
    const int64_t cus = std::max(1, c->prop.multiProcessorCount);      // one resident workgroup per CU
    auto pieces = [&](int planes, int most) {
        int best = 1;
        double best_cost = 1e300;
        for (int n = 1; n <= std::max(1, most); ++n) {
            const double cost = (double)((ntile * n + cus - 1) / cus) * ((planes + n - 1) / n + 2.5);
            if (cost < best_cost) { best_cost = cost; best = n; }
        }
        return best;
    };
    if (slab) {
        // one plane more than the rows the neighbours wait for (3 planes + 1 on a 1025^2 plane)
        const int zb = (int)std::max<int64_t>(4, (boundary_rows + L.g.plane - 1) / L.g.plane + 1);
        if (nk >= 2 * zb + 8) {
            const int inner = nk - 2 * zb;
            int m = c->fuse_segments > 2 ? c->fuse_segments - 2 : pieces(inner, inner / 8);
            m = std::max(1, std::min(m, inner));
            p.zb = zb; p.seglen = (inner + m - 1) / m; p.nseg = 2 + (inner + p.seglen - 1) / p.seglen;
            return p;
        }
        p.zb = nk; p.seglen = nk; p.nseg = 1;      // too thin to split: one segment, exchanges in sequence
        return p;
    }
    int n = c->fuse_segments > 0 ? c->fuse_segments : pieces(nk, nk / 16);
    n = std::max(1, std::min(n, nk));
    if (n <= 2) {
        p.zb = (nk + n - 1) / n; p.seglen = nk; p.nseg = n;
    } else {
        p.zb = std::max(1, nk / n);
        const int inner = nk - 2 * p.zb;
        p.seglen = (inner + n - 3) / (n - 2);
        p.nseg = 2 + (inner + p.seglen - 1) / p.seglen;
    }
    return p;
}

template <int NW, int LPW>
int launch_jacobi2c_t(mg_context* c, const J2Args& a, int nseg, bool finest) {
    const int64_t items = (int64_t)a.ntx * a.nty * nseg;
    if (items >= ((int64_t)1 << 31) - 4096) return fail("too many tiles");
    J2Args b = a;
    b.nitems = (unsigned)items;
    b.xcd_chunk = (unsigned)c->fuse_xcd_chunk;
    const int64_t group = 8 * (int64_t)b.xcd_chunk;
    const unsigned grid = (unsigned)(((items + group - 1) / group) * group);
    constexpr size_t lds = j2c_lds_bytes<NW, LPW>();
    void (*const kern[2])(J2Args) = {sdia_jacobi2c<NW, LPW>, sdia_jacobi2c_finest<NW, LPW>};
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern[finest ? 1 : 0]), lds));
    hipLaunchKernelGGL(kern[finest ? 1 : 0], dim3(grid), dim3(NW * WAVE), lds, c->stream, b);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int R, int NW, int LPW>
int launch_jacobi2p_t(mg_context* c, const J2Args& a, int nseg, bool finest) {
    const int64_t items = (int64_t)a.ntx * a.nty * nseg;
    if (items >= ((int64_t)1 << 31) - 4096) return fail("too many tiles");
    J2Args b = a;
    b.nitems = (unsigned)items;
    b.xcd_chunk = (unsigned)c->fuse_xcd_chunk;
    const int64_t group = 8 * (int64_t)b.xcd_chunk;
    const unsigned grid = (unsigned)(((items + group - 1) / group) * group);
    constexpr size_t lds = j2p_lds_bytes<NW, LPW>();
    void (*const kern[2])(J2Args) = {sdia_jacobi2p<R, NW, LPW>, sdia_jacobi2p_finest<R, NW, LPW>};
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern[finest ? 1 : 0]), lds));
    hipLaunchKernelGGL(kern[finest ? 1 : 0], dim3(grid), dim3(NW * WAVE), lds, c->stream, b);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int R, int NW, int LPW>
int launch_jacobi2_t(mg_context* c, const J2Args& a, int nseg, bool finest) {
    static_assert(NW * LPW == kJ2Lines, "tile height");
    const int64_t items = (int64_t)a.ntx * a.nty * nseg;
    if (items >= ((int64_t)1 << 31) - 512) return fail("too many tiles");
    J2Args b = a;
    b.nitems = (unsigned)items;
    const unsigned grid = (unsigned)((items + 255) / 256) * 256u;      // whole groups of 8 XCDs x 32 items
    constexpr size_t lds = j2_lds_bytes<NW, LPW>();
    void (*const kern[4])(J2Args) = {sdia_jacobi2<R, NW, LPW, false>, sdia_jacobi2<R, NW, LPW, true>,
                                     sdia_jacobi2_finest<R, NW, LPW, false>, sdia_jacobi2_finest<R, NW, LPW, true>};
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern[(finest ? 2 : 0) + (c->fuse_nontemporal ? 1 : 0)]), lds));
    hipLaunchKernelGGL(kern[(finest ? 2 : 0) + (c->fuse_nontemporal ? 1 : 0)], dim3(grid), dim3(NW * WAVE), lds, c->stream, b);
    HIP_TRY(hipGetLastError());
    return 0;
}

// out = two Jacobi sweeps applied to x, for `count` plane segments seg0, seg0 + stride, ... of `plan`
// (slabs: the second sweep is stored for the rows [st_lo, st_hi) only and the once-relaxed iterate of the rows
// within reach of the others goes to v1_rows, see smooth())
int launch_jacobi2(mg_context* c, const Level& L, const J2Plan& plan, int seg0, int stride, int count, const double* x_rows,
                   const double* f_rows, double* out_rows, int64_t st_lo = 0, int64_t st_hi = INT64_MAX,
                   double* v1_rows = nullptr) {
    if (count <= 0) return 0;
    J2Args a{};
    a.vals = L.dvals; a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.nloc = L.nloc; a.mlead = L.mlead; a.P = L.g.plane;
    a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk; a.omega = c->omega;
    a.zero = L.f.raw;
    a.xlo = -(L.halo_lo ? (int64_t)c->halo_planes * L.g.plane : 0); a.xhi = L.nloc + (L.halo_hi ? (int64_t)c->halo_planes * L.g.plane : 0);
    a.slo = a.xlo;
    a.st_lo = st_lo; a.st_hi = st_hi; a.v1out = v1_rows;
    a.k1_lo = st_lo + L.g.plane + L.g.nx + 2; a.k1_hi = st_hi - L.g.plane - L.g.nx - 2;
    a.ntx = plan.ntx; a.nty = plan.nty; a.nseg = plan.nseg; a.zb = plan.zb; a.seglen = plan.seglen;
    a.seg0 = seg0; a.seg_stride = stride;
    const int n = count;
    // 8 waves x 2 grid lines each (16 waves x 1 line measured slower and does not fit 128 registers)
    const bool finest = c->nlev > 1 && &L == &c->L[c->nlev - 1];
    if (cls_full(L) && c->fuse_classes) {
        a.cls = L.cls; a.ctab = L.ctab; a.clead = L.cls_lead;
        a.ncls = L.ncls; a.cmain = L.cmain; a.wi = plan.wi;
        for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
        switch (c->fuse_shape) {
            case 0: return launch_jacobi2c_t<8, 2>(c, a, n, finest);       // 16 lines, 512 threads
            case 2: return launch_jacobi2c_t<16, 1>(c, a, n, finest);      // 16 lines, 1024 threads
            case 3: return launch_jacobi2c_t<8, 3>(c, a, n, finest);       // 24 lines, 512 threads
            default: return launch_jacobi2c_t<12, 2>(c, a, n, finest);     // 24 lines, 768 threads (measured best)
        }
    }
    if (c->fuse_plain == 2) {            // round-2 structure of the pass on the stored rows (sdia_jacobi2p)
        a.wi = plan.wi;
        if (c->fuse_plain_shape == 0) {  // 12 waves x 1 grid line (155 VGPRs, no spills: measured best)
            if (L.R == 2) return launch_jacobi2p_t<2, 12, 1>(c, a, n, finest);
            if (L.R == 1) return launch_jacobi2p_t<1, 12, 1>(c, a, n, finest);
            return launch_jacobi2p_t<4, 12, 1>(c, a, n, finest);
        }
        if (c->fuse_plain_shape == 2) {  // 16 waves x 1 grid line
            if (L.R == 2) return launch_jacobi2p_t<2, 16, 1>(c, a, n, finest);
            if (L.R == 1) return launch_jacobi2p_t<1, 16, 1>(c, a, n, finest);
            return launch_jacobi2p_t<4, 16, 1>(c, a, n, finest);
        }
        if (L.R == 2) return launch_jacobi2p_t<2, 8, 2>(c, a, n, finest);
        if (L.R == 1) return launch_jacobi2p_t<1, 8, 2>(c, a, n, finest);
        return launch_jacobi2p_t<4, 8, 2>(c, a, n, finest);
    }
    if (L.R == 2) return launch_jacobi2_t<2, 8, 2>(c, a, n, finest);
    if (L.R == 1) return launch_jacobi2_t<1, 8, 2>(c, a, n, finest);
    return launch_jacobi2_t<4, 8, 2>(c, a, n, finest);
}

// ---- row classes of the neighbours' planes (slabs, K-sweep march) --------------------------------------------------
__global__ void bytes_to_doubles(const unsigned char* __restrict__ in, double* __restrict__ out, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = (double)in[t];
}
__global__ void doubles_to_classes(const double* __restrict__ in, const int* __restrict__ map, unsigned char* __restrict__ out, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = (unsigned char)map[(int)in[t] & 255];
}

// The march relaxes the K - 1 planes of either neighbour next to this slab along with its own (mg_jacobik3d.hip.h), so
// it needs their rows: as class bytes in front of / behind the slab's own in `cls`, in THIS rank's class numbering.
// Every rank builds its dictionary by itself, so the neighbours send their class bytes together with their tables, and
// a class is translated by looking its row up in the own table (bit for bit; an unknown row is appended).  Collective:
// every rank of a distributed level calls it at the same point; if one of them cannot translate (more than 255 classes),
// all of them leave the march alone on this level.
int ensure_class_halos(mg_context* c, Level& L) {
    if (L.cls_halo != 0) return 0;
    L.cls_halo = -1;
    Comm& cm = c->comm;
    const bool lo = cm.rank > 0, hi = cm.rank + 1 < cm.world;
    const size_t n = (size_t)L.hd * (size_t)L.g.plane, nt = 256 * CLS_W + 8;
    struct Buf {
        double* p = nullptr;
        ~Buf() { if (p) (void)hipFree(p); }
    } send, recv, tsend, trecv, dmap;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&send.p), 2 * n * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&recv.p), 2 * n * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&tsend.p), nt * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&trecv.p), 2 * nt * sizeof(double)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&dmap.p), 2 * 256 * sizeof(int)));
    const unsigned nb = (unsigned)std::min<size_t>(4096, (n + 255) / 256);
    int ok = cls_full(L) && L.ncls > 0 && L.sdia && L.wu == 4 && L.up[1] == 1 && L.up[2] == L.g.nx && (int64_t)L.up[3] == L.g.plane ? 1 : 0;
    std::vector<double> mine(nt, 0.0), theirs(2 * nt, 0.0);
    if (ok) {
        hipLaunchKernelGGL(bytes_to_doubles, dim3(nb), dim3(256), 0, c->stream, L.cls + L.cls_lead, send.p, (int64_t)n);
        hipLaunchKernelGGL(bytes_to_doubles, dim3(nb), dim3(256), 0, c->stream, L.cls + L.cls_lead + L.nloc - (int64_t)n, send.p + n, (int64_t)n);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(mine.data() + 8, L.ctab, 256 * CLS_W * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        mine[0] = (double)L.ncls;
    }
    // (a rank without classes still takes part in the exchanges: ncls = 0 tells the neighbours)
    HIP_TRY(hipMemcpyAsync(tsend.p, mine.data(), nt * sizeof(double), hipMemcpyHostToDevice, c->stream));
    if (!ok) HIP_TRY(hipMemsetAsync(send.p, 0, 2 * n * sizeof(double), c->stream));
    MG_TRY(exchange_raw(c, send.p, send.p + n, recv.p, recv.p + n, n, c->stream));
    MG_TRY(exchange_raw(c, tsend.p, tsend.p, trecv.p, trecv.p + nt, nt, c->stream));
    HIP_TRY(hipMemcpyAsync(theirs.data(), trecv.p, 2 * nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int> map(512, 0);
    int ncls = L.ncls;
    bool grown = false;
    for (int side = 0; side < 2 && ok; ++side) {
        if (!(side == 0 ? lo : hi)) continue;
        const double* t = theirs.data() + (size_t)side * nt;
        const int n_theirs = (int)t[0];
        if (n_theirs <= 0 || n_theirs > 256) { ok = 0; break; }
        for (int j = 0; j < n_theirs && ok; ++j) {
            const double* row = t + 8 + (size_t)j * CLS_W;
            int found = -1;
            for (int i = 0; i < ncls && found < 0; ++i)
                if (std::memcmp(mine.data() + 8 + (size_t)i * CLS_W, row, 7 * sizeof(double)) == 0) found = i;
            if (found < 0) {
                if (ncls >= 256) { ok = 0; break; }
                std::memcpy(mine.data() + 8 + (size_t)ncls * CLS_W, row, 7 * sizeof(double));
                mine[8 + (size_t)ncls * CLS_W + 7] = 0.0;
                found = ncls++;
                grown = true;
            }
            map[(size_t)side * 256 + j] = found;
        }
    }
    // every rank or none
    double flag = ok ? 0.0 : 1.0;
    HIP_TRY(hipMemcpyAsync(c->scalars + 7, &flag, sizeof(double), hipMemcpyHostToDevice, c->stream));
    MG_TRY(allreduce_sum(c, c->scalars + 7, 1));
    HIP_TRY(hipMemcpyAsync(&flag, c->scalars + 7, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (flag != 0.0) return 0;
    if (grown) {
        HIP_TRY(hipMemcpyAsync(L.ctab, mine.data() + 8, 256 * CLS_W * sizeof(double), hipMemcpyHostToDevice, c->stream));
        L.ncls = ncls;
    }
    HIP_TRY(hipMemcpyAsync(dmap.p, map.data(), 512 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int* dm = reinterpret_cast<const int*>(dmap.p);
    if (lo) hipLaunchKernelGGL(doubles_to_classes, dim3(nb), dim3(256), 0, c->stream, recv.p, dm, L.cls + L.cls_lead - (int64_t)n, (int64_t)n);
    if (hi) hipLaunchKernelGGL(doubles_to_classes, dim3(nb), dim3(256), 0, c->stream, recv.p + n, dm + 256, L.cls + L.cls_lead + L.nloc, (int64_t)n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    L.cls_halo = 1;
    return 0;
}

// K sweeps per pass (mg_jacobik3d.hip.h): seven-point levels with row classes that use the two-sweep pass -- whole levels,
// and slabs whose vectors have room for K halo planes ("halo_depth").
bool sweepsk_ok(const mg_context* c, const Level& L, bool ignore_size = false) {
    if (c->fuse_k < 3 || !L.cls || !c->fuse_classes) return false;
    if (L.cls_escape) {
        // (nothing else on such a level uses its classes: the march wherever it runs at all, whatever the level's size)
        if (L.esc_kmax < 3) return false;
        ignore_size = true;
    }
    const bool slab = !L.replicated && c->comm.active();
    if (slab && (L.hd < 2 || L.cls_halo < 0 || c->halo_planes != 1)) return false;
    if (!ignore_size && (slab ? min_slab_rows(L) < c->fuse_k_slab_min_rows : L.nloc < c->fuse_k_min_rows)) return false;
    // (the size test of the K-sweep pass is the one above: "fuse_min_rows" is the pair pass's)
    return fused_sweeps_ok(c, L, true) && L.g.nk >= 8;
}

// sweeps per pass on a whole level (with the 64 x 32 tiles of 16 waves five sweeps per pass measured best on 1025^3 and 513^3
// rows, four on 257^3, profiles/r03_ksweep_levels.txt; "fuse_k4_min_rows" caps smaller levels at three)
int sweepsk_max(const mg_context* c, const Level& L) {
    const int k = std::min(std::min(c->fuse_k, 5), L.nloc < c->fuse_k4_min_rows ? 3 : L.nloc < c->fuse_k_small_rows ? 5 : L.nloc < c->fuse_k5_min_rows ? 4 : 5);
    return L.cls_escape ? std::min(k, L.esc_kmax) : k;
}

// the most pool rows a tile of the escape variant of the march takes within K + 2 planes (jk3_escape_window)
template <int K>
int escape_window_k(mg_context* c, const Level& L, const unsigned char* cls, int64_t clead, unsigned* most) {
    constexpr int WI = 64 - 2 * K, HY = 16 * 2 - 2 * K + 2;
    JK3WindowArgs a{};
    a.cls = cls; a.clead = clead; a.P = L.g.plane; a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk;
    a.ntx = (a.nx + WI - 1) / WI; a.nty = (a.ny + HY - 1) / HY;
    a.most = reinterpret_cast<unsigned*>(c->partials);
    HIP_TRY(hipMemsetAsync(a.most, 0, sizeof(unsigned), c->stream));
    hipLaunchKernelGGL((jk3_escape_window<K, 16, 2, 1>), dim3((unsigned)(a.ntx * a.nty)), dim3(256), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(most, a.most, sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (std::getenv("MG_DEBUG_STORAGE"))
        std::fprintf(stderr, "[mg] escape rows: at most %u in %d planes of a tile (%d sweeps per pass, pool of %d)\n", *most, K + 2, K, jk3_pool(K));
    return 0;
}

int escape_kmax(mg_context* c, const Level& L, const unsigned char* cls, int64_t clead, int* kmax) {
    *kmax = 0;
    if (L.wu != 4 || L.g.nk < 8 || L.g.nx < 32 || L.g.ny < 32) return 0;
    unsigned most = 0;
    MG_TRY(escape_window_k<5>(c, L, cls, clead, &most));
    if (most <= (unsigned)jk3_pool(5)) { *kmax = 5; return 0; }
    MG_TRY(escape_window_k<4>(c, L, cls, clead, &most));
    if (most <= (unsigned)jk3_pool(4)) { *kmax = 4; return 0; }
    MG_TRY(escape_window_k<3>(c, L, cls, clead, &most));
    if (most <= (unsigned)jk3_pool(3)) *kmax = 3;
    return 0;
}

// plane ranges of one launch of the march: [za0, za1) and then [zb0, zb1)
struct JK3Range { int za0, za1, zb0, zb1; };

template <int K, int NW, int LPW, int M, bool DPP, int PF = 1, int WPE = (NW == 12 ? 3 : 2), int TR = 256, bool ESC = false>
int launch_jacobikc_t(mg_context* c, JK3Args a, bool finest, const JK3Range& zr, int seglen) {
    constexpr int EX = 64 * M, EY = NW * LPW, WI = EX - 2 * K, HY = EY - 2 * K + 2;
    a.ntx = (a.nx + WI - 1) / WI;
    a.nty = (a.ny + HY - 1) / HY;
    const int64_t ntile = (int64_t)a.ntx * a.nty;
    constexpr size_t lds = jk3_lds_bytes<K, NW, LPW, M, TR, ESC>();
    static_assert(lds <= 160 * 1024, "one CU's LDS");
    a.za0 = zr.za0; a.za1 = zr.za1; a.zb0 = zr.zb0; a.zb1 = zr.zb1;
    const int planes = std::max(0, zr.za1 - zr.za0) + std::max(0, zr.zb1 - zr.zb0);
    if (planes <= 0) return 0;
    a.ta = (int)ntile; a.seglen_b = 1;
    int64_t items = 0;
    if (seglen <= 0) {
        // Plane segments.  The items run in rounds of the resident workgroups (by LDS and by waves per SIMD), each paying 2K
        // steps of warm-up (which do about two thirds of a step's work).  A last round that only a few tiles are left for
        // would keep most CUs idle for a whole segment: those tiles -- fewer than there are CUs -- are cut into as many shorter
        // segments as fill the round once, and come last ("fuse_k_tail" 0: every tile alike).
        int64_t cus = std::max(1, c->prop.multiProcessorCount) * (int64_t)std::max<size_t>(1, std::min<size_t>(160 * 1024 / lds, (size_t)(4 * WPE / NW)));
        if (c->fuse_k_tail > 1) cus = c->fuse_k_tail;       // (tests: a tail on grids of a few tiles)
        const double warm = 1.4 * K;
        int best = 1, best_ta = (int)ntile, best_m = 1;
        double best_cost = 1e300;
        const bool tail = c->fuse_k_tail && zr.zb1 <= zr.zb0;
        for (int n = 1; n <= std::max(1, planes / 8); ++n) {
            const int len = (planes + n - 1) / n;
            const int64_t full = ntile * n / cus;
            int64_t ta = tail ? std::min<int64_t>(ntile, full * cus / n) : ntile;
            if (ta <= 0 || (ntile - ta) * 2 > cus) ta = ntile;           // (a tail of more than half a round is a round)
            const int64_t tb = ntile - ta;
            const int m = tb > 0 ? (int)std::max<int64_t>(1, std::min<int64_t>(cus / tb, planes / 4)) : 1;
            const double cost = (double)((ta * n + cus - 1) / cus) * (len + warm) + (tb > 0 ? (planes + m - 1) / m + warm : 0.0);
            if (cost < best_cost) { best_cost = cost; best = n; best_ta = (int)ta; best_m = m; }
        }
        if (c->fuse_k_segments > 0) { best = std::min(c->fuse_k_segments, planes); best_ta = (int)ntile; }
        seglen = (planes + best - 1) / best;
        a.ta = best_ta;
        a.seglen_b = best_ta < ntile ? (planes + best_m - 1) / best_m : 1;
    }
    a.seglen = seglen;
    {
        const int nseg = (std::max(0, zr.za1 - zr.za0) + seglen - 1) / seglen + (std::max(0, zr.zb1 - zr.zb0) + seglen - 1) / seglen;
        items = (int64_t)a.ta * nseg;
        if (a.ta < ntile) items += (ntile - a.ta) * (int64_t)((planes + a.seglen_b - 1) / a.seglen_b);
    }
    if (items >= ((int64_t)1 << 31) - 4096) return fail("too many tiles");
    a.nitems = (unsigned)items;
    a.xcd_chunk = (unsigned)c->fuse_xcd_chunk;
    const int64_t group = 8 * (int64_t)a.xcd_chunk;
    const unsigned grid = (unsigned)(((items + group - 1) / group) * group);
    if (a.ncls > TR) return fail("more row classes than this tile shape keeps in LDS");
    void (*kern)(JK3Args) = finest ? sdia_jacobikc_finest<K, NW, LPW, M, DPP, PF, WPE, TR> : sdia_jacobikc<K, NW, LPW, M, DPP, PF, WPE, TR>;
    if constexpr (ESC) kern = sdia_jacobikc_escape<K, NW, LPW, M, DPP, PF, WPE, TR>;
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * WAVE), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int K, int PF>
int launch_jacobikc_kp(mg_context* c, const JK3Args& a, bool finest, const JK3Range& zr, int seglen) {
    if (!c->fuse_k_dpp) return launch_jacobikc_t<K, 12, 2, 2, false, PF>(c, a, finest, zr, seglen);     // (experiment: -1 / +1 neighbours through LDS)
    // shapes 3..5: 64 x 24 tiles, two workgroups per CU (class table of 64 rows); levels with more classes take shape 1
    int shape = c->fuse_k_shape >= 3 && a.ncls > 64 ? 1 : c->fuse_k_shape;
    // planes with fewer 64 x 48 tiles than the GPU has CUs (the 513^2 and 257^2 planes of a slab's coarser levels): half
    // as tall tiles, two workgroups per CU
    if (shape == 1 && a.ncls <= 64 && K <= 4 && c->fuse_k_small_tiles) {
        constexpr int WI = 64 - 2 * K, HY = 48 - 2 * K + 2;
        if ((int64_t)((a.nx + WI - 1) / WI) * ((a.ny + HY - 1) / HY) < std::max(1, c->prop.multiProcessorCount)) shape = 4;
    }
    switch (shape) {
        case 1: return launch_jacobikc_t<K, 12, 4, 1, true, PF>(c, a, finest, zr, seglen);
        case 2: return launch_jacobikc_t<K, 8, 3, 2, true, PF>(c, a, finest, zr, seglen);
        case 3: return launch_jacobikc_t<K, 6, 4, 1, true, PF, 3, 64>(c, a, finest, zr, seglen);
        case 4: return launch_jacobikc_t<K, 8, 3, 1, true, PF, 4, 64>(c, a, finest, zr, seglen);
        case 5: if constexpr (K <= 4) return launch_jacobikc_t<K, 4, 6, 1, true, PF, 2, 64>(c, a, finest, zr, seglen);
        case 6: if constexpr (K <= 4 && PF == 1) return launch_jacobikc_t<K, 16, 3, 1, true, 1, 4>(c, a, finest, zr, seglen);      // 64 x 48 by 16 waves
        case 7: return launch_jacobikc_t<K, 16, 2, 1, true, PF, 4>(c, a, finest, zr, seglen);                                      // 64 x 32 by 16 waves
        default: return launch_jacobikc_t<K, 12, 2, 2, true, PF>(c, a, finest, zr, seglen);
    }
}

template <int K>
int launch_jacobikc_k(mg_context* c, const JK3Args& a, bool finest, const JK3Range& zr, int seglen) {
    // levels with rows of class CLS_ESCAPE: the variant that fetches them, one tile shape (the one escape_window() checked)
    if (a.escape) {
        if constexpr (K >= 3) return launch_jacobikc_t<K, 16, 2, 1, true, 1, 4, 256, true>(c, a, finest, zr, seglen);
        else return fail("levels with escape rows take three to five sweeps per pass");
    }
    // a second plane of x staged in registers ("fuse_k_pf" 2) fits the register budget with three sweeps only
    if constexpr (K == 3 || K == 4) {
        if (c->fuse_k_pf == 2 && (K == 3 || c->fuse_k_shape == 7)) return launch_jacobikc_kp<K, 2>(c, a, finest, zr, seglen);
    }
    // (two sweeps per pass: slabs only, where a pass also saves an exchange; one tile shape)
    if constexpr (K == 2) return launch_jacobikc_t<K, 12, 4, 1, true, 1>(c, a, finest, zr, seglen);
    else return launch_jacobikc_kp<K, 1>(c, a, finest, zr, seglen);
}

// out = K Jacobi sweeps applied to x (2 <= K <= 5) on the owned planes [zr) of the level (null: all of them); on slabs the
// level's K halo planes of x and K - 1 of f must be valid
int launch_jacobikc(mg_context* c, const Level& L, int K, const double* x_rows, const double* f_rows, double* out_rows,
                    const JK3Range* zr = nullptr, int seglen = 0) {
    JK3Args a{};
    a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.cls = L.cls; a.ctab = L.ctab; a.clead = L.cls_lead; a.ncls = L.ncls; a.cmain = L.cmain;
    for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
    a.P = L.g.plane; a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk; a.omega = c->omega;
    const bool dist = !L.replicated && c->comm.active();
    a.plo = dist && c->comm.rank > 0 ? K : 0;
    a.phi = dist && c->comm.rank + 1 < c->comm.world ? K : 0;
    if (dist && (L.hd < K || L.cls_halo != 1)) return fail("the level's halos are not prepared for that many sweeps per pass");
    if (L.cls_escape && K > L.esc_kmax) return fail("too many escape rows per tile for that many sweeps per pass");
    a.force_form = c->timing_force_form;
    a.nt_store = c->fuse_k_nt_store;
    a.escape = L.cls_escape ? 1 : 0; a.dvals = L.dvals; a.mlead = L.mlead; a.sshift = L.R == 1 ? 6 : (L.R == 2 ? 7 : 8);
    const JK3Range whole{0, L.g.nk, 0, 0};
    const bool finest = c->nlev > 1 && &L == &c->L[c->nlev - 1];
    switch (K) {
        case 2: return launch_jacobikc_k<2>(c, a, finest, zr ? *zr : whole, seglen);
        case 3: return launch_jacobikc_k<3>(c, a, finest, zr ? *zr : whole, seglen);
        case 4: return launch_jacobikc_k<4>(c, a, finest, zr ? *zr : whole, seglen);
        case 5: return launch_jacobikc_k<5>(c, a, finest, zr ? *zr : whole, seglen);
        default: return fail("sweeps per pass must be in 2..5");
    }
}

// Is lattice_color(COLOR_LATTICE9) a valid Gauss-Seidel colouring of the level's matrix?  (structure of the stored
// non-zeros, in whatever format the level has)
int check_coloring(mg_context* c, Level& L) {
    EllArgs a{};
    a.vals = L.vals; a.cols = L.cols; a.codes = L.codes; a.offsets = L.offsets; a.W = L.W;
    a.nloc = L.nloc; a.lead = L.g.lead; a.dinv = L.dinv; a.color_kind = COLOR_LATTICE9; a.grow0 = L.row0; a.gnx = L.g.nx; a.gny = L.g.ny;
    int* d_flag = reinterpret_cast<int*>(c->partials);
    HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(int), c->stream));
    const dim3 grid(blocks_for(L.nloc, 256)), blk(256);
    if (L.sdia) {
        a.vals = L.dvals; a.mlead = L.mlead;
        for (int t = 0; t < 8; ++t) a.up[t] = L.up[t];
        switch (L.R) {
            case 1: hipLaunchKernelGGL(sdia_check_coloring<1>, grid, blk, 0, c->stream, a, L.wu, d_flag); break;
            case 2: hipLaunchKernelGGL(sdia_check_coloring<2>, grid, blk, 0, c->stream, a, L.wu, d_flag); break;
            default: hipLaunchKernelGGL(sdia_check_coloring<4>, grid, blk, 0, c->stream, a, L.wu, d_flag); break;
        }
    } else {
        switch (L.R) {
            case 1: hipLaunchKernelGGL(ell_check_coloring<1>, grid, blk, 0, c->stream, a, L.nslices, d_flag); break;
            case 2: hipLaunchKernelGGL(ell_check_coloring<2>, grid, blk, 0, c->stream, a, L.nslices, d_flag); break;
            default: hipLaunchKernelGGL(ell_check_coloring<4>, grid, blk, 0, c->stream, a, L.nslices, d_flag); break;
        }
    }
    HIP_TRY(hipGetLastError());
    int flag = 1;
    HIP_TRY(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    L.mc_ok = flag ? 0 : 1;
    return 0;
}

// K sweeps per launch on 2-D levels (mg_jacobi2.hip.h, sdia_jacobik2d): whole, undistributed five-point levels
// with row classes whose stored diagonals are exactly {0, +1, +nx}.
bool sweeps2d_ok(const mg_context* c, const Level& L) {
    if (!c->fuse_2d || !L.sdia || L.wu != 3 || !cls_full(L) || !c->fuse_classes || L.flat || !L.replicated) return false;
    return L.g.ny == 1 && L.up[1] == 1 && L.up[2] == L.g.nx && L.g.nx >= 8 && L.g.nz >= 8;
}

// Lines per region ("fuse_2d_lines"; 0 = chosen here).  Regions of 32 lines (four cells per thread, 56 registers, 80 KB of LDS)
// run two workgroups per CU, which hides the barriers of one behind the other: 0.039 ms per five sweeps on 2049^2 rows against
// 0.049 with 64 lines although a sweep keeps 22 of 32 lines instead of 54 of 64 -- BASELINE config 2 451 -> 592 cycles/s
// (profiles/r03_2d_lines.txt).  A launch on a small level takes as long as ONE workgroup does, about 2 us plus 1.3 us per cell
// a thread owns: 16 lines (two cells) while all tiles still run at once.
template <int K, int H>
int launch_jacobik_th(mg_context* c, const JKArgs& a) {
    constexpr size_t lds = jk_lds_bytes<H>();
    void (*const kern)(JKArgs) = sdia_jacobik2d<K, H>;
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern), lds));
    JKArgs b = a;
    b.ntx = (a.nx + JK_W - 2 * K - 1) / (JK_W - 2 * K);
    b.nty = (a.nlines + H - 2 * K - 1) / (H - 2 * K);
    hipLaunchKernelGGL(kern, dim3((unsigned)(b.ntx * b.nty)), dim3(1024), lds, c->stream, b);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <int K>
int launch_jacobik_t(mg_context* c, const JKArgs& a) {
    const int64_t cus = std::max(1, c->prop.multiProcessorCount);
    const int64_t ntx = (a.nx + JK_W - 2 * K - 1) / (JK_W - 2 * K);
    int best = c->fuse_2d_lines;
    if (!best) best = K <= 5 && ntx * ((a.nlines + 16 - 2 * K - 1) / (16 - 2 * K)) <= 2 * cus ? 16 : 32;
    switch (best) {
        case 16: return launch_jacobik_th<K, 16>(c, a);
        case 32: return launch_jacobik_th<K, 32>(c, a);
        default: return launch_jacobik_th<K, 64>(c, a);
    }
}

// out = K Jacobi sweeps applied to x (2 <= K <= 5)
int launch_jacobik(mg_context* c, const Level& L, int K, const double* x_rows, const double* f_rows, double* out_rows) {
    JKArgs a{};
    a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.cls = L.cls + L.cls_lead; a.ctab = L.ctab; a.ncls = L.ncls; a.cmain = L.cmain;
    for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
    a.n = L.nloc; a.nx = L.g.nx; a.nlines = L.g.nz; a.omega = c->omega;
    switch (K) {
        case 2: return launch_jacobik_t<2>(c, a);
        case 3: return launch_jacobik_t<3>(c, a);
        case 4: return launch_jacobik_t<4>(c, a);
        case 5: return launch_jacobik_t<5>(c, a);
        default: return fail("sweeps per launch must be in 2..5");
    }
}

// One sweep as a plane march (sdia_sweep1c): whole, undistributed 3-D seven-point levels with row classes, large enough
// for the march to pay ("march_min_rows").
bool sweep1c_ok(const mg_context* c, const Level& L) {
    if (!c->march_sweeps || !cls_full(L) || !c->class_sweeps || !L.sdia || L.wu != 4 || L.flat || !L.replicated) return false;
    if (L.g.nx < 32 || L.g.ny < 32 || L.g.nk < 8) return false;
    if (L.up[1] != 1 || L.up[2] != L.g.nx || (int64_t)L.up[3] != L.g.plane) return false;
    return L.nloc >= c->march_min_rows;
}

template <int NW, int LPW>
int launch_sweep1c_t(mg_context* c, J2Args& a, int mode) {
    constexpr int EY = NW * LPW;
    a.ntx = (a.nx + J2_EX - 3) / (J2_EX - 2);
    a.nty = (a.ny + EY - 1) / EY;
    const int64_t ntile = (int64_t)a.ntx * a.nty;
    // plane segments: enough work items for a few rounds of the CUs, each paying ~2.5 plane-times of warm-up
    const int64_t cus = std::max(1, c->prop.multiProcessorCount);
    int best = 1;
    double best_cost = 1e300;
    for (int n = 1; n <= std::max(1, a.nz / 16); ++n) {
        const double cost = (double)((ntile * n + cus - 1) / cus) * ((a.nz + n - 1) / n + 2.5);
        if (cost < best_cost) { best_cost = cost; best = n; }
    }
    if (c->fuse_segments > 0) best = std::min(c->fuse_segments, a.nz);
    a.seglen = (a.nz + best - 1) / best;
    const int nseg = (a.nz + a.seglen - 1) / a.seglen;
    const int64_t items = ntile * nseg;
    if (items >= ((int64_t)1 << 31) - 4096) return fail("too many tiles");
    a.nitems = (unsigned)items;
    a.xcd_chunk = (unsigned)c->fuse_xcd_chunk;
    const int64_t group = 8 * (int64_t)a.xcd_chunk;
    const unsigned grid = (unsigned)(((items + group - 1) / group) * group);
    constexpr size_t lds = j1c_lds_bytes<NW, LPW>();
    void (*kern)(J2Args) = mode == MODE_RESIDUAL ? sdia_sweep1c<NW, LPW, MODE_RESIDUAL>
                         : mode == MODE_GS ? sdia_sweep1c<NW, LPW, MODE_GS> : sdia_sweep1c<NW, LPW, MODE_JACOBI>;
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern), lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * WAVE), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int launch_sweep1c(mg_context* c, const Level& L, int mode, const double* x_rows, const double* f_rows, double* out_rows,
                   int color) {
    J2Args a{};
    a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.nloc = L.nloc; a.P = L.g.plane; a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk; a.omega = c->omega;
    a.cls = L.cls; a.ctab = L.ctab; a.clead = L.cls_lead; a.ncls = L.ncls; a.cmain = L.cmain;
    for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
    a.color = color; a.color_kind = c->smoother == MG_SMOOTH_MCGS ? COLOR_LATTICE9 : COLOR_PARITY; a.grow0 = L.row0;
    if (c->march_shape == 1) return launch_sweep1c_t<16, 2>(c, a, mode);
    return launch_sweep1c_t<12, 2>(c, a, mode);
}

// All sweeps of a small level in one launch (mg_jacobi2.hip.h, sdia_jacobi_small): whole five- / seven-point levels with
// row classes that fit one CU's LDS.
// K sweeps per launch on blocks resident on the CU (mg_jacobiblk.hip.h): whole seven-point levels with row classes, sizes
// below the plane marches'
bool block_sweeps_ok(const mg_context* c, const Level& L) {
    if (!c->fuse_block || !L.sdia || L.wu != 4 || !cls_full(L) || !c->fuse_classes || L.flat || !L.replicated) return false;
    if (!L.grid_decoupled || L.ncls > 128 || L.g.nx < 8 || L.g.ny < 8 || L.g.nk < 8 || L.nloc >= ((int64_t)1 << 28)) return false;
    return c->fuse_block >= 2 || (L.nloc >= c->fuse_block_min_rows && L.nloc < c->fuse_block_max_rows);
}

// sweeps per launch and planes per block: the launch runs in rounds of one workgroup per CU, a workgroup loads EZ planes
// of 32 x 32 cells (about 0.33 us per plane) and relaxes them K times (0.13 us per plane and sweep), of which it keeps
// (32 - 2K)^2 (EZ - 2K) cells
struct JBPlan { int K, EZ; };

JBPlan block_plan(const mg_context* c, const Level& L) {
    const int64_t cus = std::max(1, c->prop.multiProcessorCount);
    const int EZ = c->fuse_block_ez ? c->fuse_block_ez : 11;
    auto blocks = [&](int K) {
        const int ow = JB_E - 2 * K, oz = EZ - 2 * K;
        return (int64_t)((L.g.nx + ow - 1) / ow) * ((L.g.ny + ow - 1) / ow) * ((L.g.nk + oz - 1) / oz);
    };
    // three sweeps per launch; four where the blocks then still all run at once (65^3 rows: 3.4 against 3.9 us per sweep)
    const int K = c->fuse_block_k ? c->fuse_block_k : blocks(4) <= cus ? 4 : 3;
    return JBPlan{K, EZ};
}

template <int K, int EZ>
int launch_jacobi_block_t(mg_context* c, const Level& L, JBArgs a) {
    constexpr int ow = JB_E - 2 * K, oz = EZ - 2 * K;
    a.nbx = (a.nx + ow - 1) / ow;
    a.nby = (a.ny + ow - 1) / ow;
    const int64_t nb = (int64_t)a.nbx * a.nby * ((a.nz + oz - 1) / oz);
    if (nb >= ((int64_t)1 << 31)) return fail("too many blocks");
    const size_t lds = jb_lds_bytes<EZ>(a.ncls);
    if (lds > (size_t)160 * 1024) return fail("block pass: class table too large");
    void (*const kern)(JBArgs) = sdia_jacobi_block<K, EZ>;
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern), (size_t)160 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(1024), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// out = K Jacobi sweeps applied to x (2 <= K <= 4), blocks of EZ planes (11 or 19)
int launch_jacobi_block(mg_context* c, const Level& L, int K, int EZ, const double* x_rows, const double* f_rows, double* out_rows) {
    JBArgs a{};
    a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.cls = L.cls + L.cls_lead; a.ctab = L.ctab; a.ncls = L.ncls; a.cmain = L.cmain;
    for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
    a.omega = c->omega; a.nx = L.g.nx; a.ny = L.g.ny; a.nz = L.g.nk; a.P = L.g.plane;
    switch (K * 100 + EZ) {
        case 211: return launch_jacobi_block_t<2, 11>(c, L, a);
        case 219: return launch_jacobi_block_t<2, 19>(c, L, a);
        case 311: return launch_jacobi_block_t<3, 11>(c, L, a);
        case 319: return launch_jacobi_block_t<3, 19>(c, L, a);
        case 411: return launch_jacobi_block_t<4, 11>(c, L, a);
        case 419: return launch_jacobi_block_t<4, 19>(c, L, a);
        default: return fail("block pass: 2..4 sweeps per launch on blocks of 11 or 19 planes");
    }
}

bool small_level_ok(const mg_context* c, const Level& L) {
    if (!c->fuse_small || !L.sdia || !cls_full(L) || !c->fuse_classes || L.flat || !L.replicated) return false;
    if (L.wu != 3 && L.wu != 4) return false;
    if (L.up[1] != 1 || L.up[2] <= 1 || (L.wu == 4 && L.up[3] <= L.up[2])) return false;
    const int pad = L.wu == 4 ? L.up[3] : L.up[2];
    // (2-D levels of more than "fuse_small_2d_rows" rows are faster through the K-sweep 2-D kernel -- ten launches of a dozen
    //  small workgroups on as many CUs instead of one launch of one workgroup: 65^2 rows 55 against 72 us per 50 sweeps)
    if (L.wu == 3 && L.nloc > c->fuse_small_2d_rows && sweeps2d_ok(c, L)) return false;
    return L.nloc <= 16384 && js_lds_bytes((int)L.nloc, pad) <= (size_t)150 * 1024;
}

int launch_jacobi_small(mg_context* c, const Level& L, int nw, const double* x_rows, const double* f_rows, double* out_rows) {
    JSArgs a{};
    a.x = x_rows; a.f = f_rows; a.out = out_rows;
    a.cls = L.cls + L.cls_lead; a.ctab = L.ctab; a.ncls = L.ncls; a.cmain = L.cmain;
    for (int t = 0; t < 8; ++t) a.cm[t] = L.cm[t];
    a.n = (int)L.nloc; a.nw = nw; a.up1 = L.up[1]; a.up2 = L.up[2]; a.up3 = L.wu == 4 ? L.up[3] : 0; a.omega = c->omega;
    const size_t lds = js_lds_bytes(a.n, L.wu == 4 ? a.up3 : a.up2);
    // rows per thread: the smallest instance that covers the level
    const int need = (a.n + 1023) / 1024;
    void (*kern)(JSArgs) = nullptr;
    auto pick = [&](auto wu_tag) {
        constexpr int WU = decltype(wu_tag)::value;
        if (need <= 1) kern = sdia_jacobi_small<WU, 1>;
        else if (need <= 2) kern = sdia_jacobi_small<WU, 2>;
        else if (need <= 3) kern = sdia_jacobi_small<WU, 3>;
        else if (need <= 4) kern = sdia_jacobi_small<WU, 4>;
        else if (need <= 5) kern = sdia_jacobi_small<WU, 5>;
        else if (need <= 6) kern = sdia_jacobi_small<WU, 6>;
        else if (need <= 8) kern = sdia_jacobi_small<WU, 8>;
        else if (need <= 12) kern = sdia_jacobi_small<WU, 12>;
        else kern = sdia_jacobi_small<WU, 16>;
    };
    if (L.wu == 4) pick(std::integral_constant<int, 4>{});
    else pick(std::integral_constant<int, 3>{});
    MG_TRY(allow_large_lds(c, reinterpret_cast<const void*>(kern), (size_t)150 * 1024));
    hipLaunchKernelGGL(kern, dim3(1), dim3(1024), lds, c->stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

// a slab level on which calls of enough sweeps take the K-sweep march with K-plane exchanges (the same answer on every rank)
bool slab_ksweep_level(const mg_context* c, const Level& L) {
    return !L.replicated && c->comm.active() && c->fuse_k >= 3 && c->fuse_sweeps && c->fuse_classes && c->use_classes && c->use_sdia &&
           L.hd >= 2 && c->halo_planes == 1 && !L.flat && L.g.nx >= 32 && L.g.ny >= 32 &&
           min_slab_rows(L) >= std::max<int64_t>(8 * L.g.plane, c->fuse_k_slab_min_rows);
}

// nw Jacobi sweeps; v halos must be valid on entry and are valid on exit.
int smooth(mg_context* c, int level, int nw) {
    Level& L = c->L[level];
    if (c->smoother == MG_SMOOTH_RBGS) {
        // red-black Gauss-Seidel: two in-place half sweeps per sweep, halos refreshed after each colour
        if (!L.rb_ok)
            return fail("red-black ordering is not a valid two-colouring of level " + std::to_string(level) +
                        " (needs a pruned grid matrix with an odd number of nodes per axis)");
        for (int s = 0; s < nw; ++s)
            for (int color = 0; color < 2; ++color) {
                MG_TRY(launch_ell(c, L, MODE_GS, false, L.v.base, L.f.rows, L.v.rows, nullptr, nullptr, nullptr, 0, -1, color));
                MG_TRY(exchange_halo(c, L, L.v));
            }
        return 0;
    }
    if (c->smoother == MG_SMOOTH_MCGS) {
        // nine-colour Gauss-Seidel for P2 rows (lattice_color): one in-place launch per colour, in ascending order
        if (L.flat) return fail("multi-colour Gauss-Seidel needs a grid level");
        if (L.mc_ok < 0) MG_TRY(check_coloring(c, L));
        if (!L.mc_ok)
            return fail("the nine lattice colours are not a valid colouring of level " + std::to_string(level) +
                        " (needs a pruned P1 / P2 grid matrix)");
        // whole 3-D lattice levels: two colours per launch, out of place (v -> v2, then the two swap)
        const bool pairs = c->lattice_gs2 && c->class_sweeps && lat_march_ok(c, L) && (L.replicated || !c->comm.active());
        for (int s = 0; s < nw; ++s) {
            if (pairs) {
                for (int color = 0; color < 9; color += 2)
                    MG_TRY(launch_lat_gs2(c, L, color, color + 1 < 9 ? color + 1 : -1, L.v.rows, L.v2.rows));
                std::swap(L.v, L.v2);
                continue;
            }
            for (int color = 0; color < 9; ++color) {
                if (c->dim == 2 && (color & 2)) continue;            // (nx, 1, nz) storage: j is always 0
                MG_TRY(launch_ell(c, L, MODE_GS, false, L.v.base, L.f.rows, L.v.rows, nullptr, nullptr, nullptr, 0, -1, color));
                MG_TRY(exchange_halo(c, L, L.v));
            }
        }
        return 0;
    }
    const bool dist = !L.replicated && c->comm.active();
    if (!dist && nw >= 2 && small_level_ok(c, L)) {
        MG_TRY(launch_jacobi_small(c, L, nw, L.v.rows, L.f.rows, L.v2.rows));
        std::swap(L.v, L.v2);
        return 0;
    }
    if (!dist && nw >= 2 && block_sweeps_ok(c, L)) {
        // middle 3-D levels: K sweeps per launch on blocks resident on the CU, the rest (at most one) as a single sweep
        const JBPlan plan = block_plan(c, L);
        int left = nw;
        while (left >= 2) {
            int k = std::min(plan.K, left);
            if (left - k == 1 && k > 2) --k;                // 4 = 2 + 2 rather than 3 + 1
            MG_TRY(launch_jacobi_block(c, L, k, plan.EZ, L.v.rows, L.f.rows, L.v2.rows));
            std::swap(L.v, L.v2);
            left -= k;
        }
        nw = left;
    }
    if (!dist && sweeps2d_ok(c, L)) {
        // 2-D levels: up to fuse_2d_k sweeps per launch, the rest (at most one) as a single sweep
        int left = nw;
        while (left >= 2) {
            int k = std::min(c->fuse_2d_k, left);
            if (left - k == 1 && k > 2) --k;                // 6 = 3 + 3 rather than 5 + 1
            MG_TRY(launch_jacobik(c, L, k, L.v.rows, L.f.rows, L.v2.rows));
            std::swap(L.v, L.v2);
            left -= k;
        }
        nw = left;
    }
    // slices that hold rows of the first / last owned plane: their results are what the neighbours need
    const int64_t S = (int64_t)WAVE * L.R;
    const int64_t hplanes = (int64_t)c->halo_planes * L.g.plane;
    const int64_t lo_end = std::min(L.nslices, (hplanes + S - 1) / S);
    const int64_t hi_begin = std::max<int64_t>(lo_end, (L.nloc - hplanes) / S);
    // below a few million rows a sweep is shorter than the extra launches and event hops of the overlapped
    // form: exchange in-stream there
    // (taken alike on every rank: every slab of a distributed level has at least 2^level >= 2 planes)
    const bool two_planes = min_slab_rows(L) >= 2 * hplanes;
    const bool overlap = dist && c->overlap && two_planes && hi_begin > lo_end && c->comm_stream &&
                         min_slab_rows(L) >= c->overlap_min_rows;
    // (the paired pass on slabs is written for one halo plane)
    const bool fused = fused_sweeps_ok(c, L) && (!dist || (c->halo_planes == 1 && two_planes && hi_begin > lo_end));
    J2Plan plan{};
    if (fused && nw > 1) {
        if (dist) MG_TRY(vec_alloc(c, L, &L.sw));
        plan = jacobi2_plan(c, L, dist, lo_end * S + L.g.plane + L.g.nx + 2);
    }
    // Slabs with room for K halo planes ("halo_depth"): K sweeps per pass AND per exchange -- K planes of the iterate travel
    // once per pass (the same volume as one plane per sweep) and the march relaxes the neighbours' K - 1 planes next to the
    // slab itself, so there is no boundary chain: two launches and one grouped send / receive per K sweeps.  With the
    // overlap on, the planes the neighbours wait for are relaxed first (one launch), and travel on the communication
    // stream while a second launch relaxes the rest.  (Everything that decides is the same on every rank.)
    if (dist && nw >= c->fuse_k_slab_min_sweeps && slab_ksweep_level(c, L)) {
        if (L.cls_halo == 0) MG_TRY(ensure_class_halos(c, L));
        if (L.cls_halo == 1) {
            const int kmax = std::min(std::min(c->fuse_k, 5), L.hd);
            auto next_k = [&](int left) -> int {
                if (left < 2) return 0;
                const int k = left >= kmax + 2 || left == kmax ? kmax : left == kmax + 1 ? kmax - 1 : left;
                return std::max(2, std::min(k, kmax));
            };
            const bool lo = c->comm.rank > 0, hi = c->comm.rank + 1 < c->comm.world;
            const int nk = L.g.nk;
            int left = nw, k = next_k(left);
            MG_TRY(exchange_halo(c, L, L.f, nullptr, std::max(1, kmax - 1)));
            MG_TRY(exchange_halo(c, L, L.v, nullptr, k));
            while (k) {
                const int kn = next_k(left - k);                    // the pass after this one
                const int e = kn ? kn : c->halo_planes;             // planes of the new iterate the neighbours need next
                const bool split = c->overlap && c->comm_stream && min_slab_rows(L) >= c->overlap_min_rows &&
                                   min_slab_rows(L) >= (int64_t)(4 * e + 8) * L.g.plane;
                if (split) {
                    const int lob = lo ? e : 0, hib = hi ? e : 0;
                    const JK3Range edge{0, lob, nk - hib, nk}, rest{lob, nk - hib, 0, 0};
                    MG_TRY(launch_jacobikc(c, L, k, L.v.rows, L.f.rows, L.v2.rows, &edge, e));
                    HIP_TRY(hipEventRecord(c->ev_boundary, c->stream));
                    HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
                    MG_TRY(exchange_halo(c, L, L.v2, c->comm_stream, e));
                    HIP_TRY(hipEventRecord(c->ev_halo, c->comm_stream));
                    MG_TRY(launch_jacobikc(c, L, k, L.v.rows, L.f.rows, L.v2.rows, &rest));
                    HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
                } else {
                    MG_TRY(launch_jacobikc(c, L, k, L.v.rows, L.f.rows, L.v2.rows));
                    MG_TRY(exchange_halo(c, L, L.v2, nullptr, e));
                }
                std::swap(L.v, L.v2);
                left -= k;
                k = kn;
            }
            nw = left;
        }
    }
    if (!dist && nw >= 3 && sweepsk_ok(c, L)) {
        // whole levels: K sweeps per pass while that leaves no single sweep over (50 = 12 x 4 + 2, 7 = 4 + 3, 5 = 3 + 2)
        const int kmax = sweepsk_max(c, L);
        int left = nw;
        while (left >= 3) {
            int k = left >= kmax + 2 || left == kmax ? kmax : left == kmax + 1 ? kmax - 1 : left;
            if (k < 3) break;
            MG_TRY(launch_jacobikc(c, L, k, L.v.rows, L.f.rows, L.v2.rows));
            std::swap(L.v, L.v2);
            left -= k;
        }
        nw = left;
    }
    for (int s = 0; s < nw; ++s) {
        if (fused && s + 1 < nw) {
            if (!dist) {
                MG_TRY(launch_jacobi2(c, L, plan, 0, 1, plan.nseg, L.v.rows, L.f.rows, L.v2.rows));
                std::swap(L.v, L.v2);
                ++s;
                continue;
            }
            // Slabs.  The rows of the slices that hold the first / last owned plane need the neighbours' once-relaxed
            // planes for their second sweep: the pass leaves them out and parks v1 around them in `sw`, whose halos
            // are then exchanged like any iterate's, and the one-sweep kernel finishes those slices.
            const bool lo = c->comm.rank > 0, hi = c->comm.rank + 1 < c->comm.world;
            const int64_t st_lo = lo ? lo_end * S : 0, st_hi = hi ? hi_begin * S : INT64_MAX;
            auto boundary_chain = [&](hipStream_t stream) -> int {
                MG_TRY(exchange_halo(c, L, L.sw, stream));
                // (the first and the last slices in one launch where the slab has both neighbours)
                if (lo && hi)
                    MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.sw.base, L.f.rows, L.v2.rows, nullptr, nullptr, nullptr, 0,
                                      lo_end + L.nslices - hi_begin, 0, lo_end, hi_begin - lo_end));
                else if (lo)
                    MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.sw.base, L.f.rows, L.v2.rows, nullptr, nullptr, nullptr, 0, lo_end));
                else if (hi)
                    MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.sw.base, L.f.rows, L.v2.rows, nullptr, nullptr, nullptr, hi_begin,
                                      L.nslices - hi_begin));
                MG_TRY(exchange_halo(c, L, L.v2, stream));
                return 0;
            };
            // slices whose once-relaxed values the finishing sweep of the first / last slices reads
            const int64_t reach = L.g.plane + L.g.nx + 2;
            const int64_t lo2 = std::min(L.nslices, (lo_end * S + reach + S - 1) / S);
            const int64_t hi2 = std::max<int64_t>(0, (hi_begin * S - reach) / S);
            if (overlap && c->slab_pair_form == 1 && plan.nseg >= 3) {
                // ("slab_pair_form" 1.)  The two boundary segments of the pass first, alone on the GPU -- launched beside
                // the interior they are dispatched AFTER it, the event hop delays them, and finish late --, then the
                // interior segments on the main stream while the chain runs on the communication stream; the once-
                // relaxed boundary planes come from the pass (`sw`).
                MG_TRY(launch_jacobi2(c, L, plan, 0, plan.nseg - 1, 2, L.v.rows, L.f.rows, L.v2.rows, st_lo, st_hi, L.sw.rows));
                HIP_TRY(hipEventRecord(c->ev_boundary, c->stream));
                HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
                MG_TRY(launch_jacobi2(c, L, plan, 1, 1, plan.nseg - 2, L.v.rows, L.f.rows, L.v2.rows, st_lo, st_hi, L.sw.rows));
                std::swap(c->stream, c->comm_stream);
                const int rc = boundary_chain(c->stream);
                std::swap(c->stream, c->comm_stream);
                MG_TRY(rc);
                HIP_TRY(hipEventRecord(c->ev_halo, c->comm_stream));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
            } else if (overlap && c->slab_pair_form != 1 && hi2 > lo2) {
                // Everything that waits for the neighbours -- the first sweep of the planes next to them, the exchange of
                // its boundary planes, the second sweep of the first / last slices, the exchange of the result -- runs
                // on the (high-priority) communication stream from the start of the pair and needs nothing from the
                // pass itself: four small operations beside ONE launch of the pass over the whole slab, which stores the
                // second sweep of all other rows.  (Measured on one slab of eight, tools/slab_rank_probe.py: a separate
                // boundary launch of the pass in front costs 0.1 ms of the pair's 0.8; beside the interior launch it
                // is dispatched after it and finishes late.)  The first sweep of those few planes is computed twice.
                HIP_TRY(hipEventRecord(c->ev_boundary, c->stream));
                HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
                const J2Plan whole = jacobi2_plan(c, L, false, 0);
                MG_TRY(launch_jacobi2(c, L, whole, 0, 1, whole.nseg, L.v.rows, L.f.rows, L.v2.rows, st_lo, st_hi, nullptr));
                std::swap(c->stream, c->comm_stream);
                const int rc = [&]() -> int {
                    if (lo && hi)
                        MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.sw.rows, nullptr, nullptr, nullptr, 0,
                                          lo2 + L.nslices - hi2, 0, lo2, hi2 - lo2));
                    else if (lo)
                        MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.sw.rows, nullptr, nullptr, nullptr, 0, lo2));
                    else if (hi)
                        MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.sw.rows, nullptr, nullptr, nullptr, hi2,
                                          L.nslices - hi2));
                    return boundary_chain(c->stream);
                }();
                std::swap(c->stream, c->comm_stream);
                MG_TRY(rc);
                HIP_TRY(hipEventRecord(c->ev_halo, c->comm_stream));
                HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
            } else {
                MG_TRY(launch_jacobi2(c, L, plan, 0, 1, plan.nseg, L.v.rows, L.f.rows, L.v2.rows, st_lo, st_hi, L.sw.rows));
                MG_TRY(boundary_chain(c->stream));
            }
            std::swap(L.v, L.v2);
            ++s;
            continue;
        }
        if (!overlap) {
            MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.v2.rows, nullptr, nullptr));
            std::swap(L.v, L.v2);
            MG_TRY(exchange_halo(c, L, L.v));
            continue;
        }
        // boundary planes first, then their exchange on the communication stream while the interior runs
        MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.v2.rows, nullptr, nullptr, nullptr, 0,
                          lo_end + L.nslices - hi_begin, 0, lo_end, hi_begin - lo_end));
        HIP_TRY(hipEventRecord(c->ev_boundary, c->stream));
        HIP_TRY(hipStreamWaitEvent(c->comm_stream, c->ev_boundary, 0));
        MG_TRY(exchange_halo(c, L, L.v2, c->comm_stream));
        HIP_TRY(hipEventRecord(c->ev_halo, c->comm_stream));
        MG_TRY(launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.v2.rows, nullptr, nullptr, nullptr, lo_end,
                          hi_begin - lo_end));
        HIP_TRY(hipStreamWaitEvent(c->stream, c->ev_halo, 0));
        std::swap(L.v, L.v2);
    }
    return 0;
}

int residual(mg_context* c, int level) {
    Level& L = c->L[level];
    return launch_ell(c, L, MODE_RESIDUAL, false, L.v.base, L.f.rows, L.v2.rows, nullptr, nullptr);
}

// Grid of the coarse planes this rank produces when restricting from `fine`: its own slab if the
// coarse level is distributed, otherwise the planes matching its fine slab inside the replica.
Grid coarse_target_grid(const mg_context* c, const Level& C, const Level& F) {
    Grid g = C.g;
    if (C.replicated && !F.replicated) {
        g.k0 = C.splits[c->comm.rank];
        g.nk = C.splits[c->comm.rank + 1] - g.k0;
        g.lead = (int64_t)g.k0 * g.plane;
    }
    return g;
}

int restrict_to(mg_context* c, int level, int kind) {
    Level& F = c->L[level];
    Level& C = c->L[level - 1];
    Grid gc = coarse_target_grid(c, C, F);
    if (kind == MG_RESTRICT_TABLE) {
        if (!c->rtab_count) return fail("no restriction table (mg_set_restriction_table)");
        // (the transpose of the P2 prolongation reaches three fine planes: on slabs the residual's halo must have room for them)
        if (!F.replicated && c->comm.active()) {
            if (F.hd < 3) return fail("the table restriction reaches three fine planes: \"halo_depth\" must be at least 3 on slabs");
            MG_TRY(exchange_halo(c, F, F.v2, nullptr, 3));
        }
        const RestrictTable t{c->rtab_count, c->rtab_off, c->rtab_w, c->rtab_m};
        hipLaunchKernelGGL(restrict_table, grid3(gc, gc.nk), dim3(kPlaneBlock), 0, c->stream, gc, F.g, t, F.v2.base, C.f.base);
    } else if (kind == MG_RESTRICT_FULL_WEIGHTING) {
        MG_TRY(exchange_halo(c, F, F.v2));
        hipLaunchKernelGGL(restrict_full_weighting, grid3(gc, gc.nk), dim3(kPlaneBlock), 0, c->stream, gc, F.g, F.v2.base,
                           C.f.base);
    } else {
        hipLaunchKernelGGL(restrict_inject, grid3(gc, gc.nk), dim3(kPlaneBlock), 0, c->stream, gc, F.g, F.v2.base, C.f.base);
    }
    HIP_TRY(hipGetLastError());
    if (C.replicated && !F.replicated) MG_TRY(allgather_planes(c, C.splits, C.g.plane, C.f.base));
    return 0;
}

// r = f - A v evaluated at the coarse nodes only and injected (one launch instead of residual + restrict)
int residual_restrict_fused(mg_context* c, int level) {
    Level& F = c->L[level];
    Level& C = c->L[level - 1];
    const Grid gc = coarse_target_grid(c, C, F);
    FusedRestrictArgs a{};
    a.vals = F.vals; a.cols = F.cols; a.codes = F.codes; a.offsets = F.offsets;
    a.x = F.v.base; a.f = F.f.rows; a.fc = C.f.base;
    a.W = F.W; a.R = F.R; a.coded = F.coded ? 1 : 0; a.gc = gc; a.gf = F.g;
    if (F.coded && F.scls && c->class_sweeps) {
        a.coded = 4; a.cls = F.scls; a.s_off = F.s_off; a.s_val = F.s_val; a.s_cnt = F.s_cnt;
    }
    if (F.sdia) {
        a.vals = F.dvals; a.W = F.wu; a.coded = 2; a.mlead = F.mlead;
        for (int t = 0; t < 8; ++t) a.up[t] = F.up[t];
        if (cls_full(F) && c->class_sweeps) {
            a.coded = 3; a.cls = F.cls + F.cls_lead; a.ctab = F.ctab;
        }
    }
    hipLaunchKernelGGL(residual_inject, grid3(gc, gc.nk), dim3(kPlaneBlock), 0, c->stream, a);
    HIP_TRY(hipGetLastError());
    if (C.replicated && !F.replicated) MG_TRY(allgather_planes(c, C.splits, C.g.plane, C.f.base));
    return 0;
}

int prolong(mg_context* c, int level, int add) {
    Level& F = c->L[level];
    Level& C = c->L[level - 1];
    const bool keep = !add || c->keep_err;
    if (keep) MG_TRY(vec_alloc(c, F, &F.err));
    if (c->ptab_count) {
        if (!F.replicated && c->comm.active() && c->halo_planes < 2)
            return fail("the table prolongation reaches two coarse planes: halo_planes must be 2 on slabs");
        const ProlongTable t{c->ptab_count, c->ptab_off, c->ptab_w};
        const dim3 grid = grid3(F.g, F.g.nk), blk(kPlaneBlock);
        if (add && keep)
            hipLaunchKernelGGL((prolong_table<true, true>), grid, blk, 0, c->stream, C.g, F.g, t, C.v.base, F.v.base, F.err.base);
        else if (add)
            hipLaunchKernelGGL((prolong_table<true, false>), grid, blk, 0, c->stream, C.g, F.g, t, C.v.base, F.v.base, (double*)nullptr);
        else
            hipLaunchKernelGGL((prolong_table<false, true>), grid, blk, 0, c->stream, C.g, F.g, t, C.v.base, F.v.base, F.err.base);
        HIP_TRY(hipGetLastError());
        if (add) MG_TRY(exchange_halo(c, F, F.v));
        return 0;
    }
    // one thread per pair of fine nodes along x
    const int64_t pairs = (int64_t)((F.g.nx + 1) / 2) * F.g.ny;
    const dim3 grid((unsigned)((pairs + kPlaneBlock - 1) / kPlaneBlock), (unsigned)F.g.nk, 1u);
    if (add && keep)
        hipLaunchKernelGGL((prolong_correct<true, true>), grid, dim3(kPlaneBlock), 0, c->stream, C.g, F.g, C.v.base, F.v.base, F.err.base);
    else if (add)
        hipLaunchKernelGGL((prolong_correct<true, false>), grid, dim3(kPlaneBlock), 0, c->stream, C.g, F.g, C.v.base, F.v.base, (double*)nullptr);
    else
        hipLaunchKernelGGL((prolong_correct<false, true>), grid, dim3(kPlaneBlock), 0, c->stream, C.g, F.g, C.v.base, F.v.base, F.err.base);
    HIP_TRY(hipGetLastError());
    if (add) MG_TRY(exchange_halo(c, F, F.v));
    return 0;
}

int zero_vec(mg_context* c, const Level& L, DVector& v) {
    HIP_TRY(hipMemsetAsync(v.base, 0, (size_t)L.xlen * sizeof(double), c->stream));
    return 0;
}

// sum over owned rows of x.y on the device -> c->scalars[slot]; all-reduced over slabs
int dot_device(mg_context* c, const Level& L, const double* x_rows, const double* y_rows, int slot) {
    const int np = (int)std::min<int64_t>(kMaxParts, std::max<int64_t>(1, (L.nloc + 2 * BLOCK - 1) / (2 * BLOCK)));
    hipLaunchKernelGGL(dot_partial, dim3(np), dim3(BLOCK), 0, c->stream, x_rows, y_rows, L.nloc, c->partials);
    hipLaunchKernelGGL(reduce_partials, dim3(1), dim3(BLOCK), 0, c->stream, c->partials, np, c->scalars + slot);
    HIP_TRY(hipGetLastError());
    if (!L.replicated) MG_TRY(allreduce_sum(c, c->scalars + slot, 1));
    return 0;
}

int norm2(mg_context* c, const Level& L, const double* x_rows, double* out) {
    MG_TRY(dot_device(c, L, x_rows, x_rows, 0));
    HIP_TRY(hipMemcpyAsync(c->h_scalars, c->scalars, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *out = std::sqrt(c->h_scalars[0]);
    return 0;
}

void free_direct(mg_context* c) {
    DirectSolver& d = c->direct;
    const size_t pw = (size_t)d.g.nb * d.g.p * (d.g.W > 0 ? d.g.W : 1), np = (size_t)d.g.nb * d.g.p;
    dev_free(c, d.T, np * d.g.p);
    dev_free(c, d.lcol, pw); dev_free(c, d.ucol, pw); dev_free(c, d.lval, pw); dev_free(c, d.uval, pw);
    dev_free(c, d.y, np); dev_free(c, d.x, np); dev_free(c, d.w, (size_t)d.g.p);
    d = DirectSolver();
}

EllView ell_view(const Level& L) {
    EllView e{};
    e.vals = L.vals; e.cols = L.cols; e.codes = L.codes; e.offsets = L.offsets;
    e.W = L.W; e.R = L.R; e.coded = L.coded ? 1 : 0; e.n = L.nloc;
    return e;
}

// Factor the coarsest level once: T_k = (D_k - L_k T_{k-1} U_{k-1})^-1 for every block of planes.
int build_direct(mg_context* c) {
    DirectSolver& d = c->direct;
    d.tried = true;
    Level& L = c->L[0];
    if (!c->use_direct || L.flat || L.g.lead != 0) return 0;
    const int64_t plane = L.g.plane;
    const int nz = L.g.nz;
    // planes per block: at least "direct_block_rows" rows (a solve is 3 dependent launches per block, each of them a few
    // microseconds whatever the block's size: fewer, larger blocks while the dense inverses stay small), and at least as many
    // planes as the stencil reaches (P2 rows reach two: the blocks must couple to their neighbours only)
    int64_t reach = 1;
    if (L.coded && L.ntable > 0) {
        std::vector<int> offs(256);
        HIP_TRY(hipMemcpy(offs.data(), L.offsets, 256 * sizeof(int), hipMemcpyDeviceToHost));
        for (int t = 0; t < L.ntable; ++t) reach = std::max<int64_t>(reach, (std::llabs((long long)offs[t]) + plane / 2) / plane);
    }
    int64_t G = std::max<int64_t>(reach, std::min<int64_t>(nz, (c->direct_block_rows + plane - 1) / plane));
    while (G > reach && G > 1 && G * plane > 2304) --G;               // (the dense blocks' size limit below)
    const int64_t p = G * plane;
    const int64_t nb = (nz + G - 1) / G;
    if (p > 2304 || nb * p * p * 8 > ((int64_t)3 << 29)) return 0;     // too large to store densely: PCG
    d.g.n = L.nloc; d.g.p = (int)p; d.g.plane = (int)plane; d.g.nb = (int)nb; d.g.W = L.W;
    const size_t pw = (size_t)nb * p * L.W, np = (size_t)nb * p, pp = (size_t)p * p;
    double *A = nullptr, *B = nullptr, *P = nullptr;       // A: block being inverted, B: row panel, P: pivot block
    int* d_status = nullptr;
    int rc = [&]() -> int {
        MG_TRY(dev_alloc(c, &d.T, np * p));
        MG_TRY(dev_alloc(c, &d.lcol, pw)); MG_TRY(dev_alloc(c, &d.ucol, pw));
        MG_TRY(dev_alloc(c, &d.lval, pw)); MG_TRY(dev_alloc(c, &d.uval, pw));
        MG_TRY(dev_alloc(c, &d.y, np)); MG_TRY(dev_alloc(c, &d.x, np)); MG_TRY(dev_alloc(c, &d.w, (size_t)p));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&A), pp * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&B), (size_t)GJ_NB * p * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&P), (size_t)GJ_NB * GJ_NB * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_status), sizeof(int)));
        HIP_TRY(hipMemsetAsync(d_status, 0, sizeof(int), c->stream));
        const EllView e = ell_view(L);
        const dim3 rows_grid(blocks_for(p, 128)), rows_blk(128);
        const dim3 gj_blk(256);
        for (int k = 0; k < (int)nb; ++k) {
            const size_t ko = (size_t)k * p * L.W;
            HIP_TRY(hipMemsetAsync(A, 0, pp * 8, c->stream));
            hipLaunchKernelGGL(bt_extract_dense, rows_grid, rows_blk, 0, c->stream, e, d.g, k, A, d_status);
            hipLaunchKernelGGL(bt_extract_coupling, rows_grid, rows_blk, 0, c->stream, e, d.g, k, d.lcol + ko, d.lval + ko,
                               d.ucol + ko, d.uval + ko);
            if (k > 0) {
                const size_t po = (size_t)(k - 1) * p * L.W;
                hipLaunchKernelGGL(bt_schur_update, rows_grid, rows_blk, 0, c->stream, d.g, d.lcol + ko, d.lval + ko,
                                   d.ucol + po, d.uval + po, d.T + (size_t)(k - 1) * pp, A);
            }
            for (int k0 = 0; k0 < (int)p; k0 += GJ_NB) {
                const int nbk = std::min<int>(GJ_NB, (int)p - k0);
                hipLaunchKernelGGL(gj_pivot_block, dim3(1), gj_blk, 0, c->stream, (int)p, k0, nbk, A, P);
                hipLaunchKernelGGL(gj_row_panel, dim3(blocks_for(p, 256), (unsigned)nbk), gj_blk, 0, c->stream, (int)p, k0,
                                   nbk, A, P, B);
                hipLaunchKernelGGL(gj_trailing, dim3(blocks_for(p, 256), (unsigned)p), gj_blk, 0, c->stream, (int)p, k0, nbk,
                                   A, B);
                hipLaunchKernelGGL(gj_finish, dim3(blocks_for(p, 128)), dim3(128), 0, c->stream, (int)p, k0, nbk, A, P, B);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(d.T + (size_t)k * pp, A, pp * 8, hipMemcpyDeviceToDevice, c->stream));
        }
        int status = 0;
        HIP_TRY(hipMemcpyAsync(&status, d_status, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        d.ok = status == 0;
        return 0;
    }();
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(P); (void)hipFree(d_status);
    if (rc || !d.ok) { free_direct(c); d.tried = true; }
    return rc;
}

int direct_solve(mg_context* c);

// The factorisation uses no pivoting, which is safe for the SPD / M-matrices this path is meant for but not
// for an arbitrary hand-off: solve once against a known right-hand side (all ones) and keep the direct
// solver only if the residual is at round-off; otherwise the coarsest level falls back to PCG.
int validate_direct(mg_context* c) {
    DirectSolver& d = c->direct;
    if (!d.ok) return 0;
    Level& L = c->L[0];
    double* saved = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&saved), (size_t)L.xlen * 8));
    int rc = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(saved, L.f.base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
        std::vector<double> ones((size_t)L.nloc, 1.0);
        HIP_TRY(hipMemcpyAsync(L.f.rows, ones.data(), (size_t)L.nloc * 8, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        MG_TRY(direct_solve(c));
        MG_TRY(launch_ell(c, L, MODE_RESIDUAL, false, L.v.base, L.f.rows, L.v2.rows, nullptr, nullptr));
        double rn = 0.0;
        MG_TRY(norm2(c, L, L.v2.rows, &rn));
        const double rel = rn / std::sqrt((double)L.nloc);
        if (!(rel <= 1e-9)) d.ok = false;           // also catches NaN
        HIP_TRY(hipMemcpyAsync(L.f.base, saved, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    (void)hipFree(saved);
    if (rc || !d.ok) { free_direct(c); d.tried = true; }
    return rc;
}

int direct_solve(mg_context* c) {
    DirectSolver& d = c->direct;
    Level& L = c->L[0];
    const int p = d.g.p, nb = d.g.nb;
    const size_t pp = (size_t)p * p, pw = (size_t)p * d.g.W;
    const int64_t first = std::min<int64_t>(L.nloc, p);
    HIP_TRY(hipMemsetAsync(d.y, 0, (size_t)p * 8, c->stream));
    HIP_TRY(hipMemcpyAsync(d.y, L.f.rows, (size_t)first * 8, hipMemcpyDeviceToDevice, c->stream));
    const dim3 wave_grid(blocks_for(p, WAVES_PER_BLOCK)), blk(BLOCK);
    // wide rows (P2: ~20 couplings into the previous block per row): T_{k-1} y_{k-1} once, then the sparse part -- 19 + 3 us
    // per block instead of 136 us at p = 2178; narrow rows keep the single launch (fewer launches on the 2-D levels)
    const bool two_step = d.g.W > 12;
    for (int k = 1; k < nb; ++k) {
        if (two_step) {
            hipLaunchKernelGGL(bt_matvec, wave_grid, blk, 0, c->stream, p, d.T + (size_t)(k - 1) * pp, d.y + (size_t)(k - 1) * p, d.w);
            hipLaunchKernelGGL(bt_forward_sparse, dim3(blocks_for(p, 128)), dim3(128), 0, c->stream, d.g, k, d.lcol + k * pw,
                               d.lval + k * pw, d.w, L.f.rows, d.y);
        } else {
            hipLaunchKernelGGL(bt_forward, wave_grid, blk, 0, c->stream, d.g, k, d.lcol + k * pw, d.lval + k * pw,
                               d.T + (size_t)(k - 1) * pp, L.f.rows, d.y);
        }
    }
    for (int k = nb - 1; k >= 0; --k) {
        hipLaunchKernelGGL(bt_backward_rhs, dim3(blocks_for(p, 128)), dim3(128), 0, c->stream, d.g, k, d.ucol + k * pw,
                           d.uval + k * pw, d.y, d.x, d.w);
        hipLaunchKernelGGL(bt_backward, wave_grid, blk, 0, c->stream, d.g, k, d.T + (size_t)k * pp, d.w, d.x, L.v.rows);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int pcg_solve(mg_context* c, int* iters_out, double* rel_out);

// Coarsest level: v = A^-1 f, standing in for the reference's exact spsolve (multigrid.py:239-241):
// the block-tridiagonal LU where the level allows it, otherwise Jacobi-preconditioned CG.
int coarse_solve(mg_context* c, int* iters_out, double* rel_out) {
    if (!c->direct.tried) { MG_TRY(build_direct(c)); MG_TRY(validate_direct(c)); }
    if (c->direct.ok) {
        if (iters_out) *iters_out = 0;
        if (rel_out) *rel_out = 0.0;
        return direct_solve(c);
    }
    return pcg_solve(c, iters_out, rel_out);
}

// Jacobi-preconditioned CG run to `coarse_rtol`.  Iterations are enqueued in batches; only the 4-byte
// convergence flag crosses to the host between batches.
int pcg_solve(mg_context* c, int* iters_out, double* rel_out) {
    Level& L = c->L[0];
    if (L.g.lead != 0) return fail("coarsest level must not be distributed");
    const int64_t n = L.nloc;
    if (!c->pcg_r) {
        MG_TRY(dev_alloc(c, &c->pcg_r, (size_t)n + 4));
        MG_TRY(dev_alloc(c, &c->pcg_z, (size_t)n + 4));
        MG_TRY(dev_alloc(c, &c->pcg_q, (size_t)n + 4));
        MG_TRY(vec_alloc(c, L, &c->pcg_p));
        c->pcg_parts = (int)std::min<int64_t>(512, std::max<int64_t>(1, (n + BLOCK - 1) / BLOCK));
        c->pcg_parts_a = (int)std::max<unsigned>(blocks_for(L.nslices, WAVES_PER_BLOCK), (unsigned)c->pcg_parts);
        MG_TRY(dev_alloc(c, &c->pcg_part_a, (size_t)c->pcg_parts_a));
        MG_TRY(dev_alloc(c, &c->pcg_part_b, (size_t)2 * c->pcg_parts));
    }
    PcgArgs a{};
    a.x = L.v.rows; a.r = c->pcg_r; a.z = c->pcg_z; a.p = c->pcg_p.rows; a.q = c->pcg_q;
    a.b = L.f.rows; a.dinv = L.dinv; a.part_a = c->pcg_part_a; a.part_b = c->pcg_part_b;
    a.nparts = c->pcg_parts; a.sc = c->scalars; a.done = c->done; a.n = n;
    a.rtol2 = c->coarse_rtol * c->coarse_rtol;
    const dim3 grid(c->pcg_parts), blk(BLOCK);
    hipLaunchKernelGGL(pcg_init, grid, blk, 0, c->stream, a);
    hipLaunchKernelGGL(pcg_init_finish, dim3(1), blk, 0, c->stream, a);
    int it = 0;
    int batch = c->pcg_predict > 0 ? c->pcg_predict : c->pcg_chunk;
    int* h_done = reinterpret_cast<int*>(c->h_scalars + 8);
    for (;;) {
        batch = std::min(batch, c->coarse_maxit - it);
        for (int b = 0; b < batch; ++b) {
            unsigned np_spmv = 0;
            MG_TRY(launch_ell(c, L, MODE_SPMV, true, c->pcg_p.base, nullptr, c->pcg_q, c->pcg_part_a, c->done, &np_spmv));
            hipLaunchKernelGGL(pcg_update, grid, blk, 0, c->stream, a, (int)np_spmv, it + b);
            hipLaunchKernelGGL(pcg_direction, grid, blk, 0, c->stream, a, it + b);
        }
        it += batch;
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_done, c->done, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(c->h_scalars, c->scalars, 8 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        if (*h_done || it >= c->coarse_maxit) break;
        batch = c->pcg_chunk;
    }
    const int used = (int)c->h_scalars[5];
    const double bb = c->h_scalars[3], rr = c->h_scalars[2];
    const double rel = bb > 0.0 ? std::sqrt(rr / bb) : 0.0;
    if (iters_out) *iters_out = used;
    if (rel_out) *rel_out = rel;
    if (!*h_done)
        return fail("coarse solve did not converge: " + std::to_string(used) + " iterations, relative residual " +
                    std::to_string(rel));
    c->pcg_predict = used;
    return 0;
}

// One V(mu1, mu2) cycle on `level` (V_cycle_scheme, multigrid.py:231-268): pre-smooth, residual,
// restrict, recurse from a zero guess, interpolate + correct, post-smooth; exact solve on level 0.
int vcycle(mg_context* c, int level) {
    if (level == 0) return coarse_solve(c, nullptr, nullptr);
    Level& C = c->L[level - 1];
    MG_TRY(smooth(c, level, c->mu1));
    if (c->restriction == MG_RESTRICT_INJECTION && c->fuse_restrict) {
        MG_TRY(residual_restrict_fused(c, level));
    } else {
        MG_TRY(residual(c, level));
        MG_TRY(restrict_to(c, level, c->restriction));
    }
    MG_TRY(zero_vec(c, C, C.v));
    MG_TRY(vcycle(c, level - 1));
    MG_TRY(prolong(c, level, 1));
    MG_TRY(smooth(c, level, c->mu2));
    return 0;
}

void drop_graphs(mg_context* c) {
    for (auto& g : c->graphs) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
    }
    c->graphs.clear();
}

// A V-cycle is a fixed sequence of launches once the coarsest solve is the direct one (no host round
// trip), so it is captured into a hipGraph the first time and replayed afterwards: the launch-bound coarse
// levels then cost a kernel boundary instead of a host launch each.  The Jacobi ping-pong swaps the
// MG_VEC_V buffers mu1+mu2 times per level; the captured pointer state is part of the cache key and the
// swaps are re-applied on the host after a replay.
// Everything a V-cycle from `level` builds or allocates on first use (the direct coarsest solve and its validation, the
// colouring checks, the parked-sweep and error vectors): afterwards a cycle -- and its capture -- only enqueues work.
int prepare_cycle(mg_context* c, int level) {
    if (!c->direct.tried) { MG_TRY(build_direct(c)); MG_TRY(validate_direct(c)); }
    if (c->smoother == MG_SMOOTH_MCGS)                   // (the check synchronises: not inside a capture)
        for (int l = 1; l <= level; ++l)
            if (c->L[l].mc_ok < 0 && !c->L[l].flat) MG_TRY(check_coloring(c, c->L[l]));
    if (c->comm.active())
        for (int l = 1; l <= level; ++l)
            if (!c->L[l].replicated) {
                MG_TRY(vec_alloc(c, c->L[l], &c->L[l].sw));
                // (the neighbours' class bytes next to the slab: allocations, exchanges and a vote -- not inside a capture)
                if (c->L[l].cls_halo == 0 && slab_ksweep_level(c, c->L[l]) && std::max(c->mu1, c->mu2) >= c->fuse_k_slab_min_sweeps)
                    MG_TRY(ensure_class_halos(c, c->L[l]));
            }
    if (c->keep_err)
        for (int l = 1; l <= level; ++l) MG_TRY(vec_alloc(c, c->L[l], &c->L[l].err));
    return 0;
}

int vcycle_graphed(mg_context* c, int level) {
    MG_TRY(prepare_cycle(c, level));
    // Slabs: RCCL's point-to-point calls are stream operations and are captured with the kernels (both streams of the
    // overlapped sweeps join the capture through their events).  Opt-in ("graph_comm"): every rank must take the same
    // decision, and this build could only be validated against the in-process stand-in (tests/fake_rccl).
    const bool slabs = c->comm.active();
    if (!c->use_graph || level == 0 || !c->direct.ok || (slabs && !(c->graph_comm && c->comm.nccl))) return vcycle(c, level);
    std::vector<double*> pre(level + 1);
    for (int l = 0; l <= level; ++l) pre[l] = c->L[l].v.raw;
    for (auto& g : c->graphs) {
        if (g.level != level || g.epoch != c->epoch || g.pre != pre) continue;
        HIP_TRY(hipGraphLaunch(g.exec, c->stream));
        ++c->graph_replays;
        for (int l = 0; l <= level; ++l)
            if (c->L[l].v.raw != g.post[l]) std::swap(c->L[l].v, c->L[l].v2);
        return 0;
    }
    if (c->graphs.size() >= 8) drop_graphs(c);
    CycleGraph g;
    g.level = level; g.epoch = c->epoch; g.pre = pre;
    HIP_TRY(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    const int rc = vcycle(c, level);
    const hipError_t e = hipStreamEndCapture(c->stream, &g.graph);
    if (rc != 0 || e != hipSuccess) {
        if (g.graph) (void)hipGraphDestroy(g.graph);
        (void)hipGetLastError();
        if (rc != 0) return rc;
        return fail(std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    g.post.resize(level + 1);
    for (int l = 0; l <= level; ++l) g.post[l] = c->L[l].v.raw;
    HIP_TRY(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(g.exec, c->stream));
    c->graphs.push_back(std::move(g));
    return 0;
}

int ensure_stage(mg_context* c, int64_t elems) {
    if (c->stage_elems >= elems) return 0;
    dev_free(c, c->stage, (size_t)c->stage_elems);
    c->stage_elems = 0;
    MG_TRY(dev_alloc(c, &c->stage, (size_t)elems));
    c->stage_elems = elems;
    return 0;
}

int upload_vector(mg_context* c, Level& L, DVector& v, const double* host) {
    MG_TRY(ensure_stage(c, L.n_global));
    ++c->uploads;
    HIP_TRY(hipMemcpyAsync(c->stage, host, (size_t)L.n_global * 8, hipMemcpyHostToDevice, c->stream));
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (L.n_global + 255) / 256);
    hipLaunchKernelGGL(scatter_in, dim3(nb), dim3(256), 0, c->stream, c->stage, L.perm, L.n_global, L.row0, L.g.lead,
                       L.xlen, v.base);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

// Replace the int32 column array by offset codes when the level has <= 255 distinct col - row
// values and a specialised width; otherwise the int32 kernels stay in use.
int encode_level(mg_context* c, Level& L) {
    if (!c->use_codes || !coded_width(L.W)) return 0;
    int* d_table = nullptr;
    int* d_count = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_table), kDeltaSlots * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_count), 2 * sizeof(int)));
    std::vector<int> h_table(kDeltaSlots, kDeltaEmpty);
    int h_counts[2] = {0, 0};
    int rc = [&]() -> int {
        HIP_TRY(hipMemcpyAsync(d_table, h_table.data(), kDeltaSlots * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipMemsetAsync(d_count, 0, 2 * sizeof(int), c->stream));
        const int64_t total = L.nslices * L.W * (WAVE * L.R);
        const dim3 grid(blocks_for(total, 256)), blk(256);
        switch (L.R) {
            case 1: hipLaunchKernelGGL(ell_collect_deltas<1>, grid, blk, 0, c->stream, L.cols, L.nslices, L.W, L.nloc, L.g.lead, d_table, d_count); break;
            case 2: hipLaunchKernelGGL(ell_collect_deltas<2>, grid, blk, 0, c->stream, L.cols, L.nslices, L.W, L.nloc, L.g.lead, d_table, d_count); break;
            default: hipLaunchKernelGGL(ell_collect_deltas<4>, grid, blk, 0, c->stream, L.cols, L.nslices, L.W, L.nloc, L.g.lead, d_table, d_count); break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h_counts, d_count, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(h_table.data(), d_table, kDeltaSlots * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    (void)hipFree(d_table);
    (void)hipFree(d_count);
    if (rc) return rc;
    const int h_count = h_counts[0];
    if (h_counts[1] || h_count > 255) return 0;     // irregular numbering: keep int32 columns
    std::vector<int> offs;
    for (int v : h_table)
        if (v != kDeltaEmpty) offs.push_back(v);
    std::sort(offs.begin(), offs.end());
    if ((int)offs.size() != h_count) return fail("offset table is inconsistent");
    const auto zero = std::find(offs.begin(), offs.end(), 0);
    if (zero == offs.end()) return fail("offset table lacks the diagonal");
    L.ntable = (int)offs.size();
    L.dcode = (int)(zero - offs.begin());
    L.rb_ok = !L.flat && (L.g.nx & 1) && (L.g.ny & 1);
    for (int o : offs)
        if (o != 0 && (o & 1) == 0) L.rb_ok = false;       // a coupling inside one colour
    offs.resize(256, 0);
    MG_TRY(dev_alloc(c, &L.offsets, 256));
    HIP_TRY(hipMemcpyAsync(L.offsets, offs.data(), 256 * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int CW = (L.W + 7) / 8;
    MG_TRY(dev_alloc(c, &L.codes, (size_t)L.nslices * CW * (WAVE * L.R)));
    const dim3 grid(blocks_for(L.nslices * (WAVE * L.R), 256)), blk(256);
    switch (L.R) {
        case 1: hipLaunchKernelGGL(ell_encode<1>, grid, blk, 0, c->stream, L.cols, L.codes, L.nslices, L.W, L.nloc, L.g.lead, L.offsets, L.ntable, L.dcode); break;
        case 2: hipLaunchKernelGGL(ell_encode<2>, grid, blk, 0, c->stream, L.cols, L.codes, L.nslices, L.W, L.nloc, L.g.lead, L.offsets, L.ntable, L.dcode); break;
        default: hipLaunchKernelGGL(ell_encode<4>, grid, blk, 0, c->stream, L.cols, L.codes, L.nslices, L.W, L.nloc, L.g.lead, L.offsets, L.ntable, L.dcode); break;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    dev_free(c, L.cols, (size_t)L.nslices * L.W * (WAVE * L.R));      // 4*W bytes per row back
    L.coded = true;
    return 0;
}

// Row classes (mg_jacobi2.hip.h, "row classes"): a dictionary of the level's distinct FULL rows (five- or seven-point),
// built and verified on the device; levels with more than 255 distinct non-zero rows go without.
int build_row_classes_q(mg_context* c, Level& L, int qbits);

int build_row_classes(mg_context* c, Level& L) {
    if (!c->use_classes || !L.sdia || (L.wu != 3 && L.wu != 4)) return 0;
    MG_TRY(build_row_classes_q(c, L, c->storage_qbits));
    // "storage_auto": rows that are almost repetitive (entries assembled with row-dependent round-off) get one more try
    // in which entries within 4 units in the last place (2^-50 relative: one ulp either way, across a power of two) count as equal
    if (!L.cls && c->storage_auto && c->storage_qbits == 0) MG_TRY(build_row_classes_q(c, L, 2));
    return 0;
}

int build_row_classes_q(mg_context* c, Level& L, int qbits) {
    // scratch: hash tags | slot values | count, flag, slot_class[CLS_SLOTS], hist[256]
    struct Scratch {
        char* p = nullptr;
        ~Scratch() { if (p) (void)hipFree(p); }
    } scratch;
    const size_t tag_bytes = CLS_SLOTS * sizeof(unsigned long long), val_bytes = (size_t)CLS_SLOTS * CLS_W * sizeof(double);
    const size_t int_bytes = (2 + CLS_SLOTS + 256) * sizeof(int);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scratch.p), tag_bytes + val_bytes + int_bytes));
    HIP_TRY(hipMemsetAsync(scratch.p, 0, tag_bytes + val_bytes + int_bytes, c->stream));
    int* const ints = reinterpret_cast<int*>(scratch.p + tag_bytes + val_bytes);
    // padded like the vectors (vec_reach) so that the pass can read whole tiles around the level unpredicated
    const int64_t cls_lead = ((std::max(L.mlead, vec_reach(L) + (int64_t)L.hd * L.g.plane) + 255) / 256) * 256;
    const int64_t cls_rows = cls_lead + L.nloc + cls_lead + 512;
    unsigned char* cls = nullptr;
    double* ctab = nullptr;
    MG_TRY(dev_alloc(c, &cls, (size_t)cls_rows));
    MG_TRY(dev_alloc(c, &ctab, 256 * CLS_W));
    ClsArgs a{};
    a.dvals = L.dvals; a.wu = L.wu; a.nloc = L.nloc; a.mlead = L.mlead;
    for (int t = 0; t < 4; ++t) a.up[t] = L.up[t];
    a.tags = reinterpret_cast<unsigned long long*>(scratch.p);
    a.svals = reinterpret_cast<double*>(scratch.p + tag_bytes);
    a.count = ints; a.flag = ints + 1; a.slot_class = ints + 2;
    a.hist = reinterpret_cast<unsigned*>(ints + 2 + CLS_SLOTS);
    a.ctab = ctab; a.cls = cls; a.crows = cls_rows; a.clead = cls_lead; a.qbits = qbits;
    const dim3 grid(blocks_for(L.nloc, 256)), egrid(blocks_for(cls_rows, 256)), blk(256);
    int h[2] = {0, 0};
    std::vector<unsigned> hist(256, 0u);
    std::vector<double> tab(256 * CLS_W, 0.0);
    int rc = [&]() -> int {
        switch (L.R) {
            case 1: hipLaunchKernelGGL(cls_insert<64>, grid, blk, 0, c->stream, a); break;
            case 2: hipLaunchKernelGGL(cls_insert<128>, grid, blk, 0, c->stream, a); break;
            default: hipLaunchKernelGGL(cls_insert<256>, grid, blk, 0, c->stream, a); break;
        }
        hipLaunchKernelGGL(cls_assign, dim3(1), dim3(64), 0, c->stream, a);
        switch (L.R) {
            case 1: hipLaunchKernelGGL(cls_encode<64>, egrid, blk, 0, c->stream, a); break;
            case 2: hipLaunchKernelGGL(cls_encode<128>, egrid, blk, 0, c->stream, a); break;
            default: hipLaunchKernelGGL(cls_encode<256>, egrid, blk, 0, c->stream, a); break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, ints, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(hist.data(), a.hist, 256 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipMemcpyAsync(tab.data(), ctab, 256 * CLS_W * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    if (!rc) { L.rep_distinct = h[0]; L.rep_cls_qbits = qbits; }
    bool escape = false;
    if (!rc && (h[0] > 255 || h[1]) && c->cls_escape && L.wu == 4 && !L.flat && L.nloc >= (1 << 16)) {
        // More than 255 distinct rows.  If most rows are still copies of a few (a uniform mesh with some odd rows: other
        // material, other boundary condition), the 254 most frequent rows -- found on a sample of 2^20 rows spread over the
        // level -- keep their class bytes and every other row gets CLS_ESCAPE: the K-sweep march reads such a row from the
        // symmetric diagonal storage.  Worth it while at least three quarters of the rows have a class.
        rc = [&]() -> int {
            unsigned* slot_count = nullptr;
            struct Free { unsigned*& p; ~Free() { if (p) (void)hipFree(p); } } guard{slot_count};
            // (slot counts | one bit per hash value: seen once)
            HIP_TRY(hipMalloc(reinterpret_cast<void**>(&slot_count), (CLS_SLOTS + CLS_SEEN_BITS / 32) * sizeof(unsigned)));
            HIP_TRY(hipMemsetAsync(slot_count, 0, (CLS_SLOTS + CLS_SEEN_BITS / 32) * sizeof(unsigned), c->stream));
            HIP_TRY(hipMemsetAsync(scratch.p, 0, tag_bytes + val_bytes + int_bytes, c->stream));
            const int64_t nsample = std::min<int64_t>(L.nloc, (int64_t)1 << 20);
            const dim3 sgrid(blocks_for(nsample, 256));
            switch (L.R) {
                case 1: hipLaunchKernelGGL(cls_sample_insert<64>, sgrid, blk, 0, c->stream, a, nsample, slot_count, slot_count + CLS_SLOTS); break;
                case 2: hipLaunchKernelGGL(cls_sample_insert<128>, sgrid, blk, 0, c->stream, a, nsample, slot_count, slot_count + CLS_SLOTS); break;
                default: hipLaunchKernelGGL(cls_sample_insert<256>, sgrid, blk, 0, c->stream, a, nsample, slot_count, slot_count + CLS_SLOTS); break;
            }
            HIP_TRY(hipGetLastError());
            std::vector<unsigned> cnt(CLS_SLOTS);
            std::vector<double> sv((size_t)CLS_SLOTS * CLS_W);
            HIP_TRY(hipMemcpyAsync(cnt.data(), slot_count, CLS_SLOTS * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(sv.data(), a.svals, val_bytes, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            std::vector<int> order(CLS_SLOTS);
            for (int i = 0; i < CLS_SLOTS; ++i) order[i] = i;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cnt[x] > cnt[y]; });
            std::vector<int> slot_class(CLS_SLOTS, 0);
            std::fill(tab.begin(), tab.end(), 0.0);
            for (int id = 1; id < CLS_ESCAPE && cnt[order[id - 1]] > 0; ++id) {
                slot_class[order[id - 1]] = id;
                for (int t = 0; t < 7; ++t) tab[(size_t)CLS_W * id + t] = sv[(size_t)CLS_W * order[id - 1] + t];
            }
            HIP_TRY(hipMemcpyAsync(a.slot_class, slot_class.data(), CLS_SLOTS * sizeof(int), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(ctab, tab.data(), 256 * CLS_W * sizeof(double), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(hipMemsetAsync(a.hist, 0, 256 * sizeof(unsigned), c->stream));
            switch (L.R) {
                case 1: hipLaunchKernelGGL(cls_encode_escape<64>, egrid, blk, 0, c->stream, a); break;
                case 2: hipLaunchKernelGGL(cls_encode_escape<128>, egrid, blk, 0, c->stream, a); break;
                default: hipLaunchKernelGGL(cls_encode_escape<256>, egrid, blk, 0, c->stream, a); break;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(hist.data(), a.hist, 256 * sizeof(unsigned), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            return 0;
        }();
        escape = !rc && (double)hist[CLS_ESCAPE] <= 0.25 * (double)L.nloc;
        int kmax = 0;
        // ... and no tile of the march meets more of them at a time than its pool holds
        if (escape && hist[CLS_ESCAPE] > 0) {
            rc = escape_kmax(c, L, cls, cls_lead, &kmax);
            escape = !rc && kmax >= 3;
        }
        if (std::getenv("MG_DEBUG_STORAGE"))
            std::fprintf(stderr, "[mg] row dictionary with escape: %lld rows, tolerance %d bits, %u escape rows, sweeps per pass that fit %d, rc %d\n",
                         (long long)L.nloc, qbits, hist[CLS_ESCAPE], kmax, rc);
        if (escape) { h[0] = 254; h[1] = 0; L.esc_kmax = kmax; L.rep_escape = hist[CLS_ESCAPE]; }
    }
    if (rc || h[0] > 255 || h[1]) {                 // too many distinct rows (or a hash collision): plain pass
        dev_free(c, cls, (size_t)cls_rows);
        dev_free(c, ctab, 256 * CLS_W);
        return rc;
    }
    L.cls = cls; L.ctab = ctab; L.ncls = h[0] + 1; L.cls_lead = cls_lead; L.cls_rows = cls_rows;
    L.cls_escape = escape && L.rep_escape > 0;
    if (escape) hist[CLS_ESCAPE] = 0;               // (never the most frequent class)
    L.cmain = 0;
    for (int k = 1; k < L.ncls; ++k)
        if (L.cmain == 0 || hist[k] > hist[L.cmain]) L.cmain = k;
    for (int t = 0; t < 8; ++t) L.cm[t] = tab[(size_t)CLS_W * L.cmain + t];
    // what the block pass (mg_jacobiblk.hip.h) needs to know: blocks are cut in grid coordinates
    if (L.wu == 4 && !L.cls_escape && !L.flat && L.up[1] == 1 && L.up[2] == L.g.nx && (int64_t)L.up[3] == L.g.plane) {
        JBCheckArgs k{};
        k.cls = L.cls + L.cls_lead; k.ctab = L.ctab; k.nx = L.g.nx; k.ny = L.g.ny; k.nz = L.g.nk; k.P = L.g.plane; k.n = L.nloc;
        k.flag = reinterpret_cast<int*>(c->partials);
        HIP_TRY(hipMemsetAsync(k.flag, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(sdia_grid_decoupled, dim3(blocks_for(L.nloc, 256)), dim3(256), 0, c->stream, k);
        HIP_TRY(hipGetLastError());
        int flag = 1;
        HIP_TRY(hipMemcpyAsync(&flag, k.flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        L.grid_decoupled = flag == 0;
    }
    return 0;
}

// Stencil classes of an offset-coded level that did not qualify for symmetric diagonals (wide stencils: P2): a
// dictionary of its distinct rows -- the W (offset, value) pairs in stored order, bit for bit -- built and verified on
// the device; levels with more than 255 distinct rows go without.
// What the plane march of mg_lattice.hip.h needs on top of the stencil classes: every entry's offset as (di, dj, dk)
// with |d| <= 2 (else the level keeps the gathering kernel), and the LM_K most frequent classes.
int prepare_lat_march(mg_context* c, Level& L) {
    L.lm_ntop = 0;
    if (L.flat || L.g.ny < 5 || L.g.nx < 5 || !L.scls) return 0;
    const size_t n = (size_t)256 * L.W;
    std::vector<int> off(n), cnt(256), pack(n, 0), hist(256, 0);
    HIP_TRY(hipMemcpy(off.data(), L.s_off, n * sizeof(int), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(cnt.data(), L.s_cnt, 256 * sizeof(int), hipMemcpyDeviceToHost));
    const int64_t P = L.g.plane, nx = L.g.nx;
    for (int cl = 0; cl < L.nscls; ++cl)
        for (int t = 0; t < cnt[cl]; ++t) {
            const int64_t o = off[(size_t)cl * L.W + t];
            const int64_t dk = (o + (o >= 0 ? P / 2 : -(P / 2))) / P, rem = o - dk * P;
            const int64_t dj = (rem + (rem >= 0 ? nx / 2 : -(nx / 2))) / nx, di = rem - dj * nx;
            if (std::llabs(dk) > 2 || std::llabs(dj) > 2 || std::llabs(di) > 2) return 0;
            pack[(size_t)cl * L.W + t] = (int)(((dk + 2) << 16) | ((dj + 2) << 8) | (di + 2));
        }
    int* d_hist = reinterpret_cast<int*>(c->partials);
    HIP_TRY(hipMemsetAsync(d_hist, 0, 256 * sizeof(int), c->stream));
    hipLaunchKernelGGL(lm_class_histogram, dim3(1024), dim3(256), 0, c->stream, L.scls, L.nloc, d_hist);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(hist.data(), d_hist, 256 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    std::vector<int> order(L.nscls);
    for (int i = 0; i < L.nscls; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return hist[x] > hist[y]; });
    MG_TRY(dev_alloc(c, &L.s_pack, n));
    HIP_TRY(hipMemcpy(L.s_pack, pack.data(), n * sizeof(int), hipMemcpyHostToDevice));
    L.lm_ntop = std::min<int>(LM_K, L.nscls);
    for (int t = 0; t < L.lm_ntop; ++t) L.lm_top[t] = order[t];
    return 0;
}

int build_stencil_classes(mg_context* c, Level& L) {
    if (!c->use_classes || !L.coded || L.sdia || L.flat || L.W < 8) return 0;
    struct Scratch {
        char* p = nullptr;
        ~Scratch() { if (p) (void)hipFree(p); }
    } scratch;
    const size_t tag_bytes = SCLS_SLOTS * sizeof(unsigned long long), row_bytes = SCLS_SLOTS * sizeof(int64_t);
    const size_t int_bytes = (2 + SCLS_SLOTS) * sizeof(int);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&scratch.p), tag_bytes + row_bytes + int_bytes));
    HIP_TRY(hipMemsetAsync(scratch.p, 0, tag_bytes + row_bytes + int_bytes, c->stream));
    int* const ints = reinterpret_cast<int*>(scratch.p + tag_bytes + row_bytes);
    unsigned char* cls = nullptr;
    int *s_off = nullptr, *s_cnt = nullptr;
    double* s_val = nullptr;
    const size_t crows = (size_t)L.nslices * WAVE * L.R;
    MG_TRY(dev_alloc(c, &cls, crows));
    MG_TRY(dev_alloc(c, &s_off, (size_t)256 * L.W));
    MG_TRY(dev_alloc(c, &s_val, (size_t)256 * L.W));
    MG_TRY(dev_alloc(c, &s_cnt, 256));
    HIP_TRY(hipMemsetAsync(cls, 0, crows, c->stream));
    HIP_TRY(hipMemsetAsync(s_off, 0, (size_t)256 * L.W * sizeof(int), c->stream));
    HIP_TRY(hipMemsetAsync(s_val, 0, (size_t)256 * L.W * sizeof(double), c->stream));
    HIP_TRY(hipMemsetAsync(s_cnt, 0, 256 * sizeof(int), c->stream));
    SclsArgs a{};
    a.vals = L.vals; a.codes = L.codes; a.offsets = L.offsets; a.W = L.W; a.dcode = L.dcode; a.nloc = L.nloc; a.nslices = L.nslices;
    a.tags = reinterpret_cast<unsigned long long*>(scratch.p);
    a.slot_row = reinterpret_cast<int64_t*>(scratch.p + tag_bytes);
    a.count = ints; a.flag = ints + 1; a.slot_class = ints + 2;
    a.cls = cls; a.s_off = s_off; a.s_val = s_val; a.s_cnt = s_cnt;
    const dim3 grid(blocks_for(L.nloc, 256)), blk(256);
    int h[2] = {0, 0};
    int rc = [&]() -> int {
        switch (L.R) {
            case 1: hipLaunchKernelGGL(scls_insert<1>, grid, blk, 0, c->stream, a); hipLaunchKernelGGL(scls_assign<1>, dim3(1), blk, 0, c->stream, a);
                    hipLaunchKernelGGL(scls_encode<1>, grid, blk, 0, c->stream, a); break;
            case 2: hipLaunchKernelGGL(scls_insert<2>, grid, blk, 0, c->stream, a); hipLaunchKernelGGL(scls_assign<2>, dim3(1), blk, 0, c->stream, a);
                    hipLaunchKernelGGL(scls_encode<2>, grid, blk, 0, c->stream, a); break;
            default: hipLaunchKernelGGL(scls_insert<4>, grid, blk, 0, c->stream, a); hipLaunchKernelGGL(scls_assign<4>, dim3(1), blk, 0, c->stream, a);
                     hipLaunchKernelGGL(scls_encode<4>, grid, blk, 0, c->stream, a); break;
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, ints, 2 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return 0;
    }();
    if (rc || h[0] > 255 || h[1]) {
        dev_free(c, cls, crows); dev_free(c, s_off, (size_t)256 * L.W); dev_free(c, s_val, (size_t)256 * L.W); dev_free(c, s_cnt, 256);
        return rc;
    }
    L.scls = cls; L.s_off = s_off; L.s_val = s_val; L.s_cnt = s_cnt; L.nscls = h[0];
    return prepare_lat_march(c, L);
}

// Symmetric diagonal storage: possible when the level is offset-coded, its offsets come in +/- pairs
// and every lower entry equals its transposed partner bit for bit (mg_kernels.hip.h, sdia_*).
// The coarsest level keeps the coded form (the direct solver reads it).
int repack_sdia(mg_context* c, Level& L, int level) {
    if (!c->use_sdia || !L.coded || level == 0 || L.flat) return 0;
    std::vector<int> offs(256);
    HIP_TRY(hipMemcpy(offs.data(), L.offsets, 256 * sizeof(int), hipMemcpyDeviceToHost));
    offs.resize(L.ntable);
    std::vector<int> up{0};
    for (int o : offs) {
        if (o > 0) up.push_back(o);
        if (o != 0 && std::find(offs.begin(), offs.end(), -o) == offs.end()) return 0;     // not a symmetric pattern
    }
    const int wu = (int)up.size();
    if (2 * wu - 1 != L.ntable || wu > 8) return 0;
    if (up.back() > vec_reach(L)) return 0;                     // x slack would not cover the reach
    const int wu_t = wu <= 3 ? 3 : (wu <= 4 ? 4 : 8);          // template width actually launched
    const int64_t S = (int64_t)WAVE * L.R;
    const int64_t mlead = ((up.back() + S - 1) / S) * S;
    const int64_t mslices = (L.nloc + mlead + S - 1) / S;
    SdiaArgs a{};
    a.vals = L.vals; a.codes = L.codes; a.offsets = L.offsets; a.W = L.W; a.WU = wu_t; a.NU = wu; a.dcode = L.dcode;
    for (int t = 0; t < 8; ++t) a.up[t] = t < wu ? up[t] : 0;
    a.nloc = L.nloc; a.mlead = mlead;
    double* dvals = nullptr;
    MG_TRY(dev_alloc(c, &dvals, (size_t)mslices * wu_t * S));
    a.dvals = dvals;
    int* d_flag = nullptr;              // flag | pad | report[2] (first asymmetric row + 1, largest ulp distance of a pair)
    int flag = 0;
    unsigned long long report[2] = {~0ull, 0ull};
    int qbits = c->storage_qbits;
    int rc = [&]() -> int {
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_flag), 8 + 2 * sizeof(unsigned long long)));
        unsigned long long* d_report = reinterpret_cast<unsigned long long*>(d_flag + 2);
        HIP_TRY(hipMemsetAsync(dvals, 0, (size_t)mslices * wu_t * S * sizeof(double), c->stream));
        const dim3 grid(blocks_for(L.nloc, 256)), blk(256);
        switch (L.R) {
            case 1: hipLaunchKernelGGL(sdia_fill<1>, grid, blk, 0, c->stream, a); break;
            case 2: hipLaunchKernelGGL(sdia_fill<2>, grid, blk, 0, c->stream, a); break;
            default: hipLaunchKernelGGL(sdia_fill<4>, grid, blk, 0, c->stream, a); break;
        }
        for (int attempt = 0; attempt < 2; ++attempt) {
            // ("storage_auto": pairs that differ by at most 4 units in the last place -- round-off of an assembly that sums its
            //  element contributions in varying order -- get a second try with that tolerance; the upper half of a pair is kept)
            HIP_TRY(hipMemsetAsync(d_flag, 0, 8, c->stream));
            HIP_TRY(hipMemcpyAsync(d_report, report, sizeof(report), hipMemcpyHostToDevice, c->stream));
            switch (L.R) {
                case 1: hipLaunchKernelGGL(sdia_check<1>, grid, blk, 0, c->stream, a, d_flag, qbits, d_report); break;
                case 2: hipLaunchKernelGGL(sdia_check<2>, grid, blk, 0, c->stream, a, d_flag, qbits, d_report); break;
                default: hipLaunchKernelGGL(sdia_check<4>, grid, blk, 0, c->stream, a, d_flag, qbits, d_report); break;
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(&flag, d_flag, sizeof(int), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipMemcpyAsync(report, d_report, sizeof(report), hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(hipStreamSynchronize(c->stream));
            if (!(flag && attempt == 0 && c->storage_auto && qbits == 0 && report[1] <= 4)) break;
            qbits = 2;
            report[0] = ~0ull; report[1] = 0ull;
        }
        return 0;
    }();
    (void)hipFree(d_flag);
    if (!rc) {
        L.rep_first_asym = report[0] == ~0ull ? -1 : (int64_t)report[0] - 1;
        L.rep_max_ulps = report[1] >= (1ull << 62) ? -1 : (int64_t)report[1];
        L.rep_sym = flag ? 0 : (report[0] == ~0ull ? 1 : 2);
        L.rep_sym_qbits = flag ? 0 : qbits;
    }
    if (rc || flag) {                                           // not symmetric: keep the coded form
        dev_free(c, dvals, (size_t)mslices * wu_t * S);
        return rc;
    }
    L.dvals = dvals; L.sdia = true; L.wu = wu_t; L.mlead = mlead; L.mslices = mslices;
    for (int t = 0; t < 8; ++t) L.up[t] = t < wu ? up[t] : 0;
    // padded template slots (wu < wu_t) hold zeros and offset 0: they add 0 * x[row]
    dev_free(c, L.vals, (size_t)L.nslices * L.W * S);
    dev_free(c, L.codes, (size_t)L.nslices * ((L.W + 7) / 8) * S);
    return build_row_classes(c, L);
}

int finish_level(mg_context* c, Level& L) {
    MG_TRY(alloc_level_vectors(c, L));
    L.set = true;
    return 0;
}

int alloc_ell(mg_context* c, Level& L) {
    L.R = c->rows_per_lane;
    L.nslices = (L.nloc + (int64_t)WAVE * L.R - 1) / ((int64_t)WAVE * L.R);
    const size_t ell = (size_t)L.nslices * L.W * (WAVE * L.R);
    MG_TRY(dev_alloc(c, &L.vals, ell));
    MG_TRY(dev_alloc(c, &L.cols, ell));
    MG_TRY(dev_alloc(c, &L.dinv, (size_t)L.nslices * WAVE * L.R));
    const unsigned nb = blocks_for((int64_t)ell, 256);
    switch (L.R) {
        case 1: hipLaunchKernelGGL(ell_fill_padding<1>, dim3(nb), dim3(256), 0, c->stream, L.vals, L.cols, L.nslices, L.W, L.nloc, L.g.lead); break;
        case 2: hipLaunchKernelGGL(ell_fill_padding<2>, dim3(nb), dim3(256), 0, c->stream, L.vals, L.cols, L.nslices, L.W, L.nloc, L.g.lead); break;
        case 4: hipLaunchKernelGGL(ell_fill_padding<4>, dim3(nb), dim3(256), 0, c->stream, L.vals, L.cols, L.nslices, L.W, L.nloc, L.g.lead); break;
        default: return fail("rows_per_lane must be 1, 2 or 4");
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

void sorted_offsets(int dim, int out[15][3], int* n) {
    std::vector<std::array<int, 3>> offs;
    if (dim == 2) {
        const int b[3][2] = {{1, 0}, {0, 1}, {1, 1}};
        offs.push_back({0, 0, 0});
        for (auto& o : b) {
            offs.push_back({o[0], 0, o[1]});
            offs.push_back({-o[0], 0, -o[1]});
        }
    } else {
        const int b[7][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {1, 1, 0}, {1, 0, 1}, {0, 1, 1}, {1, 1, 1}};
        offs.push_back({0, 0, 0});
        for (auto& o : b) {
            offs.push_back({o[0], o[1], o[2]});
            offs.push_back({-o[0], -o[1], -o[2]});
        }
    }
    std::sort(offs.begin(), offs.end(), [](const std::array<int, 3>& a, const std::array<int, 3>& b) {
        if (a[2] != b[2]) return a[2] < b[2];
        if (a[1] != b[1]) return a[1] < b[1];
        return a[0] < b[0];
    });
    *n = (int)offs.size();
    for (int t = 0; t < *n; ++t)
        for (int d = 0; d < 3; ++d) out[t][d] = offs[t][d];
}

}  // namespace

// =====================================================================================================
extern "C" {

const char* mg_last_error(void) { return g_err.c_str(); }

int mg_create(int n_levels, int dim, int device, mg_handle* out) {
    if (!out) return fail("null output handle");
    *out = nullptr;
    if (n_levels < 1 || n_levels > 32) return fail("n_levels must be in 1..32");
    if (dim != 2 && dim != 3) return fail("dim must be 2 or 3");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("no HIP device visible: this library has no CPU path");
    if (device < 0 || device >= ndev) return fail("device index out of range");
    HIP_TRY(hipSetDevice(device));
    mg_context* c = new mg_context();
    if (const char* e = std::getenv("MG_COMM_PRIORITY")) c->comm_priority = std::atoi(e) != 0;      // experiments only
    c->dim = dim;
    c->nlev = n_levels;
    c->device = device;
    c->L.resize(n_levels);
    HIP_TRY(hipGetDeviceProperties(&c->prop, device));
    HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    {
        // the communication stream outranks the main one: the exchange chain of a slab sweep runs beside a pass whose
        // workgroups fill every CU (measured on one slab of eight it makes no difference: 53.2 against 53.7 ms per cycle)
        int least = 0, greatest = 0;
        HIP_TRY(hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(hipStreamCreateWithPriority(&c->comm_stream, hipStreamNonBlocking, c->comm_priority ? greatest : least));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ev_boundary, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->ev_halo, hipEventDisableTiming));
    MG_TRY(dev_alloc(c, &c->partials, 2 * kMaxParts));
    MG_TRY(dev_alloc(c, &c->scalars, 8));
    MG_TRY(dev_alloc(c, &c->done, 1));
    HIP_TRY(hipMemsetAsync(c->done, 0, sizeof(int), c->stream));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_scalars), 16 * sizeof(double), hipHostMallocDefault));
    *out = c;
    return 0;
}

int mg_destroy(mg_handle c) {
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    drop_graphs(c);
    for (auto& L : c->L) free_level(c, L);
    free_level(c, c->mass);
    (void)hipFree(c->mass_out.raw); (void)hipFree(c->diff.raw); (void)hipFree(c->uexact.raw);
    (void)hipFree(c->ptab_count); (void)hipFree(c->ptab_off); (void)hipFree(c->ptab_w);
    (void)hipFree(c->rtab_count); (void)hipFree(c->rtab_off); (void)hipFree(c->rtab_w);
    (void)hipFree(c->partials);
    (void)hipFree(c->scalars);
    (void)hipFree(c->done);
    (void)hipFree(c->pcg_r);
    (void)hipFree(c->pcg_z);
    (void)hipFree(c->pcg_q);
    (void)hipFree(c->pcg_p.raw);
    (void)hipFree(c->pcg_part_a);
    (void)hipFree(c->pcg_part_b);
    (void)hipFree(c->stage);
    if (c->h_scalars) (void)hipHostFree(c->h_scalars);
    if (c->comm.h_stage) (void)hipHostFree(c->comm.h_stage);
    if (c->comm.nccl) g_rccl.CommDestroy(c->comm.nccl);
    if (c->ev_boundary) (void)hipEventDestroy(c->ev_boundary);
    if (c->ev_halo) (void)hipEventDestroy(c->ev_halo);
    if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
    (void)hipStreamDestroy(c->stream);
    delete c;
    return 0;
}

int mg_device_info(mg_handle c, char* buf, size_t buflen) {
    if (!c || !buf || buflen == 0) return fail("bad arguments");
    snprintf(buf, buflen, "%s %s %d CUs %.1f GiB", c->prop.name, c->prop.gcnArchName, c->prop.multiProcessorCount,
             (double)c->prop.totalGlobalMem / (1024.0 * 1024.0 * 1024.0));
    return 0;
}

int mg_comm_unique_id(void* id_out, size_t id_bytes) {
    if (!id_out || id_bytes < sizeof(ncclUniqueId)) return fail("id buffer must hold 128 bytes");
    MG_TRY(load_rccl());
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return 0;
}

int mg_comm_selftest(int device) {
    MG_TRY(load_rccl());
    HIP_TRY(hipSetDevice(device));
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    ncclComm_t comm = nullptr;
    NCCL_TRY(g_rccl.CommInitRank(&comm, 1, id, 0));
    hipStream_t s = nullptr;
    double* d = nullptr;
    const int n = 1024;
    std::vector<double> h(3 * n);
    for (int i = 0; i < n; ++i) { h[i] = 1.0 + i; h[n + i] = 0.0; h[2 * n + i] = -1.0; }
    int rc = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), 3 * n * sizeof(double)));
        HIP_TRY(hipMemcpyAsync(d, h.data(), 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
        NCCL_TRY(g_rccl.AllReduce(d, d, n, ncclDouble, ncclSum, comm, s));            // one rank: unchanged
        NCCL_TRY(g_rccl.GroupStart());
        NCCL_TRY(g_rccl.Broadcast(d, d, n, ncclDouble, 0, comm, s));
        NCCL_TRY(g_rccl.GroupEnd());
        NCCL_TRY(g_rccl.GroupStart());
        NCCL_TRY(g_rccl.Send(d, n, ncclDouble, 0, comm, s));                          // to self
        NCCL_TRY(g_rccl.Recv(d + n, n, ncclDouble, 0, comm, s));
        NCCL_TRY(g_rccl.GroupEnd());
        HIP_TRY(hipMemcpyAsync(h.data(), d, 3 * n * sizeof(double), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        for (int i = 0; i < n; ++i)
            if (h[i] != 1.0 + i || h[n + i] != 1.0 + i || h[2 * n + i] != -1.0) return fail("RCCL self-test: wrong data");
        return 0;
    }();
    if (d) (void)hipFree(d);
    if (s) (void)hipStreamDestroy(s);
    g_rccl.CommDestroy(comm);
    return rc;
}

static int comm_common(mg_handle c, int rank, int world, int64_t replicate_below) {
    if (!c) return fail("null handle");
    if (world < 1 || rank < 0 || rank >= world) return fail("bad rank/world");
    for (auto& L : c->L)
        if (L.set) return fail("mg_set_comm must precede level set-up");
    c->comm.rank = rank;
    c->comm.world = world;
    c->comm.replicate_below = replicate_below;
    return 0;
}

int mg_set_comm(mg_handle c, int rank, int world, const void* nccl_unique_id, size_t id_bytes, int64_t replicate_below) {
    MG_TRY(comm_common(c, rank, world, replicate_below));
    if (world == 1) return 0;
    if (!nccl_unique_id || id_bytes < sizeof(ncclUniqueId)) return fail("missing RCCL unique id");
    MG_TRY(load_rccl());
    HIP_TRY(hipSetDevice(c->device));
    ncclUniqueId id;
    memcpy(&id, nccl_unique_id, sizeof(id));
    NCCL_TRY(g_rccl.CommInitRank(&c->comm.nccl, world, id, rank));
    return 0;
}

int mg_set_comm_callbacks(mg_handle c, int rank, int world, mg_exchange_fn ex, mg_allreduce_fn ar, mg_allgatherv_fn ag,
                          void* user, int64_t replicate_below) {
    MG_TRY(comm_common(c, rank, world, replicate_below));
    if (world > 1 && (!ex || !ar || !ag)) return fail("all three callbacks are required");
    c->comm.ex = ex;
    c->comm.ar = ar;
    c->comm.ag = ag;
    c->comm.user = user;
    return 0;
}

int mg_set_params(mg_handle c, int mu1, int mu2, double omega, int restriction, int smoother, double coarse_rtol,
                  int coarse_maxit, int keep_err) {
    if (!c) return fail("null handle");
    if (mu1 < 0 || mu2 < 0) return fail("mu1/mu2 must be >= 0");
    if (restriction != MG_RESTRICT_INJECTION && restriction != MG_RESTRICT_FULL_WEIGHTING && restriction != MG_RESTRICT_TABLE)
        return fail("unknown restriction");
    if (smoother != MG_SMOOTH_JACOBI && smoother != MG_SMOOTH_RBGS && smoother != MG_SMOOTH_MCGS) return fail("unknown smoother");
    const double rtol = coarse_rtol > 0 ? coarse_rtol : c->coarse_rtol;
    const int maxit = coarse_maxit > 0 ? coarse_maxit : c->coarse_maxit;
    // captured V-cycles (vcycle_graphed) are keyed on the epoch: callers such as the Python shim set the same
    // parameters before every cycle, which must not throw the graphs away
    if (mu1 == c->mu1 && mu2 == c->mu2 && omega == c->omega && restriction == c->restriction && smoother == c->smoother &&
        rtol == c->coarse_rtol && maxit == c->coarse_maxit && keep_err == c->keep_err)
        return 0;
    ++c->epoch;
    c->mu1 = mu1; c->mu2 = mu2; c->omega = omega; c->restriction = restriction; c->smoother = smoother;
    c->coarse_rtol = rtol; c->coarse_maxit = maxit;
    c->keep_err = keep_err;
    return 0;
}

int mg_set_prolongation_table(mg_handle c, const int* count, const int* offsets, const double* weights) {
    if (!c) return fail("null handle");
    HIP_TRY(hipSetDevice(c->device));
    ++c->epoch;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->ptab_count); (void)hipFree(c->ptab_off); (void)hipFree(c->ptab_w);
    c->ptab_count = nullptr; c->ptab_off = nullptr; c->ptab_w = nullptr;
    if (!count && !offsets && !weights) return 0;                 // back to the reference's interpolation
    if (!count || !offsets || !weights) return fail("null table");
    for (int r = 0; r < 64; ++r) {
        if (count[r] < 0 || count[r] > 10) return fail("a residue has more than 10 entries");
        for (int e = 0; e < count[r]; ++e)
            for (int d = 0; d < 3; ++d)
                if (offsets[(r * 10 + e) * 3 + d] < 0 || offsets[(r * 10 + e) * 3 + d] > 2) return fail("table offsets must be 0..2");
    }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ptab_count), 64 * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ptab_off), 64 * 10 * 3 * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->ptab_w), 64 * 10 * sizeof(double)));
    HIP_TRY(hipMemcpy(c->ptab_count, count, 64 * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->ptab_off, offsets, 64 * 10 * 3 * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->ptab_w, weights, 64 * 10 * sizeof(double), hipMemcpyHostToDevice));
    return 0;
}

int mg_set_restriction_table(mg_handle c, int max_entries, const int* count, const int* offsets, const double* weights) {
    if (!c) return fail("null handle");
    HIP_TRY(hipSetDevice(c->device));
    ++c->epoch;
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(c->rtab_count); (void)hipFree(c->rtab_off); (void)hipFree(c->rtab_w);
    c->rtab_count = nullptr; c->rtab_off = nullptr; c->rtab_w = nullptr; c->rtab_m = 0;
    if (!count && !offsets && !weights) return 0;
    if (!count || !offsets || !weights || max_entries < 1 || max_entries > 4096) return fail("bad restriction table");
    for (int t = 0; t < 8; ++t) {
        if (count[t] < 0 || count[t] > max_entries) return fail("a type has more entries than max_entries");
        for (int e = 0; e < count[t]; ++e)
            for (int d = 0; d < 3; ++d)
                if (std::abs(offsets[((size_t)t * max_entries + e) * 3 + d]) > 4) return fail("table offsets must be within +-4");
    }
    const size_t n = (size_t)8 * max_entries;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->rtab_count), 8 * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->rtab_off), n * 3 * sizeof(int)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c->rtab_w), n * sizeof(double)));
    HIP_TRY(hipMemcpy(c->rtab_count, count, 8 * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->rtab_off, offsets, n * 3 * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->rtab_w, weights, n * sizeof(double), hipMemcpyHostToDevice));
    c->rtab_m = max_entries;
    return 0;
}

int mg_set_tuning(mg_handle c, const char* key, int64_t value) {
    if (!c || !key) return fail("bad arguments");
    ++c->epoch;
    const std::string k(key);
    if (k == "graph") {
        c->use_graph = value != 0;
        return 0;
    }
    if (k == "slab_pair_form") {
        if (value != 0 && value != 1) return fail("slab_pair_form must be 0 or 1");
        c->slab_pair_form = (int)value;
        return 0;
    }
    if (k == "graph_comm") {
        c->graph_comm = value != 0;
        return 0;
    }
    if (k == "rows_per_lane") {
        if (value != 1 && value != 2 && value != 4) return fail("rows_per_lane must be 1, 2 or 4");
        for (auto& L : c->L)
            if (L.set) return fail("rows_per_lane must be chosen before level set-up");
        c->rows_per_lane = (int)value;
    } else if (k == "xcd_chunk") {
        if (value < 1 || value > 4096) return fail("xcd_chunk out of range");
        c->chunk = (unsigned)value;
    } else if (k == "offset_codes") {
        for (auto& L : c->L)
            if (L.set) return fail("offset_codes must be chosen before level set-up");
        c->use_codes = value != 0;
    } else if (k == "symmetric_storage") {
        for (auto& L : c->L)
            if (L.set) return fail("symmetric_storage must be chosen before level set-up");
        c->use_sdia = value != 0;
    } else if (k == "storage_ulps") {
        for (auto& L : c->L)
            if (L.set) return fail("storage_ulps must be chosen before level set-up");
        if (value < 0 || value > 4096) return fail("storage_ulps must be in 0..4096");
        int q = 0;
        while ((1ll << q) < value) ++q;
        c->storage_qbits = value > 0 ? std::max(1, q) : 0;
    } else if (k == "fuse_block") {
        if (value < 0 || value > 2) return fail("fuse_block must be 0, 1 or 2");
        c->fuse_block = (int)value;
    } else if (k == "fuse_block_min_rows") {
        c->fuse_block_min_rows = value;
    } else if (k == "fuse_block_max_rows") {
        c->fuse_block_max_rows = value;
    } else if (k == "fuse_block_k") {
        if (value != 0 && (value < 2 || value > 4)) return fail("fuse_block_k must be 0 or 2..4");
        c->fuse_block_k = (int)value;
    } else if (k == "fuse_block_ez") {
        if (value != 0 && value != 11 && value != 19) return fail("fuse_block_ez must be 0, 11 or 19");
        c->fuse_block_ez = (int)value;
    } else if (k == "direct_block_rows") {
        if (value < 1 || value > 2304) return fail("direct_block_rows must be in 1..2304");
        if (c->direct.tried) return fail("direct_block_rows must be chosen before the coarsest level is factored");
        c->direct_block_rows = (int)value;
    } else if (k == "fuse_2d_lines") {
        if (value != 0 && value != 16 && value != 32 && value != 64) return fail("fuse_2d_lines must be 0, 16, 32 or 64");
        c->fuse_2d_lines = (int)value;
    } else if (k == "gen_odd_rows") {
        if (value < 0 || value > 10000) return fail("gen_odd_rows: rows in 10000");
        c->gen_odd_rows = (int)value;
    } else if (k == "row_escape") {
        for (auto& L : c->L)
            if (L.set) return fail("row_escape must be chosen before level set-up");
        c->cls_escape = value != 0;
    } else if (k == "storage_auto") {
        for (auto& L : c->L)
            if (L.set) return fail("storage_auto must be chosen before level set-up");
        c->storage_auto = value != 0;
    } else if (k == "row_classes") {
        for (auto& L : c->L)
            if (L.set) return fail("row_classes must be chosen before level set-up");
        c->use_classes = value != 0;
    } else if (k == "strip_slices") {
        if (value < 0 || value > 4096 || value % 4) return fail("strip_slices must be a multiple of 4 in 0..4096");
        c->strip_slices = (int)value;
    } else if (k == "nontemporal") {
        c->nontemporal = value != 0;
    } else if (k == "lds_pad") {
        if (value < 0 || value > 160 * 1024) return fail("lds_pad out of range");
        c->lds_pad = (int)value;
    } else if (k == "halo_planes") {
        if (value != 1 && value != 2) return fail("halo_planes must be 1 or 2");
        for (auto& L : c->L)
            if (L.set) return fail("halo_planes must be chosen before level set-up");
        c->halo_planes = (int)value;
    } else if (k == "halo_depth") {
        if (value < 0 || value > 5) return fail("halo_depth must be in 0..5");
        for (auto& L : c->L)
            if (L.set) return fail("halo_depth must be chosen before level set-up");
        c->halo_depth = (int)value;
    } else if (k == "overlap") {
        c->overlap = value != 0;
    } else if (k == "overlap_min_rows") {
        c->overlap_min_rows = value;
    } else if (k == "require_diagonal") {
        c->require_diagonal = value != 0;
    } else if (k == "fuse_restrict") {
        c->fuse_restrict = value != 0;
    } else if (k == "fuse_sweeps") {
        c->fuse_sweeps = value != 0;
    } else if (k == "fuse_min_rows") {
        c->fuse_min_rows = value;
    } else if (k == "fuse_shape") {
        if (value < 0 || value > 3) return fail("fuse_shape must be 0..3");
        c->fuse_shape = (int)value;
    } else if (k == "march_sweeps") {
        c->march_sweeps = value != 0;
    } else if (k == "march_min_rows") {
        c->march_min_rows = value;
    } else if (k == "march_shape") {
        if (value < 0 || value > 1) return fail("march_shape must be 0 or 1");
        c->march_shape = (int)value;
    } else if (k == "fuse_small_2d_rows") {
        c->fuse_small_2d_rows = value;
    } else if (k == "fuse_small") {
        c->fuse_small = value != 0;
    } else if (k == "fuse_2d") {
        c->fuse_2d = value != 0;
    } else if (k == "fuse_2d_k") {
        if (value < 2 || value > 5) return fail("fuse_2d_k must be in 2..5");
        c->fuse_2d_k = (int)value;
    } else if (k == "lattice_march") {
        c->lattice_march = value != 0;
    } else if (k == "lattice_march_min_rows") {
        c->lattice_march_min_rows = value;
    } else if (k == "lattice_gs2") {
        c->lattice_gs2 = value != 0;
    } else if (k == "lattice_tile") {
        if (value < 0 || value > 2) return fail("lattice_tile must be 0, 1 or 2");
        c->lattice_tile = (int)value;
    } else if (k == "lattice_segments") {
        if (value < 0 || value > 4096) return fail("lattice_segments must be in 0..4096");
        c->lattice_segments = (int)value;
    } else if (k == "fuse_xcd_chunk") {
        if (value < 1 || value > 512) return fail("fuse_xcd_chunk must be in 1..512");
        c->fuse_xcd_chunk = (int)value;
    } else if (k == "cls_blocks_per_cu") {
        if (value < 1 || value > 64) return fail("cls_blocks_per_cu must be in 1..64");
        c->cls_blocks_per_cu = (int)value;
    } else if (k == "fuse_wi") {
        c->fuse_wi = (int)value;
    } else if (k == "fuse_even") {
        c->fuse_even = value != 0;
    } else if (k == "class_sweeps") {
        c->class_sweeps = value != 0;
    } else if (k == "fuse_plain_shape") {
        if (value < 0 || value > 2) return fail("fuse_plain_shape must be 0..2");
        c->fuse_plain_shape = (int)value;
    } else if (k == "fuse_plain") {
        if (value != 1 && value != 2) return fail("fuse_plain must be 1 or 2");
        c->fuse_plain = (int)value;
    } else if (k == "fuse_k") {
        if (value < 0 || value > 5) return fail("fuse_k must be in 0..5 (below 3: pairs of sweeps only)");
        c->fuse_k = (int)value;
    } else if (k == "fuse_k_shape") {
        if (value < 0 || value > 7) return fail("fuse_k_shape must be 0..7");
        c->fuse_k_shape = (int)value;
    } else if (k == "fuse_k_slab_min_rows") {
        c->fuse_k_slab_min_rows = value;
    } else if (k == "fuse_k_slab_min_sweeps") {
        if (value < 2) return fail("fuse_k_slab_min_sweeps must be at least 2");
        c->fuse_k_slab_min_sweeps = (int)value;
    } else if (k == "fuse_k_min_rows") {
        c->fuse_k_min_rows = value;
    } else if (k == "fuse_k_small_rows") {
        c->fuse_k_small_rows = value;
    } else if (k == "fuse_k5_min_rows") {
        c->fuse_k5_min_rows = value;
    } else if (k == "fuse_k4_min_rows") {
        c->fuse_k4_min_rows = value;
    } else if (k == "fuse_k_nt_store") {
        c->fuse_k_nt_store = value != 0;
    } else if (k == "fuse_k_tail") {
        if (value < 0) return fail("fuse_k_tail must be >= 0");
        c->fuse_k_tail = (int)value;            // (> 1: the number of resident workgroups to plan for -- tests)
    } else if (k == "fuse_k_small_tiles") {
        c->fuse_k_small_tiles = value != 0;
    } else if (k == "fuse_k_pf") {
        if (value != 1 && value != 2) return fail("fuse_k_pf must be 1 or 2");
        c->fuse_k_pf = (int)value;
    } else if (k == "fuse_k_dpp") {
        c->fuse_k_dpp = value != 0;
    } else if (k == "fuse_k_segments") {
        if (value < 0) return fail("fuse_k_segments must be >= 0");
        c->fuse_k_segments = (int)value;
    } else if (k == "fuse_classes") {
        c->fuse_classes = value != 0;
    } else if (k == "fuse_nontemporal") {
        c->fuse_nontemporal = value != 0;
    } else if (k == "fuse_segments") {
        if (value < 0) return fail("fuse_segments must be >= 0");
        c->fuse_segments = (int)value;
    } else if (k == "coarse_direct") {
        c->use_direct = value != 0;
        free_direct(c);
    } else if (k == "pcg_chunk") {
        if (value < 1) return fail("pcg_chunk must be positive");
        c->pcg_chunk = (int)value;
    } else {
        return fail("unknown tuning key " + k);
    }
    return 0;
}

namespace {

// Temporary device buffer that frees itself (set-up paths have many early exits).
struct DevTemp {
    void* p = nullptr;
    DevTemp() = default;
    DevTemp(const DevTemp&) = delete;
    DevTemp& operator=(const DevTemp&) = delete;
    ~DevTemp() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(&p, std::max<size_t>(1, bytes)));
        return 0;
    }
};

// grid_index[dof] (int64, caller) -> L.perm (int32, device), validated as a permutation of the nodes.
int upload_permutation(mg_context* c, Level& L, const int64_t* grid_index, int64_t n_rows) {
    if (!grid_index) return 0;
    std::vector<int> p32((size_t)n_rows);
    std::vector<char> seen((size_t)n_rows, 0);
    for (int64_t d = 0; d < n_rows; ++d) {
        const int64_t p = grid_index[d];
        if (p < 0 || p >= n_rows || seen[(size_t)p]) return fail("grid_index is not a permutation of the grid nodes");
        seen[(size_t)p] = 1;
        p32[(size_t)d] = (int)p;
    }
    MG_TRY(dev_alloc(c, &L.perm, (size_t)n_rows));
    HIP_TRY(hipMemcpy(L.perm, p32.data(), (size_t)n_rows * 4, hipMemcpyHostToDevice));
    return 0;
}

// `local_cols` (per-rank hand-off): the matrix holds this rank's owned rows only, its column indices are local ids in
// [0, n_cols) and local_cols[id] is the global lexicographic node of each (ids 0 .. n_rows-1 are the rows themselves);
// otherwise rows and columns are global DoF numbers and `grid_index` (or the identity) maps them to nodes.
int build_level_from_csr(mg_context* c, int level, Level& L, int64_t n_rows, int64_t nnz, const void* indptr,
                         int indptr_is_64, const int32_t* indices, const double* data, const int64_t* grid_index,
                         int prune_zeros, bool vectors = true, const int64_t* local_cols = nullptr, int64_t n_cols = 0) {
    // upload the hand-off
    DevTemp d_ptr, d_idx, d_val;
    const size_t ptr_bytes = (size_t)(n_rows + 1) * (indptr_is_64 ? 8 : 4);
    MG_TRY(d_ptr.alloc(ptr_bytes));
    MG_TRY(d_idx.alloc((size_t)nnz * 4));
    MG_TRY(d_val.alloc((size_t)nnz * 8));
    HIP_TRY(hipMemcpyAsync(d_ptr.p, indptr, ptr_bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_idx.p, indices, (size_t)nnz * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipMemcpyAsync(d_val.p, data, (size_t)nnz * 8, hipMemcpyHostToDevice, c->stream));
    DevTemp d_map;
    if (local_cols) {
        // vectors keep crossing in the caller's GLOBAL numbering (grid_index, n_global entries); the matrix kernels see the
        // local ids through their own map
        MG_TRY(upload_permutation(c, L, grid_index, L.n_global));
        std::vector<int> m32((size_t)n_cols);
        std::vector<char> seen((size_t)L.nloc, 0);
        const int64_t lo = L.row0 - L.halo_lo, hi = L.row0 + L.nloc + L.halo_hi;
        for (int64_t d = 0; d < n_cols; ++d) {
            const int64_t p = local_cols[d];
            if (d < n_rows) {
                if (p < L.row0 || p >= L.row0 + L.nloc || seen[(size_t)(p - L.row0)])
                    return fail("the handed-over rows are not exactly this rank's slab of the level");
                seen[(size_t)(p - L.row0)] = 1;
            } else if (p < lo || p >= hi) {
                return fail("a ghost column lies outside the slab's halo");
            }
            m32[(size_t)d] = (int)p;
        }
        MG_TRY(d_map.alloc((size_t)n_cols * sizeof(int)));
        HIP_TRY(hipMemcpyAsync(d_map.p, m32.data(), (size_t)n_cols * sizeof(int), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else {
        MG_TRY(upload_permutation(c, L, grid_index, n_rows));
    }
    CsrArgs a{};
    a.indptr = d_ptr.p; a.indptr64 = indptr_is_64; a.indices = static_cast<const int*>(d_idx.p); a.data = static_cast<const double*>(d_val.p);
    a.perm = local_cols ? static_cast<const int*>(d_map.p) : L.perm;
    a.n = n_rows; a.row0 = L.row0; a.nloc = L.nloc; a.lead = L.g.lead; a.xlen = L.xlen;
    a.prune = prune_zeros;
    // pass 1: widest kept row, kept entries, sanity flags
    unsigned long long* d_stats = reinterpret_cast<unsigned long long*>(c->partials);
    HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * sizeof(unsigned long long), c->stream));
    hipLaunchKernelGGL(csr_scan, dim3(blocks_for(n_rows, 256)), dim3(256), 0, c->stream, a, d_stats);
    unsigned long long stats[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(stats, d_stats, sizeof(stats), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (stats[2] & 1ull)
        return fail("a matrix row couples unknowns more than one grid plane apart (not a slab-local stencil)");
    if ((stats[2] & 2ull) && c->require_diagonal) return fail("a matrix row has a zero or missing diagonal");
    L.W = (int)std::max<unsigned long long>(1, stats[0]);
    L.nnz_stored = stats[1];
    // pass 2: tiles
    MG_TRY(alloc_ell(c, L));
    const dim3 g1(blocks_for(n_rows, 256)), b1(256);
    switch (L.R) {
        case 1: hipLaunchKernelGGL(csr_to_ell<1>, g1, b1, 0, c->stream, a, L.vals, L.cols, L.dinv, L.W); break;
        case 2: hipLaunchKernelGGL(csr_to_ell<2>, g1, b1, 0, c->stream, a, L.vals, L.cols, L.dinv, L.W); break;
        default: hipLaunchKernelGGL(csr_to_ell<4>, g1, b1, 0, c->stream, a, L.vals, L.cols, L.dinv, L.W); break;
    }
    HIP_TRY(hipGetLastError());
    // true non-zeros among the kept entries (== kept when pruned); counted on the caller's copy (set-up only)
    L.nnz_nonzero = L.nnz_stored;
    if (!prune_zeros && (!c->comm.active() || L.replicated) && !local_cols) {
        unsigned long long nz = 0;
        for (int64_t q = 0; q < nnz; ++q) nz += data[q] != 0.0;
        L.nnz_nonzero = nz;
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    MG_TRY(encode_level(c, L));
    MG_TRY(repack_sdia(c, L, level));
    MG_TRY(build_stencil_classes(c, L));
    L.has_matrix = true;
    if (!vectors) { L.set = true; return 0; }
    return finish_level(c, L);
}

}  // namespace

int mg_set_level_csr(mg_handle c, int level, int N, int64_t n_rows, int64_t nnz, const void* indptr, int indptr_is_64,
                     const int32_t* indices, const double* data, const int64_t* grid_index, int prune_zeros) {
    MG_TRY(check_level(c, level, false));
    if (!indptr || !indices || !data) return fail("null CSR arrays");
    if (n_rows <= 0 || nnz < 0) return fail("bad matrix dimensions");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    free_level(c, L);
    MG_TRY(setup_geometry(c, L, level, N, n_rows));
    if (n_rows != L.n_global)
        return fail("matrix has " + std::to_string(n_rows) + " rows, grid has " + std::to_string(L.n_global));
    const int rc = build_level_from_csr(c, level, L, n_rows, nnz, indptr, indptr_is_64, indices, data, grid_index,
                                        prune_zeros);
    if (rc) {
        const std::string why = g_err;
        free_level(c, L);           // never leave a half-built level behind
        g_err = why;
    }
    return rc;
}

int mg_level_slab(mg_handle c, int level, int N, int64_t* row0, int64_t* n_local, int64_t* halo_lo, int64_t* halo_hi) {
    MG_TRY(check_level(c, level, false));
    if (N <= 0) return fail("elements_per_dim must be positive");
    Level tmp;
    MG_TRY(setup_geometry(c, tmp, level, N));
    if (row0) *row0 = tmp.row0;
    if (n_local) *n_local = tmp.nloc;
    // (what the rows may couple to: "halo_planes" planes, however many the vectors have room for)
    if (halo_lo) *halo_lo = tmp.halo_lo ? (int64_t)c->halo_planes * tmp.g.plane : 0;
    if (halo_hi) *halo_hi = tmp.halo_hi ? (int64_t)c->halo_planes * tmp.g.plane : 0;
    return 0;
}

int mg_set_level_csr_local(mg_handle c, int level, int N, int64_t n_rows, int64_t n_cols, int64_t nnz, const void* indptr,
                           int indptr_is_64, const int32_t* indices, const double* data, const int64_t* col_nodes,
                           const int64_t* grid_index, int prune_zeros) {
    MG_TRY(check_level(c, level, false));
    if (!indptr || !indices || !data || !col_nodes) return fail("null CSR arrays");
    if (N <= 0 || n_rows <= 0 || n_cols < n_rows || nnz < 0) return fail("bad matrix dimensions");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    free_level(c, L);
    MG_TRY(setup_geometry(c, L, level, N));
    if (n_rows != L.nloc)
        return fail("the rank owns " + std::to_string(L.nloc) + " rows of this level (mg_level_slab), " + std::to_string(n_rows) +
                    " were handed over");
    const int rc = build_level_from_csr(c, level, L, n_rows, nnz, indptr, indptr_is_64, indices, data, grid_index, prune_zeros,
                                        true, col_nodes, n_cols);
    if (rc) {
        const std::string why = g_err;
        free_level(c, L);
        g_err = why;
    }
    return rc;
}

int mg_set_level_grid(mg_handle c, int level, int N, int64_t n_rows, const int64_t* grid_index) {
    MG_TRY(check_level(c, level, false));
    if (N < 0) return fail("elements_per_dim must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    free_level(c, L);
    MG_TRY(setup_geometry(c, L, level, N, n_rows));
    if (n_rows != L.n_global)
        return fail("vector has " + std::to_string(n_rows) + " entries, grid has " + std::to_string(L.n_global));
    int rc = upload_permutation(c, L, grid_index, n_rows);
    if (!rc) rc = finish_level(c, L);
    if (rc) {
        const std::string why = g_err;
        free_level(c, L);
        g_err = why;
    }
    return rc;
}

int mg_gen_poisson_level(mg_handle c, int level, int N, int prune_zeros) {
    MG_TRY(check_level(c, level, false));
    if (N <= 0) return fail("elements_per_dim must be positive");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    free_level(c, L);
    MG_TRY(setup_geometry(c, L, level, N));
    GenArgs a{};
    a.g = L.g; a.N = N; a.dim = c->dim; a.prune = prune_zeros; a.odd = c->gen_odd_rows;
    a.h = 1.0 / (double)N;
    a.w = c->dim == 2 ? 1.0 : a.h;
    a.diag = c->dim == 2 ? 4.0 : 6.0 * a.h;
    a.fh = (c->dim == 2 ? -6.0 : -12.0) * std::pow(a.h, (double)c->dim);
    sorted_offsets(c->dim, a.off, &a.noff);
    L.W = prune_zeros ? (c->dim == 2 ? 5 : 7) : a.noff;
    a.W = L.W;
    MG_TRY(alloc_ell(c, L));
    MG_TRY(alloc_level_vectors(c, L));
    unsigned long long* d_counts = reinterpret_cast<unsigned long long*>(c->partials);
    HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(unsigned long long), c->stream));
    const dim3 grid = grid3(L.g, L.g.nk), blk(kPlaneBlock);
    switch (L.R) {
        case 1: hipLaunchKernelGGL(gen_poisson<1>, grid, blk, 0, c->stream, a, L.vals, L.cols, L.dinv, L.f.rows, d_counts); break;
        case 2: hipLaunchKernelGGL(gen_poisson<2>, grid, blk, 0, c->stream, a, L.vals, L.cols, L.dinv, L.f.rows, d_counts); break;
        default: hipLaunchKernelGGL(gen_poisson<4>, grid, blk, 0, c->stream, a, L.vals, L.cols, L.dinv, L.f.rows, d_counts); break;
    }
    HIP_TRY(hipGetLastError());
    unsigned long long counts[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    L.nnz_stored = counts[0];
    L.nnz_nonzero = counts[1];
    MG_TRY(encode_level(c, L));
    MG_TRY(repack_sdia(c, L, level));
    MG_TRY(build_stencil_classes(c, L));
    // the generated right-hand side is also this level's true right-hand side for mg_fmg
    if (level + 1 < c->nlev) {
        MG_TRY(vec_alloc(c, L, &L.ftrue));
        HIP_TRY(hipMemcpyAsync(L.ftrue.base, L.f.base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    L.set = true;
    L.has_matrix = true;
    return 0;
}

int mg_gen_lattice_level(mg_handle c, int level, int N, int width, const int* count, const int* offsets, const double* values,
                         const double* load) {
    MG_TRY(check_level(c, level, false));
    if (N <= 0 || (N & 1)) return fail("lattice levels need an even, positive number of lattice steps per dimension");
    if (!count || !offsets || !values || !load) return fail("null stencil tables");
    if (width < 1 || width > LAT_MAX) return fail("stencil width out of range");
    if (c->comm.active() && c->halo_planes < 2)
        return fail("lattice levels reach two planes: set mg_set_tuning(\"halo_planes\", 2) before mg_set_comm / level set-up");
    for (int p = 0; p < 8; ++p)
        if (count[p] < 0 || count[p] > width) return fail("a class has more entries than the stated width");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    free_level(c, L);
    MG_TRY(setup_geometry(c, L, level, N));
    DevTemp d_count, d_off, d_val, d_load;
    MG_TRY(d_count.alloc(8 * sizeof(int)));
    MG_TRY(d_off.alloc((size_t)8 * LAT_MAX * 3 * sizeof(int)));
    MG_TRY(d_val.alloc((size_t)8 * LAT_MAX * sizeof(double)));
    MG_TRY(d_load.alloc(8 * sizeof(double)));
    HIP_TRY(hipMemcpy(d_count.p, count, 8 * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_off.p, offsets, (size_t)8 * LAT_MAX * 3 * sizeof(int), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_val.p, values, (size_t)8 * LAT_MAX * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(d_load.p, load, 8 * sizeof(double), hipMemcpyHostToDevice));
    LatticeArgs a{};
    a.g = L.g; a.N = N; a.dim = c->dim; a.W = width;
    a.count = static_cast<const int*>(d_count.p); a.off = static_cast<const int*>(d_off.p);
    a.val = static_cast<const double*>(d_val.p); a.load = static_cast<const double*>(d_load.p);
    L.W = width;
    MG_TRY(alloc_ell(c, L));
    MG_TRY(alloc_level_vectors(c, L));
    unsigned long long* d_counts = reinterpret_cast<unsigned long long*>(c->partials);
    HIP_TRY(hipMemsetAsync(d_counts, 0, 2 * sizeof(unsigned long long), c->stream));
    const dim3 grid = grid3(L.g, L.g.nk), blk(kPlaneBlock);
    switch (L.R) {
        case 1: hipLaunchKernelGGL(gen_lattice<1>, grid, blk, 0, c->stream, a, L.vals, L.cols, L.dinv, L.f.rows, d_counts); break;
        case 2: hipLaunchKernelGGL(gen_lattice<2>, grid, blk, 0, c->stream, a, L.vals, L.cols, L.dinv, L.f.rows, d_counts); break;
        default: hipLaunchKernelGGL(gen_lattice<4>, grid, blk, 0, c->stream, a, L.vals, L.cols, L.dinv, L.f.rows, d_counts); break;
    }
    HIP_TRY(hipGetLastError());
    unsigned long long counts[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    L.nnz_stored = counts[0];
    L.nnz_nonzero = counts[1];
    MG_TRY(encode_level(c, L));
    MG_TRY(repack_sdia(c, L, level));
    MG_TRY(build_stencil_classes(c, L));
    if (level + 1 < c->nlev) {
        MG_TRY(vec_alloc(c, L, &L.ftrue));
        HIP_TRY(hipMemcpyAsync(L.ftrue.base, L.f.base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    L.set = true;
    L.has_matrix = true;
    return 0;
}

int mg_jacobi_split(int device, int64_t n_rows, int64_t nnz, const void* indptr, int indptr_is_64,
                    const int32_t* indices, const double* data, double* dinv, double* scaled, unsigned char* keep) {
    if (!indptr || !indices || !data || !dinv || !scaled || !keep) return fail("null argument");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail("no HIP device visible: this library has no CPU path");
    HIP_TRY(hipSetDevice(device));
    void* d_ptr = nullptr;
    int* d_idx = nullptr;
    double *d_val = nullptr, *d_dinv = nullptr, *d_scaled = nullptr;
    unsigned char* d_keep = nullptr;
    const size_t ptr_bytes = (size_t)(n_rows + 1) * (indptr_is_64 ? 8 : 4);
    const size_t nz = std::max<size_t>(1, (size_t)nnz);
    auto cleanup = [&]() {
        (void)hipFree(d_ptr); (void)hipFree(d_idx); (void)hipFree(d_val);
        (void)hipFree(d_dinv); (void)hipFree(d_scaled); (void)hipFree(d_keep);
    };
    int rc = [&]() -> int {
        HIP_TRY(hipMalloc(&d_ptr, ptr_bytes));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_idx), nz * 4));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_val), nz * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_dinv), std::max<size_t>(1, (size_t)n_rows) * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_scaled), nz * 8));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d_keep), nz));
        HIP_TRY(hipMemcpy(d_ptr, indptr, ptr_bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_idx, indices, (size_t)nnz * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_val, data, (size_t)nnz * 8, hipMemcpyHostToDevice));
        CsrArgs a{};
        a.indptr = d_ptr; a.indptr64 = indptr_is_64; a.indices = d_idx; a.data = d_val; a.n = n_rows;
        hipLaunchKernelGGL(jacobi_split, dim3(blocks_for(n_rows, 256)), dim3(256), 0, 0, a, d_dinv, d_scaled, d_keep);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpy(dinv, d_dinv, (size_t)n_rows * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(scaled, d_scaled, (size_t)nnz * 8, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(keep, d_keep, (size_t)nnz, hipMemcpyDeviceToHost));
        return 0;
    }();
    cleanup();
    return rc;
}

int mg_level_info(mg_handle c, int level, int64_t* n_global, int64_t* n_local, int64_t* row0, int64_t* nnz_stored,
                  int64_t* nnz_nonzero, int* ell_width, int* replicated, int* offset_codes) {
    MG_TRY(check_level(c, level));
    const Level& L = c->L[level];
    if (n_global) *n_global = L.n_global;
    if (n_local) *n_local = L.nloc;
    if (row0) *row0 = L.row0;
    if (nnz_stored) *nnz_stored = (int64_t)L.nnz_stored;
    if (nnz_nonzero) *nnz_nonzero = (int64_t)L.nnz_nonzero;
    if (ell_width) *ell_width = L.W;
    if (replicated) *replicated = L.replicated ? 1 : 0;
    if (offset_codes) *offset_codes = L.sdia ? -L.wu : (L.coded ? L.ntable : 0);
    return 0;
}

int mg_level_storage(mg_handle c, int level, int* symmetric, int64_t* first_asymmetric_row, int64_t* max_pair_ulps,
                     int* distinct_rows, int* ulps_used, int64_t* escape_rows) {
    MG_TRY(check_level(c, level));
    const Level& L = c->L[level];
    if (symmetric) *symmetric = L.rep_sym;
    if (first_asymmetric_row) *first_asymmetric_row = L.rep_first_asym;
    if (max_pair_ulps) *max_pair_ulps = L.rep_max_ulps;
    if (distinct_rows) *distinct_rows = L.rep_distinct;
    if (escape_rows) *escape_rows = L.cls_escape ? L.rep_escape : 0;
    if (ulps_used) *ulps_used = std::max(L.sdia ? (L.rep_sym_qbits ? 1 << L.rep_sym_qbits : 0) : 0, L.cls ? (L.rep_cls_qbits ? 1 << L.rep_cls_qbits : 0) : 0);
    return 0;
}

int mg_level_row_classes(mg_handle c, int level, int* classes) {
    MG_TRY(check_level(c, level));
    if (!classes) return fail("bad arguments");
    *classes = c->L[level].cls ? c->L[level].ncls : c->L[level].nscls;
    return 0;
}

int mg_set_vector(mg_handle c, int level, int which, const double* host) {
    MG_TRY(check_level(c, level));
    if (!host) return fail("null host vector");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    DVector* v = pick(L, which);
    if (!v) return fail("unknown vector selector");
    MG_TRY(vec_alloc(c, L, v));
    return upload_vector(c, L, *v, host);
}

int mg_set_rhs_true(mg_handle c, int level, const double* host) {
    MG_TRY(check_level(c, level));
    if (!host) return fail("null host vector");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    MG_TRY(vec_alloc(c, L, &L.ftrue));
    return upload_vector(c, L, L.ftrue, host);
}

int mg_get_vector(mg_handle c, int level, int which, double* host, int gather) {
    MG_TRY(check_level(c, level));
    if (!host) return fail("null host vector");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    DVector* v = pick(L, which);
    if (!v || !v->raw) return fail("vector is not available on this level");
    MG_TRY(ensure_stage(c, L.n_global));
    ++c->downloads;
    const unsigned nb = (unsigned)std::min<int64_t>(4096, (L.n_global + 255) / 256);
    if (gather && !L.replicated && c->comm.active()) {
        // all-gather slabs in lexicographic order inside a scratch vector, then permute
        double* full = nullptr;
        MG_TRY(dev_alloc(c, &full, (size_t)L.n_global));
        HIP_TRY(hipMemcpyAsync(full + L.row0, v->rows, (size_t)L.nloc * 8, hipMemcpyDeviceToDevice, c->stream));
        int rc = allgather_planes(c, L.splits, L.g.plane, full);
        if (!rc) {
            hipLaunchKernelGGL(gather_out, dim3(nb), dim3(256), 0, c->stream, full, L.perm, L.n_global, (int64_t)0,
                               L.n_global, (int64_t)0, c->stage);
        }
        hipError_t e = hipStreamSynchronize(c->stream);
        dev_free(c, full, (size_t)L.n_global);
        if (rc) return rc;
        HIP_TRY(e);
        HIP_TRY(hipMemcpy(host, c->stage, (size_t)L.n_global * 8, hipMemcpyDeviceToHost));
        return 0;
    }
    if (!L.replicated && c->comm.active()) {
        // owned rows only: pre-load the caller's buffer so untouched entries survive
        HIP_TRY(hipMemcpyAsync(c->stage, host, (size_t)L.n_global * 8, hipMemcpyHostToDevice, c->stream));
    }
    hipLaunchKernelGGL(gather_out, dim3(nb), dim3(256), 0, c->stream, v->base, L.perm, L.n_global, L.row0, L.nloc,
                       L.g.lead, c->stage);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(host, c->stage, (size_t)L.n_global * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int mg_zero_vector(mg_handle c, int level, int which) {
    MG_TRY(check_level(c, level));
    Level& L = c->L[level];
    DVector* v = pick(L, which);
    if (!v) return fail("unknown vector selector");
    MG_TRY(vec_alloc(c, L, v));
    return zero_vec(c, L, *v);
}

int mg_copy_vector(mg_handle c, int level, int dst_which, int src_which) {
    MG_TRY(check_level(c, level));
    Level& L = c->L[level];
    DVector* d = pick(L, dst_which);
    DVector* s = pick(L, src_which);
    if (!d || !s || !s->raw) return fail("bad vector selector");
    MG_TRY(vec_alloc(c, L, d));
    if (d->raw == s->raw) return 0;
    HIP_TRY(hipMemcpyAsync(d->base, s->base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
    return 0;
}

int mg_smooth(mg_handle c, int level, int nw) {
    MG_TRY(need_matrix(c, level));
    if (nw < 0) return fail("nw must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    MG_TRY(exchange_halo(c, c->L[level], c->L[level].v));
    return smooth(c, level, nw);
}

int mg_smooth_split(mg_handle c, int level, int nw) {
    MG_TRY(need_matrix(c, level));
    if (nw < 0) return fail("nw must be >= 0");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    if (!L.replicated && c->comm.active()) return fail("mg_smooth_split is single-GPU only");
    if (!L.err.raw) return fail("MG_VEC_ERR must hold the diagonal of D^-1");
    const unsigned nb = (unsigned)std::min<int64_t>(2048, (L.nloc + 255) / 256);
    for (int s = 0; s < nw; ++s) {
        MG_TRY(launch_ell(c, L, MODE_SPMV, false, L.v.base, nullptr, L.v2.rows, nullptr, nullptr));
        hipLaunchKernelGGL(jacobi_split_combine, dim3(nb), dim3(256), 0, c->stream, L.v.rows, L.f.rows, L.err.rows,
                           L.v2.rows, L.v2.rows, L.nloc, c->omega);
        std::swap(L.v, L.v2);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

int mg_residual(mg_handle c, int level) {
    MG_TRY(need_matrix(c, level));
    HIP_TRY(hipSetDevice(c->device));
    MG_TRY(exchange_halo(c, c->L[level], c->L[level].v));
    return residual(c, level);
}

int mg_restrict(mg_handle c, int level, int kind) {
    MG_TRY(check_level(c, level));
    if (level == 0) return fail("level 0 has no coarser level");
    MG_TRY(need_grid(c, level));
    MG_TRY(need_grid(c, level - 1));
    if (kind != MG_RESTRICT_INJECTION && kind != MG_RESTRICT_FULL_WEIGHTING && kind != MG_RESTRICT_TABLE) return fail("unknown restriction");
    HIP_TRY(hipSetDevice(c->device));
    return restrict_to(c, level, kind);
}

int mg_prolong(mg_handle c, int level, int add) {
    MG_TRY(check_level(c, level));
    if (level == 0) return fail("level 0 has no coarser level");
    MG_TRY(need_grid(c, level));
    MG_TRY(need_grid(c, level - 1));
    HIP_TRY(hipSetDevice(c->device));
    MG_TRY(exchange_halo(c, c->L[level - 1], c->L[level - 1].v));
    return prolong(c, level, add);
}

int mg_coarse_solve(mg_handle c, int* iterations, double* rel_residual) {
    MG_TRY(need_matrix(c, 0));
    HIP_TRY(hipSetDevice(c->device));
    return coarse_solve(c, iterations, rel_residual);
}

int mg_norm2(mg_handle c, int level, int which, double* out) {
    MG_TRY(check_level(c, level));
    if (!out) return fail("null output");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    DVector* v = pick(L, which);
    if (!v || !v->raw) return fail("vector is not available on this level");
    return norm2(c, L, v->rows, out);
}

int mg_quadratic_form(mg_handle c, int level, int which, double* out) {
    MG_TRY(need_matrix(c, level));
    if (!out) return fail("null output");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    DVector* v = pick(L, which);
    if (!v || !v->raw || v == &L.v2) return fail("vector is not available for a quadratic form");
    MG_TRY(exchange_halo(c, L, *v));
    const unsigned grid = blocks_for(L.nslices, WAVES_PER_BLOCK);
    double* parts = nullptr;
    MG_TRY(dev_alloc(c, &parts, grid));
    int rc = launch_ell(c, L, MODE_SPMV, true, v->base, nullptr, L.v2.rows, parts, nullptr);
    if (!rc) {
        hipLaunchKernelGGL(reduce_partials, dim3(1), dim3(BLOCK), 0, c->stream, parts, (int)grid, c->scalars);
        if (!L.replicated) rc = allreduce_sum(c, c->scalars, 1);
    }
    if (!rc) {
        hipError_t e = hipMemcpyAsync(c->h_scalars, c->scalars, sizeof(double), hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(hipGetErrorString(e));
    }
    dev_free(c, parts, grid);
    if (rc) return rc;
    *out = c->h_scalars[0];
    return 0;
}

namespace {

__global__ void vec_diff(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int64_t n) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        out[t] = a[t] - b[t];
}

// c->scalars[slot] = x^T M x with the handle's mass matrix (x: a vector of the mass matrix's level, halos valid)
int mass_form(mg_context* c, Level& L, DVector& x, int slot) {
    Level& M = c->mass;
    MG_TRY(vec_alloc(c, L, &c->mass_out));
    const unsigned nparts = blocks_for(M.nslices, WAVES_PER_BLOCK);
    if (nparts > 2 * kMaxParts) {
        // more partial sums than the handle's scratch holds: a buffer of its own for this call
        double* parts = nullptr;
        MG_TRY(dev_alloc(c, &parts, nparts));
        unsigned grid = 0;
        int rc = launch_ell(c, M, MODE_SPMV, true, x.base, nullptr, c->mass_out.rows, parts, nullptr, &grid);
        if (!rc) hipLaunchKernelGGL(reduce_partials, dim3(1), dim3(BLOCK), 0, c->stream, parts, (int)grid, c->scalars + slot);
        if (!rc && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail("mass_form: stream error");
        dev_free(c, parts, nparts);
        MG_TRY(rc);
    } else {
        unsigned grid = 0;
        MG_TRY(launch_ell(c, M, MODE_SPMV, true, x.base, nullptr, c->mass_out.rows, c->partials, nullptr, &grid));
        hipLaunchKernelGGL(reduce_partials, dim3(1), dim3(BLOCK), 0, c->stream, c->partials, (int)grid, c->scalars + slot);
    }
    HIP_TRY(hipGetLastError());
    if (!L.replicated) MG_TRY(allreduce_sum(c, c->scalars + slot, 1));
    return 0;
}

// after a cycle on the top level: residual norm (and error norm) of the iterate in the chosen norm, one small
// device -> host copy for both
int fmg_norms(mg_context* c, int l, int norm, bool want_err, double* rn, double* en) {
    Level& L = c->L[l];
    MG_TRY(residual(c, l));
    if (norm == MG_NORM_MASS) {
        MG_TRY(exchange_halo(c, L, L.v2));
        MG_TRY(mass_form(c, L, L.v2, 0));
    } else {
        MG_TRY(dot_device(c, L, L.v2.rows, L.v2.rows, 0));
    }
    if (want_err) {
        MG_TRY(vec_alloc(c, L, &c->diff));
        const unsigned nb = (unsigned)std::min<int64_t>(4096, (L.nloc + 255) / 256);
        hipLaunchKernelGGL(vec_diff, dim3(nb), dim3(256), 0, c->stream, L.v.rows, c->uexact.rows, c->diff.rows, L.nloc);
        HIP_TRY(hipGetLastError());
        if (norm == MG_NORM_MASS) {
            MG_TRY(exchange_halo(c, L, c->diff));
            MG_TRY(mass_form(c, L, c->diff, 1));
        } else {
            MG_TRY(dot_device(c, L, c->diff.rows, c->diff.rows, 1));
        }
    }
    HIP_TRY(hipMemcpyAsync(c->h_scalars, c->scalars, 2 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    *rn = std::sqrt(std::max(0.0, c->h_scalars[0]));
    if (want_err) *en = std::sqrt(std::max(0.0, c->h_scalars[1]));
    return 0;
}

}  // namespace

int mg_set_mass_csr(mg_handle c, int level, int64_t n_rows, int64_t nnz, const void* indptr, int indptr_is_64,
                    const int32_t* indices, const double* data) {
    MG_TRY(check_level(c, level));
    MG_TRY(need_grid(c, level));
    if (!indptr || !indices || !data) return fail("null CSR arrays");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    if (n_rows != L.n_global) return fail("mass matrix has " + std::to_string(n_rows) + " rows, level has " + std::to_string(L.n_global));
    Level& M = c->mass;
    // the work vector of mass_form is sized for the level the previous mass matrix belonged to
    if (c->mass_out.raw && c->mass_level >= 0) vec_free(c, c->L[c->mass_level], &c->mass_out);
    free_level(c, M);
    c->mass_level = -1;
    M = Level();
    MG_TRY(setup_geometry(c, M, level, L.N));
    int rc = 0;
    if (L.perm) {                       // the level's DoF numbering
        rc = dev_alloc(c, &M.perm, (size_t)L.n_global);
        if (!rc && hipMemcpy(M.perm, L.perm, (size_t)L.n_global * sizeof(int), hipMemcpyDeviceToDevice) != hipSuccess)
            rc = fail("copy of the level's permutation failed");
    }
    // (level index 1: any level but the coarsest may use symmetric diagonal storage)
    if (!rc) rc = build_level_from_csr(c, 1, M, n_rows, nnz, indptr, indptr_is_64, indices, data, nullptr, 1, false);
    if (rc) {
        const std::string why = g_err;
        free_level(c, M);
        g_err = why;
        return rc;
    }
    c->mass_level = level;
    return 0;
}

int mg_set_exact(mg_handle c, int level, const double* host) {
    MG_TRY(check_level(c, level));
    if (!host) return fail("null host vector");
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    if (c->uexact.raw && c->uexact_level != level) {
        vec_free(c, c->L[c->uexact_level], &c->uexact);
        if (c->diff.raw) vec_free(c, c->L[c->uexact_level], &c->diff);
    }
    MG_TRY(vec_alloc(c, L, &c->uexact));
    c->uexact_level = level;
    return upload_vector(c, L, c->uexact, host);
}

int mg_counters(mg_handle c, int64_t* uploads, int64_t* downloads, int64_t* graph_replays, int* graphs_cached) {
    if (!c) return fail("null handle");
    if (uploads) *uploads = c->uploads;
    if (downloads) *downloads = c->downloads;
    if (graph_replays) *graph_replays = c->graph_replays;
    if (graphs_cached) *graphs_cached = (int)c->graphs.size();
    return 0;
}

int mg_prepare_cycle(mg_handle c, int level) {
    MG_TRY(check_level(c, level));
    for (int l = 0; l <= level; ++l) {
        MG_TRY(need_matrix(c, l));
        if (level > 0) MG_TRY(need_grid(c, l));
    }
    HIP_TRY(hipSetDevice(c->device));
    MG_TRY(prepare_cycle(c, level));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int mg_vcycle(mg_handle c, int level, int ncycles, double* resid_l2) {
    MG_TRY(check_level(c, level));
    for (int l = 0; l <= level; ++l) {
        MG_TRY(need_matrix(c, l));
        if (level > 0) MG_TRY(need_grid(c, l));
    }
    HIP_TRY(hipSetDevice(c->device));
    Level& L = c->L[level];
    MG_TRY(exchange_halo(c, L, L.v));
    for (int k = 0; k < ncycles; ++k) {
        MG_TRY(vcycle_graphed(c, level));
        if (resid_l2) {
            MG_TRY(residual(c, level));
            MG_TRY(norm2(c, L, L.v2.rows, &resid_l2[k]));
        }
    }
    return 0;
}

int mg_fmg_ex(mg_handle c, int top, int mu0, double tol, int max_cycles, int norm, double* resid_hist, double* err_hist,
              int* cycles_done) {
    MG_TRY(check_level(c, top));
    for (int l = 0; l <= top; ++l) {
        MG_TRY(need_matrix(c, l));
        if (top > 0) MG_TRY(need_grid(c, l));
    }
    if (norm != MG_NORM_L2 && norm != MG_NORM_MASS) return fail("unknown norm");
    if (norm == MG_NORM_MASS && c->mass_level != top) return fail("mg_fmg_ex: no mass matrix on the top level (mg_set_mass_csr)");
    if (err_hist && (!c->uexact.raw || c->uexact_level != top)) return fail("mg_fmg_ex: no exact solution on the top level (mg_set_exact)");
    HIP_TRY(hipSetDevice(c->device));
    for (int l = 0; l < top; ++l) {
        Level& L = c->L[l];
        if (!L.ftrue.raw) return fail("mg_fmg needs the true right-hand side of level " + std::to_string(l));
        HIP_TRY(hipMemcpyAsync(L.f.base, L.ftrue.base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
    }
    MG_TRY(coarse_solve(c, nullptr, nullptr));
    int done_cycles = 0;
    const bool norms = resid_hist != nullptr || err_hist != nullptr;
    for (int l = 1; l <= top; ++l) {
        Level& L = c->L[l];
        if (l < top)   // cycles on level l-1 consumed F[l-1..0]; F[l] is still the true right-hand side
            HIP_TRY(hipMemcpyAsync(L.f.base, L.ftrue.base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
        // v_h = Interpolation(v_2h) (multigrid.py:283-284): interpolate into ERR, then copy to V
        MG_TRY(exchange_halo(c, c->L[l - 1], c->L[l - 1].v));
        MG_TRY(prolong(c, l, 0));
        HIP_TRY(hipMemcpyAsync(L.v.base, L.err.base, (size_t)L.xlen * 8, hipMemcpyDeviceToDevice, c->stream));
        MG_TRY(exchange_halo(c, L, L.v));
        const bool until_tol = l == top && tol > 0.0;
        const int ncyc = until_tol ? max_cycles : mu0;
        for (int k = 0; k < ncyc; ++k) {
            MG_TRY(vcycle_graphed(c, l));
            if (l < top) continue;
            ++done_cycles;
            if (!until_tol && !norms) continue;
            // residual (and error) of this cycle's iterate in the caller's norm (multigrid.py:291-295)
            double rn = 0.0, en = 0.0;
            MG_TRY(fmg_norms(c, l, norm, err_hist != nullptr, &rn, &en));
            if (resid_hist) resid_hist[k] = rn;
            if (err_hist) err_hist[k] = en;
            if (until_tol && rn <= tol) break;
        }
    }
    if (cycles_done) *cycles_done = done_cycles;
    return 0;
}

int mg_fmg(mg_handle c, int top, int mu0, double tol, int max_cycles, double* resid_l2, int* cycles_done) {
    return mg_fmg_ex(c, top, mu0, tol, max_cycles, MG_NORM_L2, resid_l2, nullptr, cycles_done);
}

int mg_time_kernel(mg_handle c, const char* kernel, int level, int reps, double* avg_ms) {
    MG_TRY(need_matrix(c, level));
    if (!kernel || !avg_ms || reps < 1) return fail("bad arguments");
    HIP_TRY(hipSetDevice(c->device));
    const std::string k(kernel);
    Level& L = c->L[level];
    struct Events {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        ~Events() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
    } ev;
    HIP_TRY(hipEventCreate(&ev.e0));
    HIP_TRY(hipEventCreate(&ev.e1));
    const hipEvent_t e0 = ev.e0, e1 = ev.e1;
    auto once = [&]() -> int {
        if (k == "jacobi") return launch_ell(c, L, MODE_JACOBI, false, L.v.base, L.f.rows, L.v2.rows, nullptr, nullptr);
        // "jacobi2": only where mg_smooth itself pairs sweeps on this level; "jacobi2!": wherever the kernel applies
        if (k == "jacobi2" || k == "jacobi2!") {
            if (!fused_sweeps_ok(c, L, k == "jacobi2!")) return fail("level does not use the two-sweep kernel");
            const J2Plan plan = jacobi2_plan(c, L, false, 0);
            return launch_jacobi2(c, L, plan, 0, 1, plan.nseg, L.v.rows, L.f.rows, L.v2.rows);
        }
        if (k.rfind("jacobik3", 0) == 0) {     // K = fuse_k sweeps per pass on a whole class-coded 3-D level ("!": wherever it applies)
            const bool forced = k.find(":form") != std::string::npos;
            if (!sweepsk_ok(c, L, forced || k.find('!') != std::string::npos)) return fail("level does not use the K-sweep pass");
            c->timing_force_form = forced ? k.back() - '0' : -1;
            const bool slab = !L.replicated && c->comm.active();
            if (slab && L.cls_halo != 1) return fail("the slab's class halos are not built yet (first smoother call)");
            const int rc = launch_jacobikc(c, L, slab ? std::min(std::min(c->fuse_k, 5), L.hd) : sweepsk_max(c, L), L.v.rows, L.f.rows, L.v2.rows);
            c->timing_force_form = -1;
            return rc;
        }
        if (k == "jacobiblk" || k == "jacobiblk!") {      // K sweeps per launch on resident blocks ("!": whatever the level's size)
            const int keep = c->fuse_block;
            if (k.back() == '!') c->fuse_block = 2;
            const bool ok = block_sweeps_ok(c, L);
            c->fuse_block = keep;
            if (!ok) return fail("level does not use the block pass");
            const JBPlan plan = block_plan(c, L);
            return launch_jacobi_block(c, L, plan.K, plan.EZ, L.v.rows, L.f.rows, L.v2.rows);
        }
        if (k == "jacobi_small") {           // all mu1 sweeps of a small level in one launch
            if (!small_level_ok(c, L) || c->mu1 < 2) return fail("level does not use the one-launch smoother");
            return launch_jacobi_small(c, L, c->mu1, L.v.rows, L.f.rows, L.v2.rows);
        }
        if (k == "jacobik") {
            if (!sweeps2d_ok(c, L)) return fail("level does not use the K-sweep 2-D kernel");
            return launch_jacobik(c, L, c->fuse_2d_k, L.v.rows, L.f.rows, L.v2.rows);
        }
        if (k == "gs") {            // one full Gauss-Seidel sweep (all colours) with the configured colouring
            if (c->smoother == MG_SMOOTH_JACOBI) return fail("the configured smoother is Jacobi");
            return smooth(c, level, 1);
        }
        if (k == "residual") return residual(c, level);
        if (k == "restrict") return level > 0 ? restrict_to(c, level, c->restriction) : fail("level 0");
        if (k == "prolong") return level > 0 ? prolong(c, level, 1) : fail("level 0");
        if (k == "norm2") return dot_device(c, L, L.v.rows, L.v.rows, 1);
        return fail("unknown kernel " + k);
    };
    MG_TRY(once());   // warm-up
    HIP_TRY(hipEventRecord(e0, c->stream));
    for (int r = 0; r < reps; ++r) MG_TRY(once());
    HIP_TRY(hipEventRecord(e1, c->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *avg_ms = (double)ms / reps;
    return 0;
}

int mg_sync(mg_handle c) {
    if (!c) return fail("null handle");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return 0;
}

int mg_memory_bytes(mg_handle c, int64_t* bytes) {
    if (!c || !bytes) return fail("bad arguments");
    *bytes = c->bytes;
    return 0;
}

}  // extern "C"
