/*
 * mg_hip.h -- C ABI of the MI355X (gfx950) geometric-multigrid V-cycle library
 * (libmg_hip.so).
 *
 * This is the drop-in boundary for the hot path of nikhilTkur/Multigrid_dolfinx.
 * The reference has no FFI of its own: its boundary is the Python function surface
 * of multigrid.py.  Each entry point below names the reference interface it
 * replaces (file:line in the reference).  The only intended caller is the ctypes
 * shim multigrid_dolfinx_amd/_capi.py; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - plain C types, host pointers, caller owns every buffer it passes in;
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from mg_last_error() (thread-local);
 *   - levels are numbered 0 (coarsest) .. n_levels-1 (finest); level l has
 *     N_l = N_0 * 2^l elements per dimension and (N_l+1)^dim unknowns
 *     (multigrid.py:247-248);
 *   - vectors crossing the boundary are fp64 in the CALLER's DoF numbering; the
 *     library keeps them in lexicographic grid numbering internally and permutes
 *     at this edge using the grid_index given to mg_set_level_csr;
 *   - one handle = one hierarchy on one GPU (or one slab of it, see mg_set_comm);
 *     calls on a handle are serialised on the handle's HIP stream; there is no
 *     global state in the library (the reference's module globals,
 *     multigrid.py:10-45, live in the Python shim only).
 */
#ifndef MG_HIP_H
#define MG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mg_context* mg_handle;

/* which per-level device vector an accessor refers to */
enum mg_vec {
    MG_VEC_V = 0,   /* iterate / correction v_h                                    */
    MG_VEC_F = 1,   /* right-hand side f_h (restricted residual on coarse levels)   */
    MG_VEC_R = 2,   /* residual r_h = f_h - A v_h of the last mg_residual / V-cycle */
    MG_VEC_ERR = 3  /* interpolated coarse correction err_h (multigrid.py:258-259),  */
                    /* kept only when mg_set_params(... keep_err = 1)                */
};

enum mg_restriction {
    MG_RESTRICT_INJECTION = 0,      /* Restriction2D_direct, multigrid.py:123-132 (the live path, :251-252) */
    MG_RESTRICT_FULL_WEIGHTING = 1, /* Restriction2D,        multigrid.py:135-198                            */
    MG_RESTRICT_TABLE = 2           /* transpose of the table prolongation (mg_set_restriction_table); no reference */
};

enum mg_smoother {
    MG_SMOOTH_JACOBI = 0,           /* jacobiRelaxation, multigrid.py:223-228 */
    MG_SMOOTH_RBGS = 1,             /* red-black Gauss-Seidel / SOR with factor omega (no reference;
                                       BASELINE.json config 5): per sweep one in-place half sweep per
                                       colour, colour = parity of the lexicographic node index.  Needs
                                       pruned grid matrices (P1 stencils are then bipartite). */
    MG_SMOOTH_MCGS = 2              /* nine-colour Gauss-Seidel / SOR for P2 rows (BASELINE.json config 5; no
                                       reference): the seven parity classes of the lattice's mid-points plus the
                                       vertices split red / black; one in-place launch per colour in ascending
                                       order.  Valid for pruned P2 and P1 grid matrices (checked per level). */
};

/* ---- life cycle ----------------------------------------------------------------
 * Replaces the module-global problem state set by initialize_problem
 * (multigrid.py:28-45) with an explicit handle. */
int mg_create(int n_levels, int dim, int device, mg_handle* out);
int mg_destroy(mg_handle h);
const char* mg_last_error(void);
/* "gfx950 <n_cus> CUs" style description of the device the handle runs on. */
int mg_device_info(mg_handle h, char* buf, size_t buflen);

/* ---- multi-GPU (no reference counterpart) -----------------------
 * One process per GPU.  Must be called before any level is set.  The finest
 * levels are split into contiguous slabs of grid planes (slowest axis), aligned so
 * that coarse plane K lives with fine plane 2K; levels with fewer than
 * `replicate_below` unknowns are replicated on every rank (no communication below
 * that size).  `nccl_unique_id` is the 128-byte RCCL id obtained on rank 0 with
 * mg_comm_unique_id and distributed by the caller (bench.py uses torch.distributed
 * for that rendezvous only).  Halo planes and scalar reductions then travel over
 * RCCL/xGMI. */
int mg_comm_unique_id(void* id_out, size_t id_bytes);
int mg_set_comm(mg_handle h, int rank, int world, const void* nccl_unique_id, size_t id_bytes,
                int64_t replicate_below);
/* Exercises every RCCL entry point the library uses (id, init, all-reduce, broadcast, grouped
 * send/recv, destroy) on a one-rank communicator of `device`; 0 = all results correct. */
int mg_comm_selftest(int device);
/* Host-staged transport for tests without RCCL peers: the library stages device
 * buffers through host memory and calls back into the caller (who moves the bytes,
 * e.g. over gloo).  exchange: send `count` doubles to rank-1 / rank+1 (NULL pointer
 * = no such neighbour) and receive as many from each.  allreduce: in-place sum of
 * `count` doubles over all ranks.  allgatherv: counts[r] doubles from every rank r
 * into recv (rank order). */
typedef int (*mg_exchange_fn)(void* user, const double* send_lo, const double* send_hi,
                              double* recv_lo, double* recv_hi, int64_t count);
typedef int (*mg_allreduce_fn)(void* user, double* inout, int64_t count);
typedef int (*mg_allgatherv_fn)(void* user, const double* send, int64_t send_count,
                                double* recv, const int64_t* counts);
int mg_set_comm_callbacks(mg_handle h, int rank, int world, mg_exchange_fn ex, mg_allreduce_fn ar,
                          mg_allgatherv_fn ag, void* user, int64_t replicate_below);

/* ---- hierarchy set-up --------------------------------------------------------------
 * mg_set_level_csr: hand over one level's stiffness matrix exactly as
 * scipy.sparse.csr_matrix((av, aj, ai)) holds it after PETSc getValuesCSR()
 * (Multigrid_prototype.py:95-99): fp64 values, int32 column indices (possibly
 * unsorted, possibly with explicit zeros), row pointers int32 or int64
 * (indptr_is_64).  `grid_index[dof]` is the lexicographic node index
 * i + (N+1)(j + (N+1)k) of each DoF -- the integer form of the reference's
 * coordinate dictionaries (Multigrid_prototype.py:68-74); NULL means the DoFs are
 * already lexicographic.  The library renumbers once (P A P^T), optionally drops
 * explicit zeros (prune_zeros; the reference's own smoother matrix drops them too,
 * multigrid.py:52-55, App. A Q7) and stores sliced-ELL tiles.  This subsumes
 * getJacobiMatrices (multigrid.py:48-56): D^-1 is extracted here and the smoother
 * streams A itself (v + w D^-1 (f - A v)), so no second matrix is stored.
 * In slab mode every rank passes the full matrix (mg_set_level_csr_local takes the rank's rows only) and keeps its own planes. */
int mg_set_level_csr(mg_handle h, int level, int elements_per_dim, int64_t n_rows, int64_t nnz,
                     const void* indptr, int indptr_is_64, const int32_t* indices,
                     const double* data, const int64_t* grid_index, int prune_zeros);
/* Per-rank hand-off for slabs: the rank passes ONLY the rows it owns (mg_level_slab says which: the lexicographic
 * nodes [row0, row0 + n_local) of the level, plus how many halo nodes it may couple to on either side), in a local
 * numbering like a distributed assembly has it (PETSc's owned + ghost layout behind getValuesCSR(),
 * Multigrid_prototype.py:95-96): row r of the CSR is local id r, column indices are local ids in [0, n_cols), and
 * col_nodes[id] names the global lexicographic node of every local id (ids < n_rows: the owned rows, in any order; the
 * rest: ghosts).  grid_index (n_global entries, or NULL for lexicographic) still maps the caller's GLOBAL DoF numbers to
 * nodes for the vector calls.  Replicated levels take the whole matrix this way too (n_rows = n_global).  Results are
 * those of the full-matrix hand-off bit for bit.  No reference counterpart (the reference is serial). */
int mg_level_slab(mg_handle h, int level, int elements_per_dim, int64_t* row0, int64_t* n_local, int64_t* halo_lo,
                  int64_t* halo_hi);
int mg_set_level_csr_local(mg_handle h, int level, int elements_per_dim, int64_t n_rows, int64_t n_cols, int64_t nnz,
                           const void* indptr, int indptr_is_64, const int32_t* indices, const double* data,
                           const int64_t* col_nodes, const int64_t* grid_index, int prune_zeros);
/* A level with grid geometry and numbering only (no matrix): enough for the transfer
 * operators, which the reference exposes as free functions taking two coordinate
 * dictionaries (Interpolation2D / Restriction2D(_direct), multigrid.py:59, :123, :135). */
int mg_set_level_grid(mg_handle h, int level, int elements_per_dim, int64_t n_rows,
                      const int64_t* grid_index);   /* elements_per_dim == 0: flat vector space */
/* Device-side synthetic generator (bench/tests; no reference counterpart): writes
 * the same tiles and right-hand side that poisson.make_level + mg_set_level_csr
 * would produce for the P1 Poisson problem of Multigrid_prototype.py:77-110
 * (lexicographic numbering), without a host CSR -- the only way to set up 1025^3
 * unknowns.  Also fills MG_VEC_F with the lifted right-hand side. */
int mg_gen_poisson_level(mg_handle h, int level, int elements_per_dim, int prune_zeros);
/* The same for elements whose unknowns fill a lattice with one row per parity class of the lattice point -- P2 on the
 * structured simplicial mesh, BASELINE.json config 5 (no reference counterpart: the reference is P1 only).  The
 * caller hands over, for each of the eight classes (i & 1) | (j & 1) << 1 | (k & 1) << 2, the interior stencil as
 * count[class] (offset, value) pairs -- offsets[class][t][3] = (di, dj, dk) in ascending column order, at most
 * 64 per class, values[class][t] for this level's mesh width -- and the load of the constant source; the device writes
 * the tiles with the reference's boundary treatment (identity rows, zeroed columns, lifted right-hand side,
 * Multigrid_prototype.py:77-108).  `lattice_steps` = lattice points per dimension - 1 (even).  Single GPU. */
int mg_gen_lattice_level(mg_handle h, int level, int lattice_steps, int width, const int* count, const int* offsets,
                         const double* values, const double* load);
/* getJacobiMatrices (multigrid.py:48-56) as a stand-alone set-up kernel, for callers that
 * want the reference's split operands back: for every stored entry a_ij of the CSR matrix
 * writes scaled[q] = a_ij / a_ii computed as (1/a_ii) * a_ij, keep[q] = 1 unless the entry
 * is the diagonal or an explicit zero (SciPy's CSR - DIA subtraction drops both, App. A Q7),
 * and dinv[i] = 1 / a_ii.  The caller compacts the kept entries (the Python shim does, in
 * SciPy's reversed per-row order) into D^-1 (A - D). */
int mg_jacobi_split(int device, int64_t n_rows, int64_t nnz, const void* indptr, int indptr_is_64,
                    const int32_t* indices, const double* data, double* dinv, double* scaled,
                    unsigned char* keep);
/* mu1/mu2/omega: Multigrid_prototype.py:43-46 via initialize_problem
 * (multigrid.py:38-41).  coarse_rtol/coarse_maxit steer the device PCG that stands
 * in for spsolve on the coarsest level (multigrid.py:239). */
int mg_set_params(mg_handle h, int mu1, int mu2, double omega, int restriction, int smoother,
                  double coarse_rtol, int coarse_maxit, int keep_err);
/* Prolongation from a table instead of the reference's bilinear / trilinear Interpolation2D (multigrid.py:59-120): a fine
 * lattice point (i, j, k) combines the count[r] coarse lattice points 2 * floor((i, j, k) / 4) + offsets[r][t] (each
 * offset component 0..2, at most 10 entries) with weights[r][t], r = (i mod 4) + 4 (j mod 4) + 16 (k mod 4); 2-D levels
 * use the residues with j mod 4 = 0.  With poisson.p2_prolongation_table this is the natural embedding of the coarse P2
 * space into the fine one (BASELINE.json config 5; no reference counterpart).  All three NULL: back to the reference's
 * interpolation.  Slabs need halo_planes = 2. */
int mg_set_prolongation_table(mg_handle h, const int* count /*[64]*/, const int* offsets /*[64][10][3]*/,
                              const double* weights /*[64][10]*/);
/* Restriction from a table (MG_RESTRICT_TABLE), gathered per coarse lattice point: an interior coarse point A of type
 * (a & 1) + 2 (b & 1) + 4 (c & 1) sums weights[type][t] * r[2 A + offsets[type][t]] over the interior fine points (offset
 * components within +-4, max_entries per type); boundary coarse points take the coincident fine value.  With
 * poisson.p2_restriction_table this is the transpose of the P2 prolongation, <R r, v> = <r, P v>: the canonical restriction of
 * a finite-element residual (the reference injects it, multigrid.py:251-252, which under-scales the coarse correction:
 * SURVEY.md App. A Q1).  Whole levels only.  No reference counterpart. */
int mg_set_restriction_table(mg_handle h, int max_entries, const int* count /*[8]*/, const int* offsets /*[8][max][3]*/,
                             const double* weights /*[8][max]*/);
/* Tuning and format knobs (defaults in parentheses; DESIGN.md sections 4-6 explain each):
 *   before any level is set:
 *     "rows_per_lane"      1 | 2 | 4 rows of a slice per lane (2)
 *     "offset_codes"       0 keeps int32 column indices (1)
 *     "symmetric_storage"  0 keeps lower entries even for bit-for-bit symmetric matrices (1)
 *     "require_diagonal"   0 accepts operators without a diagonal, e.g. D^-1 R for mg_smooth_split (1)
 *     "halo_planes"        grid planes exchanged with each slab neighbour: 1, or 2 for stencils that reach two planes
 *                          (P2 lattice levels) (1)
 *     "storage_ulps"       k > 0: matrix entries that agree within about k units in the last place count as equal in the
 *                          symmetry test (the upper entry of a pair is kept) and in the row dictionary (a class holds the first
 *                          such row seen): assemblies whose rows differ by round-off (h not a power of two, varying
 *                          summation order) then still get the compact formats.  This PERTURBS the matrix by up to k ulps
 *                          per entry -- below the round-off of the assembly itself for small k -- so results agree with the
 *                          exact-storage ones to ~k * 1e-16 relative, not bit for bit.  (0: everything bit for bit)
 *     "storage_auto"       1 = a level whose exact symmetry test or row dictionary fails is tried once more with entries within
 *                          4 units in the last place counting as equal (what an assembly with row-dependent round-off needs
 *                          to reach the compact formats: 2.6 x on the headline pass); the perturbation -- at most 4 ulps per
 *                          entry, below the assembly's own round-off -- is reported by mg_level_storage.  0 = exact storage
 *                          only (1)
 *     "fuse_block"         1 = whole seven-point levels with row classes of "fuse_block_min_rows" <= rows < "fuse_block_max_rows" --
 *                          the middle levels, too small for the plane marches -- are relaxed K sweeps per launch on blocks of
 *                          32 x 32 x EZ cells that stay on the CU (mg_jacobiblk.hip.h), each block recomputing a halo of K
 *                          cells; bit-identical to single sweeps.  2 = whatever the level's size (tests), 0 = off (1: alone the
 *                          pass is no faster per sweep than one launch per sweep, but a cycle has a third of the launches --
 *                          BASELINE config 3 132.8 -> 137.5 cycles/s)
 *     "fuse_block_min_rows"  see "fuse_block" (2^15)
 *     "fuse_block_max_rows"  see "fuse_block" (2^23)
 *     "fuse_block_k"       sweeps per launch of the block pass, 2..4; 0 = three, four on levels whose blocks then all run at
 *                          once (65^3 rows) (0)
 *     "fuse_block_ez"      planes per block of the block pass, 11 or 19 (19 spills registers and is slower) (11)
 *     "fuse_2d_lines"      lines per region of the 2-D K-sweep kernel ("fuse_2d"): 64, 32 or 16; 0 = chosen per level -- 32 (two
 *                          workgroups per CU: BASELINE config 2 451 -> 592 cycles/s against 64 lines), 16 on levels whose
 *                          tiles then still all run at once (621) (0)
 *     "direct_block_rows"  the coarsest level's exact solve (block-tridiagonal LU over groups of grid planes / lines) takes blocks
 *                          of at least this many rows, at most 2304: a solve is three dependent launches per block, so fewer,
 *                          larger blocks are faster while their dense inverses (rows^2 x 8 B each) stay small.  Before the first
 *                          cycle (2048)
 *     "gen_odd_rows"       mg_gen_poisson_level gives this many of 10000 interior rows, picked by a hash of their grid index, a
 *                          reaction term of their own on the diagonal (a.diag * (1 + r), 0 <= r < 1): rows unlike any other,
 *                          for measuring what "row_escape" costs.  Levels generated afterwards (0)
 *     "row_escape"         1 = a symmetric seven-point level with more than 255 distinct rows keeps row classes when at least three
 *                          quarters of its rows are copies of the 254 most frequent ones: the other rows ("escape rows":
 *                          another material, a perturbed coefficient) are read from the stored matrix by the K-sweep march
 *                          ("fuse_k"), which is then the only kernel that uses the classes -- same arithmetic, same results bit
 *                          for bit.  Needs whole levels (not slabs) and no 64 x 32 tile of the march with more than 1024 such rows in
 *                          K + 2 planes (512 for K = 5; the march takes the most sweeps per pass that fit); else, and with 0, such a level has no row classes.  Before level set-up (1)
 *     "row_classes"        0 skips the dictionary of distinct rows on symmetric 5- and 7-point levels (1)
 *     "halo_depth"         halo planes every vector of a slab has ROOM for, >= "halo_planes" (0 .. 5; 0: just those).  With K
 *                          planes of room the K-sweep march ("fuse_k") runs on slabs too: K planes of the iterate travel once per
 *                          K sweeps and the slab relaxes its neighbours' K - 1 planes next to it itself -- one grouped exchange
 *                          and two launches per K sweeps instead of a boundary chain per pair; bit-identical to the single
 *                          handle.  Every rank alike, before mg_set_comm / level set-up.  (0)
 *   any time:
 *     "xcd_chunk"          consecutive tiles per XCD in the chunked block -> tile map (8)
 *     "strip_slices"       slices per XCD strip, 0 = chunked map only (64)
 *     "nontemporal"        streaming loads for once-read matrix / right-hand-side data (1)
 *     "lds_pad"            dynamic LDS bytes per block on large levels = occupancy cap (32768)
 *     "overlap"            halo exchange on the communication stream behind the interior sweep (1)
 *     "overlap_min_rows"   ... only on levels with at least this many owned rows (4194304)
 *     "slab_pair_form"     overlapped pair sweeps on slabs: 0 = the exchange chain computes its own first sweep of the boundary
 *                          planes and runs beside ONE launch of the pass over the whole slab; 1 = the boundary segments of the
 *                          pass first, then its interior segments beside the chain (0); bit-identical, every rank alike
 *     "graph_comm"         1 = V-cycles on slabs are captured into hipGraphs too, RCCL exchanges included (the calls are
 *                          stream operations on both streams of the overlapped sweeps); every rank must set it alike.
 *                          Needs the RCCL transport (mg_comm_init_rccl).  (0: validated against the in-process stand-in of
 *                          tests/fake_rccl only, not against RCCL on a multi-GPU node)
 *     "fuse_restrict"      residual evaluated at the coarse nodes only when injecting (1)
 *     "fuse_sweeps"        Jacobi sweeps in pairs, two per pass over the matrix, on 3-D 7-point levels in
 *                          symmetric diagonal storage (1); bit-identical to single sweeps
 *     "fuse_min_rows"      ... only on levels with at least this many owned rows (16777216)
 *     "fuse_segments"      plane segments per tile of that pass, 0 = chosen by the cost model (0)
 *     "fuse_nontemporal"   streaming loads in that pass (0: measured slower)
 *     "fuse_k"             Jacobi sweeps per pass of the K-sweep march (mg_jacobik3d.hip.h) on whole seven-point levels with
 *                          row classes: a smoother call of nw sweeps (multigrid.py:223-228) runs as passes of 3 .. "fuse_k"
 *                          sweeps and at most one pair (50 = 10 x 5); 0 .. 2 = pairs only (5); bit-identical to single sweeps
 *     "fuse_k_shape"       tile of that march: 0 = 128 x 24 cells (12 waves x 2 grid lines), 1 = 64 x 48 (12 waves x 4 lines),
 *                          2 = 128 x 24 (8 waves x 3 lines), 3 / 4 / 5 = 64 x 24 by 6 / 8 / 4 waves with two workgroups per CU
 *                          (levels of at most 64 row classes), 6 / 7 = 64 x 48 / 64 x 32 by 16 waves (7: no register spills up to five sweeps, measured best)
 *     "fuse_k_segments"    plane segments per tile of that march, 0 = chosen by its cost model (0)
 *     "fuse_k_slab_min_rows"  ... on slabs (below): levels whose smallest slab has at least this many rows (1048576)
 *     "fuse_k_slab_min_sweeps"  ... and smoother calls of at least this many sweeps (4)
 *     "fuse_k_min_rows"    ... on whole levels with at least this many rows; smaller levels keep single sweeps (16777216: with
 *                          2097152 the 129^3 level takes the march too -- 12.2 us per sweep against 14.3 alone, but whole cycles
 *                          of BASELINE config 3 are no faster; 65^3 rows: 11.6 against 4.0)
 *     "fuse_k_small_rows"  ... five sweeps per pass again on levels with fewer rows than this (4194304: 129^3 rows, 12.2 us per
 *                          sweep with five against 15.5 with four)
 *     "fuse_k4_min_rows"   ... more than three sweeps per pass only on levels with at least this many rows (0)
 *     "fuse_k5_min_rows"   ... more than four only on levels with at least this many rows (67108864: on 257^3 rows four
 *                          sweeps per pass measured best)
 *     "fuse_k_nt_store"    non-temporal stores of that march's result (experiment) (0)
 *     "fuse_k_tail"        the tiles left over for a last, nearly empty round of workgroups (fewer tiles than half the CUs) are cut
 *                          into shorter plane segments that fill that round once (1); bit-identical
 *     "fuse_k_small_tiles" 64 x 24 tiles (shape 4) on levels whose planes hold fewer 64 x 48 tiles than the GPU has CUs (0:
 *                          measured slower)
 *     "fuse_k_pf"          register sets for the planes of x that arrive: 2 = a second set keeps x staged one step longer
 *                          (three sweeps per pass only) (1: measured no slower)
 *     "fuse_k_dpp"         0 = the -1 / +1 neighbours of that march come through LDS instead of the neighbouring lanes'
 *                          registers (experiment, tile 0 only: measured 1.4 x slower) (1)
 *     "fuse_classes"       that pass reads one class byte per row instead of the 32-byte row where the level has
 *                          row classes (1); bit-identical either way
 *     "fuse_plain"         that pass on levels WITHOUT row classes: 2 = round-2 structure (sdia_jacobi2p), 1 = round 1's (2);
 *                          bit-identical either way
 *     "fuse_plain_shape"   ... its launch shape: 0 = 12 waves x 1 grid line, 1 = 8 x 2, 2 = 16 x 1 (0: the only one without
 *                          register spills, measured best)
 *     "class_sweeps"       the one-sweep kernels (residual, single sweeps, SpMV, Gauss-Seidel colours) read the class
 *                          byte too where the level has row classes (1); bit-identical either way
 *     "fuse_shape"         launch shape of the class-coded pass: 0 = 8 waves x 2 grid lines, 1 = 12 waves x 2 lines,
 *                          2 = 16 waves x 1 line, 3 = 8 waves x 3 lines (1)
 *     "fuse_even"          1 = the tile columns of the class-coded pass are equally wide, as narrow as covers the grid;
 *                          0 = always 124 cells, the last column nearly empty (0: measured faster); bit-identical
 *     "fuse_wi"            experiments: that width given directly (0 = chosen per level as above)
 *     "march_sweeps"       the one-sweep class kernels (residual, single Jacobi sweeps, Gauss-Seidel colours) run as a
 *                          plane march with x in LDS on whole 3-D seven-point levels (1); bit-identical either way
 *     "march_min_rows"     ... only on levels with at least this many rows (4194304)
 *     "march_shape"        ... 0 = 12 waves x 2 grid lines per workgroup, 1 = 16 x 2 (0)
 *     "fuse_small"         all Jacobi sweeps of a smoother call in ONE launch on levels with row classes that fit one CU's
 *                          LDS (a few thousand rows: the reference's own 65^2 / 33^2 levels) (1); bit-identical
 *     "fuse_small_2d_rows" ... on 2-D levels only up to this many rows: larger ones (65^2) are faster through the K-sweep 2-D
 *                          kernel, ten launches of a dozen small workgroups instead of one launch of one workgroup (2048)
 *     "fuse_2d"            up to "fuse_2d_k" Jacobi sweeps per launch on 2-D five-point levels with row classes (1);
 *                          bit-identical to single sweeps
 *     "fuse_2d_k"          ... at most this many per launch, 2..5 (5)
 *     "fuse_xcd_chunk"     consecutive tiles of that pass given to one XCD at a time (32)
 *     "cls_blocks_per_cu"  persistent blocks per CU of the one-sweep class kernels (4)
 *     "coarse_direct"      exact block-tridiagonal coarsest solve, 0 = PCG (1)
 *     "pcg_chunk"          PCG iterations enqueued between convergence checks (16)
 *     "lattice_march"      wide lattice stencils (3-D P2 levels with stencil classes) as a plane march with five planes of x in
 *                          LDS instead of gathers from global memory (1); bit-identical either way
 *     "lattice_march_min_rows"  ... only on levels with at least this many owned rows (4194304)
 *     "lattice_gs2"        1 = the nine-colour Gauss-Seidel sweep on whole 3-D lattice levels runs two colours per launch, out of
 *                          place (five passes over the vector instead of nine; the level's two iterate buffers swap);
 *                          bit-identical, but measured slower (its nine plane slots only fit 32-wide tiles) (0)
 *     "lattice_tile"       tile of that march: 0 / 1 = 64 x 16 cells (256 threads, two workgroups per CU), 2 = 128 x 16 where the
 *                          grid is 128 wide (512 threads, one per CU: fewer rim cells and partly used cache lines, but
 *                          measured slower) (0)
 *     "lattice_segments"   plane segments per tile of that march, 0 = chosen from the tile count (0)
 *     "graph"              replay V-cycles as hipGraphs on a single GPU (1)
 * None of them changes results beyond round-off; the tests pin which ones are bit-for-bit neutral. */
int mg_set_tuning(mg_handle h, const char* key, int64_t value);

/* ---- level queries ------------------------------------------------------------------- */
int mg_level_info(mg_handle h, int level, int64_t* n_global, int64_t* n_local, int64_t* row0,
                  int64_t* nnz_stored, int64_t* nnz_nonzero, int* ell_width, int* replicated,
                  int* offset_codes /* 0 = int32 columns; > 0 = offset codes (number of distinct
                                       offsets); < 0 = symmetric diagonal storage (-stored diagonals) */);
/* Number of distinct full rows (the all-zero row included) when the level's symmetric diagonal storage also
 * carries one class byte per row (the row's five or seven entries, bit for bit, from a table), 0 when it does not
 * (more than 255 distinct rows, another format, "row_classes" 0).  No reference counterpart: storage detail of jacobiRelaxation (multigrid.py:223-228). */
int mg_level_row_classes(mg_handle h, int level, int* classes);
/* Why a level has the storage it has (what `A.getValuesCSR()`, Multigrid_prototype.py:95-96, really delivered):
 *   symmetric             1 = every pair a_ij, a_ji agrees bit for bit, 2 = within `ulps_used` units in the last place (the
 *                         upper half is kept), 0 = not symmetric (the level keeps both halves), -1 = not tested (pattern not
 *                         symmetric, coarsest level, "symmetric_storage" 0)
 *   first_asymmetric_row  lowest row (lexicographic grid numbering, local to the rank) with a pair that differs, -1 = none
 *   max_pair_ulps         largest distance between the halves of a pair in units in the last place; -1 = a pair with one half
 *                         missing or of the other sign
 *   distinct_rows         distinct non-zero rows the row dictionary saw (more than 255: no row classes), -1 = not built
 *   ulps_used             0 = the stored matrix is the handed-over one bit for bit; k > 0: entries within k units in the last
 *                         place were identified ("storage_ulps", or the automatic second try "storage_auto" with k = 4)
 *   escape_rows           rows that have no class of their own although the level has row classes: the level had more than 255
 *                         distinct rows, the 254 most frequent ones became classes and these rows are read from the stored
 *                         matrix ("row_escape"); 0 = none
 * Any pointer may be null.  No reference counterpart: storage detail. */
int mg_level_storage(mg_handle h, int level, int* symmetric, int64_t* first_asymmetric_row, int64_t* max_pair_ulps,
                     int* distinct_rows, int* ulps_used, int64_t* escape_rows);

/* ---- vectors ---------------------------------------------------------------------------
 * Host <-> device copies in the caller's DoF numbering, (n,1) fp64 C-contiguous as the
 * reference's vectors (Multigrid_prototype.py:110).  In slab mode `host` still has
 * global length n: set reads the entries of owned (and halo) rows; get writes owned
 * rows only unless gather != 0, in which case slabs are all-gathered first. */
int mg_set_vector(mg_handle h, int level, int which, const double* host);
int mg_get_vector(mg_handle h, int level, int which, double* host, int gather);
int mg_zero_vector(mg_handle h, int level, int which);
int mg_copy_vector(mg_handle h, int level, int dst_which, int src_which);

/* ---- the hot path, device-resident --------------------------------------------------------
 * mg_smooth    nw weighted-Jacobi sweeps on MG_VEC_V            jacobiRelaxation   multigrid.py:223-228
 * mg_residual  MG_VEC_R = MG_VEC_F - A MG_VEC_V                 multigrid.py:244, :291
 * mg_restrict  MG_VEC_F[level-1] = R MG_VEC_R[level]            Restriction2D(_direct) multigrid.py:123-198
 * mg_prolong   MG_VEC_ERR[level] = P MG_VEC_V[level-1]; if add, MG_VEC_V[level] += it
 *                                                               Interpolation2D    multigrid.py:59-120, :260
 * mg_coarse_solve  MG_VEC_V[0] = A_0^-1 MG_VEC_F[0]             spsolve            multigrid.py:239-241
 * mg_vcycle    ncycles V(mu1,mu2) cycles on `level` starting from MG_VEC_V[level];
 *              resid_l2[c] (optional) = ||f - A v||_2 after cycle c   V_cycle_scheme multigrid.py:231-268
 * mg_norm2     out = ||x||_2 (all-reduced over slabs)           replaces the l2 part of res_calculator :203-208
 * mg_fmg       FullMultiGrid(_test): coarsest solve, interpolate up, mu0 cycles per level,
 *              on the finest either exactly mu0 cycles (tol <= 0; FullMultiGrid_test
 *              :312-339) or until ||r||_2 <= tol (FullMultiGrid :285-302).  Runs on levels
 *              0..top_level: MG_VEC_F[top_level] is the right-hand side there, the coarser
 *              levels use the true right-hand sides given with mg_set_rhs_true (b_dict,
 *              multigrid.py:279).  elements_per_dim == 0 in mg_set_level_csr declares a
 *              "flat" level (any square matrix, smoother/residual only, single GPU). */
int mg_smooth(mg_handle h, int level, int nw);
/* jacobiRelaxation on the reference's own split operands (multigrid.py:223-228), for callers that hand
 * over (D^-1 R, D^-1) rather than A: the level's matrix is D^-1 (A - D) (set with
 * mg_set_tuning("require_diagonal", 0)), MG_VEC_ERR holds the diagonal of D^-1, and every sweep is
 * (1-w) v + w (D^-1 f) - w (D^-1 R) v evaluated in the reference's order. */
int mg_smooth_split(mg_handle h, int level, int nw);
int mg_residual(mg_handle h, int level);
int mg_restrict(mg_handle h, int level, int kind);
int mg_prolong(mg_handle h, int level, int add);
int mg_coarse_solve(mg_handle h, int* iterations, double* rel_residual);
int mg_vcycle(mg_handle h, int level, int ncycles, double* resid_l2);
/* Builds and allocates now what the first V-cycle from `level` would build lazily (direct coarsest solve, colouring checks,
 * work vectors) and waits for it: the cycles that follow only enqueue work.  Optional; useful before timing, and on slabs
 * with "graph_comm" so that every rank enters its first (captured) cycle without set-up work of its own in between.  No
 * reference counterpart. */
int mg_prepare_cycle(mg_handle h, int level);
int mg_norm2(mg_handle h, int level, int which, double* out);
/* out = x^T A x for the level's matrix (one tile SpMV fused with the dot product).  With a P1 mass
 * matrix handed over as the level's matrix this is the square of the reference's L2(Omega) norm
 * (res_calculator / err_calculator, multigrid.py:203-218). */
int mg_quadratic_form(mg_handle h, int level, int which, double* out);
int mg_set_rhs_true(mg_handle h, int level, const double* host);
int mg_fmg(mg_handle h, int top_level, int mu0, double tol, int max_cycles, double* resid_l2,
           int* cycles_done);
/* FullMultiGrid with the reference's own norms, device-resident (multigrid.py:285-302: the stop test `resn <= 1e-11`
 * and the two histories use res_calculator / err_calculator = sqrt(int r_h^2), sqrt(int (u_h - u_exact)^2),
 * multigrid.py:203-218).  norm = MG_NORM_MASS evaluates them as sqrt(x^T M x) with the P1 mass matrix M of the
 * top level handed over by mg_set_mass_csr (same DoF numbering and CSR conventions as mg_set_level_csr; it stands
 * in for the dolfinx function space `V_fine_dolfx`); MG_NORM_L2 is the Euclidean norm.  err_hist (optional)
 * needs the exact solution's nodal values (mg_set_exact = `u_exact_fine`).  Per cycle only two doubles cross to
 * the host.  resid_hist / err_hist hold max(mu0, max_cycles) entries. */
enum { MG_NORM_L2 = 0, MG_NORM_MASS = 1 };
int mg_set_mass_csr(mg_handle h, int level, int64_t n_rows, int64_t nnz, const void* indptr, int indptr_is_64,
                    const int32_t* indices, const double* data);
int mg_set_exact(mg_handle h, int level, const double* host);
int mg_fmg_ex(mg_handle h, int top_level, int mu0, double tol, int max_cycles, int norm, double* resid_hist,
              double* err_hist, int* cycles_done);
/* Bookkeeping for tests: whole-vector host -> device / device -> host copies made through this handle so far,
 * V-cycles replayed from a captured hipGraph, graphs currently cached.  No reference counterpart. */
int mg_counters(mg_handle h, int64_t* uploads, int64_t* downloads, int64_t* graph_replays, int* graphs_cached);

/* ---- measurement -----------------------------------------------------------------------------
 * mg_time_kernel: average duration in milliseconds of `reps` back-to-back launches of
 * one kernel of the path on `level`, measured with HIP events on the handle's own
 * stream ("jacobi", "residual", "restrict", "prolong", "norm2"; "jacobi2" = the two-sweep pass, an
 * error on levels where mg_smooth does not use it; "jacobi2!" = the same wherever the kernel applies;
 * "jacobi_small" = the mu1 sweeps of a small level in one launch, an error where mg_smooth does not do that;
 * "jacobik" = one launch of the K-sweep 2-D kernel with K = "fuse_2d_k", an error on levels that do not use it;
 * "jacobik3" = one launch of the K-sweep plane march on a 3-D level ("jacobik3!": wherever it applies), "jacobiblk" = one launch
 * of the block pass ("fuse_block"; "jacobiblk!": whatever the level's size), errors on levels that do not use them;
 * "gs" = one full Gauss-Seidel sweep, all colours, with the configured Gauss-Seidel smoother).
 * Used by bench.py for the roofline figure.  mg_sync waits for the handle's stream. */
int mg_time_kernel(mg_handle h, const char* kernel, int level, int reps, double* avg_ms);
int mg_sync(mg_handle h);
/* bytes of device memory held by the handle */
int mg_memory_bytes(mg_handle h, int64_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* MG_HIP_H */
