/*
 * nvalchemiops_hip.h -- C ABI of libnvalchemiops_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the reference's L2 seam: the `torch.library.custom_op` wrappers that launch
 * Warp kernels (reference file:line cited per entry point).  All pointers are DEVICE pointers unless
 * marked [host]; every buffer is owned by the caller (the Python layer allocates through torch's caching
 * allocator, exactly as the reference wrappers do); `stream` is a hipStream_t passed as void*.
 * dtype: 0 = float32, 1 = float64.  Every function returns 0 on success or a negative MI_E* code;
 * mi_last_error() gives a message for the calling thread.
 *
 * State between calls: none that a result depends on.  Three things outlive a call, all documented where they are declared:
 *   - FFT plans (mi_fft_plan_*): explicit handles owned by the caller, each holding its rocFFT work area;
 *   - mi_d3's atom-order heuristic: per device, 64 bytes of pinned host memory with an earlier call's measurement of how spatially
 *     coherent the caller's atom numbering is (it picks between two code paths with bit-identical outputs; see mi_d3);
 *   - the optional per-kernel timing records of mi_timing_enable (bench.py only).
 *
 * A reference maintainer binds these with ctypes; no torch types cross this line.
 */
#ifndef NVALCHEMIOPS_HIP_H
#define NVALCHEMIOPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MI_OK 0
#define MI_EINVAL (-1)   /* bad argument                                   */
#define MI_EHIP (-2)     /* a HIP runtime call failed                      */
#define MI_EWORKSPACE (-3) /* workspace too small                          */
#define MI_ECOMM (-4)    /* RCCL missing, or an RCCL call failed           */

#define MI_F32 0
#define MI_F64 1

/* ---- library ------------------------------------------------------------------------------- */
int mi_version(void);                 /* ABI version, currently 1                                  */
const char* mi_last_error(void);      /* [host] message of the last failure on this thread         */
/* Optional per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the live
 * roofline figure).  mi_timing_report waits for the events and writes "<kernel> <launches> <total_ms>\n" lines. */
int mi_timing_enable(int on);
/* record only the kernel brackets called `name` (NULL / "": all of them): an event record is a packet of its own in the stream, ~5 us of
 * bubble on either side of a kernel, so a timed region that needs ONE kernel's duration selects it and leaves the rest back to back      */
int mi_timing_select(const char* name /*[host]*/);
int mi_timing_report(char* buf /*[host]*/, int cap);
/* same records with per-launch statistics: "<kernel> <launches> <total_ms> <median_ms> <min_ms> <max_ms>\n" */
int mi_timing_report_stats(char* buf /*[host]*/, int cap);
/* HBM calibration streams for the benchmark (no reference counterpart: the reference's protocol, benchmarks/utils.py:133-240, has no
 * in-run calibration): 16-byte-per-lane read / fill / copy streams over caller-owned device buffers, `bytes` a multiple of 16. */
int mi_calibrate_copy(const void* src, void* dst, size_t bytes, void* stream);
int mi_calibrate_fill(void* dst, size_t bytes, float value, void* stream);
int mi_calibrate_read(const void* src, size_t bytes, float* sink /* 1 float, never written in practice */, void* stream);

/* ---- neighbour list -------------------------------------------------------------------------
 * Replaces: nvalchemiops::build_cell_list + ::query_cell_list (neighborlist/cell_list.py:725,892),
 * ::batch_build_cell_list + ::batch_query_cell_list (batch_cell_list.py:739,915), the padded-matrix
 * fill of cell_list()/batch_cell_list() (cell_list.py:1358-1373), the naive ops (naive.py:221,299) and
 * get_neighbor_list_from_neighbor_matrix (neighbor_utils.py:362-441).
 *
 * One fused pipeline: per-system grid setup -> cell keys -> radix sort by cell -> cell ranges ->
 * cell-ordered position copy -> wave64-per-atom full-shell query with ballot/popcount compaction.
 * Rows are written only by their owner wave (no atomics, deterministic order), padding included.
 */
#define MI_NL_MODE_MATRIX 0   /* write neighbor_matrix / shifts / num_neighbors (+ padding)          */
#define MI_NL_MODE_COUNT 1    /* write num_neighbors only (first pass of direct CSR emission)        */
#define MI_NL_MODE_CSR 2      /* write list_ij / list_shifts at neighbor_ptr[i] (second pass)         */

#define MI_NL_HALF_FILL 1     /* keep one of (i,j,S)/(j,i,-S): S lexicographically > 0, or S==0 and j>i */
#define MI_NL_NAIVE_EXPR 2    /* distance expression / cutoff^2 rounding / image range of naive.py:37-182 */
#define MI_NL_REUSE_GRID 4    /* skip the binning stages: workspace still holds the grid of the previous call */
#define MI_NL_NO_SHIFTS 8     /* matrix mode: do not write the shifts tensor (non-periodic naive output)  */
#define MI_NL_NO_PAD 16       /* matrix mode: leave slots >= num_neighbors untouched (query_cell_list: the
                                 caller pre-filled the outputs, cell_list.py:892-1034)                    */

size_t mi_nl_workspace_bytes(int n_atoms, int n_systems, int dtype);

int mi_nl_neighbors(const void* positions,          /* [n_atoms,3] dtype                              */
                    int n_atoms,
                    const void* cell,               /* [n_systems,3,3] dtype, rows = lattice vectors   */
                    const uint8_t* pbc,             /* [n_systems,3] bool                              */
                    const int32_t* batch_idx,       /* [n_atoms] or NULL (single system)               */
                    int n_systems, double cutoff, int dtype, int mode, int flags,
                    int32_t* neighbor_matrix,       /* [n_atoms,max_neighbors]      (MATRIX)           */
                    int32_t* neighbor_matrix_shifts,/* [n_atoms,max_neighbors,3]    (MATRIX)           */
                    int32_t* num_neighbors,         /* [n_atoms]                    (MATRIX, COUNT)    */
                    int max_neighbors, int fill_value,
                    const int32_t* neighbor_ptr,    /* [n_atoms+1]                  (CSR)              */
                    int32_t* list_ij,               /* [2,n_pairs]                  (CSR)              */
                    int32_t* list_shifts,           /* [n_pairs,3]                  (CSR)              */
                    long long n_pairs,
                    const void* bin_origin,         /* [n_systems,3] dtype or NULL: origin subtracted for BINNING
                                                       only (non-periodic inputs far from 0); distances use the
                                                       caller's coordinates unchanged                        */
                    void* workspace, size_t workspace_bytes, void* stream);

/* Matrix-mode search that also leaves a PACKED COMPANION of the padded matrix (round 5): one 32-bit word per slot -- neighbour index in
 * bits 0-25, unit shift + 1 in three 2-bit fields (bits 26-31), 0xffffffff = padding -- behind a 256-byte header whose first int32 is
 * raised (non-zero) when a stored shift lies outside {-1, 0, 1} (the companion is then unusable).  It is the format the D3 passes stream
 * (mi_d3_packed below): the search has index and shift of every hit in registers when it stores the 16 bytes of the API format, so the
 * 4-byte word costs one more store there and saves the consumer a 16 B/slot read + 4 B/slot write.  No reference counterpart: the
 * reference's dftd3 kernels re-read neighbor_matrix + neighbor_matrix_shifts in every pass (dftd3.py:833-1260).  The companion describes
 * exactly what was written to neighbor_matrix / neighbor_matrix_shifts by THIS call; keeping the two in step afterwards is the caller's
 * business (the Python layer keys it on the tensors' version counters).  Requires shifts, padding, a full (not half-filled) list and
 * n_atoms < 2^26.  packed_out: mi_nl_packed_bytes(n_atoms, max_neighbors) bytes (0 = not available for these sizes).                    */
size_t mi_nl_packed_bytes(int n_atoms, int max_neighbors);
int mi_nl_neighbors_packed(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                           double cutoff, int dtype, int flags, int32_t* neighbor_matrix, int32_t* neighbor_matrix_shifts,
                           int32_t* num_neighbors, int max_neighbors, int fill_value, const void* bin_origin, void* workspace,
                           size_t workspace_bytes, void* packed_out, size_t packed_bytes, void* stream);

/* mi_nl_neighbors_packed that ALSO sums the DFT-D3 coordination numbers of the list it writes (round 6):
 *   CN_i = sum over the stored entries (j, S) of row i of 1 / (1 + exp(-k1 ((rcov[Z_i] + rcov[Z_j]) / |r_j - r_i + S.cell| - 1)))
 * -- what the reference's first D3 pass computes by walking the whole list again (`_cn_kernel_nm`, interactions/dispersion/dftd3.py:833-941,
 * `_cn_counting` :608-645).  The search has the squared distance of every hit in registers and both atoms' records in LDS, so the sum is a
 * handful of instructions inside a kernel that waits for its stores; mi_d3_packed_cn below then skips its CN pass.  Atoms with Z <= 0 or
 * Z >= nz take part in the search but neither have nor contribute a coordination number (dftd3.py:871-885).
 * cn_out: mi_nl_cn_bytes(n_atoms) bytes = a 1 KiB header + float cn[n_atoms] (caller's atom order).  Header: int32[0] is raised when the
 * numbers must not be used (a row overflowed max_neighbors: the stored list's own sum is smaller); float[2] = cutoff, float[3] = k1 log2(e);
 * 64 uint64 checksum slots at byte 256 whose wrapping sum fingerprints the inputs the numbers were computed from (every atom's index,
 * system, position bits and covalent radius, the cell, k1).  mi_d3_packed_cn recomputes the fingerprint from ITS inputs on the device and
 * only adopts the numbers on a match, so a consumer called with other positions (an MD step that reuses the list), other species or
 * another cell silently runs its own pass -- there is no host-side bookkeeping to get wrong.
 * `request` is read on the host; its pointers are device pointers.  Not combinable with MI_NL_REUSE_GRID / half-fill.                    */
typedef struct mi_nl_cn_request {
  const int32_t* numbers;       /* [n_atoms] atomic numbers                                   */
  const float* covalent_radii;  /* [nz] fp32, index 0 = padding (D3Parameters.rcov as fp32)    */
  int nz;
  float k1;                     /* steepness of the counting function (dftd3 default 16)       */
} mi_nl_cn_request;
size_t mi_nl_cn_bytes(int n_atoms);
int mi_nl_neighbors_packed_cn(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                              double cutoff, int dtype, int flags, int32_t* neighbor_matrix, int32_t* neighbor_matrix_shifts,
                              int32_t* num_neighbors, int max_neighbors, int fill_value, const void* bin_origin, void* workspace,
                              size_t workspace_bytes, void* packed_out, size_t packed_bytes, const mi_nl_cn_request* request /* [host] */,
                              void* cn_out, size_t cn_bytes, void* stream);

/* Single-sweep dual-cutoff search: ONE walk over the candidate pairs fills two padded matrices, the short list nested in the long one and
 * both with the image range of the long cutoff.  Replaces: _fill_naive_neighbor_matrix[_pbc]_dual_cutoff and the batch variants
 * (neighborlist/naive_dual_cutoff.py:36,115 / batch_naive_dual_cutoff.py:37,126).  flags as for mi_nl_neighbors (matrix mode). */
int mi_nl_neighbors_dual(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc, const int32_t* batch_idx, int n_systems,
                         double cutoff_short, double cutoff_long, int dtype, int flags,
                         int32_t* neighbor_matrix_short, int32_t* neighbor_matrix_shifts_short, int32_t* num_neighbors_short,
                         int max_neighbors_short,
                         int32_t* neighbor_matrix_long, int32_t* neighbor_matrix_shifts_long, int32_t* num_neighbors_long,
                         int max_neighbors_long, int fill_value, const void* bin_origin, void* workspace, size_t workspace_bytes,
                         void* stream);

/* padded matrix -> COO/CSR (neighbor_utils.py:362-441): entries != fill_value, row-major.
 * neighbor_ptr = [0, cumsum(num_neighbors)] supplied by the caller; shifts may be NULL.              */
int mi_nl_matrix_to_coo(const int32_t* neighbor_matrix, const int32_t* neighbor_matrix_shifts,
                        const int32_t* neighbor_ptr, int n_atoms, int max_neighbors, int fill_value,
                        int32_t* list_ij, int32_t* list_shifts, long long n_pairs, void* stream);

/* Reference-format cell-list cache (cell_list.py:725-889 / batch_cell_list.py:739-912): fills
 * cells_per_dimension, atom_periodic_shifts, atom_to_cell_mapping, atoms_per_cell_count,
 * cell_atom_start_indices, cell_atom_list with the reference's binning rule for `max_total_cells`
 * cells (atoms inside a cell are listed in ascending index order).                                   */
int mi_nl_build_cell_cache(const void* positions, int n_atoms, const void* cell, const uint8_t* pbc,
                           const int32_t* batch_idx, int n_systems, double cutoff, int dtype,
                           int max_total_cells, int32_t* cells_per_dimension, int32_t* atom_periodic_shifts,
                           int32_t* atom_to_cell_mapping, int32_t* atoms_per_cell_count,
                           int32_t* cell_atom_start_indices, int32_t* cell_atom_list, void* workspace,
                           size_t workspace_bytes, void* stream);

/* estimate_cell_list_sizes kernels (cell_list.py:35-99 / batch_cell_list.py:36-99)                   */
int mi_nl_estimate_sizes(const void* cell, const uint8_t* pbc, int n_systems, double cutoff, int max_nbins,
                         int dtype, int32_t* number_of_cells /*[n_systems]*/,
                         int32_t* neighbor_search_radius /*[n_systems,3]*/, void* stream);

/* Bounding boxes of non-periodic systems: cell_out[s] = diag(max(hi - lo, 1) * 1.001), origin_out[s] = lo, the binning frame a
 * free-space search uses when the caller gives no cell (reference: `neighbor_list` fabricates a unit cell, neighborlist.py:220-226;
 * SURVEY Appendix B.2).  `scratch`: 6 * n_systems 8-byte words.  batch_idx NULL = one system.                               */
int mi_nl_bounding_cells(const void* positions, const int32_t* batch_idx, int n_atoms, int n_systems, int dtype, void* cell_out,
                         void* origin_out, void* scratch, void* stream);

/* Rebuild detection for MD loops (neighborlist/rebuild_detection.py:37-170): `flag` (one byte, zeroed by the caller)
 * is set when any atom's cell (reference binning of mi_nl_build_cell_cache) differs from atom_to_cell_mapping, resp. when any atom
 * moved farther than `threshold` from its reference position.                                                                    */
int mi_nl_cells_changed(const void* positions, const void* cell /*[3,3]*/, const int32_t* atom_to_cell_mapping,
                        const int32_t* cells_per_dimension /*[3]*/, const uint8_t* pbc /*[3]*/, int n_atoms, int dtype,
                        uint8_t* flag, void* stream);
int mi_nl_moved_beyond_skin(const void* reference_positions, const void* current_positions, double threshold, int n_atoms, int dtype,
                            uint8_t* flag, void* stream);

/* ---- DFT-D3(BJ) -----------------------------------------------------------------------------
 * Replaces nvalchemiops::dftd3_nm (dftd3.py:1792-2122) and ::dftd3_nl (:2125-2465): CN pass, fused
 * C6-interpolation + BJ damping + energy + direct force + dE/dCN pass, chain-rule force pass.  The
 * reference's pass 0 (materialised cartesian_shifts) is folded into the passes.  Outputs are float32
 * whatever `dtype` (the positions/cell dtype) is.  Layout selector: neighbor_ptr == NULL -> padded matrix
 * (entries >= fill_value are padding), else CSR (idx_j[neighbor_ptr[i]..neighbor_ptr[i+1])).
 */
typedef struct {
  const float* rcov;   /* [nz]            */
  const float* r4r2;   /* [nz]            */
  const float* c6ab;   /* [nz,nz,5,5]     */
  const float* cn_ref; /* [nz,nz,5,5]     */
  int nz;              /* max_Z + 1       */
  float a1, a2, s6, s8, k1, k3, s5_on, s5_off;
} mi_d3_params;

size_t mi_d3_workspace_bytes(int n_atoms, int n_systems, int nz);
/* Optional larger workspace for a periodic padded matrix: base + 4 bytes per slot.  Given at least this much, mi_d3 lets its CN pass
 * leave a packed copy of the list (index + unit shift in one word) that the energy and chain passes stream instead of the caller's
 * 16-byte-per-slot arrays; lists with shifts outside {-1,0,1} or >= 2^26 atoms fall back to the arrays on the device. Results are
 * identical either way.  Equals mi_d3_workspace_bytes when max_neighbors <= 0 (CSR).                                              */
size_t mi_d3_workspace_bytes_packed(int n_atoms, int n_systems, int nz, int max_neighbors);
/* The same for any layout, by number of stored entries (CSR: neighbor_ptr[n_atoms] = mi_d3's `n_list_entries`).              */
size_t mi_d3_workspace_bytes_entries(int n_atoms, int n_systems, int nz, long long n_entries);

/* Atom-order heuristic (the one thing mi_d3 remembers between calls): with the packed list and >= 2048 atoms the passes can walk spatially
 * ordered copies of the per-atom records, which pays when the caller numbers its atoms incoherently and costs a little when it does not.
 * The choice is made on the host from a measurement an EARLIER call left in pinned host memory (asynchronous copy, no synchronisation,
 * records kept per device and per (n_atoms, n_systems) generation); both paths produce bit-identical outputs, so the state can change
 * run-to-run timing only.  Nothing is measured or published while `stream` is being captured into a HIP graph: the choice made at capture
 * time is part of the graph.  NVALCHEMIOPS_D3_SORT=0|1 in the environment forces the choice.                                         */
int mi_d3(const void* positions, const int32_t* numbers, int n_atoms, int dtype,
          const int32_t* idx_j,        /* matrix [n_atoms,max_neighbors] or CSR values [n_pairs]      */
          const int32_t* unit_shifts,  /* same layout x3, or NULL (non-periodic)                      */
          const int32_t* neighbor_ptr, /* NULL => matrix layout                                       */
          int max_neighbors,           /* matrix row width (ignored for CSR)                          */
          long long n_list_entries,    /* CSR: length of idx_j, 0 = unknown (no packed copy); ignored for the matrix layout */
          int fill_value,
          const void* cell,            /* [n_systems,3,3] dtype or NULL                               */
          const int32_t* batch_idx,    /* [n_atoms] or NULL                                           */
          int n_systems, const mi_d3_params* params /* [host] */, int compute_virial,
          float* energy /*[n_systems]*/, float* forces /*[n_atoms,3]*/, float* coord_num /*[n_atoms]*/,
          float* virial /*[n_systems,3,3] or NULL*/, void* workspace, size_t workspace_bytes, void* stream);

/* mi_d3 on a periodic padded matrix whose packed companion the search already wrote (mi_nl_neighbors_packed): the CN pass streams the
 * 4 B/slot companion instead of the 16 B/slot arrays and writes no copy of its own; the energy and chain passes read the companion too.
 * neighbor_matrix / neighbor_matrix_shifts must still be the arrays the companion was built with: they are what the passes read when the
 * companion's flag is raised (device-side choice, no host round trip), and the results are bit-identical to mi_d3's either way.
 * `workspace` as for mi_d3 (with mi_d3_workspace_bytes_packed's size the spatial order stays available for incoherently numbered atoms).
 * Requires cell, shifts, fill_value >= n_atoms, n_atoms < 2^26.                                                                          */
int mi_d3_packed(const void* positions, const int32_t* numbers, int n_atoms, int dtype, const int32_t* neighbor_matrix,
                 const int32_t* neighbor_matrix_shifts, int max_neighbors, int fill_value, const void* cell, const int32_t* batch_idx,
                 int n_systems, const mi_d3_params* params, int compute_virial, float* energy, float* forces, float* coord_num, float* virial,
                 void* workspace, size_t workspace_bytes, const void* packed_list, size_t packed_bytes /* >= mi_nl_packed_bytes(...) */,
                 void* stream);

/* mi_d3_packed with two additions (round 6):
 *  - `cn_block` (optional, NULL = none): the coordination numbers the search summed while it wrote the list (mi_nl_neighbors_packed_cn).
 *    They are adopted -- and the CN pass skipped -- iff the block's flag is clear AND the fingerprint in its header equals the one this call
 *    computes on the device from its own positions, numbers, params->rcov, params->k1, cell and batch_idx; otherwise the CN pass runs as in
 *    mi_d3_packed.  Either way coord_num is the reference's sum over the stored list to fp32 rounding (the two evaluations differ in
 *    summation order and in one rounding of the exponent: <= 1e-6 relative, tests/test_search_cn_gpu.py).
 *  - a sampled consistency check of the companion against the arrays it claims to describe: one row in every block of `verify_stride`
 *    consecutive rows (at offset (verify_phase + hash(block)) mod verify_stride: consecutive phases visit every row once in verify_stride
 *    calls, and no regular edit pattern can hide between the samples) is re-derived from neighbor_matrix / neighbor_matrix_shifts and
 *    compared word by word; on a
 *    mismatch the call falls back to the 16 B/slot arrays (device-side flag, no host round trip, results = mi_d3's on the arrays as they
 *    are NOW).  verify_stride = 1 checks every row, 0 disables the check (mi_d3_packed itself uses stride 64, phase 0).  This catches
 *    arrays edited behind the companion's back in bulk (a filtering kernel, a copy from elsewhere); a single edited entry in an unsampled
 *    row is not seen -- callers who edit lists through raw pointers must drop the companion.                                           */
int mi_d3_packed_cn(const void* positions, const int32_t* numbers, int n_atoms, int dtype, const int32_t* neighbor_matrix,
                    const int32_t* neighbor_matrix_shifts, int max_neighbors, int fill_value, const void* cell, const int32_t* batch_idx,
                    int n_systems, const mi_d3_params* params, int compute_virial, float* energy, float* forces, float* coord_num, float* virial,
                    void* workspace, size_t workspace_bytes, const void* packed_list, size_t packed_bytes, const void* cn_block, size_t cn_bytes,
                    int verify_stride, int verify_phase, void* stream);

/* ---- Ewald real space -----------------------------------------------------------------------
 * Replaces the 12 alchemiops::_[batch_]ewald_real_space_* ops (ewald.py:263-1365; kernels
 * ewald_kernels.py:266-1495): erfc(A&S 7.1.26)-damped pair sum over the stored neighbour entries.
 * energies are float64 whatever dtype is (the wrapper casts, ewald.py:577).  The reference adds -f to atom i and
 * atomically +f to atom j for every stored entry (ewald_kernels.py:518-544; charge gradients :864-873).  Over a
 * symmetric (full) list that is 2x the row owner's sum, which is what the fast path writes (no atomics).
 * `scratch` (device memory, may be NULL) has two optional parts, used when `scratch_bytes` covers them:
 *   [0, mi_ewald_symmetry_scratch_bytes())            list-symmetry checksums: when given, the same pass checksums the list in both
 * directions and, if it is NOT symmetric (half lists, rows truncated by overflow, one-sided lists), two fix-up launches
 * redo forces / charge gradients with the reference's scatter; they exit at once otherwise.  Without it the caller
 * guarantees a symmetric list.
 *   [.., mi_ewald_real_scratch_bytes(n_atoms, dtype))  {x,y,z,q} records: the pair loop then gathers one record per neighbour
 *                                                      instead of three coordinates + the charge (same arithmetic, same results).
 */
#define MI_EW_FORCES 1
#define MI_EW_CHARGE_GRAD 2
size_t mi_ewald_symmetry_scratch_bytes(void);
size_t mi_ewald_real_scratch_bytes(int n_atoms, int dtype);
int mi_ewald_real(const void* positions, const void* charges, const void* cell, const void* alpha /*[n_systems]*/,
                  const int32_t* batch_idx, int n_atoms, int dtype, const int32_t* idx_j,
                  const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors, int mask_value,
                  int flags, double* energies /*[n_atoms]*/, void* forces /*[n_atoms,3] dtype*/,
                  double* charge_grads /*[n_atoms]*/, void* scratch /*or NULL*/, size_t scratch_bytes, void* stream);
/* mi_ewald_real for a padded matrix that is KNOWN to be the unmodified output of a full (not half-filled) neighbour search: the caller passes
 * that search's num_neighbors array (counts that keep counting past the row width, as mi_nl_neighbors writes them).  A full list is symmetric
 * unless a row overflowed, so the per-entry checksums of the symmetry test are not computed (0.170 -> 0.138 ms on the 9 A headline list);
 * instead a row with search_num_neighbors[i] > max_neighbors raises the "not symmetric" mark, and every verify_stride-th row (0: none; rounded up to a power of two),
 * rotating with verify_phase, looks one of its entries (j, S) up in row j as (i, -S) and raises the mark when it is missing.  A marked call
 * takes the general scatter path of mi_ewald_real: results are those of the arrays as they ARE.  Rows are read up to their count only.
 * search_num_neighbors == NULL, a CSR list, or a scratch too small for checksums + records: exactly mi_ewald_real.  The Python host passes
 * the array only while matrix, shifts and counts carry the version counters the search left (neighborlist/_engine.py::FullListRecord).      */
int mi_ewald_real_listed(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx, int n_atoms,
                         int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors,
                         int mask_value, int flags, double* energies, void* forces, double* charge_grads, void* scratch, size_t scratch_bytes,
                         const int32_t* search_num_neighbors, int verify_stride, int verify_phase, void* stream);

/* Explicit-k reciprocal-space Ewald.  Replaces `alchemiops::_[batch_]ewald_reciprocal_space_energy[_forces
 * [_charge_grad]]` (ewald.py:1365-2318; kernels ewald_kernels.py:1496-2480).  Two passes, no [K,N] phase tables:
 *   mi_ewald_structure_factors : S[b][k] = 8pi/V exp(-k^2/4a^2)/k^2 * sum_j w_j exp(i k.r_j) (interleaved re,im; k^2<1e-10 -> 0)
 *                                total_charge[b] = sum_j w_j / V (0 when n_k < 2, as the reference's k_idx==1 accumulation)
 *   mi_ewald_recip_gather      : phi_i = sum_k Re(S_k exp(-i k.r_i)), kforce_i = sum_k (S_re sin - S_im cos) k, and from them
 *                                E_i = q phi/2 - a q^2/sqrt(pi) - pi q Q/(2a^2), F_i = q kforce_i, dE/dq_i = phi - 2a q/sqrt(pi) - pi Q/a^2
 * k_vectors [n_systems,n_k,3] in `dtype` (half-space set); system_ptr [n_systems+1] atom ranges (NULL: one system);
 * total_charge NULL in the gather = no self/background corrections (used by the adjoint).  Any output pointer may be NULL.  */
int mi_ewald_structure_factors(const void* positions, const void* weights /*[n_atoms] dtype*/, const void* k_vectors, const void* cell,
                               const void* alpha, const int32_t* system_ptr, int n_atoms, int n_systems, int n_k,
                               int max_atoms_per_system, int dtype, double* structure_factors /*[n_systems,n_k,2]*/,
                               double* total_charge /*[n_systems]*/, void* stream);
int mi_ewald_recip_gather(const void* positions, const void* charges, const void* k_vectors, const void* alpha, const int32_t* batch_idx,
                          const double* structure_factors, const double* total_charge, int n_atoms, int n_k, int dtype,
                          double* potential /*[n_atoms]*/, double* kforce /*[n_atoms,3]*/, double* energies /*[n_atoms]*/,
                          void* forces /*[n_atoms,3] dtype*/, double* charge_grads /*[n_atoms]*/, void* stream);

/* out[i] = sum_k k (weights[i] . k) (S_re cos + S_im sin)(k.r_i), float64 [n_atoms,3]: position derivative of sum_i weights_i . kforce_i at
 * fixed structure factors; one term of the adjoint of the explicit-k FORCES ("forces" is in the grad_arrays of
 * alchemiops::_[batch_]ewald_reciprocal_space_energy_forces*, ewald.py:1486-1496; the reference replays its tape).                       */
int mi_ewald_recip_gather_kk(const void* positions, const void* k_vectors, const int32_t* batch_idx, const double* structure_factors,
                             const double* weights /*[n_atoms,3]*/, int n_atoms, int n_k, int dtype, double* out /*[n_atoms,3]*/, void* stream);

/* Adjoint of mi_ewald_real w.r.t. positions / charges / cell / alpha for L = sum_i g_i E_i (replaces the Warp-tape backward of
 * the real-space ops, autograd.py:525-665 + the generated adjoints of ewald_kernels.py:266-1495).  Owner-only like the forward:
 *   dL/dr_i = sum_j (g_i+g_j) fm_ij sep_ij ; dL/dq_i = sum_j 1/2 (g_i+g_j) q_j erfc(a r)/r ;
 *   dL/dcell[s][a][b] = -sum_i g_i sum_j fm_ij sep_ij[b] S_ij[a] ; dL/dalpha[s] = -sum_i g_i sum_j q_i q_j exp(-a^2 r^2)/sqrt(pi).
 * grad_cell / grad_alpha ([n_systems,3,3] / [n_systems], float64, zeroed by the caller) may be NULL.  `symmetry_scratch` as in
 * mi_ewald_real: a list that is not symmetric gets the general adjoint (entry (i -> j) carries g_i to both ends, atomics).      */
size_t mi_ewald_real_bwd_scratch_bytes(int n_systems); /* symmetry checksums + slotted partials of the per-system sums (grad_cell / grad_alpha) */
int mi_ewald_real_bwd(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                      int n_atoms, int n_systems, int dtype, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr,
                      int max_neighbors, int mask_value, const void* grad_energies /*[n_atoms] dtype*/,
                      void* grad_positions /*[n_atoms,3] dtype*/, void* grad_charges /*[n_atoms] dtype*/,
                      double* grad_cell, double* grad_alpha, void* symmetry_scratch /*or NULL*/,
                      size_t scratch_bytes /* >= mi_ewald_symmetry_scratch_bytes(); with mi_ewald_real_bwd_scratch_bytes(n_systems) the per-system
                                              sums go through 64 slots per system instead of one atomic address */,
                      void* stream);

/* Adjoint of the explicit FORCES and CHARGE GRADIENTS of mi_ewald_real for L = sum_k w_k . F_k + sum_k v_k cg_k (second derivatives of the
 * pair sum; the reference differentiates these outputs through the Warp tape: "forces" and "charge_gradients" are in the grad_arrays of the
 * `_energy_forces*` ops, ewald.py:343-348, :606-612).  grad_forces / grad_charge_grads: either may be NULL.  Entry-wise scatter, valid for
 * any list; with `scratch` (mi_ewald_real_bwd_scratch_bytes(n_systems)) the list is checksummed in an owner-only pass first (no atomics: the
 * mirrored entry of a symmetric list contributes the same amount) and the scatter pass only runs when the list is NOT symmetric.  All gradient outputs are float64 and zeroed by the library; grad_cell / grad_alpha may be NULL.                                */
int mi_ewald_real_forces_bwd(const void* positions, const void* charges, const void* cell, const void* alpha, const int32_t* batch_idx,
                             int n_atoms, int n_systems, int dtype, const int32_t* idx_j, const int32_t* unit_shifts,
                             const int32_t* neighbor_ptr, int max_neighbors, int mask_value, const void* grad_forces /*[n_atoms,3] dtype or NULL*/,
                             const void* grad_charge_grads /*[n_atoms] dtype or NULL*/, double* grad_positions /*[n_atoms,3]*/,
                             double* grad_charges /*[n_atoms]*/, double* grad_cell /*[n_systems,3,3]*/, double* grad_alpha /*[n_systems]*/,
                             void* scratch /*or NULL; mi_ewald_real_bwd_scratch_bytes(n_systems): slotted per-system sums*/, size_t scratch_bytes, void* stream);

/* ---- cut-off Coulomb ---------------------------------------------------------------------------
 * Replaces the eight alchemiops::_[batch_]coulomb_energy[_forces]_{list,matrix} ops (interactions/electrostatics/coulomb.py:716-1330;
 * kernels :133-708).  float64 only, as the reference upcasts before the launch (:1423-1426).  Pair term over each stored entry
 * (i, j): r_ij = r_i - r_j - cell^T S, skipped when r >= cutoff or r < 1e-10; phi = erfc_AS(alpha r)/r for alpha > 0, else 1/r.
 *   energies[i] = energy_prefactor * sum_j q_i q_j phi          (plain store; the reference atomically adds into zeros)
 *   forces      : += f_ij on i and -= f_ij on j with f_ij = 1/2 q_i q_j (-phi'(r)/r) r_ij -- the reference's scatter, valid for
 *                 full, half and asymmetric lists (fp64 atomics).  NULL = energy only.  Zeroed by the library.
 * CSR when neighbor_ptr != NULL (idx_j = neighbor_list[1]); otherwise a [n_atoms,max_neighbors] matrix whose entries
 * j >= fill_value or j >= n_atoms are padding (:325).  energy_prefactor is 0.5 for every reference kernel except the energy-only
 * matrix kernels, which omit the 1/2 (:340, :623): the caller passes 1.0 for those to stay result-identical.
 * mi_coulomb_bwd is the adjoint of `energies` for L = sum_i g_i E_i w.r.t. positions / charges / cell (replaces the Warp-tape
 * backward, coulomb.py:716-1330 via autograd.py:525-665); outputs float64, zeroed by the library.                           */
int mi_coulomb(const double* positions, const double* charges, const double* cell /*[n_systems,3,3]*/, const int32_t* batch_idx,
               int n_atoms, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors,
               int fill_value, double cutoff, double alpha, double energy_prefactor, double* energies /*[n_atoms]*/,
               double* forces /*[n_atoms,3] or NULL*/, void* stream);
int mi_coulomb_bwd(const double* positions, const double* charges, const double* cell, const int32_t* batch_idx, int n_atoms,
                   int n_systems, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors,
                   int fill_value, double cutoff, double alpha, double energy_prefactor, const double* grad_energies /*[n_atoms]*/,
                   double* grad_positions /*[n_atoms,3]*/, double* grad_charges /*[n_atoms]*/, double* grad_cell /*[n_systems,3,3]*/,
                   void* stream);

/* Adjoint of the `forces` output for L = sum_k grad_forces_k . F_k (the reference lists `forces` in the ops' grad_arrays, coulomb.py:785-790,
 * :937-945 and replays its tape): second derivatives of the pair term, entry-wise scatter, any list; outputs float64, zeroed by the library. */
int mi_coulomb_forces_bwd(const double* positions, const double* charges, const double* cell, const int32_t* batch_idx, int n_atoms,
                          int n_systems, const int32_t* idx_j, const int32_t* unit_shifts, const int32_t* neighbor_ptr, int max_neighbors,
                          int fill_value, double cutoff, double alpha, const double* grad_forces /*[n_atoms,3]*/,
                          double* grad_positions /*[n_atoms,3]*/, double* grad_charges /*[n_atoms]*/, double* grad_cell /*[n_systems,3,3]*/,
                          void* stream);

/* ---- B-spline spread / gather ---------------------------------------------------------------
 * Replaces alchemiops::_[batch_]spline_spread / _gather / _gather_vec3 (spline.py:1500-2107; kernels
 * :497-676, :763-959).  Orders 1-4 use the reference's piecewise polynomials; orders 5-6 use the true
 * cardinal B-spline recursion (the reference returns 0 there: SURVEY F2).  mesh is [n_systems,nx,ny,nz].
 * `batched` selects the reference's batch-kernel weight threshold (w > 1e-8 instead of w > 0 in spread).
 *
 * Every `order` argument below may carry MI_SPLINE_REFERENCE_ORDERS: orders 5 and 6 are then evaluated as the reference evaluates them --
 * all weights ZERO (spline.py:150-193 implements orders 1-4 only: spread leaves a zero mesh, every gather returns zeros) and the
 * structure-factor exponent capped at 4 (pme_kernels.py:213-225) -- instead of as true B-splines with exponent = order.  Orders 1-4 are
 * the same either way.  The host API sets the bit per call (nvalchemiops.spline.reference_spline_orders); no library state is involved.
 */
#define MI_SPLINE_REFERENCE_ORDERS 0x100
/* Per-system cell geometry for the mesh ops in one launch: cell_inv_t = (cell^-1)^T, reciprocal_cell = 2 pi cell^-1, volume = |det|
 * (the torch.linalg.inv_ex / det calls of `_pme_reciprocal_space_impl`, pme.py:1382-1395).  All [n_systems,...] in `dtype`.       */
int mi_cell_geometry(const void* cell, int n_systems, int dtype, void* cell_inv_t, void* reciprocal_cell, void* volume, void* stream);
/* The same plus total_charge[s] = sum of the charges of system s (charges.sum() / scatter_add_, pme.py:1225,1241-1244) in ONE launch
 * (total_charge is zeroed by the library): everything the fused PME step needs before the spread.                                  */
int mi_pme_prepare(const void* cell, const void* charges, const int32_t* batch_idx, int n_atoms, int n_systems, int dtype,
                   void* cell_inv_t, void* reciprocal_cell, void* volume, void* total_charge /*[n_systems]*/, void* stream);
/* 1 when mi_spline_spread (given its workspace) runs tile-owned for this mesh / order: every mesh point is then written, so the
 * caller may skip the zero-fill of `mesh`.                                                                                          */
int mi_spline_spread_is_tiled(int n_systems, int nx, int ny, int nz, int order);
/* 1 when the tile-owned run is also the FASTER one for this atom count (host policy: >= 12 000 atoms and >= 128 mesh tiles over all systems;
 * below that the zero-fill + atomic kernel + per-atom gather cost fewer launches).  A caller that follows it passes workspace = NULL (and a
 * zeroed mesh, and spread_workspace = NULL to mi_pme_gather_finish) when this returns 0.                                                   */
int mi_spline_spread_prefers_tiles(int n_atoms, int n_systems, int nx, int ny, int nz, int order);
int mi_spline_spread(const void* positions, const void* values, const int32_t* batch_idx,
                     const void* cell_inv_t /*[n_systems,3,3]*/, int n_atoms, int n_systems, int nx, int ny,
                     int nz, int order, int batched, int dtype, void* mesh /* zeroed by caller unless mi_spline_spread_is_tiled */,
                     void* workspace /* mi_spline_spread_workspace_bytes, or NULL */, size_t workspace_bytes, void* stream);
/* With a workspace the spread runs tile-owned when every mesh dimension has a divisor e with max(order - 1, 2) <= e <= 8 (atoms
 * binned by ex*ey*ez mesh tile, LDS accumulation, every mesh point written once, no global atomics); otherwise order^2 threads per
 * atom add into the zeroed mesh.                                                                                               */
size_t mi_spline_spread_workspace_bytes(int n_atoms, int n_systems, int nx, int ny, int nz);
/* The exact size for one (order, dtype): the tile-box scratch, by far the largest part, is (e + order - 1)^3 values of the mesh dtype per
 * tile instead of the any-order fp64 bound above (256^3, fp32, order 4: 178 MB instead of 576).  mi_spline_spread accepts either size.   */
size_t mi_spline_spread_workspace_bytes_for(int n_atoms, int n_systems, int nx, int ny, int nz, int order, int dtype);
/* byte offset, inside the workspace mi_spline_spread was given, of int32[4 + n_atoms] that the tile-owned spread leaves behind (valid until
 * the workspace is reused): [0] = number of consecutive atom pairs (i, i+1) that sit in neither the same nor neighbouring mesh tiles (a
 * measure of how spatially incoherent the caller's atom order is), [1..3] unused, [4..] = the atom ids grouped by mesh tile.
 * -1 when this mesh / order runs the atomic kernel, which does not sort. */
long long mi_spline_spread_order_offset(int n_atoms, int n_systems, int nx, int ny, int nz, int order);
/* channels = 1: out[n_atoms] += sum_g mesh[g] w ; channels = 3 (mesh [..,3] interleaved) or 4 planar:
 * see mi_pme_gather below for the fused PME form.                                                     */
int mi_spline_gather(const void* positions, const void* mesh, const int32_t* batch_idx, const void* cell_inv_t,
                     int n_atoms, int n_systems, int nx, int ny, int nz, int order, int dtype,
                     void* out /*[n_atoms]*/, void* stream);
int mi_spline_gather_vec3(const void* positions, const void* charges, const void* mesh_vec3 /*[B,nx,ny,nz,3]*/,
                          const int32_t* batch_idx, const void* cell_inv_t, int n_atoms, int n_systems, int nx,
                          int ny, int nz, int order, int dtype, void* out /*[n_atoms,3]*/, void* stream);

/* Gradient of a gather w.r.t. the fractional coordinate: out[i][a] = sum_g mesh[g] * d w_i(g) / d frac_a  (frac = cell_inv_t . r;
 * the factor mesh_dims[a] of d theta / d frac is included).  Building block of the spread / gather adjoints
 * (reference: _bspline_gather_gradient_kernel, spline.py:680-760, and the Warp adjoints of the spread/gather kernels).      */
int mi_spline_gather_grad(const void* positions, const void* mesh, const int32_t* batch_idx, const void* cell_inv_t, int n_atoms,
                          int n_systems, int nx, int ny, int nz, int order, int dtype, void* out /*[n_atoms,3]*/, void* stream);

/* Second-order building blocks: the adjoint of `spline_gather_gradient` (alchemiops::_[batch_]spline_gather_gradient lists positions,
 * charges, mesh and cell_inv_t in its grad_arrays, spline.py:1750-1840, :2110-2200; the reference replays its Warp tape).  With
 * W_i(g) the 3-D weight of atom i at mesh point g and frac = cell_inv_t . r (derivatives include the mesh_dims factors):
 *   mi_spline_gather_hess_dot  out[i][b] = sum_g mesh[g] sum_a vec[i][a] d^2 W_i(g) / dfrac_a dfrac_b
 *   mi_spline_spread_grad      mesh[g]   = sum_i sum_a vec[i][a] d W_i(g) / dfrac_a          (mesh is zeroed by the call)        */
int mi_spline_gather_hess_dot(const void* positions, const void* mesh, const int32_t* batch_idx, const void* cell_inv_t,
                              const void* vec /*[n_atoms,3]*/, int n_atoms, int n_systems, int nx, int ny, int nz, int order, int dtype,
                              void* out /*[n_atoms,3]*/, void* stream);
int mi_spline_spread_grad(const void* positions, const void* vec /*[n_atoms,3]*/, const int32_t* batch_idx, const void* cell_inv_t,
                          int n_atoms, int n_systems, int nx, int ny, int nz, int order, int dtype, void* mesh /*[B,nx,ny,nz]*/,
                          void* stream);

/* ---- FFT plans (csrc/fft.cpp: hipFFT on rocFFT) ------------------------------------------------------------------------
 * 3-D real <-> complex transforms over a batch of contiguous meshes: real [batch][nx][ny][nz], complex [batch][nx][ny][nz/2+1]
 * interleaved.  Replace torch.fft.rfftn(norm="backward") / irfftn(norm="forward") of `_pme_reciprocal_space_impl` (pme.py:1398,
 * :1422, :1455-1457): both directions UNSCALED.  `inverse` = 0: R2C forward, 1: C2R inverse -- which may OVERWRITE its input
 * spectrum (rocFFT's multi-dimensional C2R; this is why torch clones before irfftn -- here the spectrum is scratch of the same step).
 * A plan owns its work area (mi_fft_plan_work_bytes, allocated at creation on the then-current device): the one object of this ABI
 * that holds device memory.  mi_fft_plan_exec allocates nothing, never synchronises, is HIP-graph capturable.                     */
int mi_fft_plan_create(int nx, int ny, int nz, int batch, int dtype, int inverse, void** plan_out);
size_t mi_fft_plan_work_bytes(const void* plan);
int mi_fft_plan_exec(void* plan, void* in, void* out, void* stream);
int mi_fft_plan_destroy(void* plan);

/* hipFFT versions as major * 10000 + minor * 100 + patch: compiled-against and loaded at run time (they can differ inside a Python process,
 * where torch has already loaded its own copy of the same SONAME).                                                                  */
int mi_fft_library_versions(int* compiled, int* loaded);

/* ---- dense DFT (csrc/dft.hip) -------------------------------------------------------------------------------------------
 * The same transforms as mi_fft_plan_exec -- real [batch][nx][ny][nz] <-> complex [batch][nx][ny][nz/2+1], both directions unscaled --
 * for ANY mesh size, evaluated from the definition (three passes of 1-D DFTs, each split once into two dense levels, n1 + n2 multiply-adds
 * per output; sincospi twiddles in double): no plan, no library
 * behind it, no state.  O(sqrt n) per output: this is the transform of last resort, used when a hipFFT plan fails its known-answer test at
 * creation (rocFFT on this stack can return a wrong transform for some shapes depending on what the process planned before),
 * and the cross-check of every other FFT path in the tests.  inverse != 0: `in` (complex) is transformed in place along x and y before
 * the z pass writes `out` -- it is scratch, as for hipFFT's multi-dimensional C2R.  Replaces torch.fft.rfftn / irfftn of
 * interactions/electrostatics/pme.py:1398, :1422, :1455-1457 on that path.                                                          */
#define MI_DFT_MAX_N 1024
int mi_dft3d(void* in, void* out, int nx, int ny, int nz, int batch, int dtype, int inverse, void* stream);

/* ---- PME reciprocal-space mesh kernels ------------------------------------------------------
 * mi_pme_green_sf: alchemiops::_[batch_]pme_green_structure_factor (pme.py:273-553, pme_kernels.py:121-331)
 * mi_pme_convolve: the torch elementwise block of _pme_reciprocal_space_impl (pme.py:1418-1419,1455-1457):
 *   conv = spec / sf2 * G ; E_d = -i k_d conv, fused into one pass that writes 1 or 4 spectra.  k and k^2 are evaluated in registers
 *   from recip_cell unless the caller's precomputed arrays are passed (pme_reciprocal_space(k_vectors=, k_squared=), pme.py:1386-1392).
 *   exponent of the sinc product before squaring, in both: `order` (true order-5/6 splines), or min(order, 4) as in the reference
 *   (pme_kernels.py:213-225) when `order` carries MI_SPLINE_REFERENCE_ORDERS.
 * mi_pme_gather_finish: spline_gather + pme_energy_corrections[_with_charge_grad] + gather_vec3 + "x2"
 *   (pme.py:1429-1477; pme_kernels.py:340-657) fused over the 4 planar real-space meshes.  add_energies / add_forces /
 *   add_charge_grads (NULL ok): the real-space part (mi_ewald_real outputs: float64 energies and charge gradients, forces in
 *   `dtype`) added in the epilogue -- the `real + reciprocal` sums of particle_mesh_ewald (pme.py:1975-1990).
 *   spread_workspace selects the tile-staged kernel (round 4); results equal the per-atom kernel's up to the order of the order^3 additions.
 */
int mi_pme_green_sf(const void* k_squared /*[B,nx,ny,nzr]*/, const void* alpha /*[B]*/, const void* volume /*[B]*/,
                    int n_systems, int nx, int ny, int nz, int sf_exponent, int dtype, void* green /*[B,nx,ny,nzr]*/,
                    void* sf_sq /*[nx,ny,nzr]*/, void* stream);
int mi_pme_convolve(const void* spec /*complex [B,nx,ny,nzr]*/, const void* recip_cell /*[B,3,3] = 2pi inv(cell)*/,
                    const void* alpha /*[B]*/, const void* volume /*[B]*/, int n_systems, int nx, int ny, int nz,
                    int sf_exponent, int with_field, int dtype,
                    const void* k_vectors /*[(B,)nx,ny,nzr,3] or NULL*/, const void* k_squared /*[(B,)nx,ny,nzr] or NULL*/,
                    int k_batched /*the k arrays carry a leading system dimension*/, void* out /*complex [B,(1|4),nx,ny,nzr]*/, void* stream);
/* Fused mesh solve: everything between the spread and the gather of pme.py:1398-1461 -- rfftn, (spec / sf^2) G, the three field spectra
 * -i k_d conv, and the four inverse transforms (norm="forward": unscaled) -- for a mesh whose sizes are products of 2, 3 and 5 (round 6: mixed-radix lines; powers of two only before), in four kernels (+ one tiny table
 * launch into the scratch) that keep a whole (y,z) plane resp. 16 x-columns in LDS (csrc/fft_lds.h): mesh [n_systems][nx][ny][nz] real -> real_out [n_systems][1|4][nx][ny][nz]
 * (potential, then E_x, E_y, E_z when with_field), exactly what mi_fft_plan_exec (R2C) -> mi_pme_convolve -> mi_fft_plan_exec (C2R) leave,
 * without the charge spectrum as a by-product (callers that need it -- the adjoint of the autograd node -- keep the three-call form, or
 * mi_pme_solve_keep below).  No state outlives the call.
 * The inverse follows numpy / torch irfftn in not reading the imaginary parts of the DC and Nyquist bins along z.
 * supported: nx, ny in {8..256}, nz in {8..512} and even, every size a product of 2, 3 and 5 (96, 100, 120 ... as well as the powers of two), one (ny, nz/2+1) complex plane + its tables within 160 KB of LDS
 * (fp64: 128 x 128, 64 x 256; fp32: 256 x 128, 128 x 256).  k is evaluated from recip_cell (2 pi cell^-1, mi_pme_prepare).            */
int mi_pme_solve_supported(int n_systems, int nx, int ny, int nz, int dtype);
/* 1 when the fused solve is the path to take (host policy, measured: since round 5 wherever it is supported -- at parity with hipFFT's
 * batched plans for batches of small meshes, ahead for single and large meshes, and independent of rocFFT) */
int mi_pme_solve_preferred(int n_systems, int nx, int ny, int nz, int dtype);
size_t mi_pme_solve_scratch_bytes(int n_systems, int nx, int ny, int nz, int n_channels /*1 | 4*/, int dtype);
int mi_pme_solve(const void* mesh, const void* recip_cell /*[n_systems,3,3]*/, const void* alpha /*[n_systems]*/,
                 const void* volume /*[n_systems]*/, int n_systems, int nx, int ny, int nz, int order /* as mi_pme_convolve */,
                 int with_field, int dtype, void* scratch, size_t scratch_bytes, void* real_out, void* stream);
/* The same, and additionally the charge spectrum itself: spectrum_out (NULL ok) [n_systems][nx][ny][nz/2+1] complex, UNSCALED forward
 * transform of `mesh` in natural frequency order -- what mi_fft_plan_exec (R2C) / torch.fft.rfftn leave -- written by the forward column
 * kernel as a by-product (one 16-byte store per bin).  For callers that need it later (the backward of the autograd node).  Prepared at the
 * end of round 4: index arithmetic checked on the host (tests/test_fft_lds_cpu.py), not yet run or timed on a GPU; the Python host code
 * uses it only under NVALCHEMIOPS_PME_SOLVE_AUTOGRAD=1.                                                                                  */
int mi_pme_solve_keep(const void* mesh, const void* recip_cell, const void* alpha, const void* volume, int n_systems, int nx, int ny, int nz,
                      int order, int with_field, int dtype, void* scratch, size_t scratch_bytes, void* real_out, void* spectrum_out,
                      void* stream);
/* The mesh solve's kernels as 3-D transforms on their own (round 6): real [batch][nx][ny][nz] <-> half spectrum [batch][nx][ny][nz/2+1]
 * (interleaved re, im; natural frequency order), UNSCALED in both directions -- layout and scaling of mi_fft_plan_exec / of
 * torch.fft.rfftn(norm="backward") and irfftn(norm="forward") (pme.py:1398, :1422, :1455-1457) -- without hipFFT: inverse = 0: plane kernel +
 * column kernel (R2C), inverse = 1: column kernel + plane kernel (C2R; like numpy / torch it does not read the imaginary parts of the DC
 * and Nyquist bins along z, and unlike hipFFT's multi-dimensional C2R it leaves its input untouched).  Supported
 * (mi_fft_lds_supported) where mi_pme_solve_supported(batch, nx, ny, nz, dtype) is and the plane kernel's two extra index tables still fit.  scratch: mi_fft_lds_scratch_bytes (one half spectrum + the per-shape tables);
 * nothing outlives the call, nothing is allocated or synchronised: capturable in a HIP graph.  The backward of the autograd node, the
 * composed differentiable path and the spectrum-keeping callers take these instead of a hipFFT plan wherever the mesh allows.          */
int mi_fft_lds_supported(int batch, int nx, int ny, int nz, int dtype);
size_t mi_fft_lds_scratch_bytes(int batch, int nx, int ny, int nz, int dtype);
int mi_fft_lds(const void* in, void* out, int batch, int nx, int ny, int nz, int dtype, int inverse, void* scratch, size_t scratch_bytes,
               const void* tables /*NULL, or a block filled by mi_fft_lds_tables for this (nx, ny, nz, dtype)*/, void* stream);
/* The per-shape tables of the in-LDS kernels (unit roots, Miller index and sinc of every slot: a few KB; they depend on nx, ny, nz and the
 * dtype only).  Every call of mi_fft_lds / mi_pme_solve* computes them into its scratch with one small launch unless the caller passes a
 * block it filled ONCE with mi_fft_lds_tables -- caller-owned memory, read-only afterwards, shareable by any number of calls and streams
 * once that launch has completed (saves ~6 us per call: the Python host keeps one block per mesh shape and device).                       */
size_t mi_fft_lds_tables_bytes(int nx, int ny, int nz, int dtype);
int mi_fft_lds_tables(int nx, int ny, int nz, int dtype, void* tables, void* stream);
/* mi_pme_solve_keep with the tables handed in (NULL = computed by the call, as in mi_pme_solve / mi_pme_solve_keep) */
int mi_pme_solve_tabled(const void* mesh, const void* recip_cell, const void* alpha, const void* volume, int n_systems, int nx, int ny, int nz,
                        int order, int with_field, int dtype, void* scratch, size_t scratch_bytes, void* real_out, void* spectrum_out,
                        const void* tables, void* stream);
/* Adjoint of the k-space pass (the backward of the fused forward under autograd, pme.py `_FusedPME` / `_FusedReciprocal`).  `spec`: unscaled
 * spectrum of the charge mesh [n_systems][nx][ny][nz/2+1]; `weight_spec`: spectra of the spread upstream weights, channel-major
 * [n_channels][n_systems][...] -- channel 0 for a loss on the energies (weights g_E q), channels 1..3 (n_channels = 4) for a loss on the explicit
 * forces (weights 2 g_F,d q).  Writes conv_out = D (A_E_hat + sum_d i k_d A_d_hat), whose unscaled inverse transform is dL/d(charge mesh), and per
 * system and block 20 sums: {-V dL/dV, dL/dalpha, dL/d(2 pi cell^-1)[3][3] through k^2, the same through the explicit k_d of the field}: closed
 * form of what the reference's Warp tape + torch autograd produce for pme_kernels.py:121-331 / pme.py:1398-1457.  partial:
 * [n_systems][mi_pme_convolve_bwd_blocks()][20] doubles, every entry written; the caller folds the block dimension.                     */
int mi_pme_convolve_bwd(const void* spec, const void* weight_spec, int n_channels, const void* recip_cell, const void* alpha, const void* volume,
                        int n_systems, int nx, int ny, int nz, int order, int dtype, void* conv_out, double* partial, void* stream);
int mi_pme_convolve_bwd_blocks(void);

int mi_pme_gather_finish(const void* positions, const void* charges, const int32_t* batch_idx, const void* cell_inv_t,
                         const void* meshes /*[B,(1|4),nx,ny,nz] real*/, const void* alpha, const void* volume,
                         const void* total_charge /*[B]*/, int n_atoms, int n_systems, int nx, int ny, int nz,
                         int order, int with_field, int dtype, void* energies /*[n_atoms]*/,
                         void* forces /*[n_atoms,3] or NULL*/, void* charge_grads /*[n_atoms] or NULL*/,
                         const double* add_energies, const void* add_forces, const double* add_charge_grads,
                         const void* spread_workspace /*NULL, or the workspace of the tile-owned mi_spline_spread of the SAME step (same positions,
                           mesh, order; see mi_spline_spread_is_tiled): atoms grouped by mesh tile + stencil starts + fractional offsets -> the
                           tile-staged gather (mesh boxes through LDS); NULL -> the per-atom gather*/,
                         void* stream);
int mi_pme_corrections(const void* raw, const void* charges, const int32_t* batch_idx, const void* volume,
                       const void* alpha, const void* total_charge, int n_atoms, int dtype, void* energies,
                       void* charge_grads /*NULL ok*/, void* stream);
/* per-system sum of charges (charges.sum() / scatter_add_, pme.py:1225,1241-1244); out zeroed by caller */
int mi_segment_sum(const void* values, const int32_t* batch_idx, int n_atoms, int dtype, void* out /*[B]*/, void* stream);

/* ---- multi-GPU: the one collective of the path (csrc/comm.cpp) ------------------------------------------------------------
 * The reference has no multi-GPU code; the path shards at SYSTEM granularity -- no pair crosses systems
 * (batch_cell_list.py:448-457, dftd3.py:823,1126, spline.py:809) -- so every rank runs the kernels above on its own contiguous range of
 * systems and the only exchange is an all-gather of per-system values (D3 energies fp32, PME energies fp32 / fp64): RCCL over xGMI, one
 * rank per GPU.  Python programs use nvalchemiops.distributed (torch.distributed, backend "nccl" = RCCL); these entry points give a plain
 * C / C++ caller of this ABI the same step.  RCCL is bound at run time (the copy already loaded in the process, else librccl.so.1 from the
 * ROCm installation); MI_ECOMM if there is none.
 *   rank 0: mi_comm_unique_id(id) -> ship the MI_COMM_ID_BYTES to every rank by any host channel (MPI, a socket, a file);
 *   every rank, with its GPU current (hipSetDevice): mi_comm_init(id, ..., n_ranks, rank, &comm)   -- collective, blocks until all joined;
 *   per step: mi_comm_allgather_f32(comm, mine, all, count_per_rank, stream)  -- all[r * count .. (r + 1) * count) = rank r's `mine`;
 *             equal counts on every rank (pad ragged shards to the largest, as nvalchemiops.distributed.all_gather_system_values does);
 *             enqueued on `stream`, in order with the kernels that produced `mine`; in place when mine == all + rank * count;
 *   mi_comm_destroy(comm) after the stream has drained.                                                                            */
#define MI_COMM_ID_BYTES 128
int mi_comm_library_version(int* version /*[host] RCCL's NCCL_VERSION_CODE*/);
int mi_comm_unique_id(void* id_out /*[host] MI_COMM_ID_BYTES*/, size_t bytes);
int mi_comm_init(const void* id /*[host]*/, size_t bytes, int n_ranks, int rank, void** comm_out);
int mi_comm_size(const void* comm, int* n_ranks /*NULL ok*/, int* rank /*NULL ok*/);
int mi_comm_allgather_f32(void* comm, const float* send, float* recv /*[n_ranks * count_per_rank]*/, size_t count_per_rank, void* stream);
int mi_comm_allgather_f64(void* comm, const double* send, double* recv, size_t count_per_rank, void* stream);
int mi_comm_destroy(void* comm);

#ifdef __cplusplus
}
#endif
#endif /* NVALCHEMIOPS_HIP_H */
