/* spconv_amd -- C ABI of the MI355X (gfx950) sparse-convolution hot path.
 *
 * This header is the drop-in boundary: every entry point takes plain device
 * pointers, sizes and a HIP stream (no torch / tensorview types) and replaces
 * one native call that traveller59/spconv's Python layer makes into its
 * pybind module `spconv.core_cc` (classes SpconvOps / ConvGemmOps, signatures in
 * spconv/core_cc/csrc/sparse/all/__init__.pyi and convops/spops.pyi).  The
 * reference interface each function replaces is cited as file:line relative to
 * the reference tree.  INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *  - All pointers are DEVICE pointers unless the name ends in `_h` (host).
 *  - All functions return 0 on success and a negative value on error; the
 *    message is available from spx_last_error() (thread local).  This replaces
 *    the C++ exceptions TV_ASSERT_RT_ERR / TV_THROW_RT_ERR of the reference.
 *  - No function allocates device memory: outputs and scratch ("ws") are
 *    supplied by the caller, exactly like the reference's ExternalAllocator
 *    contract (spconv/csrc/sparse/alloc.py:38-189, pytorch/cppcore.py:112-223).
 *    spx_*_ws_bytes() give the required scratch size (cf. all.py:1580-1605).
 *  - Work is enqueued on `stream` (a hipStream_t passed as void*; the reference
 *    passes the CUDA stream as an integer, pytorch/cppcore.py:98-99).  Nothing
 *    synchronises except spx_conv_rulebook_count (one D->H read of N_out, the
 *    same unavoidable read as indices.py:1454-1455).
 *  - indices: int32 [N, ndim+1] rows (batch, z, y, x), 1 <= ndim <= 4
 *    (pytorch/core.py:148,163).  Offsets are numbered k = (r0*K1 + r1)*K2 + r2,
 *    last spatial dim fastest (indices.py:114-136).
 *  - weight: KRSC [K, *ksize, C] contiguous (pytorch/conv.py:136-139).
 */
#ifndef SPCONV_AMD_H_
#define SPCONV_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_MAX_NDIM 4

typedef void *spx_stream_t; /* hipStream_t */

enum spx_dtype { SPX_F32 = 0, SPX_F16 = 1, SPX_BF16 = 2, SPX_I8 = 3 };

/* tv::gemm::Activation subset used by the path (csrc/sparse/inference.py:26-146) */
enum spx_act { SPX_ACT_NONE = 0, SPX_ACT_RELU = 1, SPX_ACT_SIGMOID = 2, SPX_ACT_LEAKY_RELU = 3 };
/* OR-ed into the `act` argument of spx_igemm_fwd: the result rows are stored with the default cache
 * policy instead of non-temporally.  For a layer whose output the NEXT launch reads (a BatchNorm right
 * behind the convolution): it then finds the rows in the caches (config 4: 3.54 -> 3.49 ms of kernels per
 * step).  Without it the rows leave the L2 as they are written, which is what a layer measured alone
 * wants (config 2: 37.4 -> 33.6 us per step). */
#define SPX_OUT_CACHED 0x100
/* OR-ed into the `act` argument of spx_igemm_fwd_int8 (which has no tile_order argument of its own):
 * `pair` and `mask` are stored in tile order, i.e. row t of the tables belongs to output row
 * argsort[t] (spx_permute_tables), as for tile_order = 1 of spx_igemm_fwd. */
#define SPX_TILE_ORDER 0x200

/* Error text of the last failing call on this thread ("" if none). */
const char *spx_last_error(void);

/* Library ABI version (major*1000 + minor). */
int spx_version(void);

/* Run-time override of a kernel-selection switch (same names as the SPX_* environment variables the
 * library reads, e.g. "SPX_CONV_V" = 2 | 3: generation of the strided-conv rulebook passes).  Stands in for the
 * per-process tuner state of the reference (ConvTunerSimple, csrc/sparse/convops.py:919-1466): tests and
 * the benchmark use it to run two kernel generations against each other inside one process.  Host only. */
int spx_set_option(const char *name_h, int value);

/* Diagnostics: how many launches of a kernel family this process has enqueued (or captured) so far -- "igemm_v4" (128 /
 * 64-row gather-GEMM tiles), "igemm_ws" (weight-stationary 512-row workgroups of dense C = K = 64 layers),
 * "igemm_bwd" (fused dgrad + wgrad launch), "igemm_bwd_rows" (one-gather backward), "igemm_i8_stream", "generic";
 * -1 for an unknown name.  The role of the reference's tuner record (which algorithm a layer was given:
 * ConvTunerSimple, csrc/sparse/convops.py:919-1466): lets a test or a benchmark say which kernel a call REALLY took
 * when the choice depends on an asynchronously read density class.  Host only. */
long long spx_launch_count(const char *family_h);

/* dst[r] = src[r] followed by zeros: rows of src_row_bytes bytes widened to dst_row_bytes (both even), one launch.  The
 * channel padding of layers whose widths the MFMA kernels are not instantiated for (the 3-5 channel first layer of a
 * voxel backbone: the reference pads nothing and tunes a kernel per shape, csrc/sparse/convops.py:919-1466). */
int spx_pad_rows(const void *src, void *dst, long long rows, int src_row_bytes, int dst_row_bytes,
                 spx_stream_t stream);

/* ops.get_conv_output_size / get_deconv_output_size (pytorch/ops.py:73-96). Host only. */
int spx_conv_out_shape(int ndim, const int *in_shape, const int *ksize, const int *stride,
                       const int *padding, const int *dilation, const int *out_padding,
                       int transposed, int *out_shape);

/* ------------------------------------------------------------------ rulebook */

/* Scratch bytes for spx_subm_rulebook (hash table + compaction counters). */
size_t spx_subm_rulebook_ws_bytes(int n, int kv);

/* SubM rulebook.  Replaces SpconvOps.generate_subm_conv_inds[_cpu]
 * (csrc/sparse/indices.py:1495-1599 GPU, :1639-1708 CPU; drivers all.py:628-660,
 * pytorch/ops.py:202-233,505-565).
 *   pair_fwd  [kv, n]  out: pair_fwd[k][o] = input index feeding output o through
 *                      offset k, or -1 (implicit-GEMM layout, indices.py:806-874)
 *   pair_bwd  [kv, n]  or NULL: pair_bwd[k][i] = output index fed by input i
 *   mask      [n, ceil(kv/32)] uint32: bit k set iff pair_fwd[k][o] >= 0
 *                      (centre bit always set, indices.py:1576-1577)
 *   pair_native [2, kv, n] or NULL: ConvAlgo.Native lists, identical (including
 *                      order and -1 fill) to the CPU path indices.py:1639-1708
 *   num_per_loc [kv]   or NULL: counts for k < kv/2 only, like the CPU path
 * Duplicate coordinates: the smallest index wins (CPU unordered_map::insert). */
int spx_subm_rulebook(const int32_t *indices, int n, int ndim, int batch_size,
                      const int *spatial_shape, const int *ksize, const int *dilation,
                      int32_t *pair_fwd, int32_t *pair_bwd, uint32_t *mask,
                      int32_t *pair_native, int32_t *num_per_loc,
                      void *ws, size_t ws_bytes, spx_stream_t stream);

/* Scratch bytes for the two-phase regular/transposed conv rulebook.  The output hash table is
 * sized for N * prod_i ceil(k_i * gcd(d_i, s_i) / s_i) distinct outputs (kv * N when transposed):
 * SpconvOps.get_handcrafted_max_act_out (all.py:1557-1578) with dilation taken into account;
 * `dilation` may be NULL (= all ones).  spx_conv_rulebook_count fails loudly if the bound is
 * ever exceeded. */
size_t spx_conv_rulebook_ws_bytes(int n_in, int ndim, const int *ksize, const int *stride,
                                  const int *dilation, int transposed);

/* Regular / transposed conv rulebook, phase 1: hash the candidate output
 * coordinates and count the distinct ones.  Replaces stage1 + unique
 * (indices.py:501-597,997-1016,1118-1455; pytorch/ops.py:566-642).
 * Writes the count to *n_out_h after synchronising `stream` (the one D->H read); with
 * n_out_h == NULL nothing is read and the count stays in the workspace (spx_conv_rulebook_static).
 * `ws` must be passed unchanged to spx_conv_rulebook_fill. */
int spx_conv_rulebook_count(const int32_t *indices, int n_in, int ndim, int batch_size,
                            const int *in_shape, const int *out_shape, const int *ksize,
                            const int *stride, const int *padding, const int *dilation,
                            int transposed, void *ws, size_t ws_bytes, int *n_out_h,
                            spx_stream_t stream);

/* Phase 2: number the outputs in the CPU path's first-seen order (k-major,
 * then input-major, indices.py:1742-1771) and fill every artefact.  Replaces
 * assign_output + stage2 (indices.py:417-499,599-721; ops.py:646-714).
 *   out_indices [n_out, ndim+1]
 *   pair_fwd [kv, n_out], pair_bwd [kv, n_in]  (-1 = absent)
 *   mask_fwd [n_out, W], mask_bwd [n_in, W] (or NULL)
 *   pair_native [2, kv, n_in] (or NULL), num_per_loc [kv] (or NULL): identical
 *   to SparseConvIndicesCPU::generate_conv_inds (indices.py:1710-1778). */
int spx_conv_rulebook_fill(const int32_t *indices, int n_in, int ndim, int batch_size,
                           const int *in_shape, const int *out_shape, const int *ksize,
                           const int *stride, const int *padding, const int *dilation,
                           int transposed, int n_out, int32_t *out_indices,
                           int32_t *pair_fwd, int32_t *pair_bwd, uint32_t *mask_fwd,
                           uint32_t *mask_bwd, int32_t *pair_native, int32_t *num_per_loc,
                           void *ws, size_t ws_bytes, spx_stream_t stream);

/* Static-shape form of the two phases above: room for n_out_cap outputs, NOTHING read back, every
 * launch stream-ordered -- the call can be captured in a hipGraph and replayed on new coordinates of
 * the same n_in.  It is the GPU side of the reference's bounded inference mode (num_out_act_bound,
 * ops.py:263-266,644-645; the pre-sized workspace of csrc/sparse/all.py:2030-2185).
 *   - input rows with batch index < 0 are dead rows (padding up to the static n_in): they create no
 *     output and no pair
 *   - outputs are numbered in the CPU path's first-seen order; the first n_out_cap survive, pairs
 *     into later ones are dropped (as with spx_conv_rulebook_fill and n_out < the count)
 *   - out_indices rows past the number of outputs are -1 (dead rows for the next layer), their
 *     pair_fwd column is -1 and their mask 0
 *   - pair_native / num_per_loc (or NULL): the ConvAlgo.Native lists as spx_conv_rulebook_fill writes
 *     them (training: the weight-gradient kernels read them); dead rows are in no list
 *   - n_out_dev [2] (device): {distinct outputs found -- may exceed n_out_cap --, hash-table
 *     overflow flag: more distinct candidates than the table sized for 2 x n_out_cap holds};
 *     the caller reads it whenever it next synchronises
 * spx_conv_rulebook_count with n_out_h == NULL is the same count without the read. */
int spx_conv_rulebook_static(const int32_t *indices, int n_in, int ndim, int batch_size,
                             const int *in_shape, const int *out_shape, const int *ksize,
                             const int *stride, const int *padding, const int *dilation,
                             int transposed, int n_out_cap, int32_t *out_indices,
                             int32_t *pair_fwd, int32_t *pair_bwd, uint32_t *mask_fwd,
                             uint32_t *mask_bwd, int32_t *pair_native, int32_t *num_per_loc,
                             int32_t *n_out_dev, void *ws, size_t ws_bytes, spx_stream_t stream);

/* ---- sorted-order levels ---------------------------------------------------------------------------
 * The reference's GPU path does not fix the order of a strided convolution's outputs: they come out of a
 * sort + unique of the linear coordinate keys (csrc/sparse/all.py:1533-1552 apply_thrust_unique_to_indice_pairs_uniq)
 * or out of a hash table in slot order (csrc/sparse/indices.py:1380-1425); only the CPU path is first-seen
 * (indices.py:1742-1771, what spx_conv_rulebook_fill reproduces).  The calls below produce the SORTED order
 * (batch-major linear key, last spatial dimension fastest) without a sort and without a hash table, through the
 * RANK MAP of the output level: one {occupancy bits, number of occupied cells before the word} pair per 32
 * consecutive keys (the prefix counted inside a block of 2048 words, the blocks' own offsets behind the words); row of a
 * key = block offset + prefix + popcount of the bits below it.  (Built from a byte per cell in the workspace -- plain
 * idempotent stores, no atomics -- that a prefix pass packs into the words.)  The map is the caller's buffer
 * (spx_rankmap_bytes; 0 = key space beyond 2^31 cells: keep the hash builder) and stays valid after the call:
 * spx_subm_rulebook_ranked builds the SubM rulebook of a layer BEHIND the strided one from it -- no table fill,
 * no insert, one 8-byte load per neighbour query.  Rows in key order put x-neighbours in adjacent rows, which is
 * what the gather kernels of the level gain (tools/order_probe.py).
 * Values per coordinate are those of the first-seen build; only the row numbering differs.
 * spx_conv_sorted_ok: 1 when the geometry has compact candidates (at most 8 per input and at most half of the
 * offsets: k3 s2, k2 s2, k3 s3, ...; not transposed, not stride 1, not (3,1,1)/(2,1,1)) and the key space fits. */
size_t spx_rankmap_bytes(int ndim, int batch_size, const int *shape);
int spx_conv_sorted_ok(int ndim, int batch_size, const int *in_shape, const int *out_shape, const int *ksize,
                       const int *stride, const int *padding, const int *dilation, int transposed);
size_t spx_conv_rulebook_sorted_ws_bytes(int n_in, int ndim, int batch_size, const int *out_shape,
                                         const int *ksize);
/* phase 1 (as spx_conv_rulebook_count: the one D->H read of the count), phase 2 (as spx_conv_rulebook_fill)
 * and the static-shape form (as spx_conv_rulebook_static, with n_out_dev [3]: {outputs found, 0 -- a rank map cannot
 * overflow --, live output rows = min(found, n_out_cap)}).
 * `rankmap` and `ws` must be passed unchanged from phase 1 to phase 2. */
int spx_conv_rulebook_count_sorted(const int32_t *indices, int n_in, int ndim, int batch_size,
                                   const int *in_shape, const int *out_shape, const int *ksize,
                                   const int *stride, const int *padding, const int *dilation,
                                   void *rankmap, size_t rankmap_bytes, void *ws, size_t ws_bytes,
                                   int *n_out_h, spx_stream_t stream);
int spx_conv_rulebook_fill_sorted(const int32_t *indices, int n_in, int ndim, int batch_size,
                                  const int *in_shape, const int *out_shape, const int *ksize,
                                  const int *stride, const int *padding, const int *dilation, int n_out,
                                  int32_t *out_indices, int32_t *pair_fwd, int32_t *pair_bwd,
                                  uint32_t *mask_fwd, uint32_t *mask_bwd, int32_t *pair_native,
                                  int32_t *num_per_loc, void *rankmap, size_t rankmap_bytes, void *ws,
                                  size_t ws_bytes, spx_stream_t stream);
int spx_conv_rulebook_static_sorted(const int32_t *indices, int n_in, int ndim, int batch_size,
                                    const int *in_shape, const int *out_shape, const int *ksize,
                                    const int *stride, const int *padding, const int *dilation,
                                    int n_out_cap, int32_t *out_indices, int32_t *pair_fwd,
                                    int32_t *pair_bwd, uint32_t *mask_fwd, uint32_t *mask_bwd,
                                    int32_t *pair_native, int32_t *num_per_loc, int32_t *n_out_dev,
                                    void *rankmap, size_t rankmap_bytes, void *ws, size_t ws_bytes,
                                    spx_stream_t stream);
/* The rank map of a level whose rows ALREADY are in ascending, unique key order -- level 1 of a backbone whose data
 * loader sorts the voxels it hands over (the reference's own GPU builders emit such rows for every strided level:
 * csrc/sparse/all.py:1533-1552; its voxelisers emit point / hash-slot order, pytorch/utils.py:23-160).  Row = rank:
 * two launches (a fill of the map, one pass over the rows), no marks, no scan, no atomics.  `violation` (device
 * int32 [1], or NULL) is set to 1 when a live row's key is not above its predecessor's or a live row follows a dead
 * one (batch -1 rows must trail); the caller decides when to read it.  With the map attached, the level's SubM layers
 * take spx_subm_rulebook_ranked (no hash table) -- replaces, for such levels, the insert + probe passes of
 * csrc/sparse/indices.py:723-741,806-874. */
int spx_rankmap_from_sorted(const int32_t *indices, int n, int ndim, int batch_size, const int *spatial_shape,
                            void *rankmap, size_t rankmap_bytes, int32_t *violation, spx_stream_t stream);
/* Rows in key order when the caller's are not: order[t] = the row with the t-th smallest linear coordinate key
 * (batch-major, last axis fastest), rows that are dead (batch -1) or out of range behind every live row in their own
 * order; `indices_sorted` (or NULL) receives indices[order[t]] (dead rows: -1s).  The keys of a level are unique, so
 * this is four launches: one stable radix pass on the upper key bits (<= 511 buckets of 2^sh consecutive keys,
 * sh <= 20), then a workgroup per bucket ranks its rows through an occupancy bit per key in LDS -- nothing is compared.
 * A coordinate that occurs twice still yields a permutation (its extra rows land behind their bucket's distinct keys);
 * spx_rankmap_from_sorted raises its flag on such a result.  Nothing is read back (hipGraph-safe): a captured pass
 * sorts its scene at the entry (StaticInference / StaticTrainingStep entry_sort), hands spx_rankmap_from_sorted the
 * result and runs every level in key order.  With `rankmap` (spx_rankmap_bytes() bytes, or NULL) the bucket pass leaves
 * the rank map of indices_sorted behind as well -- what spx_rankmap_from_sorted would build from it; its fill rides
 * in the sort's first launch: no pass of its own -- and `violation` (device int32 [1] or NULL: cleared by the first launch) is
 * raised when a coordinate occurs twice.  `rows` / `rows_sorted` (or NULL; row_bytes a multiple of 4): rows that travel with
 * the sort -- the level's features -- rows_sorted[t] = rows[order[t]], written by the same bucket pass (no gather launch).
 * Needs batch x grid <= 0xffe00000.  The reference sorts keys where it
 * wants this order (thrust sort + unique of the output keys, csrc/sparse/all.py:1533-1552). */
size_t spx_key_argsort_ws_bytes(int n);
int spx_key_argsort(const int32_t *indices, int n, int ndim, int batch_size, const int *spatial_shape, int32_t *order,
                    int32_t *indices_sorted, void *rankmap, size_t rankmap_bytes, int32_t *violation, const void *rows,
                    void *rows_sorted, int row_bytes, void *ws, size_t ws_bytes, spx_stream_t stream);
/* SubM rulebook (outputs as spx_subm_rulebook, bit for bit) of a level whose rows are in key order and whose
 * rank map a sorted-order build left behind: `indices` must be that build's out_indices (rows past its count:
 * batch -1). */
size_t spx_subm_rulebook_ranked_ws_bytes(int n, int kv);
int spx_subm_rulebook_ranked(const int32_t *indices, int n, int ndim, int batch_size,
                             const int *spatial_shape, const int *ksize, const int *dilation,
                             int32_t *pair_fwd, int32_t *pair_bwd, uint32_t *mask,
                             int32_t *pair_native, int32_t *num_per_loc, const void *rankmap,
                             size_t rankmap_bytes, void *ws, size_t ws_bytes, spx_stream_t stream);

/* mask_argsort: permutation that groups rows with equal masks (stable, ascending
 * mask value).  Replaces SpconvOps.sort_1d_by_key_allocator (all.py:935-991). */
size_t spx_mask_argsort_ws_bytes(int n);
int spx_mask_argsort(const uint32_t *mask, int n, int words, int32_t *argsort,
                     void *ws, size_t ws_bytes, spx_stream_t stream);
/* The same sort when the caller knows the kernel volume: only the low kv bits of a mask word can be set, so the sort
 * runs ceil(kv / 9) or ceil(kv / 8) digit passes instead of four (27 offsets: three). */
int spx_mask_argsort_kv(const uint32_t *mask, int n, int kv, int32_t *argsort, void *ws, size_t ws_bytes,
                        spx_stream_t stream);

/* Layout conversions for callers that hold only one of the two rulebook forms
 * (the reference's public ops take either the Native lists, pytorch/ops.py:811-988,
 * or the dense tables, ops.py:1450-1896).
 *  spx_native_to_table: table[k][dst] = src for every list entry (dst/src = out/in,
 *     exchanged when inverse != 0); mask [n_dst, W] optional.
 *  spx_table_to_native: Native lists from pair_fwd (subm != 0, uses the mirror
 *     symmetry) or from pair_bwd [kv, n_in] (subm == 0); order as the CPU path. */
int spx_native_to_table(const int32_t *pair_native, const int32_t *num_per_loc, int n_in,
                        int n_dst, int kv, int subm, int inverse, int32_t *table,
                        uint32_t *mask, spx_stream_t stream);
size_t spx_table_to_native_ws_bytes(int n, int kv);
int spx_table_to_native(const int32_t *table, int subm, int kv, int n, int32_t *pair_native,
                        int32_t *num_per_loc, void *ws, size_t ws_bytes, spx_stream_t stream);

/* -------------------------------------------------------------- convolution */

/* Kernel volumes 33 .. 128 (the reference's multi-word masks, indices.py:1601-1618, ops.py:448,494-503)
 * run as ceil(kv / 32) launches of the MFMA kernel whose partial sums travel through an fp32
 * [n_dst, cout] scratch: pass spx_igemm_acc_bytes() bytes as `ws` to spx_igemm_fwd (n_dst = n_out,
 * cout = K) / spx_igemm_dgrad (n_dst = n_in, cout = C).  0 for kv <= 32; without the scratch such
 * layers take the generic (one thread per output) kernel. */
size_t spx_igemm_acc_bytes(int n_dst, int cout, int kv);

/* Output-stationary implicit GEMM (atomics-free):
 *   out[o,:] = sum_k [pair[k][o] >= 0] feat[pair[k][o],:] * W[:,k,:]^T  (+bias, act)
 * Replaces ConvGemmOps.implicit_gemm forward (csrc/sparse/convops.py:2073-2243,
 * pytorch/ops.py:1450-1664).
 *   feat [n_in, C], weight KRSC [K, kv, C], out [n_out, K]: all `dtype`
 *   pair [kv, n_out]; mask [n_out, W] or NULL; argsort [n_out] or NULL
 *   identity_k: offset whose pair is the identity (SubM centre, kv/2) or -1
 *   bias [K] (dtype) or NULL; act: spx_act (inference epilogue, conv.py:463-490)
 * Every row of `out` is written (no pre-zeroing needed, cf. convops.py:2128-2134). */
int spx_igemm_fwd(const void *feat, const void *weight, void *out, const int32_t *pair,
                  const uint32_t *mask, const int32_t *argsort, int tile_order, int n_in, int n_out,
                  int C, int K, int kv, int dtype, int identity_k, const void *bias,
                  int act, float act_alpha, void *ws, size_t ws_bytes,
                  spx_stream_t stream);

/* The same launch, and the BatchNorm statistics of the rows it stores on the way out: workgroup b of the launch leaves
 * {rows, mean, M2 = sum of squared deviations} of every output channel at stats[field][channel][b], i.e. float
 * (field * K + channel) * records + b with records = *slots_used_h (fp32, field 0 rows / 1 mean / 2 M2; statistics of the
 * ROUNDED output values, i.e. of what a normalisation layer behind the convolution reads), rows >= *n_live (device,
 * static-shape tensors; NULL = every row) not counted.  *slots_used_h (host) = number of records written = the
 * launch's workgroup count, or 0 when the kernel that was dispatched leaves none (bias / activation in the epilogue,
 * kernel volumes > 32, generic kernels): spx_batchnorm_fwd_stats then starts at its merge step instead of reading the
 * rows again for a statistics pass.  `stats` holds stats_slots >= spx_igemm_fwd_stats_slots(n_out) records.
 * The reference leaves BatchNorm to torch on the feature matrix (spconv/pytorch/modules.py:127-168): two extra passes
 * over every activation; this removes the first of them. */
int spx_igemm_fwd_stats_slots(int n_out);
int spx_igemm_fwd_stats(const void *feat, const void *weight, void *out, const int32_t *pair,
                        const uint32_t *mask, const int32_t *argsort, int tile_order, int n_in, int n_out, int C,
                        int K, int kv, int dtype, int identity_k, const void *bias, int act,
                        float act_alpha, void *ws, size_t ws_bytes, float *stats, int stats_slots,
                        const int32_t *n_live, int *slots_used_h, spx_stream_t stream);

/* int8 inference forward.  Replaces the int8 branch of ConvGemmOps.implicit_gemm
 * (pytorch/ops.py:1540-1553,1631-1662, csrc/sparse/convops.py:2176-2205) as driven by the
 * quantised module (pytorch/quantization/quantized/conv.py:368-378).  Numerics pinned by the
 * reference's numpy formula (test/test_all_algo.py:272-287):
 *   v = acc_i32 * scale[k] + bias[k] + add[o][k] * add_scale;  v = act(v)
 *   out_dtype SPX_I8: clip(round_half_even(v), -128, 127);  SPX_F16 / SPX_BF16 / SPX_F32: v
 *   feat int8 [n_in, C], weight int8 KRSC [K, kv, C], scale / bias fp32 [K] (or NULL = 1 / 0),
 *   add int8 [n_out, K] or NULL (the module passes add_scale = add_q_scale / output_scale).
 * C must be a multiple of 16, K one of 16/32/64/128/256, kv <= 32 (the reference's int8 kernels
 * need C, K % 16 == 0 as well, test/test_all_algo.py:376-377). */
int spx_igemm_fwd_int8(const void *feat, const void *weight, void *out, const int32_t *pair,
                       const uint32_t *mask, const int32_t *argsort, int n_in, int n_out, int C,
                       int K, int kv, int identity_k, const float *scale, const float *bias,
                       const void *add, float add_scale, int out_dtype, int act, float act_alpha,
                       spx_stream_t stream);

/* Copies of a pair table [kv, n] and its mask words in TILE ORDER (row t <- row order[t], order =
 * spx_mask_argsort's output): with tile_order = 1, spx_igemm_fwd / _dgrad / _bwd read `pair` and
 * `mask` by tile position and use `argsort` only for the operand / output rows, so a mask-sorted
 * tile reads its table columns as contiguous runs.  (The reference reads its tables through
 * mask_argsort, ops.py:1503-1530.) */
int spx_permute_tables(const int32_t *pair, const uint32_t *mask, const int32_t *order, int kv, int n,
                       int words, int32_t *pair_t, uint32_t *mask_t, spx_stream_t stream);

/* ---- density-aware row layout: the DEFAULT row order of a SubM rulebook -----------------------
 * The reference sorts the rows of every rulebook by mask (SPCONV_DO_SORT = "1", constants.py:121;
 * pytorch/ops.py:346,550,763-785 -> SpconvOps.sort_1d_by_key_allocator, all.py:935-991) so that its
 * implicit-GEMM tiles skip the offsets none of their rows has.  spx_subm_layout does that job INSIDE
 * the rulebook build -- no sort, nothing read back: the masks of the finished tables are classified
 * on the device and, for a SPARSE rulebook (fewer than a quarter of the rows have any neighbour), the
 * rows WITH a neighbour are taken out of the row-order walk into a compact appendix, grouped by their
 * lowest neighbour offset (a stable counting partition: wave ballots + prefix sums).  The gather-GEMM
 * then runs the rows in their own order with the CENTRE pair only -- a plain streaming GEMM: no row
 * order to fetch, no pair word, one step per tile -- and a handful of appendix tiles with all their
 * offsets.  A DENSE rulebook (LiDAR) keeps everything in the row-order walk (regrouping loses there).
 * The result is ONE int32 blob (pass it as `argsort` with tile_order = SPX_ROWS_LAYOUT;
 * spx_igemm_fwd_int8: OR SPX_ROWS_LAYOUT_ACT into `act`; `pair` / `mask` stay the row-order tables):
 *   [0] class: 1 = appendix in use, 0 = none   [1] M = rows with a neighbour   [2] n   [3] kv   [4] mcap
 *   main mask  [npad]       row-order mask words; class 1: ZERO for the rows that moved to the appendix
 *                           (a zero mask word means: not this tile's row, nothing stored)
 *   order      [mcap]       appendix position -> row       (first M entries; class 1 only)
 *   mask       [mcap]       mask words of those rows
 *   pair       [kv, mcap]   their pair words
 *   npad = n rounded up to 64, mcap = spx_subm_layout_mcap(n) >= n / 4 + 256.
 * The launch is the same for both classes (hipGraph-safe): ceil(n / 4 / tile rows) appendix workgroups
 * lead the grid, read {class, M} and leave at once when there is nothing for them; the M rows are dealt to
 * them in whole 16-row blocks (ceil(M / workgroups) rows each, rounded up to 16), so that a short appendix
 * becomes many short tiles instead of a few long ones. */
size_t spx_subm_layout_mcap(int n);
#define SPX_ROWS_LAYOUT 2
/* OR-ed into `tile_order` of spx_igemm_fwd / spx_igemm_dgrad: the HOST knows that the rulebook's neighbourhoods are
 * dense (it has seen class word 0 of the rows layout, or knows its data).  C = K = 64 16-bit layers then take the
 * weight-stationary gather-GEMM (csrc/igemm_ws.hip: one 512-row workgroup per CU, weights staged nine offsets at a time
 * through LDS-DMA) instead of the 128-row tiles -- a launch SHAPE, which is why it cannot be read on the device.  Results
 * are bit-identical with and without the hint; a wrong hint only costs time.  Stands in for the reference's tuner choice
 * between implicit-GEMM tile shapes (spconv/csrc/sparse/convops.py:1150-1297). */
#define SPX_DENSE_HINT 0x100
#define SPX_ROWS_LAYOUT_ACT 0x400
/* OR-ed into `act` of spx_igemm_fwd_int8 next to SPX_ROWS_LAYOUT_ACT: the HOST knows the class word is 1 (it may
 * read it once per rulebook, outside any timed or captured region) -- a launch-shape hint only: 64-row instead
 * of 128-row tiles at 128 output channels (the role of the reference's per-problem tuner cache,
 * csrc/sparse/convops.py:1150,1283-1297).  Bits 16..31 of `act` may then carry ceil(M / 64), M = the blob's
 * word [1]: only that many appendix rows get workgroups (instead of the n / 4 the class rule allows).
 * Results never depend on either. */
#define SPX_SPARSE_HINT 0x800
#define SPX_LAYOUT_HEADER 64
size_t spx_subm_layout_bytes(int n, int kv);
size_t spx_subm_layout_ws_bytes(int n);
int spx_subm_layout(const int32_t *pair_fwd, const uint32_t *mask, int n, int kv, int32_t *layout,
                    void *ws, size_t ws_bytes, spx_stream_t stream);

/* Scratch for dgrad (always 0: the weight transpose happens inside the kernel). */
size_t spx_igemm_dgrad_ws_bytes(int C, int K, int kv, int dtype);

/* Input gradient.  Replaces the dgrad half of ConvGemmOps.implicit_gemm_backward
 * (convops.py:2245-2440, ops.py:1667-1896):
 *   din[i,:] = sum_k [pair_bwd[k][i] >= 0] dout[pair_bwd[k][i],:] * W[:,k,:]
 * For SubM pass subm=1 with the FORWARD pair/mask (mirror symmetry
 * pair_bwd[k] == pair_fwd[kv-1-k]; the reference's reverse_mask, convops.py:2327-2345). */
int spx_igemm_dgrad(const void *dout, const void *weight, void *din, const int32_t *pair,
                    const uint32_t *mask, const int32_t *argsort, int tile_order, int n_out, int n_in,
                    int C, int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                    spx_stream_t stream);

/* Scratch for wgrad (per-workgroup fp32 partials + room for a work plan). */
size_t spx_igemm_wgrad_ws_bytes(int n_in, int C, int K, int kv);

/* Work plan of wgrad: the list of (offset, chunk-of-pairs) items that exist for a
 * rulebook, so the wgrad grid holds no empty workgroups.  It depends only on
 * num_per_loc, so callers build it once per rulebook and pass it to every
 * spx_igemm_wgrad call (plan == NULL makes wgrad rebuild it in `ws`). */
size_t spx_wgrad_plan_bytes(int n_in, int kv);
int spx_wgrad_plan(const int32_t *num_per_loc, int n_in, int kv, int subm, int32_t *plan,
                   spx_stream_t stream);

/* Weight gradient.  Replaces the wgrad half of implicit_gemm_backward and of
 * ConvGemmOps.indice_conv_backward (convops.py:1749-1860):
 *   dW[:,k,:] = sum_j dout[pair_native[1][k][j],:]^T (x) feat[pair_native[0][k][j],:]
 * over the Native lists; deterministic two-stage reduction (no atomics).
 *   dw KRSC [K, kv, C] in `dtype`, fully overwritten.
 *   subm=1: centre offset is the identity over all rows and offsets k > kv/2 use
 *   num_per_loc[kv-1-k] (ops.py:962-968). */
int spx_igemm_wgrad(const void *feat, const void *dout, void *dw, const int32_t *pair_native,
                    const int32_t *num_per_loc, const int32_t *plan, int n_in, int n_out, int C,
                    int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                    spx_stream_t stream);

/* Backward of a NARROW layer (C, K in {16, 32}, 16-bit features, kernel volume <= 27) from one gather per pair
 * (csrc/igemm_bwdn.hip): a walk over 128-row tiles of the input rows feeds every gathered dout tile to the
 * input gradient AND to the weight gradient.  Same role as spx_igemm_bwd (ConvGemmOps.implicit_gemm_backward,
 * pytorch/ops.py:1667-1896), without the Native lists and the range plan:
 *   table [kv, n_in], mask [n_in]: the dgrad table (SubM: pair_fwd; regular conv: pair_bwd) and its mask words
 *   weight_t [kv, C, K]: the weights with the dout channel contiguous, weight_t[k][c][k'] = weight[k'][k][c];
 *                        mirror = 1 (SubM): table row r pairs with slice kv-1-r
 *   din [n_in, C] or NULL;  dw KRSC [K, kv, C] in `dtype`, fully overwritten (deterministic)
 *   ws: spx_igemm_bwd_rows_ws_bytes (per-workgroup fp32 partial weight gradients) */
size_t spx_igemm_bwd_rows_ws_bytes(int n_in, int C, int K, int kv);
int spx_igemm_bwd_rows(const void *feat, const void *dout, const void *weight_t, void *din, void *dw,
                       const int32_t *table, const uint32_t *mask, int n_in, int n_out, int C, int K,
                       int kv, int mirror, int dtype, void *ws, size_t ws_bytes, spx_stream_t stream);

/* Backward of one layer: din (as spx_igemm_dgrad) and dw (as spx_igemm_wgrad) from ONE kernel
 * launch plus the wgrad second stage.  Replaces ConvGemmOps.implicit_gemm_backward /
 * indice_conv_backward as a whole (pytorch/ops.py:1667-1896,1103-1447), which also return both
 * gradients from one call.  pair / mask / argsort are the dgrad table (the FORWARD table for
 * SubM), pair_native / num_per_loc / plan as for spx_igemm_wgrad, ws as spx_igemm_wgrad_ws_bytes.
 * Shapes the fused kernel does not cover fall back to the two separate calls internally. */
int spx_igemm_bwd(const void *feat, const void *dout, const void *weight, void *din, void *dw,
                  const int32_t *pair, const uint32_t *mask, const int32_t *argsort, int tile_order,
                  const int32_t *pair_native, const int32_t *num_per_loc, const int32_t *plan,
                  int n_in, int n_out, int C, int K, int kv, int dtype, int subm, void *ws,
                  size_t ws_bytes, spx_stream_t stream);

/* The weight gradient's SECOND STAGE deferred.  spx_igemm_bwd / spx_igemm_wgrad end with a small launch that reduces
 * the per-range partial tiles (fp32, in `ws`) into dw; nothing in a backward pass waits for a layer's dw, and a
 * dependent 4-9 us launch per layer is what a captured training step of a backbone is made of.  The *_deferred forms
 * run everything BUT that launch and write its description into `stage2_job` (SPX_STAGE2_JOB_BYTES bytes of host
 * memory, opaque); spx_wgrad_stage2_batch then reduces the partial tiles of up to 16 layers per launch (jobs:
 * `njobs` consecutive records, any mix of dtypes).  Between the two calls the caller keeps `ws`, the plan and dw
 * alive and must not read dw.  A call whose shapes take a path without a second stage (empty scene, odd channel counts)
 * completes dw at once and leaves an empty record, which the batch call skips.  Results are bit-identical to the
 * undeferred calls (same kernel body, same summation order).  The reference returns din and dw from one blocking
 * call (pytorch/ops.py:1667-1896); its split-K reduction is part of that call. */
#define SPX_STAGE2_JOB_BYTES 64
int spx_igemm_bwd_deferred(const void *feat, const void *dout, const void *weight, void *din, void *dw,
                           const int32_t *pair, const uint32_t *mask, const int32_t *argsort, int tile_order,
                           const int32_t *pair_native, const int32_t *num_per_loc, const int32_t *plan,
                           int n_in, int n_out, int C, int K, int kv, int dtype, int subm, void *ws,
                           size_t ws_bytes, spx_stream_t stream, void *stage2_job);
int spx_igemm_wgrad_deferred(const void *feat, const void *dout, void *dw, const int32_t *pair_native,
                             const int32_t *num_per_loc, const int32_t *plan, int n_in, int n_out, int C,
                             int K, int kv, int dtype, int subm, void *ws, size_t ws_bytes,
                             spx_stream_t stream, void *stage2_job);
int spx_wgrad_stage2_batch(const void *jobs, int njobs, spx_stream_t stream);
/* Points a pending record at another destination of the same shape and dtype (an autograd engine may have moved the
 * gradient it was handed into a tensor of its own); returns 1, or 0 for an empty record. */
int spx_stage2_job_retarget(void *stage2_job, void *dw);

/* In-place epilogues for callers that keep bias/activation separate
 * (InferenceOps.bias_add_act_inplace etc., csrc/sparse/inference.py:26-146). */
int spx_bias_act_inplace(void *out, const void *bias, int n, int K, int dtype, int act,
                         float act_alpha, spx_stream_t stream);

/* ------------------------------------------------------------------ pooling */

/* Sparse max / average pooling over the same rulebook tables.
 * Replace IndiceMaxPool::forward_implicit_gemm / backward_implicit_gemm /
 * forward_avgpool_implicit_gemm / backward_avgpool_implicit_gemm
 * (csrc/sparse/maxpool.py:96-300,397-588; pytorch/ops.py:1899-2084).
 *   pair_fwd [kv, n_out] / pair_bwd [kv, n_in] with -1 = absent; mask (one uint32 word per 32
 *   offsets and row) is optional and only used to skip absent offsets.
 *   max forward: out[o] = max over the valid pairs; init_zero != 0 starts from 0 instead of the
 *   lowest value (the reference's ConvAlgo.Native pooling does, pytorch/ops.py:1910).
 *   max backward: din[i] = sum of dout[o] over the outputs o with out[o] == feat[i].
 *   avg forward: mean over the valid pairs, count_out [n_out] (or NULL) receives their number.
 *   avg backward: din[i] = sum_o dout[o] / count[o].
 * dtypes: f32 / f16 / bf16, plus int8 for the max forward. */
int spx_maxpool_fwd(const void *feat, void *out, const int32_t *pair_fwd, const uint32_t *mask,
                    int n_out, int C, int kv, int dtype, int init_zero, spx_stream_t stream);
int spx_maxpool_bwd(const void *feat, const void *out, const void *dout, void *din,
                    const int32_t *pair_bwd, const uint32_t *mask_bwd, int n_in, int C, int kv,
                    int dtype, spx_stream_t stream);
int spx_avgpool_fwd(const void *feat, void *out, int32_t *count_out, const int32_t *pair_fwd,
                    const uint32_t *mask, int n_out, int C, int kv, int dtype,
                    spx_stream_t stream);
int spx_avgpool_bwd(const void *dout, void *din, const int32_t *count, const int32_t *pair_bwd,
                    const uint32_t *mask_bwd, int n_in, int C, int kv, int dtype,
                    spx_stream_t stream);

/* ------------------------------------------------------------ point -> voxel */

/* Voxeliser.  Replaces SpconvOps.point2voxel_cuda / point2voxel_cpu
 * (csrc/sparse/all.py:1389-1500, csrc/sparse/pointops.py; driver pytorch/utils.py:23-160).
 * Result identical to the reference's CPU loop: voxels numbered in first-seen point order, the
 * first max_points points of a voxel kept in point order, voxels beyond max_voxels dropped.
 *   points [n, nfeat] fp32, the first ndim columns are x, y, z(, t)
 *   vsize [ndim], coors_range [2 ndim] (lows, then highs), grid_size [ndim]: all in ZYX order as
 *   returned by calc_point2voxel_meta_data (all.py:1349-1386)
 *   voxels [max_voxels, max_points, nfeat] fp32, indices [max_voxels, ndim] (zyx),
 *   num_per_voxel [max_voxels], pc_voxel_id [n] int64 (-1 = dropped point)
 *   empty_mean: 1 = unused slots of a voxel receive the mean of its points; 2 = the fill as the reference's CPU
 *   loop BEHAVES (pointops.py:663-686: its accumulator is carried from voxel to voxel), bit-identical to that
 *   code executed, a sequential pass (SPCONV_AMD_REFERENCE_QUIRKS=1 selects it in the Python layer);
 *   clear_voxels: zero `voxels` first.  *n_voxels_h receives the number of voxels (one D->H read). */
size_t spx_point2voxel_ws_bytes(int n_points, int max_voxels);
int spx_point2voxel(const float *points, int n, int nfeat, int ndim, const float *vsize,
                    const float *coors_range, const int *grid_size, int max_voxels, int max_points,
                    int empty_mean, int clear_voxels, float *voxels, int32_t *indices,
                    int32_t *num_per_voxel, long long *pc_voxel_id, int *n_voxels_h, void *ws,
                    size_t ws_bytes, spx_stream_t stream);

/* ---------------------------------------------------------------- hash table */

/* Fixed-size hash table over caller-owned storage.  Replaces
 * spconv/csrc/hash/core.py HashTable as driven by spconv/pytorch/hash.py:29-170.
 *   table_keys [capacity] (4- or 8-byte integers, all-ones = empty; initialise with
 *   spx_hash_clear), table_vals [capacity] (4- or 8-byte items, copied bit for bit)
 *   insert: values may be NULL (key only); query: is_empty[i] = 1 where the key is absent;
 *   insert_exist: writes values only where the key is already present;
 *   assign_arange: every present key gets its rank in SLOT order as value, *count_out (an integer
 *   of the key width) receives the number of keys; items: entries in the same order.
 * ws >= spx_hash_ws_bytes(capacity) for assign_arange / items. */
size_t spx_hash_ws_bytes(int capacity);
int spx_hash_clear(void *table_keys, int capacity, int key_bytes, spx_stream_t stream);
int spx_hash_insert(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                    const void *keys, const void *values, int n, spx_stream_t stream);
int spx_hash_query(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                   const void *keys, void *values_out, unsigned char *is_empty, int n,
                   spx_stream_t stream);
int spx_hash_insert_exist(void *table_keys, void *table_vals, int capacity, int key_bytes,
                          int val_bytes, const void *keys, const void *values,
                          unsigned char *is_empty, int n, spx_stream_t stream);
int spx_hash_assign_arange(void *table_keys, void *table_vals, int capacity, int key_bytes,
                           int val_bytes, void *count_out, void *ws, size_t ws_bytes,
                           spx_stream_t stream);
int spx_hash_items(void *table_keys, void *table_vals, int capacity, int key_bytes, int val_bytes,
                   void *keys_out, void *vals_out, int max_out, void *count_out, void *ws,
                   size_t ws_bytes, spx_stream_t stream);

/* ---- batch normalisation (+ ReLU) over the features of a sparse tensor ---------------------------
 * The reference hands `.features` of a SparseConvTensor to torch.nn.BatchNorm1d (SparseSequential,
 * spconv/pytorch/modules.py:131-145; SparseBatchNorm / SparseReLU, :147-185).  These two calls do
 * the same arithmetic (torch.nn.BatchNorm1d semantics: biased variance to normalise, unbiased for
 * the running estimate, `momentum`, `eps`, optional affine) as three streaming launches per pass.
 *   x, y, dy, dx      [n, C] row-major, dtype f16 / bf16 / f32, C a multiple of 8 (f32: 4), C <= 256
 *   weight, bias      [C] or NULL;  running_mean / running_var [C] or NULL (updated in place when
 *                     training, read when not)
 *   save_mean / save_invstd [C] fp32: batch statistics for the backward pass (training)
 *   relu              fuse max(0, .) into the output (and its mask into the backward pass)
 *   ws                spx_batchnorm_ws_bytes(n, C) bytes
 *   n_live            NULL, or a DEVICE int32: only the first *n_live rows are rows of the scene (static-shape
 *                     tensors, see spx_conv_rulebook_static): statistics over those rows, the others come out
 *                     as zeros (y and dx) */
size_t spx_batchnorm_ws_bytes(int n, int C);
/* weight / bias / running_mean / running_var: [C] vectors of dtype `param_dtype` (SPX_F32, or the
 * 16-bit dtype of a model converted with .half() / .bfloat16()); any of them may be NULL.
 * num_batches_tracked: the module's int64 step counter (torch/nn/modules/batchnorm.py:160-175), incremented by
 * the statistics kernel in training mode, or NULL. */
int spx_batchnorm_fwd(const void *x, void *y, int n, int C, int dtype, const void *weight,
                      const void *bias, void *running_mean, void *running_var,
                      long long *num_batches_tracked, int param_dtype, int training, float momentum,
                      float eps, int relu, float *save_mean, float *save_invstd, void *ws,
                      size_t ws_bytes, const int32_t *n_live, spx_stream_t stream);
/* Training-mode forward whose statistics pass has already happened: `stats` = the {rows, mean, M2} records
 * (stats_records of them, laid out [3][C][stats_records] fp32) that spx_igemm_fwd_stats left behind the convolution producing x.  Two
 * launches (merge, apply) instead of three; semantics as spx_batchnorm_fwd with training = 1. */
int spx_batchnorm_fwd_stats(const void *x, void *y, int n, int C, int dtype, const void *weight,
                            const void *bias, void *running_mean, void *running_var,
                            long long *num_batches_tracked, int param_dtype, float momentum, float eps,
                            int relu, float *save_mean, float *save_invstd, const float *stats,
                            int stats_records, const int32_t *n_live, spx_stream_t stream);
/* use_batch_stats = 1: `mean` / `invstd` are the saved fp32 batch statistics (training);
 * 0: fp32 copies of running_mean and 1 / sqrt(running_var + eps) (evaluation mode with gradients).
 * dweight / dbias: [C] of `param_dtype`, or NULL. */
int spx_batchnorm_bwd(const void *x, const void *dy, void *dx, int n, int C, int dtype,
                      const void *weight, const void *bias, int param_dtype, const float *mean,
                      const float *invstd, int use_batch_stats, int relu, void *dweight, void *dbias,
                      void *ws, size_t ws_bytes, const int32_t *n_live, spx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPCONV_AMD_H_ */
