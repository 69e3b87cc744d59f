/*
 * wisp_hip.h - C ABI of the MI355X (gfx950) neural-field hot path.
 *
 * Every entry point takes raw DEVICE pointers (unless a parameter says "host"), element counts, and the
 * HIP stream to launch on; it allocates nothing that outlives the call, never synchronises the device
 * unless stated, and returns 0 on success or a negative code (WISP_ERR_*).  wisp_last_error() returns a
 * static string describing the last failure on the calling thread.  The caller owns all buffers
 * (the Python host layer allocates them from torch so allocator and stream semantics stay PyTorch's).
 *
 * Each declaration cites the reference interface it replaces.  Reference paths are relative to
 * NVIDIAGameWorks/kaolin-wisp; "kaolin" = NVIDIA Kaolin Core 0.13 (INSTALL.md:14), whose ops wisp calls.
 */
#ifndef WISP_HIP_H
#define WISP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* wisp_stream_t; /* hipStream_t */

enum { WISP_F32 = 0, WISP_F16 = 1, WISP_BF16 = 2 };

enum {
    WISP_OK = 0,
    WISP_ERR_INVALID = -1,   /* bad argument (null pointer, unsupported dtype / dim / level) */
    WISP_ERR_LAUNCH = -2,    /* hipGetLastError() after a launch; see wisp_last_error() */
    WISP_ERR_UNSUPPORTED = -3,
    WISP_ERR_CAPACITY = -4   /* wisp_nerf_step_*: the batch does not fit the buffers the step was created for (or is empty): nothing
                                was changed; the caller takes this batch through the per-op entry points */
};

const char* wisp_last_error(void);
/* ABI version of this library; bumped whenever a signature changes (1 = round 1; 2 = round 2: scratch arguments of the backward
 * passes, raytrace nugget cache, optimizer kinds, per-ray view codes, corner query, decoded codebook rows; 3 = round 3: per-level
 * slot scales of the hash-grid backward; 4 = round 4: workspace + row counts of the order-free trilinear / codebook backward.
 * Entry points that are only ADDED - wisp_spc_query_chain, wisp_composite_loss, wisp_codebook_trilinear_multi_bwd,
 * wisp_sdf_train_step, wisp_hashgrid_grad_coords, wisp_host_reader_*, wisp_nerf_step_* - do not bump it). */
int wisp_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Hash / dense multi-resolution grid  (replaces wisp._C.ops.hashgrid_interpolate_cuda and
 * ..._backward_cuda: wisp/csrc/ops/hashgrid_interpolate.h:18-33, .cpp:46-105, kernels
 * wisp/csrc/ops/hashgrid_interpolate_cuda.cu:19-339, index math wisp/csrc/ops/hash_utils.cuh:17-112)
 *
 *  coords        f32 [n, coord_dim]   coord_dim in {2,3}, values in [-1,1]
 *  codebook      dtype [sum_T, feature_dim]   all levels stacked (wisp/models/grids/utils.py:48-63)
 *  first_idx     i64 [num_lods+1]     DEVICE; first row of each level in codebook
 *  resolutions   i32 [num_lods]       HOST (the reference passes a CPU tensor, .cu:365)
 *  feats         dtype [n, num_lods*feature_dim]; columns >= zero_from_col are written as 0
 *                (HashGrid.interpolate 'cat' quirk, wisp/models/grids/hash_grid.py:226-229); pass
 *                num_lods*feature_dim to disable.
 * All levels are processed by ONE launch (the reference launches once per level, .cpp:61-64).
 */
int wisp_hashgrid_interpolate_fwd(const float* coords, int64_t n, int coord_dim,
                                  const void* codebook, int dtype, int feature_dim,
                                  const int64_t* first_idx, const int32_t* resolutions, int num_lods,
                                  int codebook_bitwidth, int zero_from_col,
                                  void* feats, wisp_stream_t stream);

/*  grad_feats    dtype [n, num_lods*feature_dim]
 *  grad_codebook f32 [sum_T, feature_dim]; ACCUMULATED into (caller zeroes it).  Always float32:
 *                the reference adds in the table dtype with __half2 atomics (.cu:138-150); fp32
 *                accumulation is a strict numerical improvement and is cast by the host if needed.
 *  Columns >= zero_from_col receive no gradient.  grad w.r.t. coords: wisp_hashgrid_grad_coords below.
 */
int wisp_hashgrid_interpolate_bwd(const float* coords, int64_t n, int coord_dim,
                                  const void* grad_feats, int dtype, int feature_dim,
                                  const int64_t* first_idx, const int32_t* resolutions, int num_lods,
                                  int codebook_bitwidth, int zero_from_col,
                                  float* grad_codebook, void* workspace, int64_t workspace_bytes,
                                  const float* level_cap_scale /* host [num_lods] or NULL */, wisp_stream_t stream);
/* Optional device scratch for the backward: with at least this many bytes the levels are reduced through binned
 * (index, value) records + LDS accumulation instead of memory-side atomics (0 = binning not applicable).
 * level_cap_scale (host, one float in (0, 1] per level, NULL = all 1): shrinks the record slots of a level from the no-merge
 * expectation they are sized for to what the caller has measured (wisp_hashgrid_bwd_slot_stats): consecutive samples of a ray
 * that share a cell are merged before they are written, 2x on the finest levels and 20x on the coarsest, so the default
 * slots are mostly air (2.5 GB of scratch for 0.43 GB of records at 2 M samples of nerf_hash.yaml).  The same scales must
 * be passed to the size query and to the launch.  A slot that turns out too small is not an error: its excess records go
 * through atomics. */
int64_t wisp_hashgrid_bwd_workspace_bytes(int64_t n, int coord_dim, int dtype /* of grad_feats; -1 = enough for any */,
                                          int feature_dim, const int32_t* resolutions /* host */, int num_lods,
                                          int codebook_bitwidth, const float* level_cap_scale /* host or NULL */);
/* After a wisp_hashgrid_interpolate_bwd launch with the same arguments, on the same stream, before the workspace is
 * reused: the fill of the fullest record slot of every level and the number of records the level wrote -> max_fill (DEVICE
 * u32 [2 * num_lods]: maxima, then totals; written by a small kernel), plus the slot capacity the launch used and the
 * unscaled capacity (HOST i32 [num_lods] each, written at once).
 * max_fill[l] == cap[l] means slots of level l overflowed into the atomic path.  Returns 0, or 1 when that launch was not
 * binned (nothing is written then), or a negative error code. */
int wisp_hashgrid_bwd_slot_stats(int64_t n, int coord_dim, int dtype, int feature_dim, const int32_t* resolutions /* host */,
                                 int num_lods, int codebook_bitwidth, int zero_from_col, const float* level_cap_scale,
                                 const void* workspace, int64_t workspace_bytes, uint32_t* max_fill, int32_t* cap_host,
                                 int32_t* base_cap_host, wisp_stream_t stream);

/* Gradient w.r.t. the coordinates, as hashgrid_interpolate_backward_cuda(..., require_grad_coords = true) returns it
 * (hashgrid_interpolate.cpp:69-100; kernel body hashgrid_interpolate_cuda.cu:163-196; requested by wisp/ops/grid.py:109-126 when
 * `coords` needs a gradient).  grad_coords f32 [n, 3], OVERWRITTEN.  The reference's arithmetic is reproduced term by term,
 * including what its source marks unfinished: every level reads the FIRST level's upstream gradient columns (.cu:165-166), the
 * y derivative's last term subtracts corner 6 (.cu:185-186), no res / 2 chain-rule factor, and 2-D coordinates get zeros (the
 * 2-D kernel ignores the flag).  grad_feats and codebook share `dtype`; all num_lods levels are visited (no zero_from_col: the
 * reference's backward knows nothing of the 'cat' zeroing, the zero columns' upstream gradient is zero by autograd). */
int wisp_hashgrid_grad_coords(const float* coords, int64_t n, int coord_dim, const void* grad_feats, const void* codebook,
                              int dtype, int feature_dim, const int64_t* first_idx, const int32_t* resolutions /* host */,
                              int num_lods, int codebook_bitwidth, float* grad_coords, wisp_stream_t stream);

/* wisp_hashgrid_interpolate_bwd with the table's AdamW step folded into it: where the reference runs
 * hashgrid_interpolate_backward_cuda (.cu:109-196) and then the optimizer over the whole table (base_trainer.py:205-246,
 * multiview_trainer.py:169-174), the reduce kernel's workgroup that OWNS a slice of the table has the slice's complete gradient
 * in LDS and updates param / exp_avg / exp_avg_sq (and the bf16 copy, may be NULL) right there - the gradient of those rows is
 * never written, read back or zeroed.  param, exp_avg, exp_avg_sq, bf16_shadow are indexed like grad_codebook; lr ... grad_scale
 * as in wisp_adamw_step (same arithmetic, bit for bit).  covered_rows (HOST i64 [num_lods], written at once): per level, how many
 * of its leading rows were updated that way; every other row still has its gradient ACCUMULATED in grad_codebook and is the
 * caller's to update (wisp_adamw_step_groups over the remaining ranges).  Levels that were not binned or whose buckets are
 * shared by several workgroups (the coarse ones) report 0; with feature_dim != 2 everything reports 0 (plain backward).
 * The trailing levels behind zero_from_col (no gradient reaches them; nerf_hash.yaml: the finest) are stepped by extra workgroups
 * of the same launch - every row the caller's first_idx gives those levels, whatever their spacing - and report 2^62 ("the whole
 * level": clamp to first_idx[l + 1] - first_idx[l]); WISP_ADAM_TAIL=0: they report 0 and stay the caller's.
 * Single-GPU only by construction: a data-parallel step has to exchange the gradient first. */
int wisp_hashgrid_interpolate_bwd_adamw(const float* coords, int64_t n, int coord_dim,
                                        const void* grad_feats, int dtype, int feature_dim,
                                        const int64_t* first_idx, const int32_t* resolutions, int num_lods,
                                        int codebook_bitwidth, int zero_from_col,
                                        float* grad_codebook, void* workspace, int64_t workspace_bytes,
                                        const float* level_cap_scale, float* param, float* exp_avg, float* exp_avg_sq,
                                        void* bf16_shadow, float lr, float beta1, float beta2, float eps, float weight_decay,
                                        int64_t step, float grad_scale, int64_t* covered_rows /* host */, wisp_stream_t stream);

/* Diagnostic (no counterpart in the reference's bindings): the integer cell, the position inside it and the 2^d corner rows
 * (relative to the level's first row) that ONE level assigns to every coordinate - the first lines of every reference
 * hash-grid kernel (wisp/csrc/ops/hashgrid_interpolate_cuda.cu:40-66, hash_utils.cuh:17-105), evaluated by the device code all
 * kernels of this library share.  cell i32 [n, coord_dim], frac f32 [n, coord_dim], corner_idx i32 [n, 2^coord_dim] or NULL;
 * corner j = bit (coord_dim-1-a) of j selects the upper neighbour on axis a.  Parity tests compare it with the reference's
 * kernel bodies on adversarial coordinates (cell faces +- ulps, |c| < 2^-18, +-1, denormals). */
int wisp_hashgrid_cells(const float* coords, int64_t n, int coord_dim, int32_t resolution, int codebook_bitwidth,
                        int32_t* cell, float* frac, int32_t* corner_idx, wisp_stream_t stream);

/* Corner query without the blend: wisp._C.ops.hashgrid_query_cuda / hashgrid_query_backward_cuda
 * (wisp/csrc/ops/hashgrid_query_cuda.cu:19-186, hashgrid_query.cpp:41-97, bound in bindings.cpp:31-32; Python callers
 * wisp/ops/grid.py:169-245 - nothing else in the reference uses them).  One codebook [2^codebook_bitwidth, feature_dim] per
 * level (device pointers in a HOST array), 3-D coordinates; feats / grad_feats = [N, 8, num_lods, P, feature_dim] with
 * P = 2^probe_bitwidth in the tables' dtype: corner k = dx<<2 | dy<<1 | dz, index = hash_index_3d(corner, res, 2^bw - P), the
 * row repeated for every probe slot (as the reference's forward does).  Backward adds grad_feats into the gradient tables
 * (caller zeroes them): fp32 puts probe p into row idx + p, the 16-bit dtypes all probes into row idx with packed atomics -
 * the two behaviours of the reference's kernel. */
int wisp_hashgrid_query_fwd(const float* coords, int64_t n, const void* const* codebooks, int dtype, int feature_dim,
                            const int32_t* resolutions, int num_lods, int codebook_bitwidth, int probe_bitwidth,
                            void* feats, wisp_stream_t stream);
int wisp_hashgrid_query_bwd(const float* coords, int64_t n, const void* grad_feats, int dtype, int feature_dim,
                            const int32_t* resolutions, int num_lods, int codebook_bitwidth, int probe_bitwidth,
                            void* const* grad_codebooks, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * SPC octree queries  (replace kaolin.ops.spc.unbatched_query at wisp/accelstructs/octree_as.py:162,
 * kaolin.render.spc.unbatched_raytrace at :183-185, mark_pack_boundaries at :300 and
 * kaolin._C.render.spc.inclusive_sum_cuda at :351)
 *
 *  octree  u8 [n_nodes]   occupancy byte per non-leaf node, BFS / Morton order
 *  exsum   i32 [n_nodes+1] exclusive prefix sum of popcount(octree)
 *  points  i16 [n_points,3] point hierarchy
 */
int wisp_spc_query(const uint8_t* octree, const int32_t* exsum, const float* coords, int64_t n,
                   int level, int with_parents, int64_t* pidx /* [n] or [n, level+1] */,
                   wisp_stream_t stream);

/* The columns first_level .. level of wisp_spc_query(with_parents = 1) only: chain i64 [n, level - first_level + 1]
 * (what OctreeGrid.interpolate slices out of the query, wisp/models/grids/octree_grid.py:196-199).  hint_pidx (optional,
 * i32 [ceil(n / hint_group)]): for every group of hint_group consecutive coordinates the cell of first_level a caller
 * already knows, or -1 - e.g. the nuggets of the raytrace a 'voxel' march sampled (octree_as.py:213-245: hint_group samples
 * per nugget).  A hint is used only where the coordinate really quantises into that cell; everywhere else the walk
 * starts at the root, so the result does not depend on the hints.  `points` is only read with hints. */
int wisp_spc_query_chain(const uint8_t* octree, const int32_t* exsum, const int16_t* points, const float* coords,
                         int64_t n, int level, int first_level, const int32_t* hint_pidx, int hint_group,
                         int64_t* chain, wisp_stream_t stream);

/* Occupancy bitfield of one level: bit  x | y << level | z << 2 level  of bits[] is set iff the level-`level` cell
 * (x, y, z) exists.  bits: u32 [ceil(8^level / 32)], zeroed by the call. */
int wisp_spc_build_bitfield(const int16_t* level_points, int64_t n_points, int level, uint32_t* bits,
                            wisp_stream_t stream);

/* SPC build on the device (replaces Kaolin-Core unbatched_points_to_octree + scan_octrees + generate_points as called by
 * wisp/ops/spc/conversions.py:29-40,72-88 and, once per prune, wisp/models/nefs/nerf.py:205-206).  The finest level is a
 * dense mask in Morton order (code bit 3i = z, 3i+1 = y, 3i+2 = x); `dense` is u8 [(8^(level+1)-1)/7] = the levels
 * 0..level concatenated.  mask_from_points marks cells (points i16 [n,3]; out-of-range points are ignored) in a zeroed
 * leaf mask; dense_bytes fills the node bytes of levels level-1..0 in front of the leaf mask the caller put in the last
 * 8^level entries; the non-zero entries of `dense`, in order, are the point hierarchy (compact them with
 * wisp_boundary_tile_counts / wisp_boundary_pack_starts); points_from_index decodes those positions to coordinates. */
int wisp_spc_mask_from_points(const int16_t* points, int64_t n, int level, uint8_t* leaf_mask, wisp_stream_t stream);
int wisp_spc_dense_bytes(uint8_t* dense, int level, wisp_stream_t stream);
int wisp_spc_points_from_index(const int64_t* index, int64_t n, int level, int16_t* points, wisp_stream_t stream);

/* Two-phase ray / octree intersection.  count: nuggets per ray; emit: writes them at offsets[r]
 * (exclusive scan of counts), ordered by ray then front-to-back.  depth is [M,1] or [M,2].
 * cache (optional, f32 [num_rays, cache_cap, 3], caller-allocated, contents undefined on entry): the count phase parks the
 * first cache_cap nuggets of every ray there (pidx bits, entry, exit) and the emit phase copies them instead of
 * traversing again; rays with more nuggets than cache_cap are re-traversed.  Pass NULL / 0 to both phases to disable. */
int wisp_spc_raytrace_count(const uint8_t* octree, const int16_t* points, const int32_t* exsum,
                            const float* origins, const float* dirs, int64_t num_rays, int level,
                            int32_t* counts, float* cache, int cache_cap, wisp_stream_t stream);
int wisp_spc_raytrace_emit(const uint8_t* octree, const int16_t* points, const int32_t* exsum,
                           const float* origins, const float* dirs, int64_t num_rays, int level,
                           const int64_t* offsets, int with_exit, const float* cache, int cache_cap,
                           int32_t* ridx, int32_t* pidx, float* depth, wisp_stream_t stream);

int wisp_mark_pack_boundaries_i64(const int64_t* ids, int64_t n, uint8_t* boundary, wisp_stream_t stream);
int wisp_mark_pack_boundaries_i32(const int32_t* ids, int64_t n, uint8_t* boundary, wisp_stream_t stream);

/* Exclusive scan of int32 counts into int64 offsets [n+1] (offsets[n] = total).
 * workspace: at least wisp_scan_workspace_bytes(n) bytes of device scratch. */
int64_t wisp_scan_workspace_bytes(int64_t n);
int wisp_exclusive_scan_i32(const int32_t* counts, int64_t n, int64_t* offsets, void* workspace,
                            wisp_stream_t stream);
/* int32 inclusive scan (kaolin inclusive_sum_cuda). */
int wisp_inclusive_scan_i32(const int32_t* in, int64_t n, int32_t* out, void* workspace,
                            wisp_stream_t stream);
/* boundary flags -> pack start indices.  Phase 1 counts flags per 2048-element tile into counts
 * [ceil(n/2048)]; after an exclusive scan phase 2 writes starts i64 [P]. */
int wisp_boundary_tile_counts(const uint8_t* boundary, int64_t n, int32_t* counts, wisp_stream_t stream);
int wisp_boundary_pack_starts(const uint8_t* boundary, int64_t n, const int64_t* tile_offsets,
                              int64_t* starts, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dual-octree trilinear interpolation  (replace kaolin.ops.spc.unbatched_interpolate_trilinear at
 * wisp/models/grids/octree_grid.py:147-149 and kaolin.ops.spc.coords_to_trilinear_coeffs at
 * wisp/models/grids/codebook_grid.py:164)
 *
 *  coords   f32 [V, S, 3]      S samples inside each of V voxels
 *  pidx     i32 or i64 [V]     point-hierarchy index of the voxel (-1: outside -> zeros)
 *  trinkets i32 [P, 8]         per-voxel indices of its 8 corners, local to the level's feature tensor
 *  feats    dtype [Fn, C]      level-local corner features; out / grad_out f32 [V, S, C]
 *  half_round != 0 reproduces the reference's `feats.half() ... .float()` rounding in registers.
 */
int wisp_spc_trilinear_coeffs(const float* coords, const int16_t* voxel_points /* [V,3] */, int64_t num_voxels,
                              int samples_per_voxel, int level, float* coeffs /* [V,S,8] */, wisp_stream_t stream);
int wisp_spc_trilinear_fwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                           const int32_t* trinkets, const void* feats, int dtype, int64_t num_voxels,
                           int samples_per_voxel, int channels, int level, int half_round, float* out,
                           wisp_stream_t stream);
/* Backward w.r.t. the features: grad_feats f32 [num_rows, channels] is ADDED to.  Order-free: the corner sums are taken in
 * 64-bit fixed point (scaled from the launch's largest product), so the same inputs give the same bits on every run - the
 * reference's float atomics do not.  workspace: wisp_spc_bwd_workspace_bytes(num_rows, channels, 0) bytes of device memory
 * that are ZERO before the first call; every call leaves them zero (but for a 64-byte header it resets itself), so one
 * buffer serves all calls of a stream.
 * A non-finite grad_out (overflowed loss scale) is scattered with plain float atomics so that inf / NaN reach the gradient. */
int wisp_spc_trilinear_bwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                           const int32_t* trinkets, const float* grad_out, int64_t num_voxels,
                           int samples_per_voxel, int channels, int level, int64_t num_rows,
                           float* grad_feats /* f32, accumulated */, void* workspace, int64_t workspace_bytes,
                           wisp_stream_t stream);
/* Scratch of the trilinear / codebook backward passes: total_rows = feature (logits) rows of all levels of the call,
 * dict_elems = num_lods * dict_size * feature_dim for the codebook entry points, else 0. */
int64_t wisp_spc_bwd_workspace_bytes(int64_t total_rows, int channels, int64_t dict_elems);

/* All active levels of an OctreeGrid in one launch (wisp/models/grids/octree_grid.py:183-219: one trilinear lookup per
 * level, then cat or sum).  coords f32 [N,3]; chain i64 [N, chain_stride], column l = voxel (point-hierarchy index, -1 =
 * outside) of the sample on level levels[l]; feats / grad_feats: HOST arrays of num_lods device pointers ([corners_l,
 * channels] each, dtype as given / f32); levels: HOST i32 [num_lods].  out f32 [N, num_lods*channels] (sum = 0) or
 * [N, channels] (sum = 1).  Same per-level arithmetic (and half_round meaning) as wisp_spc_trilinear_fwd/_bwd. */
int wisp_spc_trilinear_multi_fwd(const float* coords, const int64_t* chain, int64_t chain_stride, const int16_t* points,
                                 const int32_t* trinkets, const void* const* feats, int dtype, int64_t num_samples,
                                 int num_lods, const int32_t* levels, int channels, int half_round, int sum, float* out,
                                 wisp_stream_t stream);
/* rows: HOST i64 [num_lods], the row count of every grad_feats tensor; workspace as for wisp_spc_trilinear_bwd with
 * total_rows = sum(rows).  One scatter launch for all levels. */
int wisp_spc_trilinear_multi_bwd(const float* coords, const int64_t* chain, int64_t chain_stride, const int16_t* points,
                                 const int32_t* trinkets, const float* grad_out, int64_t num_samples, int num_lods,
                                 const int32_t* levels, const int64_t* rows, int channels, int sum,
                                 float* const* grad_feats, void* workspace, int64_t workspace_bytes, wisp_stream_t stream);

/* TriplanarGrid.interpolate (wisp/models/grids/triplanar_grid.py:97-146, TriplanarFeatureVolume.forward :205-233): per
 * level three bilinear plane lookups with torch.nn.functional.grid_sample semantics (align_corners=True, reflection
 * padding) - plane x sampled at (y, z), plane y at (x, z), plane z at (x, y) - laid out [x | y | z] per level, then cat
 * (sum = 0: out f32 [N, num_lods*3*feature_dim]) or sum (sum = 1: [N, 3*feature_dim]) over levels.
 * planes / grad_planes: HOST arrays of num_lods*3 device pointers (x, y, z of level 0, then level 1, ...), each plane f32
 * [feature_dim, size_l, size_l] as the reference stores it; sizes: HOST i32 [num_lods].  The backward accumulates. */
int wisp_triplane_fwd(const float* coords, int64_t num_samples, const float* const* planes, const int32_t* sizes,
                      int num_lods, int feature_dim, int sum, float* out, wisp_stream_t stream);
int wisp_triplane_bwd(const float* coords, int64_t num_samples, const float* grad_out, const int32_t* sizes, int num_lods,
                      int feature_dim, int sum, float* const* grad_planes, wisp_stream_t stream);

/* wisp._C.ops.grid_interpolate_cuda / grid_interpolate_backward_cuda (wisp/csrc/ops/grid_interpolate.h; kernels
 * grid_interpolate_cuda.cu:17-129): blend of eight gathered corner rows.  coords f32 [N,3] LOCAL coordinates in [0,1],
 * feats / grad_feats [N, 8, feature_dim] and out / grad_out [N, feature_dim] in `dtype`; corner k = dx<<2 | dy<<1 | dz. */
int wisp_grid_interpolate_fwd(const float* coords, const void* feats, int dtype, int64_t num_coords, int feature_dim,
                              void* out, wisp_stream_t stream);
int wisp_grid_interpolate_bwd(const float* coords, const void* grad_out, int dtype, int64_t num_coords, int feature_dim,
                              void* grad_feats, wisp_stream_t stream);

/* VQAD codebook lookup fused with the trilinear blend (replaces CodebookOctreeGrid._index_features + _interpolate,
 * wisp/models/grids/codebook_grid.py:103-172): logits f32 [Fn, dict_size], dictionary f32 [dict_size, feature_dim]
 * (dict_size <= 256, feature_dim <= 16).  training != 0: straight-through softmax one-hot; else argmax lookup.
 * Backward (training semantics): grad_logits [num_logit_rows = Fn, dict_size] and grad_dictionary are ADDED to; order-free
 * like wisp_spc_trilinear_bwd (same workspace rules; total_rows = logits rows, channels = feature_dim, dict_elems =
 * num_lods * dict_size * feature_dim).  feature_dim <= dict_size.  With a non-finite grad_out the corner sums land as float
 * atomics in the first feature_dim columns of grad_logits - enough for a found-inf check to see them, nothing more. */
int wisp_codebook_trilinear_fwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                                const int32_t* trinkets, const float* logits, const float* dictionary,
                                int64_t num_voxels, int samples_per_voxel, int dict_size, int feature_dim, int level,
                                int training, float* out /* [V,S,feature_dim] */, wisp_stream_t stream);
/* decoded[row] = dictionary[argmax(logits[row])] * straight-through scale (training) - the per-corner feature the VQAD grid
 * blends (wisp/models/grids/codebook_grid.py:103-127 `_index_features`), f32 [num_rows, feature_dim]; wisp_spc_trilinear_fwd
 * over it equals wisp_codebook_trilinear_fwd bit for bit at a fraction of the work. */
int wisp_codebook_decode_rows(const float* logits, const float* dictionary, int64_t num_rows, int dict_size, int feature_dim,
                              int training, float* decoded, wisp_stream_t stream);
int wisp_codebook_trilinear_bwd(const float* coords, const void* pidx, int pidx_is_i64, const int16_t* points,
                                const int32_t* trinkets, const float* logits, const float* dictionary,
                                const float* grad_out, int64_t num_voxels, int samples_per_voxel, int dict_size,
                                int feature_dim, int level, int64_t num_logit_rows, float* grad_logits,
                                float* grad_dictionary, void* workspace, int64_t workspace_bytes, wisp_stream_t stream);
/* All levels of a CodebookOctreeGrid in one scatter launch (the reference loops over the levels in Python,
 * wisp/models/grids/octree_grid.py:200-213 calling codebook_grid.py:138-172 per level): chain / levels / rows / sum as in
 * wisp_spc_trilinear_multi_bwd; logits, dictionaries, grad_logits, grad_dictionaries: HOST arrays of num_lods device pointers;
 * grad_out f32 [N, feature_dim] (sum = 1) or [N, num_lods * feature_dim]. */
int wisp_codebook_trilinear_multi_bwd(const float* coords, const int64_t* chain, int64_t chain_stride, const int16_t* points,
                                      const int32_t* trinkets, const float* const* logits, const float* const* dictionaries,
                                      const float* grad_out, int64_t num_samples, int num_lods, const int32_t* levels,
                                      const int64_t* rows, int dict_size, int feature_dim, int sum, float* const* grad_logits,
                                      float* const* grad_dictionaries, void* workspace, int64_t workspace_bytes,
                                      wisp_stream_t stream);

/* One optimisation step's forward + loss + backward of the reference's SDFTrainer (wisp/trainers/sdf_trainer.py:65-124 with
 * only_last: loss = sum((pred - gt)^2) / n) for a NeuralSDF over an OctreeGrid (wisp/models/nefs/neural_sdf.py:102-155;
 * app/nglod/configs/nglod_octree.yaml: 16 'sum' features on num_lods levels, decoder [position, features] -> Linear -> relu ->
 * Linear(hidden, 1)) in four launches instead of the modular path's twenty (the reference: ~60): walk + lookups + decoder forward
 * and backward per sample, a fixed-order sum of the decoder's weight gradients, the order-free corner scatter and its row pass.
 *  coords f32 [n,3], gts f32 [n]; octree / exsum / points / trinkets as above; feats / grad_feats: HOST arrays of num_lods device
 *  pointers (f32 [rows[l], 16]), levels strictly increasing (HOST), rows HOST i64; w1 f32 [hidden, 19], b1 [hidden], w2 [hidden],
 *  b2 [1] and their gradients, which - like grad_feats - are ADDED to; loss f32 [1] is written.
 *  scratch: wisp_sdf_train_scratch_bytes(...) bytes, contents irrelevant; workspace: as for wisp_spc_trilinear_multi_bwd.
 * Bitwise repeatable: every sum has a fixed order or is an integer sum. */
int64_t wisp_sdf_train_scratch_bytes(int64_t n, int num_lods, int channels, int hidden);
int wisp_sdf_train_step(const float* coords, const float* gts, int64_t n, const uint8_t* octree, const int32_t* exsum,
                        const int16_t* points, const int32_t* trinkets, const float* const* feats, const int32_t* levels,
                        const int64_t* rows, int num_lods, int channels, int half_round, const float* w1, const float* b1,
                        const float* w2, const float* b2, int hidden, float* const* grad_feats, float* grad_w1, float* grad_b1,
                        float* grad_w2, float* grad_b2, float* loss, void* scratch, int64_t scratch_bytes, void* workspace,
                        int64_t workspace_bytes, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Raymarch sample generation  (replace OctreeAS._raymarch_ray / _raymarch_voxel / _raymarch_uniform,
 * wisp/accelstructs/octree_as.py:188-374, wisp/ops/spc/sampling.py:35-71 and
 * wisp._C.ops.uniform_sample_cuda, wisp/csrc/ops/uniform_sample_cuda.cu:18-98)
 *
 * 'ray' mode, two phases.  count: for every ray evaluates the num_samples stratified depths, tests
 * occupancy and stores a hit mask (u32 [R, ceil(N/32)]) plus the per-ray hit count.  emit: expands the
 * mask into the packed outputs at offsets[r].  jitter: f32 [R,N] in [0,1) or NULL, in which case a
 * counter-based generator keyed by (seed, ray, step) is used (the reference uses torch.rand, unseeded).
 * occ_bits: bitfield of `level` (wisp_spc_build_bitfield) - or NULL to walk octree/exsum.
 * coarse_bits: optional bitfield of a coarser level of the SAME octree (coarse_level <= 5, <= level), or NULL.
 * It only prunes work - 64-candidate chunks whose segment meets no occupied coarse cell are not evaluated -
 * and never changes an output; pick the level whose cell size is about one chunk (64 * range / num_samples).
 */
int wisp_raymarch_ray_count(const uint32_t* occ_bits, const uint8_t* octree, const int32_t* exsum,
                            const float* origins, const float* dirs, int64_t num_rays,
                            float near, float range /* fl32(dist_max - dist_min) */, int num_samples, int level,
                            const float* jitter, uint64_t seed,
                            const uint32_t* coarse_bits, int coarse_level,
                            uint32_t* hitmask, int32_t* counts, wisp_stream_t stream);
int wisp_raymarch_ray_emit(const float* origins, const float* dirs, int64_t num_rays,
                           float near, float range, int num_samples,
                           const float* jitter, uint64_t seed,
                           const uint32_t* hitmask, const int64_t* offsets,
                           int64_t* ridx, float* samples, float* depth_samples, float* deltas,
                           uint8_t* boundary,
                           float* sample_dirs /* f32 [S,3] = dirs[ridx] (rays.dirs.index_select(0, ridx),
                                                 packed_rf_tracer.py:120), or NULL */,
                           wisp_stream_t stream);

/* 'voxel' mode: num_samples jittered samples inside every nugget; S = M*num_samples.
 * nug_ridx i32 [M], nug_depth f32 [M,2]; jitter f32 [M,N] or NULL (+seed). */
int wisp_raymarch_voxel_emit(const float* origins, const float* dirs,
                             const int32_t* nug_ridx, const float* nug_depth, int64_t num_nuggets,
                             int num_samples, const float* jitter, uint64_t seed,
                             int64_t* ridx, float* samples, float* depth_samples, float* deltas,
                             uint8_t* boundary, wisp_stream_t stream);

/* 'uniform' mode: counts[i] = ceil(scale*exit) - ceil(scale*entry) per nugget, then emit at the
 * exclusive scan of counts.  ray_first i64 [R+1] = nugget offsets per ray (the raytrace offsets). */
int wisp_raymarch_uniform_count(const float* nug_depth, int64_t num_nuggets, float scale,
                                int32_t* counts, wisp_stream_t stream);
int wisp_raymarch_uniform_emit(const float* origins, const float* dirs,
                               const int32_t* nug_ridx, const float* nug_depth, int64_t num_nuggets,
                               float scale, const int64_t* sample_offsets, const int64_t* ray_first,
                               int64_t* ridx, float* samples, float* depth_samples, uint8_t* boundary,
                               wisp_stream_t stream);

/* wisp._C.ops.uniform_sample_cuda under its own signature (wisp/csrc/ops/uniform_sample.cpp:28-42; kernel
 * uniform_sample_cuda.cu:18-59): ridx i32 [V], depth f32 [V,2], insum i32 [V] = inclusive sum of the per-nugget sample
 * counts (all > 0) -> new_ridx i64 [S], depth_samples f32 [S], boundary u8 [S], S = insum[V-1] (read by the caller). */
int wisp_uniform_sample(int scale, const int32_t* ridx, const float* depth, const int32_t* insum, int64_t num_nuggets,
                        int64_t* new_ridx, float* depth_samples, uint8_t* boundary, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Packed volume integration  (replace kaolin.render.spc.exponential_integration / sum_reduce / cumsum
 * and the scatter block of PackedRFTracer.trace, wisp/tracers/packed_rf_tracer.py:143-165)
 *
 *  pack_starts i64 [P]  first sample of every pack (== boundary.nonzero()), S = total samples
 */
int wisp_packed_sum_reduce(const float* feats, int64_t num_samples, int channels,
                           const int64_t* pack_starts, int64_t num_packs, float* out /* [P,C] */,
                           wisp_stream_t stream);
int wisp_packed_cumsum(const float* feats, int64_t num_samples, int channels,
                       const int64_t* pack_starts, int64_t num_packs, int exclusive, int reverse,
                       float* out /* [S,C] */, wisp_stream_t stream);

/* Fused tracer compositing: tau = density*delta; w_i = exp(-sum_{j<i} tau_j)(1-exp(-tau_i));
 * rgb[r] = bg*(1-sum w) + sum w*c ; alpha[r] = sum w ; depth[r] = sum w*t ; hit[r] = alpha > 0.
 * Rays without samples get bg / 0 / 0 / false.  weights f32 [S] is an output (kaolin returns it).
 * depths / out_depth may be NULL together.
 * Ray-offset mode: pass ridx = NULL, pack_starts = per-ray sample offsets i64 [R+1] and num_packs = num_rays; packs are
 * then addressed by ray (every ray is its own, possibly empty, pack) and no boundary compaction is needed. */
int wisp_composite_fwd(const float* color /* [S,3] */, const float* density /* [S] */,
                       const float* deltas /* [S] */, const float* depths /* [S] or NULL */,
                       const int64_t* ridx /* [S] */, const int64_t* pack_starts, int64_t num_packs,
                       int64_t num_samples, int64_t num_rays, const float* bg /* host [3] */,
                       float* out_rgb /* [R,3] */, float* out_alpha /* [R] */,
                       float* out_depth /* [R] or NULL */, uint8_t* out_hit /* [R] */,
                       float* weights /* [S] */, wisp_stream_t stream);
int wisp_composite_bwd(const float* grad_rgb /* [R,3] */, const float* grad_alpha /* [R] or NULL */,
                       const float* grad_depth /* [R] or NULL */,
                       const float* color, const float* density, const float* deltas, const float* depths,
                       const int64_t* ridx, const int64_t* pack_starts, int64_t num_packs,
                       int64_t num_samples, const float* bg /* host [3] */,
                       float* grad_color /* [S,3] */, float* grad_density /* [S] */,
                       wisp_stream_t stream);

/* Sphere-tracing helper (replaces wisp._C.render.find_depth_bound_cuda, wisp/csrc/render/find_depth_bound.cpp:23-36,
 * kernel find_depth_bound_cuda.cu:16-45): out[i] = first nugget index >= curr_idxes[i] whose [entry, exit] contains or
 * lies beyond query[i], or -1.  nug_depth f32 [M,2]. */
int wisp_find_depth_bound(const float* query /* [P] */, const int32_t* curr_idxes /* [P] */, const float* nug_depth,
                          int64_t num_packs, int64_t num_nugs, int32_t* out /* [P] */, wisp_stream_t stream);

/* One iteration of the sphere-tracing loop of PackedSDFTracer.trace (wisp/tracers/packed_sdf_tracer.py:118-146) for all
 * rays that own nuggets ("packs"): t += dist; convergence (|dist| < thr_close or |dist + dist_prev| / 2 < thr_avg);
 * far plane; nugget search as wisp_find_depth_bound; jump to the next cell; new query point x.  All state arrays are
 * [P] (x is [P,3]) and updated in place; curr_in / curr_out double-buffer the current nugget index. */
int wisp_sphere_trace_step(int64_t num_packs, const float* nug_o, const float* nug_d, const float* nug_depth,
                           const int32_t* nug_pidx, float dist_max, float thr_close, float thr_avg, float* t,
                           const float* dist, float* dist_prev, uint8_t* mask, uint8_t* hit, const int32_t* curr_in,
                           int32_t* curr_out, int64_t* curr_pidx, float* x, wisp_stream_t stream);

/* One marching iteration INCLUDING the field query of a NeuralSDF over an OctreeGrid ('sum' of the active levels, 16
 * feature channels) with a one-hidden-layer relu decoder on [position, features] (wisp/models/nefs/neural_sdf.py:120-155,
 * nglod_octree.yaml): everything wisp_sphere_trace_step does, then dist[p] = scale * decoder(x[p]) for the packs still
 * marching - one launch per iteration, no host decision in between (first != 0: only the query, for the start positions).
 * feats / levels: HOST arrays of num_lods device pointers / octree levels (ascending); w1 f32 [hidden, 3 + channels] in
 * nn.Linear layout, b1 [hidden], w2 [hidden] (the output layer's row), b2 [1].  any_active (optional device counter) is
 * incremented once per pack that is still marching after the step. */
int wisp_sdf_trace_step_fused(int64_t num_packs, int first, const float* nug_o, const float* nug_d, const float* nug_depth,
                              const int32_t* nug_pidx, float dist_max, float thr_close, float thr_avg, float* t, float* dist,
                              float* dist_prev, uint8_t* mask, uint8_t* hit, const int32_t* curr_in, int32_t* curr_out,
                              int64_t* curr_pidx, float* x, const uint8_t* octree, const int32_t* exsum, const int16_t* points,
                              const int32_t* trinkets, const void* const* feats, int feats_dtype, const int32_t* levels,
                              int num_lods, int channels, int half_round, const float* w1, const float* b1, const float* w2,
                              const float* b2, int hidden, float scale, int32_t* any_active, wisp_stream_t stream);

/* Compositing + photometric loss + compositing backward of a TRAINING step in one launch (what
 * wisp/tracers/packed_rf_tracer.py:143-165, wisp/trainers/multiview_trainer.py:140-154 and their autograd backward do in
 * sequence): per ray rgb = bg (1 - sum w) + sum w c with w from exponential_integration (exclusive), loss = mean over the
 * 3 num_rays elements of huber (kind 0) / l2 (1) / l1 (2) of rgb - gt, and d loss / d color [S,3], d loss / d density [S,1]
 * (the gradients wisp_composite_bwd returns for grad_rgb = wisp_rgb_loss's gradient, no alpha / depth terms).
 * ray_offsets: i64 [num_rays + 1] sample range of every ray (rays without samples see the background); out_rgb: optional
 * f32 [num_rays,3]; loss: f32 [1]; workspace: f32 [workspace_floats] - one partial sum of the loss per wave; with
 * workspace_floats >= num_rays every ray gets its own wave (best), fewer make the waves walk the rays with a stride; a
 * second, one-workgroup launch adds the partial sums in index order (reproducible).  Results equal the three separate entry points' (loss to rounding: its terms are
 * grouped per ray). */
int wisp_composite_loss(const float* color, const float* density, const float* deltas, const int64_t* ray_offsets,
                        int64_t num_rays, int64_t num_samples, const float* bg, const float* gt, int kind,
                        float* grad_color, float* grad_density, float* out_rgb, float* loss, float* workspace,
                        int64_t workspace_floats, wisp_stream_t stream);

/* Photometric loss of MultiviewTrainer.step (wisp/trainers/multiview_trainer.py:140-154) and its gradient in one launch:
 * loss[0] = mean over the num_elements entries of huber(beta = 1) (kind 0) / squared (1) / absolute (2) error of rgb
 * against gt; grad[i] = d loss / d rgb[i].  rgb, gt, grad: f32 [num_elements]; loss: f32 [1];
 * workspace: 257 dwords that belong to this entry point (dword 0 unused since ABI 3 + this revision, dwords 1.. per-workgroup partial
 * sums a one-workgroup second launch adds in index order; calls sharing a workspace must be stream-ordered). */
int wisp_rgb_loss(const float* rgb, const float* gt, int64_t num_elements, int kind, float* grad, float* loss,
                  float* workspace, wisp_stream_t stream);

/* Rays of one camera through the given pixel coordinates - generate_pinhole_rays / generate_ortho_rays
 * (wisp/ops/raygen/raygen.py:40-119).  pixel_x / pixel_y: f32 [num_pixels] (device); the camera is passed by value from
 * the host: principal-point offset (x0, y0) in pixels from the image centre, image size, scale_x / scale_y =
 * tan(fov/2) per axis (pinhole) or fov_distance * aspect, fov_distance (ortho), and the world -> camera view transform
 * as rotation R[9] (row major) + translation t[3] (HOST pointers).  Out: origins, dirs f32 [num_pixels, 3], dirs normalised. */
int wisp_generate_rays(const float* pixel_x, const float* pixel_y, int64_t num_pixels, int ortho, float x0, float y0,
                       float width, float height, float scale_x, float scale_y, const float* view_rotation,
                       const float* view_translation, float* origins, float* dirs, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused radiance-field decoder  (replaces NeuralRadianceField.rgba after grid.interpolate,
 * wisp/models/nefs/nerf.py:245-264: decoder_density (Linear-ReLU-Linear) -> relu density + 15 geometry
 * features -> cat positional-encoded view dir (wisp/models/embedders/positional_embedder.py:51-66)
 * -> decoder_color (Linear-ReLU x2, Linear) -> sigmoid)
 *
 *  feats   dtype_io [S, in_dim], rows packed (stride in_dim).  This build: 1 <= in_dim <= 32, view_freqs = 4 and
 *          hidden = 64 (the YAML decoders of every app/nerf config: in_dim 32 nerf_hash, 5 nerf_octree / nerf_codebook,
 *          12 nerf_triplanar; fp32 or bf16 compute) or hidden = 128 (the reference's best nerf_hash row and the documented
 *          VQAD command line, docs/pages/app_nerf.md:175-192; bf16 compute only); other shapes return WISP_ERR_UNSUPPORTED
 *  dirs    f32 [S,3]
 *  params  f32 packed, layout given by wisp_nerf_mlp_param_count(): W1[hid,in], b1[hid], W2[16,hid],
 *          b2[16], W3[hid, 15+pe], b3[hid], W4[hid,hid], b4[hid], W5[3,hid], b5[3]   (row-major
 *          [out,in] like nn.Linear.weight; biases present but zero when the model has bias=False)
 *  compute_dtype: WISP_F32 (exact fp32 MFMA) or WISP_BF16 (bf16 MFMA, fp32 accumulate)
 */
int64_t wisp_nerf_mlp_param_count(int in_dim, int hidden, int view_freqs);
/* floats of device scratch the hidden-64 backward needs (per-wave partial weight gradients). */
int64_t wisp_nerf_mlp_workspace_floats(void);
/* bytes of device scratch wisp_nerf_mlp_bwd needs for `num_samples` samples: the above for hidden 64; for hidden 128 the
 * partial gradient rows plus the (dY, X) operand dump of one chunk of <= 2^20 samples (1.76 KB per sample). */
int64_t wisp_nerf_mlp_bwd_workspace_bytes(int64_t num_samples, int hidden);
int wisp_nerf_mlp_fwd(const void* feats, int dtype_io, const float* dirs, int64_t num_samples,
                      int in_dim, int hidden, int view_freqs, const float* params, int compute_dtype,
                      float* rgb /* [S,3] */, float* density /* [S] */, wisp_stream_t stream);
int wisp_nerf_mlp_bwd(const void* feats, int dtype_io, const float* dirs, int64_t num_samples,
                      int in_dim, int hidden, int view_freqs, const float* params, int compute_dtype,
                      const float* grad_rgb /* [S,3] */, const float* grad_density /* [S] */,
                      void* grad_feats /* dtype_io [S,in_dim] */, float* grad_params /* accumulated */,
                      float* workspace, int64_t workspace_bytes /* >= wisp_nerf_mlp_bwd_workspace_bytes() */,
                      wisp_stream_t stream);

/* The same decoder with the view direction given per RAY.  The reference gathers a direction per sample
 * (wisp/tracers/packed_rf_tracer.py:70-76 `rays.dirs.index_select(0, ridx)`) and positional_embedder.py:61-65 encodes each
 * sample; the encoding depends on the ray only, so wisp_nerf_mlp_dir_code encodes every ray once
 * (code: bf16 [num_rays, 32], opaque layout) and the *_rays kernels gather the code by `ridx` (int64 [S], the ray of every
 * sample as the raymarch returns it).  Same arithmetic, bit-identical outputs to wisp_nerf_mlp_fwd / _bwd on gathered
 * directions.  For the bf16-compute hidden-64 kernels: 1 <= in_dim <= 32 (narrow rows of the octree / codebook / triplanar
 * fields included since round 4), view_freqs 4, f32 / f16 / bf16 features; anything else returns WISP_ERR_UNSUPPORTED (use the
 * per-sample entry points). */
int wisp_nerf_mlp_dir_code(const float* ray_dirs /* [R,3] */, int64_t num_rays, int view_freqs, void* code, wisp_stream_t stream);
int wisp_nerf_mlp_fwd_rays(const void* feats, int dtype_io, const void* dir_code, const int64_t* ridx, int64_t num_samples,
                           int in_dim, int hidden, int view_freqs, const float* params, float* rgb, float* density,
                           wisp_stream_t stream);
int wisp_nerf_mlp_bwd_rays(const void* feats, int dtype_io, const void* dir_code, const int64_t* ridx, int64_t num_samples,
                           int in_dim, int hidden, int view_freqs, const float* params, const float* grad_rgb,
                           const float* grad_density, void* grad_feats, float* grad_params, float* workspace,
                           int64_t workspace_bytes, wisp_stream_t stream);

/* One-hidden-layer relu decoder with a single output, out = W2 relu(W1 x + b1) + b2 - the NeuralSDF decoder
 * (wisp/models/nefs/neural_sdf.py:102-118; nglod_octree.yaml: 19 -> 128 -> 1; BasicDecoder.forward,
 * wisp/models/decoders/basic_decoders.py:73-101).  x f32 [n, in_dim] (in_dim <= 32), w1 [hidden, in_dim], b1 [hidden],
 * w2 [hidden], b2 [1] (nn.Linear layouts), hidden <= 256.  bwd: grad_out f32 [n]; grad_x is written, the parameter
 * gradients are accumulated into. */
int wisp_small_decoder_fwd(const float* x, int64_t n, int in_dim, int hidden, const float* w1, const float* b1,
                           const float* w2, const float* b2, float* out /* [n] */, wisp_stream_t stream);
int wisp_small_decoder_bwd(const float* x, int64_t n, int in_dim, int hidden, const float* w1, const float* b1,
                           const float* w2, const float* b2, const float* grad_out, float* grad_x, float* grad_w1,
                           float* grad_b1, float* grad_w2, float* grad_b2, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer  (replaces torch.optim.AdamW / apex FusedAdam over the flat parameter buffer,
 * wisp/trainers/base_trainer.py:205-235, wisp/config/presets/torch.py:22-58)
 * One launch over n contiguous fp32 parameters; grad_scale multiplies the gradient first
 * (1/world_size for data-parallel mean).  step is the 1-based step count.  bf16_shadow (optional, bf16 [n]) receives a
 * bf16 copy of the updated parameters so that the bf16 forward does not need a separate cast pass over the table.
 */
int wisp_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step,
                    float grad_scale, int zero_grad /* also zero grad[] */, void* bf16_shadow,
                    wisp_stream_t stream);
/* The same update for up to 4 parameter groups of ONE flat buffer in one launch - the optimizer's param groups
 * ('decoder' / 'grid' / rest with their own learning rates, base_trainer.py:216-235).  group_begin / group_len (elements;
 * a group that begins on a multiple of 4 is processed 16 bytes at a time, any other element by element), group_lr,
 * group_weight_decay, group_bf16_shadow (device pointers, bf16 [len], or NULL entries / a NULL array) are HOST arrays of
 * num_groups entries. */
int wisp_adamw_step_groups(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int num_groups,
                           const int64_t* group_begin, const int64_t* group_len, const float* group_lr,
                           const float* group_weight_decay, void* const* group_bf16_shadow, float beta1, float beta2,
                           float eps, int64_t step, float grad_scale, int zero_grad, wisp_stream_t stream);

/* The other optimizers BaseTrainer.init_optimizer is configured with (wisp/config/presets/torch.py:45-68; RMSprop:
 * app/nerf/configs/nerf_octree.yaml:85, nerf_codebook.yaml:86), same group / shadow / zeroing structure, arithmetic in
 * torch.optim's order.  kind 0 = AdamW (state1 exp_avg, state2 exp_avg_sq, hyper0/1 = beta1/beta2), 1 = Adam (coupled
 * weight decay), 2 = RMSprop (state2 square_avg, hyper0 = alpha, hyper1 = momentum, state1 = momentum buffer, may be NULL
 * when momentum == 0; not centered). */
int wisp_optim_step_groups(int kind, float* param, const float* grad, float* state1, float* state2, int num_groups,
                           const int64_t* group_begin, const int64_t* group_len, const float* group_lr,
                           const float* group_weight_decay, void* const* group_bf16_shadow, float hyper0, float hyper1,
                           float eps, int64_t step, float grad_scale, int zero_grad, wisp_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Ray-batch sampling (replaces the per-tensor index_select of SampleRays,
 * wisp/datasets/transforms/ray_sampler.py:25-35): dst[k][i, :] = src[k][index[i], :] for up to 4 row-major f32 tensors
 * of num_src_rows rows sharing one index vector (i64 [num], negative entries count from the end).  src / width / dst are
 * HOST arrays of num_tensors device pointers / row widths. */
int wisp_gather_rows(const int64_t* index, int64_t num, int64_t num_src_rows, int num_tensors, const float* const* src,
                     const int* width, float* const* dst, wisp_stream_t stream);

/* Read-back of one int64 (a sample / nugget count) without draining the compute stream: where the reference syncs to learn a size
 * (octree_as.py:288 `nonzero`, uniform_sample_cuda.cu:76 blocking cudaMemcpy), the value is copied to pinned memory on a side
 * stream that only waits for the kernel that produced it; the host waits for that copy alone.  create (on the current device) ->
 * issue (src: DEVICE int64, written by work already queued on `stream`) -> wait (blocks until the copy has landed, returns the
 * value) ... -> destroy.  A reader carries one value at a time; pool them.  Added in round 5 (no signature changed). */
void* wisp_host_reader_create(void);
int wisp_host_reader_issue(void* reader, const int64_t* src, wisp_stream_t stream);
int wisp_host_reader_wait(void* reader, int64_t* value /* HOST */);
void wisp_host_reader_destroy(void* reader);

/* ------------------------------------------------------------------------------------------------
 * One training step of the nerf_hash.yaml shape issued from native code (round 6): what MultiviewTrainer.step
 * (wisp/trainers/multiview_trainer.py:111-180) runs per iteration - PackedRFTracer.trace (packed_rf_tracer.py:84-181) over
 * OctreeAS._raymarch_ray (octree_as.py:247-309), HashGrid.interpolate (hash_grid.py:205-233), NeuralRadianceField.rgba
 * (nerf.py:219-264), compositing, the rgb loss, backward and the optimizer (base_trainer.py:205-246) - as ONE call: the function
 * issues the per-op entry points of this header in the order the Python host layer does (trainers/multiview_trainer.py::
 * _DirectNeRFStep.run), on buffers carved out of one caller-owned workspace.  Same kernels, same arguments, same bits; what it saves
 * is the host: ~0.2 ms of interpreter time per step against ~0.3 ms of kernels at the reference trainer's 2^18 samples per step.
 *
 * All pointers of the config are BORROWED for the life of the handle (device pointers unless marked host; the two host arrays are
 * copied).  struct_bytes = sizeof the struct as the caller compiled it (checked against the library's: wisp_nerf_step_config_bytes).
 * A batch is COUNTED first (occupancy test of every candidate + per-ray offsets + an asynchronous read-back of the sample total,
 * wisp_host_reader_*) into one of two slots - normally one step ahead, by the `next_*` arguments of wisp_nerf_step_run - and RUN
 * later from that slot; the ray tensors given to the count must stay alive and unchanged until the run.  Jitter comes from the
 * counter-based generator keyed by `seed` (as in wisp_raymarch_ray_count / _emit with jitter = NULL).
 * Shapes covered: 'ray' march, two-feature 'cat' table of <= 16 levels read through a 16-bit copy (dtype_table), hidden 64, four
 * view frequencies (the per-ray view code kernels), in_dim = num_lods * 2.  Anything else: the per-op entry points. */
typedef struct wisp_nerf_step_config {
    int64_t struct_bytes;
    /* occupancy structure at `level` (OctreeAS: octree_as.py:42-63) */
    const uint32_t* occ_bits;        /* bitfield of `level` (wisp_spc_build_bitfield) or NULL */
    const uint8_t* octree;
    const int32_t* exsum;
    const uint32_t* coarse_bits;     /* optional coarser bitfield (see wisp_raymarch_ray_count) or NULL */
    /* hash table */
    const void* table_lookup;        /* dtype_table [rows, feature_dim]: what the forward gathers from (the bf16 shadow under amp) */
    const int64_t* first_idx;        /* device i64 [num_lods + 1] */
    const int64_t* first_idx_host;   /* HOST copy of the same */
    const int32_t* resolutions;      /* HOST i32 [num_lods] */
    float* table_param;              /* fp32 master table / its gradient / AdamW moments / bf16 copy (may be NULL): the folded update */
    float* table_grad;
    float* table_exp_avg;
    float* table_exp_avg_sq;
    void* table_shadow;
    /* decoder: packed parameters as wisp_nerf_mlp_* take them, and where their gradient is accumulated */
    const float* dec_params;
    float* dec_grad;
    /* the flat optimizer buffers all of the above live in, and their parameter groups (elements; base_trainer.py:216-235) */
    float* flat_param;
    float* flat_grad;
    float* flat_exp_avg;
    float* flat_exp_avg_sq;
    void* grid_shadow;               /* bf16 copy of the grid group [grid_len] or NULL */
    int64_t decoder_begin, decoder_len, grid_begin, grid_len, rest_begin, rest_len;
    int64_t table_offset;            /* element offset of the hash table inside the flat buffers */
    int64_t max_rays, max_samples;   /* capacities the workspace is sized for */
    int32_t level, coarse_level, num_samples /* candidates per ray */, loss_kind /* 0 huber, 1 l2, 2 l1 */;
    int32_t dtype_table, num_lods, feature_dim, bitwidth;
    int32_t zero_from_col, in_dim, hidden, view_freqs;
    float near, range /* fl32(dist_max - dist_min) */;
    float bg[3];
    float reserved;
} wisp_nerf_step_config;

typedef struct wisp_nerf_step_hyper {
    int64_t struct_bytes;
    int64_t step;                    /* optimizer step number, 1-based (bias correction) */
    float lr_decoder, lr_grid, lr_rest, weight_decay, beta1, beta2, eps, grad_scale;
    int32_t optimizer;               /* 0: none (gradients are left accumulated: the caller exchanges / applies them);
                                        1: one AdamW launch over all groups; 2: the table's AdamW folded into its backward
                                        (wisp_hashgrid_interpolate_bwd_adamw) + one launch for everything else */
    int32_t reserved;
} wisp_nerf_step_hyper;

int64_t wisp_nerf_step_config_bytes(void);
int64_t wisp_nerf_step_workspace_bytes(const wisp_nerf_step_config* cfg);
void* wisp_nerf_step_create(const wisp_nerf_step_config* cfg, void* workspace, int64_t workspace_bytes);
/* The same handle over other buffers of the same shapes and capacities (a prune replaces the octree and its bitfields): workspace,
 * read-back slots and timing events stay; batches counted against the old configuration are dropped. */
int wisp_nerf_step_reconfigure(void* step, const wisp_nerf_step_config* cfg);
void wisp_nerf_step_destroy(void* step);
int wisp_nerf_step_count(void* step, int slot, const float* origins, const float* dirs, int64_t num_rays, uint64_t seed,
                         wisp_stream_t stream);
/* Runs the batch counted into `slot`: waits for its sample total (host), emits the samples, counts the NEXT batch into the other
 * slot when next_origins is given, then forward, loss, backward and - hp->optimizer - the update.  gts: f32 [num_rays, 3].
 * level_cap_scale / hashgrid_workspace(_bytes): as for wisp_hashgrid_interpolate_bwd (the caller keeps the slot statistics).
 * Outputs (host): *num_samples; covered_rows [num_lods] (hp->optimizer == 2, else zeros; may be NULL); *loss = DEVICE pointer to
 * the f32 loss of this step (a ring of 8: valid until the eighth step after this one).  record_timing: bracket the four roofline
 * entry points with HIP events on `stream` (read them with wisp_nerf_step_read_timing after a synchronize).
 * WISP_ERR_CAPACITY: the batch holds more than max_samples samples, or none - the slot is consumed, nothing else was touched. */
int wisp_nerf_step_run(void* step, int slot, const float* gts, const float* next_origins, const float* next_dirs,
                       int64_t next_num_rays, uint64_t next_seed, const wisp_nerf_step_hyper* hp,
                       const float* level_cap_scale, void* hashgrid_workspace, int64_t hashgrid_workspace_bytes,
                       int record_timing, int64_t* num_samples, int64_t* covered_rows, float** loss, wisp_stream_t stream);
/* ms: host f32 [max_steps][4] = hashgrid_fwd, nerf_mlp_fwd, nerf_mlp_bwd, hashgrid_bwd of every timed step since the last read;
 * units: host i64 [max_steps] packed samples of those steps; *num_steps: how many were written.  Resets the record (which holds at
 * most 512 steps: later steps go untimed until it is read). */
int wisp_nerf_step_read_timing(void* step, int max_steps, float* ms, int64_t* units, int* num_steps);

#ifdef __cplusplus
}
#endif
#endif /* WISP_HIP_H */
