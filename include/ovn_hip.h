/*
 * ovn_hip.h -- C ABI of libovn_hip.so, the MI355X (gfx950) OverlapNet inference hot path.
 *
 * The reference (PRBonn/OverlapNet) has no FFI layer of its own: its hot path is reached through the
 * Python class `Infer` (src/two_heads/infer.py:22) which hands everything to Keras/TensorFlow.  These
 * entry points are what a binding for that path has to reach; each one names the reference code it
 * replaces.  The Python host (`overlapnet_amd/infer.py`) calls them through ctypes.
 *
 * Conventions
 *   - every pointer marked "dev" is a DEVICE pointer (HIP memory owned by the caller, e.g. a torch
 *     tensor's data_ptr()); the library never takes ownership and never frees caller memory;
 *   - all tensors are float32, channels-last (NHWC), densely packed;
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); every call only
 *     ENQUEUES work on it, nothing synchronises unless stated.  One exception: a context owns ONE scratch
 *     block that grows on demand; a call that needs more than any earlier call synchronises `stream`
 *     once to re-allocate it.  All calls on one context must therefore be issued on one stream at a time
 *     (two streams would race on the scratch).  A head call may internally fork onto context-owned side
 *     streams; it joins them back into `stream` before it returns (events, no host synchronisation);
 *   - every entry point selects the context's device for its own duration and restores the caller's
 *     current HIP device before it returns;
 *   - return value 0 = success, non-zero = error, message via ovn_last_error() (thread-local);
 *   - one context per GPU per process; calls on one context are not re-entrant (the reference object
 *     is not thread-safe either: mutable feature cache, infer.py:114,185).
 */
#ifndef OVN_HIP_H
#define OVN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ovn_ctx ovn_ctx;

#define OVN_OK 0
#define OVN_ERR_ARG 1      /* bad argument / unsupported shape */
#define OVN_ERR_HIP 2      /* HIP runtime error                */
#define OVN_ERR_STATE 3    /* call order (weights missing ...)  */

/* ABI version of this header; bumped on any signature change. */
#define OVN_ABI_VERSION 6
int ovn_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* ovn_last_error(void);

/* Create / destroy a context bound to HIP device `device_id` for leg inputs of in_h x in_w x in_c
 * (64 x 900 x C in the reference, infer.py:76-82).  Replaces the model construction of
 * Infer.__init__ (infer.py:86-111). */
int ovn_create(int device_id, int in_h, int in_w, int in_c, ovn_ctx** out);
int ovn_destroy(ovn_ctx* ctx);

/* Register one convolution layer of the leg, in network order (generateNet.py:161-214: s_conv1 ...
 * s_conv10).  kernel_dev: Keras layout (kh, kw, cin, cout); bias_dev: (cout).  The library keeps its
 * own re-tiled copy (MFMA fragment order), so the caller may release its buffers after the call
 * returns (the call synchronises `stream`).  All leg layers are valid-padded + bias + ReLU.
 * Replaces `leg.load_weights(file, by_name=True)` (infer.py:119). */
int ovn_add_leg_layer(ovn_ctx* ctx, const char* name, const float* kernel_dev, const float* bias_dev,
                      int kh, int kw, int cin, int cout, int stride_h, int stride_w, void* stream);

/* Register the Delta-head weights (generateNet.py:96-114): c_conv1 (1,15,128,64) linear,
 * c_conv2 (15,1,64,128) ReLU, c_conv3 (3,3,128,256) ReLU, overlap_output Dense (123904,1) sigmoid.
 * Same ownership rule as above.  Replaces `head.load_weights(...)` (infer.py:120). */
int ovn_set_head_weights(ovn_ctx* ctx, const float* c1_kernel_dev, const float* c1_bias_dev,
                         const float* c2_kernel_dev, const float* c2_bias_dev,
                         const float* c3_kernel_dev, const float* c3_bias_dev,
                         const float* dense_kernel_dev, const float* dense_bias_dev, void* stream);

/* Head geometry: `conv1NetworkHead_conv1size` of the reference's network.yml (generateNet.py:88-99: the 1 x s / s x 1 kernels and
 * strides of c_conv1 / c_conv2; default 15, which the shipped configuration uses).  Call BEFORE ovn_set_head_weights when the
 * model was built with another value: c_conv1 is then (1, s, 128, 64), c_conv2 (s, 1, 64, 128) and the Dense kernel has
 * (360 // s - 2)^2 * 256 inputs.  The MFMA-tiled Delta kernels (both arithmetic modes) serve s = 15; any other s runs a general fp32
 * path (DeltaLayer + c_conv1 as plain FMAs without materialising the difference tensor, c_conv2 / c_conv3 through the generic fp32
 * convolution) -- correct to the same tolerance, an order of magnitude slower, no Delta cache. */
int ovn_set_head_geometry(ovn_ctx* ctx, int conv1size);

/* Validate the registered leg chain: output must be 1 x feat_w x 128 (1 x 360 x 128 in the reference).
 * Writes the leg output width to *feat_w. */
int ovn_finalize(ovn_ctx* ctx, int* feat_w);

/* Leg: images_dev (n, in_h, in_w, in_c) -> features_dev (n, feat_w, 128).
 * Replaces `leg.predict_generator` in Infer.create_feature_volumes (infer.py:262-265).
 * A scan's feature volume depends on that scan alone: every call size runs the same kernels, and the power-of-two scales of the
 * f16x3 arithmetic are taken per scan (per strip / tile inside the fused first two layers and the fused tail) -- the same scan computed alone,
 * inside any batch, at any position and next to any other scans gives the same bits (the reference's `predict_generator` results
 * do not depend on batch composition either, infer.py:262-265).  Calls of more than 1024 scans are processed in equal slices of at
 * most 1024 (scratch: 2 x slice x 770 KB at in_c = 4, i.e. up to 1.6 GB, see ovn_workspace_bytes). */
int ovn_leg(ovn_ctx* ctx, const float* images_dev, int64_t n, float* features_dev, void* stream);

/* Both heads on n pairs.  Pair p uses l = feats_l_dev[lidx[p]] and r = feats_r_dev[ridx[p]]
 * (each feature volume feat_w x 128 floats); lidx_dev == NULL means lidx[p] = p, ridx_dev == NULL means
 * ridx[p] = 0 (the 1-vs-N sweep of Infer.infer_multiple, infer.py:188-190: l = candidate, r = query).
 * Outputs (any may be NULL except overlap/yaw):
 *   overlap_dev (n) f32  = sigmoid(logit)            (generateNet.py:114)
 *   yaw_dev     (n) i32  = 180 - argmax_k corr[k]    (infer.py:158; first maximum wins)
 *   logit_dev   (n) f32  pre-sigmoid value
 *   corr_dev    (n, feat_w) f32 orientation_output   (generateNet.py:352)
 * Replaces `head.predict_generator` + post-processing (infer.py:155-158,194-198,229-233) and the pair
 * gather of ImagePairOverlapSequenceFeatureVolume.__getitem__ (:43-47).
 * What a pair's bits depend on (f16x3 arithmetic): its two volumes and the left volume's SLOT in feats_l_dev modulo 32 (lidx[p],
 * or p without an index list) -- not on n, on the chunking of the sweep, on the order of the index list or on the workgroup
 * decomposition the library picks for small n.  A shard of a pool that starts at a slot which is a multiple of 32 therefore
 * reproduces the unsharded sweep bit for bit (overlapnet_amd/distributed.py: shard_bounds(align = 32), frame_slot).  The fp32 mode
 * has one fixed summation order: its bits do not depend on the slot either. */
int ovn_heads(ovn_ctx* ctx, const float* feats_l_dev, const int32_t* lidx_dev, const float* feats_r_dev,
              const int32_t* ridx_dev, int64_t n, float* overlap_dev, int32_t* yaw_dev, float* logit_dev,
              float* corr_dev, void* stream);

/* Delta (overlap) head alone (generateNet.py:64-116): overlap (n) f32, optional logit (n); indexing as ovn_heads.
 * Used together with ovn_corr_head_spectral when candidate spectra are cached. */
int ovn_delta_head(ovn_ctx* ctx, const float* feats_l_dev, const int32_t* lidx_dev, const float* feats_r_dev,
                   const int32_t* ridx_dev, int64_t n, float* overlap_dev, float* logit_dev, void* stream);

/* Correlation (yaw) head alone (NormalizedCorrelation2D.py:43-109 with normalize='none'); same
 * indexing convention as ovn_heads. */
int ovn_corr_head(ovn_ctx* ctx, const float* feats_l_dev, const int32_t* lidx_dev, const float* feats_r_dev,
                  const int32_t* ridx_dev, int64_t n, int32_t* yaw_dev, float* corr_dev, void* stream);

/* Spectral form of the correlation head (same result as ovn_corr_head up to fp32 rounding, HBM-bound):
 * ovn_spectrum turns feature volumes (n, 360, 128) into cached spectra (n, 128, 368) -- per channel the 181
 * non-redundant DFT bins, real parts at [0..180], imaginary parts at [184..364], zero padding elsewhere;
 * ovn_corr_head_spectral evaluates corr = IDFT( sum_c L^ conj(R^) ) shifted by W/2 and the argmax for n pairs
 * (l = spec_l[lidx[p]], r = spec_r[ridx[p]]; NULL index arrays as in ovn_heads).  yaw (n) int32, corr (n,360) or NULL.
 * Replaces NormalizedCorrelation2D.call (NormalizedCorrelation2D.py:43-109) + infer.py:158 for 1-vs-N sweeps. */
int ovn_spectrum(ovn_ctx* ctx, const float* feats_dev, int64_t n, float* spectra_dev, void* stream);
int ovn_corr_head_spectral(ovn_ctx* ctx, const float* spec_l_dev, const int32_t* lidx_dev, const float* spec_r_dev,
                           const int32_t* ridx_dev, int64_t n, int32_t* yaw_dev, float* corr_dev, void* stream);

/* Delta cache: everything of a pair's Delta-head preparation that depends on the LEFT volume (the candidate of a sweep) alone --
 * its feature volume re-written as the packed hi/lo fp16 words the contraction kernel streams (at the candidate's own power-of-two
 * scale; channel-major [128][360], so that a sweep fetches only the channels alive in its query: ovn_set_head_compaction), its linear term pushed through c_conv2 (TT + b2) and its value range.  OVN_DELTA_CACHE_ELEMS floats (196,864 B) per
 * volume, cached next to the feature volume and the spectrum (the reference caches per-candidate state too: infer.py:184-185).
 * A row is used by ovn_heads_spectral for a pair whenever neither volume has a negative value and the query's largest value is
 * below the candidate's next power of two; every other pair is prepared in scratch as without a cache -- same bits either way.
 * Valid for the head weights registered when it was built (f16x3 head mode; the fp32 mode ignores it). */
#define OVN_DELTA_CACHE_ELEMS 49216
int ovn_delta_cache(ovn_ctx* ctx, const float* feats_dev, int64_t n, float* cache_dev, void* stream);

/* Both heads of a sweep whose candidates have their spectra (and optionally their Delta cache rows, dcache_l_dev, may be NULL) cached
 * next to their feature volumes: what `Infer.infer_multiple` runs per query.  Same outputs and indexing as ovn_heads (one index
 * array addresses the feature, spectrum and Delta-cache pools alike); the Delta head reads the feature volumes / cache rows, the yaw
 * head the spectra (ONE launch for the whole sweep).  The Delta cache is used in the 1-vs-N form (ridx_dev == NULL).  With
 * ovn_set_head_pipeline the call can fork the independent kernel chains over context-owned side streams; it joins them back into
 * `stream` with events before it returns: to the caller everything is ordered as if enqueued on `stream`.
 * Replaces `head.predict_generator` + post-processing for 1-vs-N sweeps (infer.py:188-198). */
int ovn_heads_spectral(ovn_ctx* ctx, const float* feats_l_dev, const float* spec_l_dev, const float* dcache_l_dev,
                       const int32_t* lidx_dev, const float* feats_r_dev, const float* spec_r_dev, const int32_t* ridx_dev, int64_t n,
                       float* overlap_dev, int32_t* yaw_dev, float* logit_dev, float* corr_dev, void* stream);

/* Launch structure of the head calls (the reference's counterpart is `batch_size`, network.yml:41, which sets how many pairs one
 * predict step materialises):
 *   chunk_pairs          pairs per pass over the context's scratch (default 1024).  The f16x3 Delta path needs 2.9 MB of scratch
 *                        per pair of a chunk (packed volumes, linear terms, the c_conv1 rows between its two kernels):
 *                        ovn_workspace_bytes ~ 3.3 GB at the default; a sweep longer than a chunk is processed in several passes;
 *   sub_chunk_pairs      0 = none; otherwise every chunk is cut into sub-chunks of this many pairs whose kernel chains
 *                        alternate between `streams` (1 or 2) streams -- the prepare / c_conv2 / c_conv3 kernels of one sub-chunk
 *                        then run beside the contraction kernel of the next;
 *   yaw_on_side_stream   ovn_heads_spectral runs its yaw kernel on a side stream.
 * Defaults: 1024, 0, 1, 0 -- the serial order.  Measured on MI355X (profiles/r3a_pipeline_matrix.md): the contraction kernel
 * occupies every CU completely (256 registers x 8 waves, 126 KB of LDS), so kernels of a second stream only run in the gaps
 * between its workgroups and none of the forked forms is faster than the serial one; the knobs remain for sweeps whose
 * kernels leave room.  Results do not depend on any of these (every pair is computed by the same kernels with per-pair scales).
 * (For experiments the environment variables OVN_HEAD_CHUNK, OVN_HEAD_SUBCHUNK, OVN_HEAD_STREAMS and OVN_YAW_SIDE preset the four
 * values when a context is created; tools/experiments/r3_pipeline_matrix.sh.) */
int ovn_set_head_pipeline(ovn_ctx* ctx, int64_t chunk_pairs, int64_t sub_chunk_pairs, int streams, int yaw_on_side_stream);
/* The current settings (any output pointer may be NULL). */
int ovn_get_head_pipeline(ovn_ctx* ctx, int64_t* chunk_pairs, int64_t* sub_chunk_pairs, int* streams, int* yaw_on_side_stream);

/* Loop-closure decision of a 1-vs-N sweep, on the device (demo/demo3_lcd.py:117-120:
 * `if np.max(overlaps) > overlap_thres: return reference_idx[np.argmax(overlaps)]`; first maximum wins, NaN never wins).
 *   overlap_dev (n) f32, yaw_dev (n) i32 or NULL: outputs of ovn_heads / ovn_delta_head (+ ovn_corr_head_spectral)
 *   ids_dev     (n) i32 candidate ids (reference_idx) or NULL -> position + index_offset (the shard's first candidate)
 *   out_dev     4 x int32, 16-byte aligned: { id of the best candidate (-1 when n == 0), float bits of its overlap, its yaw,
 *                            1 if overlap > threshold else 0 } -- written with ONE 16-byte store; it may be pinned host memory
 *                            (device-visible at the same address): the host then needs no device-to-host copy and may poll word 3
 * Ranks of a sharded sweep exchange these 16-byte records instead of N scores (overlapnet_amd/distributed.py). */
int ovn_best_match(ovn_ctx* ctx, const float* overlap_dev, const int32_t* yaw_dev, const int32_t* ids_dev, int64_t n,
                   float threshold, int64_t index_offset, int32_t* out_dev, void* stream);

/* Spherical projection + normals for a batch of scans (src/utils/utils.py:59-134 range_projection and
 * :137-186 gen_normal_map; the drivers gen_depth_data.py:24-46 etc. loop over files and call these).
 *   points_dev   concatenated (x,y,z,intensity) float32 points of all scans
 *   offsets_dev  (n_scans+1) int64 point offsets into points_dev (scan s = [offsets[s], offsets[s+1]))
 * Outputs, each may be NULL: range (n,H,W), vertex (n,H,W,4), intensity (n,H,W), idx (n,H,W) int32
 * (index among the points that pass the range filter, utils.py:117-118), normal (n,H,W,3), and
 * stacked (n,H,W,C) = the leg input assembled in the reference's channel order depth|normals|intensity
 * (ImagePairOverlapOrientationSequence.py:143-207) according to the use_* flags.  Empty pixels = -1.
 * max_points_per_scan bounds the per-scan launch (>= the largest scan). */
int ovn_project(ovn_ctx* ctx, const float* points_dev, const int64_t* offsets_dev, int n_scans,
                int64_t max_points_per_scan, int proj_h, int proj_w, double fov_up_deg, double fov_down_deg,
                double max_range, float* range_dev, float* vertex_dev, float* intensity_dev,
                int32_t* idx_dev, float* normal_dev, float* stacked_dev, int use_depth, int use_normals,
                int use_intensity, void* stream);

/* Normal map alone from given range (n,H,W) and vertex (n,H,W,4) images -> normal (n,H,W,3)
 * (src/utils/utils.py:137-186 gen_normal_map). */
int ovn_normals(ovn_ctx* ctx, const float* range_dev, const float* vertex_dev, int n_scans, int proj_h, int proj_w,
                float* normal_dev, void* stream);

/* The per-point part of range_projection alone (src/utils/utils.py:75-104), exactly as ovn_project's scatter kernel evaluates it:
 * for each of n float32 points (x,y,z,*) yaw = -np.arctan2(y, x), pitch = np.arcsin(z / depth) with NumPy's float32 results
 * (csrc/svml_f32.h) and the pixel py * proj_w + px the point falls into (-1 for a point the range filter drops).  Outputs may
 * be NULL.  A validation entry: lets a test compare the two angle functions with the reference's NumPy on millions of points. */
int ovn_projection_angles(ovn_ctx* ctx, const float* points_dev, int64_t n_points, int proj_h, int proj_w, double fov_up_deg,
                          double fov_down_deg, double max_range, float* yaw_dev, float* pitch_dev, int32_t* pixel_dev,
                          void* stream);

/* Ground-truth overlap labels (src/utils/com_overlap_yaw.py:28-46), the producer of the training / evaluation targets.
 * ovn_gt_range_images: range images (n, H, W) f32 (-1 = empty) of n scans moved by p' = inv_cur_pose . (ref_pose[s] . p)
 *   and range-projected in FLOAT64 as `range_projection` does for load_vertex's float64 points (utils.py:59-134,217-230).
 *   points/offsets as in ovn_project; ref_poses_dev (n,4,4) f64 row-major or NULL (identity); inv_cur_pose_dev (4,4) f64 or
 *   NULL (identity) -- with both NULL this is the current frame's own range image (com_overlap_yaw.py:30-31).
 * ovn_gt_overlap_counts: counts_dev[s] = #{ref_range[s] > 0 and |ref_range[s] - cur_range| < 1} for s < n, and
 *   counts_dev[n] = #{cur_range > 0} (`valid_num`, :32-33): overlap[s] = counts[s] / counts[n] (:42-45). */
int ovn_gt_range_images(ovn_ctx* ctx, const float* points_dev, const int64_t* offsets_dev, int n_scans,
                        int64_t max_points_per_scan, const double* ref_poses_dev, const double* inv_cur_pose_dev, int proj_h,
                        int proj_w, double fov_up_deg, double fov_down_deg, double max_range, float* range_dev, void* stream);
int ovn_gt_overlap_counts(ovn_ctx* ctx, const float* ref_ranges_dev, const float* cur_range_dev, int n_scans, int proj_h,
                          int proj_w, int32_t* counts_dev, void* stream);

/* Arithmetic of the Delta head's contractions (c_conv1, c_conv2, c_conv3) and of ovn_spectrum's DFT; storage and accumulation
 * are fp32 either way:
 *   0 = fp32 matrix cores (v_mfma_f32_16x16x4_f32; bit-for-bit an fp32 FMA chain),
 *   1 = scaled 3-term fp16 split on the fp16 matrix cores (x * 2^k = hi + lo, a*w ~ ah*wh + al*wh + ah*wl; 2^-21 per operand)
 *       -- the default, measured as accurate as mode 0; both modes are held to |d overlap| <= 1e-4 against the fp64 oracle on
 *       every pair of the benchmark sweep by the parity tests. */
int ovn_set_head_precision(ovn_ctx* ctx, int mode);

/* Arithmetic of the leg convolutions, same two modes as ovn_set_head_precision (default 1). */
int ovn_set_leg_precision(ovn_ctx* ctx, int mode);

/* Dead-channel compaction of the Delta head's contraction in 1-vs-N sweeps (default 1).  DeltaLayer + c_conv1
 * (generateNet.py:45-59,96-100) sum |l - r| w over the 128 feature channels; in the min form of the f16x3 path a channel that is 0
 * in all 360 columns of the QUERY (ReLU outputs: a quarter of the channels under the benchmark's weights) contributes exact zeros
 * for every candidate, so the K walk covers only ceil(live / 32) slices of 32 channels.  Exact; changes only how K is grouped into
 * MFMA steps (last-bit differences against on = 0).  Pairs with negative values and indexed pairs always walk all 128 channels. */
int ovn_set_head_compaction(ovn_ctx* ctx, int on);

/* Which float32 `np.arctan2` / `np.arcsin` (src/utils/utils.py:86-87) ovn_project / ovn_projection_angles reproduce:
 *   0 (default)  NumPy >= 1.22 on an AVX512_SKX x86-64 host: Intel SVML's 1-4 ulp kernels, bit for bit (csrc/svml_f32.h) -- the
 *                machine the reference's shipped .npy files and this repo's golden vectors were produced on;
 *   1            the correctly rounded float32 results (float64 function, rounded once): what NumPy gives where its float32 loops
 *                call a correctly rounded libm (AVX2-only x86, aarch64 ...).  The two differ in the last bit of 38 % of the angles
 *                and put a point into the neighbouring pixel about once per 200 k points. */
int ovn_set_projection_trig(ovn_ctx* ctx, int mode);

/* Per-kernel-class timing with HIP events recorded on the launch stream, for bench.py's roofline line.
 * Between begin and end every kernel group launched through this context is bracketed by an event
 * pair; ovn_profile_end waits for them and returns, per class, the summed milliseconds and the number
 * of bracketed launches.  Classes: 0 leg convolutions (one entry per layer launch), 1 correlation head,
 * 2 Delta kernel (DeltaLayer+c_conv1 contraction; fp32 mode: fused with c_conv2), 3 c_conv3, 4 dense+sigmoid, 5 projection,
 * 6 spectrum (DFT), 7 spectral correlation head, 8 Delta prepare kernels, 9 c_conv2 GEMM of the f16x3 Delta path. Arrays of 10. */
int ovn_profile_begin(ovn_ctx* ctx);
int ovn_profile_end(ovn_ctx* ctx, double* ms_by_kind, int64_t* launches_by_kind);

/* Test hook: run registered leg layer `layer` alone on in_dev (nb,h,w,cin) -> out_dev (nb,oh,ow,cout). */
int ovn_debug_conv(ovn_ctx* ctx, int layer, const float* in_dev, int nb, int h, int w, float* out_dev, void* stream);

/* Test hook: copy the c_conv2 (n,24,24,128) and c_conv3 (n,22,22,256) activations that the most recent
 * ovn_heads call left in scratch (its first chunk / sub-chunk, n <= min(pairs, chunk_pairs, sub_chunk_pairs)); either output may be NULL.
 * (generateNet.py:102-110 intermediates; the reference exposes them as Keras layer outputs.) */
int ovn_debug_head_activations(ovn_ctx* ctx, int64_t n, float* o2_dev, float* o3_dev, void* stream);

/* Measurement hook: the K walk of the Delta head's contraction in the most recent 1-vs-N sweep on this context (its last chunk).
 * out16_host (HOST memory, 16 x int32): [0] slices of 32 channels a pass of the contraction kernel walks (1..4: the largest count
 * below), [1] channels of the query that are non-zero somewhere in its 360 columns, [2..13] slices the live channels of column-group
 * pairs 0..11 need (a pair = two column groups of c_conv1, generateNet.py:96-100; a wave skips a slice none of its column groups
 * walks), [14] 1 if the dead-channel compaction applied (ovn_set_head_compaction), else 0 and every entry says 4 slices / 128
 * channels, [15] MFMA steps of the LAST slice when it is packed tap-major (it then holds <= 16 live channels: 3, 6 or 9 steps instead
 * of 15; 0 = not packed).  Synchronises `stream`.  bench.py reports roofline.k_walk_frac from it. */
int ovn_head_walk_stats(ovn_ctx* ctx, int32_t* out16_host, void* stream);

/* Device scratch currently held by the context, in bytes (grows on demand, freed by ovn_destroy). */
int64_t ovn_workspace_bytes(ovn_ctx* ctx);

/* ---- optional collective for consumers that do not use torch.distributed --------------------------
 * The 1-vs-N sweep shards the candidates over the ranks in contiguous blocks (no data-path collective); its one exchange
 * step is the gather of the (overlap, yaw) results, 8 bytes per candidate.  RCCL is loaded with dlopen on first use (no
 * link-time dependency; an RCCL already in the process, e.g. PyTorch's, is reused).  The Python package does not call these
 * (it goes through torch.distributed, backend "nccl" = RCCL).
 *   rank 0: ovn_comm_unique_id(id); the application hands the OVN_COMM_ID_BYTES bytes to the other ranks (MPI, file, socket);
 *   every rank: ovn_comm_init(ctx, rank, world_size, id)  -- collective, one communicator per context, on the context's GPU;
 *   per query:  ovn_gather_scores(...) -- collective; rank r contributes counts_host[r] candidates (its shard of the sweep, host
 *               array of world_size entries, identical on every rank); on `root` overlap_all_dev / yaw_all_dev (sum of counts)
 *               receive the shards in rank order; other ranks may pass NULL result buffers.  Enqueued on `stream`. */
#define OVN_COMM_ID_BYTES 128
int ovn_comm_unique_id(unsigned char* id_out);
int ovn_comm_init(ovn_ctx* ctx, int rank, int world_size, const unsigned char* id);
int ovn_comm_destroy(ovn_ctx* ctx);
int ovn_gather_scores(ovn_ctx* ctx, const float* overlap_dev, const int32_t* yaw_dev, const int64_t* counts_host, int root,
                      float* overlap_all_dev, int32_t* yaw_all_dev, void* stream);

/* Self-test of the MFMA fragment layouts this library relies on (runs a few tiny kernels on the
 * context's device and compares with a host matmul).  0 = all layouts as assumed. */
int ovn_selftest(ovn_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* OVN_HIP_H */
