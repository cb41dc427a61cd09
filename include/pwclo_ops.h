/*
 * pwclo_ops.h -- C ABI of libpwclo_hip.so, the MI355X (gfx950) implementation of the
 * PWCLO-Net point-cloud operator path.
 *
 * Drop-in boundary.  The reference's extension is split into C++ host files that check and
 * allocate torch tensors and then call plain C++ launchers taking raw device pointers and
 * 32-bit sizes.  Those nine launchers are the FFI of this path; section 1 exports them with the
 * reference's exact names, argument order and meaning (extern "C"; the reference declares them
 * at the cited lines inside its .cpp files, so relinking its host files against this library
 * needs only `extern "C"` around those declarations -- see INTEGRATION.md).
 *
 * All pointers are device pointers on the current HIP device.  Tensors are dense, row-major
 * ("contiguous" in the reference's CHECK_CONTIGUOUS sense), float = IEEE binary32, int = int32.
 * Calls enqueue on the stream set by pwclo_set_stream() for the calling host thread (default:
 * the null stream) and return without synchronising -- the reference launches on ATen's current
 * stream the same way (e.g. group_points_gpu.cu:34).  No entry point allocates, frees or
 * synchronises, so every call may be captured into a hipGraph.
 *
 * Errors.  The reference prints and calls exit(-1) when a launch fails (cuda_utils.h:30-39).
 * This library never exits the process: it prints one line to stderr, records the failure in a
 * per-thread sticky status readable through pwclo_last_error(), and skips the launch.  Host
 * bindings must check pwclo_last_error() after a call and raise (ours do).
 *
 * Reference paths below are relative to
 *   /root/reference/slam/models/Pointnet2_PyTorch/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/
 */
#ifndef PWCLO_OPS_H
#define PWCLO_OPS_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- 0. library state ------------------------------------------------------------------- */

/* Version of this ABI (bumped on any signature change). */
int pwclo_abi_version(void);

/* Set / get the hipStream_t (passed as void*) used by the calling host thread's launches.
 * Replaces at::cuda::getCurrentCUDAStream() inside the reference launchers. */
void pwclo_set_stream(void *hip_stream);
void *pwclo_get_stream(void);

/* 0 = no error since the last pwclo_clear_error() on this thread, else a hipError_t value or
 * one of the PWCLO_E* codes.  pwclo_last_error_message() returns a static thread-local string. */
int pwclo_last_error(void);
const char *pwclo_last_error_message(void);
void pwclo_clear_error(void);

/* Developer tool: per-workgroup trace of the kernels on the fused forward path.  records: device buffer of
 * `capacity` 32-byte records {u64 t0, t1 (100 MHz constant clock); u32 kernel id, block, blocks in grid, hw id};
 * count: device u32 the kernels bump (zero it first).  records == NULL switches the trace off (the default).
 * Synchronous (writes device symbols); not for use under graph capture. */
void pwclo_trace_enable(void *records, void *count, unsigned capacity);

/* How furthest_point_sampling launches its multi-workgroup kernel for clouds too large for one workgroup (n > 24576):
 * cooperative != 0 (default; PWCLO_FPS_COOP_LAUNCH): hipLaunchCooperativeKernel -- co-residency guaranteed by the
 * runtime, but the device has ONE cooperative queue: such launches on different streams run one after the other;
 * cooperative == 0: plain launch, for a caller that keeps two large-cloud batches in flight and bounds the workgroups in
 * flight itself (<= 256 of this kernel).  Either way a workgroup that waits for its peers beyond the spin bound reports
 * PWCLO_ECOOP_TIMEOUT through pwclo_last_error(); the indices are never silently incomplete.  Process-wide. */
void pwclo_fps_large_cloud_launch(int cooperative);
/* The same kernel's cross-workgroup exchange.  xcd_local != 0 (default; PWCLO_FPS_COOP_XCD_LOCAL): the workgroups of a
 * cloud are drawn from blocks with equal index mod 8, which the dispatcher deals to one XCD; every launch CHECKS that
 * (the workgroups exchange their XCC ids first) and only then posts with plain stores that stay in that XCD's L2, where
 * the peers' L1-bypassing polls find them (2.3 instead of 2.6 us per sample).  If the ids differ, or with xcd_local == 0,
 * the posts are agent-scope stores as before.  Same indices either way.  Process-wide. */
void pwclo_fps_large_cloud_exchange(int xcd_local);

#define PWCLO_EINVAL 10001  /* argument outside what the kernels support (message says which) */
/* Reported by a RUNNING kernel (not at launch): the cooperative large-cloud sampler gave up waiting for a
 * peer workgroup.  Kernels post such codes into a pinned word the library owns; pwclo_last_error() reads it
 * without a HIP call, so it becomes visible once the kernel has run: at the next library call, or when the
 * host calls pwclo_last_error() after synchronising the stream.  The outputs of that call are invalid. */
#define PWCLO_ECOOP_TIMEOUT 10002

/* ---- 1. the nine reference launchers ------------------------------------------------------ */

/* sampling.cpp:4-6 / sampling_gpu.cu:22-30.  out[b,c,j] = points[b,c,idx[b,j]].
 * points (b,c,n) f32, idx (b,npoints) i32, out (b,c,npoints) f32. */
void gather_points_kernel_wrapper(int b, int c, int n, int npoints, const float *points,
                                  const int *idx, float *out);

/* sampling.cpp:7-9 / sampling_gpu.cu:49-57.  grad_points[b,c,idx[b,j]] += grad_out[b,c,j];
 * grad_points (b,c,n) must be zero-filled by the caller (sampling.cpp:51-53). */
void gather_points_grad_kernel_wrapper(int b, int c, int n, int npoints, const float *grad_out,
                                       const int *idx, float *grad_points);

/* sampling.cpp:11-13 / sampling_gpu.cu:175-229.  Iterative furthest point sampling.
 * dataset (b,n,3) f32; temp (b,n) f32 scratch pre-filled with 1e10 by the caller
 * (sampling.cpp:74-76).  This implementation keeps the running distances in registers: temp may be
 * NULL for n <= 24576; for larger clouds it must be the (b,n) buffer (8-byte aligned) and is used --
 * and overwritten -- as the exchange workspace of the cooperative multi-workgroup sampler (or, with
 * PWCLO_FPS_COOP=0, as the running distances of the streaming fallback).  idxs (b,m) i32.  Index 0 is always the
 * first sample; points with x*x+y*y+z*z <= 1e-3 are never selected; ties are resolved exactly
 * as the reference's block-of-opt_n_threads(n) tree reduction does ("FPS tie rule"). */
void furthest_point_sampling_kernel_wrapper(int b, int n, int m, const float *dataset,
                                            float *temp, int *idxs);

/* group_points.cpp:4-6 / group_points_gpu.cu:30-39.  out[b,c,j,k] = points[b,c,idx[b,j,k]].
 * points (b,c,n), idx (b,npoints,nsample) i32, out (b,c,npoints,nsample). */
void group_points_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                 const float *points, const int *idx, float *out);

/* group_points.cpp:8-10 / group_points_gpu.cu:66-75.  Scatter-add of grad_out (b,c,npoints,
 * nsample) into zero-filled grad_points (b,c,n). */
void group_points_grad_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                      const float *grad_out, const int *idx, float *grad_points);

/* ball_query.cpp:4-6 / ball_query_gpu.cu:46-54.  For each centre new_xyz[b,j] the first
 * `nsample` indices k (ascending) with |new_xyz-xyz[k]|^2 < radius^2; unfilled slots repeat the
 * first hit; a centre without any hit leaves its slots untouched (caller zero-fills idx,
 * ball_query.cpp:19-21).  new_xyz (b,m,3), xyz (b,n,3), idx (b,m,nsample) i32. */
void query_ball_point_kernel_wrapper(int b, int n, int m, float radius, int nsample,
                                     const float *new_xyz, const float *xyz, int *idx);

/* interpolate.cpp:4-5 / interpolate_gpu.cu:61-68.  Three nearest `known` points of every
 * `unknown` point: dist2 (b,n,3) squared distances ascending, idx (b,n,3); ties -> lower index;
 * m < 3 leaves dist2 = +inf, idx = 0 in the unfilled slots. */
void three_nn_kernel_wrapper(int b, int n, int m, const float *unknown, const float *known,
                             float *dist2, int *idx);

/* interpolate.cpp:6-8 / interpolate_gpu.cu:103-111.
 * out[b,c,j] = sum_t points[b,c,idx[b,j,t]] * weight[b,j,t].  points (b,c,m), out (b,c,n). */
void three_interpolate_kernel_wrapper(int b, int c, int m, int n, const float *points,
                                      const int *idx, const float *weight, float *out);

/* interpolate.cpp:9-12 / interpolate_gpu.cu:145-154.  Scatter-add of grad_out (b,c,n) * weight
 * into zero-filled grad_points (b,c,m). */
void three_interpolate_grad_kernel_wrapper(int b, int c, int n, int m, const float *grad_out,
                                           const int *idx, const float *weight,
                                           float *grad_points);

/* ---- 2. native replacements of pure-PyTorch ops on the same path -------------------------- */

/* Replaces pytorch_utils.py:12-49 (knn_point = dense distance + torch.topk).  For every query
 * new_xyz[b,j] the `nsample` nearest points of xyz[b] in ascending key order, key =
 * sqrtf(((dx*dx + dy*dy) + dz*dz) + 1e-8f) with dx = query - candidate, every operation rounded
 * to binary32 (no contraction); equal keys -> lower index first.  xyz (b,n,3), new_xyz (b,s,3),
 * idx (b,s,nsample) i32, dist (b,s,nsample) f32 or NULL.  Requires 1 <= nsample <= 64 and
 * nsample <= n (else PWCLO_EINVAL). */
void knn_point_kernel_wrapper(int b, int n, int s, int nsample, const float *xyz,
                              const float *new_xyz, int *idx, float *dist);

/* Same result through an exact spatially pruned search (Morton-sorted candidate blocks with
 * bounding boxes) when 256 <= n <= 16384: needs a caller-provided device workspace of
 * knn_point_workspace_bytes(b, n) bytes (0 = not applicable: the exhaustive kernel is used and
 * `workspace` may be NULL; also used when s < 256, where the build does not amortise).
 * Bit-identical output to knn_point_kernel_wrapper. */
long long knn_point_workspace_bytes(int b, int n);
/* The two passes of the pruned search separately, so that ONE build of a cloud's search structure serves several
 * searches and the slab-pruned sampler below: workspace of knn_point_build_bytes(b, n) bytes (64 <= n <= 16384);
 * slab_tab (optional, b*32 ints): (padded first row, row count) of the cloud's knn_point_slabs(n) <= 16 x-slabs. */
long long knn_point_build_bytes(int b, int n);
int knn_point_slabs(int n);
void knn_build_kernel_wrapper(int b, int n, const float *xyz, void *workspace, int *slab_tab);
void knn_point_prebuilt_kernel_wrapper(int b, int n, int s, int nsample, const float *new_xyz, int *idx, float *dist,
                                       void *workspace);
/* The same search on clouds [first_cloud, first_cloud + b) of a structure that knn_build / knn_point_ws built for
 * built_b >= first_cloud + b clouds of n points (the pyramid builds both frames' clouds as one batch; a refinement level
 * searches one frame's half of it without rebuilding).  new_xyz (b,s,3), idx (b,s,nsample). */
void knn_point_prebuilt_slice_kernel_wrapper(int b, int n, int s, int nsample, const float *new_xyz, int *idx,
                                             float *dist, void *workspace, int first_cloud, int built_b);
/* The spatial order the sorted sampler below wants, built on the device: points (b,n,3) -> sorted (b,n,3), perm (b,n);
 * workspace of fps_spatial_order_workspace_bytes(b, n) bytes.  (32^3 Morton cells by counting sort, then every block of
 * 1024 consecutive positions sorted by sampling priority; any such order is exact for the sampler.) */
long long fps_spatial_order_workspace_bytes(int b, int n);
void fps_spatial_order_kernel_wrapper(int b, int n, const float *points, float *sorted, int *perm, void *workspace);
/* Large clouds (n > 24576) presented in a spatially coherent order: sorted (b,n,3) = dataset gathered by perm (b,n),
 * perm[p] = original index of sorted position p; inside every block of 1024 consecutive positions the positions are
 * ordered by ascending sampling priority (bitrev(k mod bs) << 23 | k div bs of the original index k).  A wave of the
 * cooperative sampler then owns a compact cell and skips its distance update -- exactly -- whenever the new sample is
 * farther from the cell's box than the largest running distance.  Same idxs (ORIGINAL numbering) / new_xyz as
 * furthest_point_sampling_xyz_kernel_wrapper, bit for bit.  temp: the (b,n) scratch, 8-byte aligned. */
void furthest_point_sampling_sorted_kernel_wrapper(int b, int n, int m, const float *dataset, const float *sorted,
                                                   const int *perm, float *temp, int *idxs, float *new_xyz);
/* furthest_point_sampling_chain_kernel_wrapper for a cloud whose search structure exists (knn_point_slabs(n) == 8,
 * n >= 4096: level 1 of the pyramid): the distance update of a wave is skipped, exactly, whenever the new sample
 * cannot lower any running distance inside the wave's x-slab (csrc/sampling.hip: fps_slab_kernel).  status: b ints
 * (device) the two kernels use to divide the clouds between them.  Same idxs / new_xyz / tie_out as the chain entry. */
void furthest_point_sampling_slab_kernel_wrapper(int b, int n, int m, const float *dataset, int *idxs,
                                                 float *new_xyz, int *tie_out, int tie_iters,
                                                 const void *knn_workspace, const int *slab_tab, int *status);
void knn_point_ws_kernel_wrapper(int b, int n, int s, int nsample, const float *xyz,
                                 const float *new_xyz, int *idx, float *dist, void *workspace);

/* Replaces PWCLO_utils.py:42-63 (warp): out = q (x) (0,xyz) (x) q^-1 + t, scalar-first
 * quaternions, q^-1 = conj(q) / (|q|^2 + 1e-10).  xyz, out (b,3,n); q (b,4); t (b,3). */
void quat_warp_kernel_wrapper(int b, int n, const float *xyz, const float *q, const float *t,
                              float *out);

/* ---- 3. fused eval-mode layers ------------------------------------------------------------- */
/* One launch per reference module forward (eval mode: BatchNorm running statistics folded into
 * the packed weights, no dropout).  Feature tensors are POINT-MAJOR (b, n, c) fp32 with c a
 * multiple of 16; `packed_w` is produced by pwclonet_pylidarslam_amd/fused.py (layout in
 * csrc/mlp_core.hpp).  Unsupported channel configurations record PWCLO_EINVAL. */

/* PointnetSAModulePWCLONet.forward after FPS/knn (pointnet2_modules.py:205-243): gather the k
 * neighbours idx[b,s,:] of every query, subtract the query centre, run the 3-layer shared MLP
 * (padded widths c1,c2,c3) and max over the neighbours.  xyz (b,n,3), new_xyz (b,s,3), feat
 * (b,n,c_feat) or NULL when c_feat == 0 (level-0 input [xyz_diff, grouped_xyz]), idx (b,s,k)
 * i32, out (b,s,c3). */
void sa_fused_kernel_wrapper(int b, int n, int s, int k, int c_feat, int c1, int c2, int c3,
                             const float *xyz, const float *new_xyz, const float *feat,
                             const int *idx, const float *packed_w, float *out);

/* furthest_point_sampling_kernel_wrapper that also writes the sampled coordinates new_xyz (b,m,3)
 * = dataset[b, idxs[b,j], :] (the gather_operation that always follows FPS in the set-abstraction
 * layer, pointnet2_modules.py:196-203); new_xyz may be NULL. */
void furthest_point_sampling_xyz_kernel_wrapper(int b, int n, int m, const float *dataset,
                                                float *temp, int *idxs, float *new_xyz);

/* The same for a CHAIN of samplers (the pyramid: each level samples the previous level's samples).
 * Sampling a cloud that is itself an FPS sample list, in sampling order, returns its prefix 0..m-1
 * whenever every arg-max of the producing call was unique; an exact two-point distance tie that the
 * producer resolved over two consecutive iterations shows up as an adjacent pair in an order decided by
 * the child's position priorities (csrc/sampling.hip, "Sampling chains").
 *   tie_out  (b, 12) i32 or NULL: per cloud [fallback flag, number of tie events, 8 event iterations, -, -]
 *            for the first tie_iters - 1 decisions of THIS call (needs m >= tie_iters + 2, else flag = 1);
 *   prefix_in (b, 12) i32 or NULL: records written by the call that produced `dataset`; clouds whose
 *            flag is 0 get their result written directly (m must be <= that call's tie_iters), all
 *            others run the full algorithm.  Outputs are identical either way. */
void furthest_point_sampling_chain_kernel_wrapper(int b, int n, int m, const float *dataset, float *temp,
                                                  int *idxs, float *new_xyz, int *tie_out, int tie_iters,
                                                  const int *prefix_in);

/* Input adapter (pwclo_net.py:125-126 and the siamese batching): xyz_f1, xyz_f2 (b,3,n) channel-major
 * -> out (2b,n,3) point-major, frame 1 first. */
void ingest_pairs_kernel_wrapper(int b, int n, const float *xyz_f1, const float *xyz_f2, float *out);

/* The same batch straight from the prediction module's inputs (slam/training/prediction_modules.py:
 * 144-160: `pcd[:, :num_points, :3]` then permute): frame1/frame2 (b, n_total, c) point-major with
 * c >= 3 floats per point; the first n points and first 3 channels of each -> out (2b, n, 3). */
void ingest_frames_kernel_wrapper(int b, int n, int n_total, int c, const float *frame1,
                                  const float *frame2, float *out);

/* Hamilton product of quaternion rows (PW/PWCLO_utils.py:83-95 mul_q_point, :117-129 mul_point_q -- the same
 * component expressions): out (b,4,n) = a (b,4,na) (x) q (b,4,nb), na and nb each 1 (broadcast) or n; conj_a / conj_b
 * != 0 use that operand's conjugate.  Products rounded before the left-to-right sums: the torch expression bit for
 * bit.  The product's gradients are products with conjugates, so the training path's backward calls this again. */
void hamilton_product_kernel_wrapper(int b, int n, int na, int nb, int conj_a, int conj_b, const float *a,
                                     const float *q, float *out);

/* quat_warp_kernel_wrapper on point-major clouds: xyz, out (b,n,3). */
void quat_warp_pm_kernel_wrapper(int b, int n, const float *xyz, const float *q, const float *t,
                                 float *out);

/* PointnetFPModulePWCLONet.forward, knn branch up to the max over neighbours
 * (pointnet2_modules.py:479-506): gather feat1 (b,n,64) and xyz1 (b,n,3) at idx (b,s,k<=8), append
 * xyz1[nbr]-xyz2[s], shared MLP 67->128->64, max over k.  out (b,s,64). */
void upconv_fused_kernel_wrapper(int b, int n, int s, int k, const float *xyz2, const float *xyz1,
                                 const float *feat1, const int *idx, const float *packed_w,
                                 float *out);

/* Shared MLP (1 or 2 layers, padded widths w1,w2; w2 = 0 for one layer) over the channel
 * concatenation of up to three per-point tensors src_i (b,s,c_i): the set-upconv post_mlp
 * (pointnet2_modules.py:508-515) and FlowPredictor.forward (PW/flowpredictor.py:53-83). */
void pointwise_fused_kernel_wrapper(int b, int s, int c0, int c1, int c2, int w1, int w2,
                                    const float *src0, const float *src1, const float *src2,
                                    const float *packed_w, float *out);

/* The same two-layer stack followed by ONE linear layer (w2 -> wt channels, bias, no activation) applied to the stack's
 * output while it is in registers and written to out_tail (b,s,wt): the hoisted partial product W1_feat . out + b1 that the
 * next refinement level's set-upconv (pointnet2_modules.py:479-506, first layer of its mlp) would otherwise request from
 * linear_jobs_kernel_wrapper (section 4); same rows bit for bit.  packed_tail holds tail_floats floats. */
void pointwise_tail_fused_kernel_wrapper(int b, int s, int c0, int c1, int c2, int w1, int w2, int wt,
                                         const float *src0, const float *src1, const float *src2,
                                         const float *packed_w, const float *packed_tail, float *out,
                                         float *out_tail, int tail_floats);

/* CostVolume.forward (PW/costvolume.py:63-190) in three launches.
 * a1: per (query s, neighbour k) pixel: mlp_convs([geometry10 | feat1[s] | feat2[idx]]) -> pix
 *     (b, s*pix_slots, 64).  pix_slots is the CALLER's choice (it allocates pix) and must be passed identically
 *     to a1 and a2: k rounded up to 8/16/32, or 6 for k == 6 (dense layout of the refinement levels).
 *     xyz1 (b,s,3), feat1 (b,s,c), xyz2 (b,n,3), feat2 (b,n,c), idx (b,s,k).
 * Packed-weight format (entry points that take `wfmt, packed_floats`): 0 = fp32 operand tiles
 *     (v_mfma_f32_16x16x4_f32), 1 = the opt-in three-term bf16 split tiles, 2 = bf16 tiles (dtype "bf16":
 *     v_mfma_f32_16x16x32_bf16 on operands rounded once, fp32 accumulate; with it the hoisted rows pre / u / v / u2 / v2
 *     and the per-pixel buffer pix are bf16 IN MEMORY too: 2 bytes per channel; csrc/mlp_core.hpp); the format is a
 *     property of the buffer, fixed when it was packed.  packed_floats = its length; a length that does not match
 *     the layout the selected kernel indexes is refused with PWCLO_EINVAL (never read out of bounds).
 * a2: mlp_conv_xyz_1(geometry10), mlp2_convs, softmax over k, sum_k w*pix -> out (b,s,64).
 * b : second aggregate over the k<=4 frame-1 neighbours idx (b,s,k) of each frame-1 point:
 *     mlp_conv_xyz_2, mlp3_convs([enc | feat1[s] | first[idx]]), softmax, sum_k w*first[idx]. */
void cv_fused_a1_kernel_wrapper(int b, int n, int s, int k, int c, const float *xyz1,
                                const float *feat1, const float *xyz2, const float *feat2,
                                const int *idx, const float *packed_w, float *pix, int pix_slots);
void cv_fused_a2_kernel_wrapper(int b, int n, int s, int k, const float *xyz1, const float *xyz2,
                                const int *idx, const float *packed_w, const float *pix, float *out,
                                int pix_slots, int wfmt, int packed_floats);
void cv_fused_b_kernel_wrapper(int b, int s, int k, int c, const float *xyz1, const float *feat1,
                               const float *first, const int *idx, const float *packed_w, float *out);

/* out[b,c] = sum_n emb[b,n,c] * softmax_n(mask[b,n,c]) for 64-channel point-major emb/mask
 * (b,n,64): F.softmax(mask, dim=2) + the masked sum of PoseCalculator.forward
 * (PW/pose_calculator.py:58). */
void masked_pool_kernel_wrapper(int b, int n, const float *emb, const float *mask, float *out);

/* Whole pose head of one pyramid level in one launch: the masked pooling above, PoseCalculator's
 * three 1x1 convolutions (w_qt (256,64), w_q (4,256), w_t (3,256) + biases; eval mode, dropout =
 * identity; PW/pose_calculator.py:47-86), and -- when q_prev (b,4) / t_prev (b,3) are given --
 * the pose composition of PoseWarpRefinement (PW/pose_warp_refinement.py:139,148).  Writes q_out
 * (b,4), t_out (b,3) and this level's row [t, q/|q|] at pose_row + i*row_stride (pwclo_net.py:195-205). */
void pose_head_fused_kernel_wrapper(int b, int n, const float *emb, const float *mask,
                                    const float *w_qt, const float *b_qt, const float *w_q,
                                    const float *b_q, const float *w_t, const float *b_t,
                                    const float *q_prev, const float *t_prev, float *q_out,
                                    float *t_out, float *pose_row, int row_stride);

/* The same, followed in the same launch by the next (finer) level's quat_warp_pm(warp_src (b,warp_n,3), q_out, t_out) ->
 * warp_out: bit-identical to the two separate launches (PW/pose_warp_refinement.py:104-106 is the first thing the next
 * level does with this pose). */
void pose_head_warp_fused_kernel_wrapper(int b, int n, const float *emb, const float *mask,
                                         const float *w_qt, const float *b_qt, const float *w_q,
                                         const float *b_q, const float *w_t, const float *b_t,
                                         const float *q_prev, const float *t_prev, float *q_out,
                                         float *t_out, float *pose_row, int row_stride, int warp_n,
                                         const float *warp_src, float *warp_out);

/* Deterministic (atomics-free) form of group_points_grad / gather_points_grad (nsample = 1): the caller
 * supplies the inverse of idx per cloud -- perm (b, npoints*nsample) i32 = positions p sorted by idx[b,p]
 * (stable: ascending p inside a source point), seg (b, n+1) i32 = segment starts -- and every
 * grad_points[b,c,i] is the sum of grad_out[b,c,perm[seg[i]..seg[i+1])] in that order.  No zero fill. */
void group_points_grad_sorted_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                             const float *grad_out, const int *perm, const int *seg,
                                             float *grad_points);

/* Channel-slice forms of the three grouping entry points.  The reference concatenates every grouped tensor with the
 * coordinate differences / the other frame's features before the shared MLP (P2/pointnet2_modules.py:222-230, 490-500;
 * PW/costvolume.py:107, 134, 172), i.e. it writes the grouped tensor, then copies it into the concatenated one, and in
 * backward copies the slice of the gradient back out.  Here `out` / `grad_out` point at channel `c_off` of cloud 0 of a
 * (b, c_total, npoints, nsample) tensor and `batch_stride` = c_total * npoints * nsample floats: the kernels group
 * straight into / differentiate straight out of the concatenated tensor.  Values are those of the dense entry points. */
void group_points_strided_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *points,
                                         const int *idx, float *out, long long batch_stride);
void group_points_grad_strided_kernel_wrapper(int b, int c, int n, int npoints, int nsample, const float *grad_out,
                                              long long batch_stride, const int *idx, float *grad_points);
void group_points_grad_sorted_strided_kernel_wrapper(int b, int c, int n, int npoints, int nsample,
                                                     const float *grad_out, long long batch_stride, const int *perm,
                                                     const int *seg, float *grad_points);

/* Inputs of the cost volume's shared MLPs (PW/costvolume.py:92-107 first aggregate, :155-172 second): the 10-channel
 * geometry encoding of every (centre p, neighbour q = src[idx]) pair, out[b, 0:10, j, t] = [p (3), q (3), q - p (3),
 * sqrt(sum (q - p)^2 + 1e-20)], and the centre's features tiled over the k neighbours (torch.tile in the reference),
 * written straight into their channel slice of the concatenated tensor (`out` = first channel of the slice in cloud 0,
 * batch_stride floats between clouds, as the *_strided grouping entry points).  centre_xyz (b,3,s), src_xyz (b,3,n),
 * idx (b,s,k) i32, feats (b,c,s).  Forward values are the reference's bit for bit.  Gradients: d_centre_xyz (b,3,s) is
 * written; d_src_xyz (b,3,n) must be zero-filled and receives atomic adds; either may be NULL (not wanted); d_pair
 * (b,3,s,k), instead of d_src_xyz, receives the neighbours' gradients pair by pair (for group_points_grad_sorted).
 * broadcast_centre_grad: d_feats (b,c,s) = sum over the k neighbours of the slice's gradient. */
void geometry_encode_kernel_wrapper(int b, int n, int s, int k, const float *centre_xyz, const float *src_xyz,
                                    const int *idx, float *out, long long batch_stride);
void geometry_encode_grad_kernel_wrapper(int b, int n, int s, int k, const float *centre_xyz, const float *src_xyz,
                                         const int *idx, const float *grad_out, long long batch_stride,
                                         float *d_centre_xyz, float *d_src_xyz, float *d_pair);
/* out[b, 0:3, j, t] = src_xyz[b, :, idx[b,j,t]] - centre_xyz[b, :, j] (P2/pointnet2_modules.py:215-218, 485-488), into a
 * channel slice like the entries above; its gradient is group_points_grad_strided (neighbours) and minus
 * broadcast_centre_grad (centres) of the same slice. */
void xyz_diff_kernel_wrapper(int b, int n, int s, int k, const float *centre_xyz, const float *src_xyz, const int *idx,
                             float *out, long long batch_stride);
void broadcast_centre_kernel_wrapper(int b, int c, int s, int k, const float *feats, float *out, long long batch_stride);
void broadcast_centre_grad_kernel_wrapper(int b, int c, int s, int k, const float *grad_out, long long batch_stride,
                                          float *d_feats);

/* On-device front end of the dataset (slam/dataset/kitti_odometry_dataset.py:375-397, filter_pcd
 * :149-160): points (n,4) f32 raw velodyne rows (x,y,z,intensity), tr (12) f64 DEVICE array = rows of the
 * 3x4 calibration matrix Tr; xyz (n,3) f32 = Tr . (x,y,z,1) evaluated in fp64, keep (n) i32 = 1 where the
 * transformed point is not ground (y <= 1.1) and within |x| < 30, |z| < 30. */
void kitti_transform_filter_kernel_wrapper(int n, const double *tr, const float *points, float *xyz, int *keep);

/* KITTI-360 variant (slam/dataset/kitti_360_dataset_2.py:113-123; no calibration transform): xyz (n,3) = the
 * first three columns, keep = not ground (z >= ground_z, the reference's -(1.73 - 0.3)) and |x| < near and
 * |y| < near, compared in fp32.  n may span several frames laid end to end (b*n rows). */
void kitti360_filter_kernel_wrapper(int n, float ground_z, float near, const float *points, float *xyz, int *keep);

/* Stable per-frame compaction after either filter: keep, pos (b,n) i32 with pos = inclusive prefix sum of
 * keep along each frame; kept row i of frame f is copied to out[f, pos-1] (out (b,cap,3) f32, zero-filled by
 * the caller: zero rows are never selected by furthest_point_sampling), counts (b) i32 = min(kept, cap). */
void compact_frames_kernel_wrapper(int b, int n, int cap, const int *keep, const int *pos, const float *xyz,
                                   float *out, int *counts);
/* The same compaction with the scan inside (no `pos` input): one workgroup per frame, ballot / popcount slots, stable. */
void compact_frames_scan_kernel_wrapper(int b, int n, int cap, const int *keep, const float *xyz, float *out, int *counts);

/* ---- 3b. module-path layers: training-mode BatchNorm, stack tails, pointwise convolution ---- */

/* Training-mode BatchNorm over x (b, c, l) f32 (l = product of the trailing dimensions), the statistics pass of
 * the module path's Conv -> BN -> ReLU stacks (P2/pytorch_utils.py:86-111 wraps torch.nn.BatchNorm{1,2}d; semantics
 * of torch.nn.functional.batch_norm(training=True)): y = (x - mean) * invstd * gamma + beta with the batch's
 * mean / biased variance per channel, running_* updated in place with `momentum` and the unbiased variance
 * (both may be NULL), save_mean / save_invstd (c) kept for the backward.  gamma / beta may be NULL (1 / 0).
 * workspace: batchnorm_train_workspace_bytes(c) bytes of device memory, 8-byte aligned.  y == NULL: statistics only
 * (running_*, save_*), no apply pass -- for conv1x1_bnrelu_forward below. */
long long batchnorm_train_workspace_bytes(int c);
/* relu != 0 fuses the stack's following ReLU: y = max(., 0), and the backward masks dy where that output was 0
 * (the mask is recomputed from x, nothing extra is saved). */
void batchnorm_train_forward_kernel_wrapper(int b, int c, int l, const float *x, const float *gamma,
                                            const float *beta, float eps, float momentum, float *running_mean,
                                            float *running_var, float *y, float *save_mean, float *save_invstd,
                                            void *workspace, int relu);
/* dx (b, c, l), dgamma (c), dbeta (c) from dy (the gradient w.r.t. the forward's output) and the saved statistics. */
void batchnorm_train_backward_kernel_wrapper(int b, int c, int l, const float *x, const float *dy,
                                             const float *gamma, const float *beta, const float *save_mean,
                                             const float *save_invstd, float *dx, float *dgamma, float *dbeta,
                                             void *workspace, int relu);

/* Tail of the module path's grouped stacks in training mode: BatchNorm (batch statistics) -> ReLU -> max over the k
 * neighbours (P2/pointnet2_modules.py: `self.mlp_module(new_features)` followed by `.max(dim=3)` /
 * F.max_pool2d(kernel=[1, nsample]); the reference materialises the (b, c, s, k) activation between them).
 * x (b, c, s, k) f32 = the last convolution's output, k in {4, 8, 16, 32}; pooled (b, c, s) f32; arg (b, c, s) u8 =
 * first neighbour reaching the maximum; xsel (b, c, s) f32 = x at arg.  Statistics, running_* update, save_* and
 * workspace as batchnorm_train_forward_kernel_wrapper.  Same values as BN -> ReLU -> max evaluated separately. */
void batchnorm_train_relu_maxk_forward_kernel_wrapper(int b, int c, int s, int k, const float *x, const float *gamma,
                                                      const float *beta, float eps, float momentum,
                                                      float *running_mean, float *running_var, float *pooled,
                                                      unsigned char *arg, float *xsel, float *save_mean,
                                                      float *save_invstd, void *workspace);
/* dx (b, c, s, k), dgamma (c), dbeta (c) from dpool (b, c, s), the gradient w.r.t. pooled: the dense gradient of the
 * activation (dpool at arg where the pooled value was positive, 0 elsewhere) is never written. */
void batchnorm_train_relu_maxk_backward_kernel_wrapper(int b, int c, int s, int k, const float *x, const float *dpool,
                                                       const unsigned char *arg, const float *xsel,
                                                       const float *gamma, const float *beta, const float *save_mean,
                                                       const float *save_invstd, float *dx, float *dgamma,
                                                       float *dbeta, void *workspace);

/* Pointwise convolution of the module path's shared MLPs, forward and both gradients, on channel-major rows
 * (P2/pytorch_utils.py:114-167: Conv2d(kernel_size=(1,1), bias=False) inside SharedMLP, pytorch_utils.py:12-37;
 * the reference runs torch.nn.Conv2d = cuDNN there).  x (b, cin, p), y (b, cout, p), p = product of the trailing
 * dimensions, p % 4 == 0, all pointers 16-byte aligned, cin and cout <= 512 with the packed weights within 150 KiB
 * of LDS (ceil(cin/16) * min(ceil(cout/16), 8) KiB).  fp32 FMAs on the matrix cores: equal to any fp32 convolution
 * up to summation order.
 *   transposed = 0: y[b][o][q] = sum_i w[o * cin + i] x[b][i][q]            (w (cout, cin) row-major: forward)
 *   transposed = 1: y[b][o][q] = sum_i w[i * cout + o] x[b][i][q]           (w (cin, cout) row-major: the input
 *                   gradient of the layer whose weight is w, called with x = dY, cin = the layer's cout). */
void conv1x1_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w, int transposed,
                                    float *y);
/* The forward with the layer's eval-mode BatchNorm and ReLU in the epilogue (P2/pytorch_utils.py:114-167: conv -> bn ->
 * ReLU blocks in eval mode): y = act(conv(x) * scale[o] + shift[o]), scale = gamma / sqrt(running_var + eps), shift =
 * beta - running_mean * scale (both (cout) f32, computed by the caller), act = ReLU when relu != 0. */
void conv1x1_affine_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                           const float *scale, const float *shift, int relu, float *y);
/* Training mode, interior layers of a stack: the convolution applies the PREVIOUS layer's BatchNorm (batch statistics)
 * and ReLU to its input while loading it, a = max(((x - mean) * invstd) * gamma + beta, 0) (the expression of
 * batchnorm_train_forward, bit for bit), so the normalised activation is never written: x (b, cin, p) is the previous
 * convolution's output, in_mean / in_invstd (cin) its statistics (batchnorm_train_forward with y == NULL computes them
 * without the apply pass), in_gamma / in_beta (cin) may be NULL (1 / 0).  The weight-gradient twin multiplies dy with the
 * same transformed input. */
void conv1x1_bnrelu_forward_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                           const float *in_mean, const float *in_invstd, const float *in_gamma,
                                           const float *in_beta, float *y);
void conv1x1_bnrelu_wgrad_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *x,
                                         const float *in_mean, const float *in_invstd, const float *in_gamma,
                                         const float *in_beta, float *dw, void *workspace);
/* Training mode, conv -> BatchNorm of the SAME block (pytorch_utils.py:114-167): y = W a (a = x, or the transformed
 * input of conv1x1_bnrelu_forward when in_mean != NULL) AND the batch statistics of y for the BatchNorm that follows --
 * save_mean, save_invstd (cout), the momentum update of running_mean / running_var (nullable, unbiased variance) --
 * from per-workgroup fp64 partial sums the convolution's epilogue leaves: the statistics pass of
 * batchnorm_train_forward_kernel_wrapper(y = NULL) without reading y again.  workspace: conv1x1_stats_workspace_bytes(). */
long long conv1x1_stats_workspace_bytes(int b, int cin, int cout, int p);
void conv1x1_forward_bnstats_kernel_wrapper(int b, int cin, int cout, int p, const float *x, const float *w,
                                            const float *in_mean, const float *in_invstd, const float *in_gamma,
                                            const float *in_beta, float *y, float eps, float momentum,
                                            float *running_mean, float *running_var, float *save_mean,
                                            float *save_invstd, void *workspace);
/* Training mode, input gradient through conv <- ReLU <- BatchNorm: da (b, cin, p) = W^T dy -- the gradient w.r.t. the
 * rectified, normalised input of the convolution -- AND dgamma / dbeta (cin) of the BatchNorm in front of it, summed in the
 * convolution's epilogue from da and bn_x (b, cin, p), the BatchNorm's input: the reduction pass of
 * batchnorm_train_backward_kernel_wrapper(relu = 1) without reading da again.  w (cout, cin), dy (b, cout, p); mean /
 * invstd (cin) the saved statistics, gamma / beta nullable.  Finish with batchnorm_train_backward_apply_kernel_wrapper
 * (the same dx).  workspace: conv1x1_stats_workspace_bytes(b, cout, cin, p). */
void conv1x1_dgrad_bnstats_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *w,
                                          const float *bn_x, const float *mean, const float *invstd, const float *gamma,
                                          const float *beta, float *da, float *dgamma, float *dbeta, void *workspace);
void batchnorm_train_backward_apply_kernel_wrapper(int b, int c, int l, const float *x, const float *dy, const float *gamma,
                                                   const float *beta, const float *save_mean, const float *save_invstd,
                                                   const float *dgamma, const float *dbeta, float *dx, int relu);
/* The apply passes of batchnorm_train_forward / batchnorm_train_relu_maxk_forward alone, for statistics obtained that
 * way: y = [relu]((x - mean) * invstd * gamma + beta); pooled / arg / xsel as documented above. */
void batchnorm_train_apply_kernel_wrapper(int b, int c, int l, const float *x, const float *gamma, const float *beta,
                                          const float *mean, const float *invstd, float *y, int relu);
void batchnorm_train_relu_maxk_apply_kernel_wrapper(int b, int c, int s, int k, const float *x, const float *gamma,
                                                    const float *beta, const float *mean, const float *invstd,
                                                    float *pooled, unsigned char *arg, float *xsel);
/* The same followed by the stack's max over the k neighbours (P2/pointnet2_modules.py: SharedMLP then .max(dim=3) /
 * max_pool2d(kernel=[1, nsample])): x (b, cin, s, k), pooled (b, cout, s) = max_k act(conv(x) * scale + shift); the
 * (b, cout, s, k) activation is not written.  k in {4, 8, 16, 32}. */
void conv1x1_affine_maxk_forward_kernel_wrapper(int b, int cin, int cout, int s, int k, const float *x, const float *w,
                                                const float *scale, const float *shift, int relu, float *pooled);
/* dw (cout, cin) = sum over b, q of dy[b][o][q] x[b][i][q], summed in a fixed order (deterministic).  workspace:
 * conv1x1_wgrad_workspace_bytes(b, cin, cout, p) bytes of device memory, 16-byte aligned. */
long long conv1x1_wgrad_workspace_bytes(int b, int cin, int cout, int p);
void conv1x1_wgrad_kernel_wrapper(int b, int cin, int cout, int p, const float *dy, const float *x, float *dw,
                                  void *workspace);

/* ---- 4. hoisted variants of section 3 ----------------------------------------------------------
 * The first layer of every grouped MLP is linear in [geometry | feat_centre[s] | feat_nbr[n]]; the
 * feature parts depend on one point only, so W_feat . feat[point] (+ bias) is computed once per
 * point by linear_jobs and gathered by the pixel kernels as accumulator seeds (csrc/mlp_core.hpp,
 * "Hoisting").  Same results as section 3 up to fp32 summation order. */

/* Up to 8 independent per-point linear maps in one launch: out_j (npts_j, cout_j) = src_j (npts_j,
 * cin_j) . W_j^T + bias_j, no activation; cin in {16,32,64}, cout in {16,32,64,128}; all seven
 * arrays are HOST arrays of length njobs (pointers inside are device pointers). */
void linear_jobs_kernel_wrapper(int njobs, const int *npts, const int *cin, const int *cout,
                                const float *const *src, const float *const *w, float *const *out,
                                const int *out_bf16 /* per job: 1 = write the rows as bf16 (dtype "bf16"); may be NULL */);

/* sa_fused with pre (b,n,c1) = W1_feat . feat + b1 (NULL at level 0, where layer 1 is whole). */
void sa_fused_h_kernel_wrapper(int b, int n, int s, int k, int c1, int c2, int c3, const float *xyz,
                               const float *new_xyz, const float *pre, const int *idx,
                               const float *packed_w, float *out, int wfmt, int packed_floats, int kmajor);
/* (kmajor = 1: level-0 stack whose 8-channel layers are packed "k-step major" -- layers 2 and 3 then issue only the two
 *  MFMA k-steps that carry data; fused.py: pack_layer(kmajor_out=True).)
 * upconv_fused with pre (b,n,128) = W1_feat . feat1 + b1. */
void upconv_fused_h_kernel_wrapper(int b, int n, int s, int k, const float *xyz2, const float *xyz1,
                                   const float *pre, const int *idx, const float *packed_w, float *out,
                                   int wfmt, int packed_floats);
/* Set-upconv INCLUDING its post-MLP (P2/pointnet2_modules.py:479-515) for up to two jobs that share the queries xyz2, the
 * coarse points xyz1, the neighbour lists idx and the fine features feat2 (b,s,c2) -- the features and the mask branch of one
 * refinement level, PW/pose_warp_refinement.py:95-103 -- in ONE launch: out_j (b,s,64) = relu(Wpost_j . [max_k stack_j | feat2]).
 * pre / packed_w / packed_post / out are HOST arrays of njobs device pointers; packed_w as for upconv_fused_h (fp32 tiles),
 * packed_post = the one packed post layer (64 + c2 -> 64).  In-lane pooling (a wave tile = 16 queries), c2 in {16,32,64}. */
void upconv_post_fused_h_kernel_wrapper(int njobs, int b, int n, int s, int k, int c2, const float *xyz2,
                                        const float *xyz1, const int *idx, const float *feat2,
                                        const float *const *pre, const float *const *packed_w,
                                        const float *const *packed_post, float *const *out,
                                        int packed_floats, int post_floats);
/* cv_fused_a1 with u (b,s,128) = W1_p . feat1 + b1 and v (b,n,128) = W1_q . feat2. */
void cv_fused_a1_h_kernel_wrapper(int b, int n, int s, int k, const float *xyz1, const float *u,
                                  const float *xyz2, const float *v, const int *idx,
                                  const float *packed_w, float *pix, int pix_slots, int wfmt,
                                  int packed_floats);
/* cv_fused_a1_h + cv_fused_a2 for nsample_q = 6 as ONE kernel (in-lane softmax, a wave tile = 16 queries): the per-pixel
 * (b,s,6,64) buffer between the two stages does not exist; first (b,s,64) is the first aggregate (PW/costvolume.py:139-141).
 * packed_a1 / packed_a2: the two stages' packed stacks as for the separate kernels (fp32 tiles; both LDS resident).
 * packed_v2 (may be NULL, then v2 must be NULL): the one packed layer 64 -> 128 of cv_b's neighbour partial product;
 * v2 (b,s,128) = W_f . first is then written as well (replaces that linear_jobs launch).  Bit-identical to the separate
 * kernels. */
void cv_fused_a_lane6_kernel_wrapper(int b, int n, int s, const float *xyz1, const float *u, const float *xyz2,
                                     const float *v, const int *idx, const float *packed_a1, const float *packed_a2,
                                     const float *packed_v2, float *first, float *v2, int a1_floats, int a2_floats,
                                     int v2_floats);
/* cv_fused_b with u2 (b,s,128) = W_p . feat1 + b, v2 (b,s,128) = W_f . first, first (b,s,64). */
void cv_fused_b_h_kernel_wrapper(int b, int s, int k, const float *xyz1, const float *u2,
                                 const float *v2, const float *first, const int *idx,
                                 const float *packed_w, float *out, int wfmt, int packed_floats);

/* ---- 4b. module path (training): softmax over K + weighted sum as one pass each way ---------------------------
 * out[r] = sum_k softmax_k(x[r,:]) * v[r,k] for rows = B*C*S contiguous rows of K logits / values (PW/costvolume.py:139-141,
 * 181-183 on (B,C,S,K) tensors); backward recomputes the probabilities: dv = dout p, dx = p dout (v - out).
 * K in {1,2,4,6,8,16,32} (softmax_wsum_supported_k), 16-byte aligned tensors. */
int softmax_wsum_supported_k(int k);
void softmax_wsum_forward_kernel_wrapper(long long rows, int k, const float *x, const float *v, float *out);
void softmax_wsum_backward_kernel_wrapper(long long rows, int k, const float *x, const float *v, const float *dout,
                                          float *dx, float *dv);

/* ---- 5. KITTI odometry evaluation of predicted poses ---------------------------
 * Replaces the per-sample host loops of /root/reference/train.py:866-893 (pose row -> 4x4 via quat2mat :762-795),
 * slam/common/kitti360_utils.py:406-431 (relative -> absolute poses), evaluation.py:198-215 (trajectory distances)
 * and evaluation.py:236-271 = slam/eval/eval_odometry.py:316-361 (segment errors).  All matrices are 4x4 row-major
 * fp64 in device memory; frames of all sequences are stored back to back, sequence s owning frames
 * [seq_start[s], seq_start[s+1]) (seq_start: nseq+1 ints in DEVICE memory). */

/* rows: n pose rows [tx ty tz qw qx qy qz] fp32, row_stride floats apart (28 for level 1 of a (B,4,7) pose_params
 * tensor) -> T (n,4,4) = [[R(q) t],[0 0 0 1]], or its inverse when invert != 0 (the reference stores the inverse
 * as the "relative pose", train.py:878). */
void odom_rows_to_transforms_kernel_wrapper(int n, int row_stride, const float *rows, double *T, int invert);
/* abs[f] = T[first] . ... . T[f] inside every sequence (kitti360_utils.py:422-426 applied to rel = T^-1). */
void odom_accumulate_kernel_wrapper(int nseq, const int *seq_start, const double *T, double *abs_out);
/* dist[f] = sum_{i<=f} |p[i] - p[i-1]| (dist[first] = 0) from the translations of `poses`. */
void odom_cumulative_distance_kernel_wrapper(int nseq, const int *seq_start, const double *poses, double *dist);
/* One slot per (sequence, first frame in 0,step,2*step.., segment length): slot_start (nseq+1 ints, DEVICE) with
 * slot_start[s+1]-slot_start[s] = ceil(n_s/step)*nlen, total_slots = slot_start[nseq].  err (total_slots,5) =
 * [first_frame, r_err/len, t_err/len, len, speed] as evaluation.py:270; valid[slot] = 0 where the sequence is too
 * short for that segment (the reference skips those).  lengths: nlen doubles in DEVICE memory. */
void odom_sequence_errors_kernel_wrapper(int nseq, int total_slots, const int *seq_start, const int *slot_start,
                                         const double *poses_gt, const double *poses_result, const double *dist,
                                         int step, int nlen, const double *lengths, double *err, int *valid);

#ifdef __cplusplus
}
#endif
#endif /* PWCLO_OPS_H */
