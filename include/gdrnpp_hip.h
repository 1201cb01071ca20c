/*
 * gdrnpp_hip.h — flat C ABI of libgdrnpp_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the GDRNPP inference hot path
 * (SURVEY.md §8b).  Two groups of entry points:
 *
 *  (1) Reference-ABI symbols: byte-for-byte the prototypes the reference's
 *      cffi modules bind (host pointers, synchronous, no status):
 *        core/csrc/fps/src/ext.h:1-14
 *        core/csrc/uncertainty_pnp/src/ext.h:1-9
 *  (2) gdrnpp_* symbols: device-pointer, stream-ordered, batched entry points
 *      that replace the reference's torch extensions (ransac_voting.cpp:112-117,
 *      nnd_cuda.cpp:86-89) and its per-ROI CPU/GL post-processing
 *      (gdrn_evaluator.py:115-153,461-573, engine_utils.py:295-333,
 *      pose_from_pred_centroid_z.py:56-154, lib/render_vispy/renderer.py).
 *
 * Conventions for group (2):
 *   - every pointer is a DEVICE pointer unless its name starts with h_;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return 0 = ok, negative = argument error (GDRNPP_E*), positive = hipError_t;
 *   - no allocation inside; scratch is passed in, sized by *_workspace_bytes();
 *   - no global state apart from the process-wide tuning switches of gdrnpp_set_option; safe from several host
 *     threads on different streams.
 */
#ifndef GDRNPP_HIP_H_
#define GDRNPP_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDRNPP_EINVAL (-1) /* bad size / null pointer */
#define GDRNPP_ELIMIT (-2) /* size above what the kernel supports */

/* library/ABI version: major*10000 + minor*100 + patch */
int gdrnpp_version(void);
/* message for the last non-zero status returned on this host thread */
const char* gdrnpp_last_error(void);
/* process-wide tuning switches, read on the launch path (no environment lookups there):
 *   "split_gemm_glds"  0 / 1   256-row tiles of the split GEMM / convolution use the register-staged kernel / the
 *                              LDS-DMA kernel (default 1; results are bitwise identical)
 *   "split_gemm_pipe"  0 / 2 / 3   256-row tiles of the linear form use the software-pipelined LDS-DMA kernel with two /
 *                              three A stages (default 3; bitwise identical to the other kernels); 0 = off
 *   "split_gemm_pipe_conv"  0 / 1   the 3x3 / stride 1 / pad 1 convolution uses it too (default 0)
 *   "split_gemm_big_tiles"  >= 1    256x128 output tiles (pipelined / LDS-DMA kernels) when the problem has at least this many
 *                                   of them (default 256 = one per CU), 128x128 tiles below
 *   "split_gemm_panel"      0, 2..64  wide layers (packed weight > 2 MB, N >= 1024) walk their tiles in panels of that many
 *                                   256-row blocks, column tile outer, for L2 reuse of both operands (default 4; 0: row-major)
 *   "split_gemm_mi4"   -1 / 0 / 1   tile height by tile count (default) / force 128 rows / force 256 rows
 *   "dwconv_tile"      -1 / 0 / 1 / 2   output pixels per thread of the depthwise 7x7 kernel: by launch size (default: 2x8 at
 *                                   the headline batch, 2x4 / 1x4 when a launch has too few tiles for the chip) / force 2x8 / 2x4 / 1x4
 *   "mlp_fused_pipe"   0 / 1       fused stage-0 MLP (gdrnpp_convnext_mlp_f32_fused): software-pipelined tile loop (default 1) or the plain loop (A/B switch)
 *   "split2_wide"      0 / 1       three-product kernels (gdrnpp_*_split2): 256x256 block tiles whenever N % 256 == 0 (A/B switch,
 *                                   default 0: bitwise identical and measured slower than 256x128)
 *   "dwconv_lds_w"     0 / 1       depthwise 7x7 + LayerNorm: [49][C] weights in LDS with persistent workgroups where they fit (default 1) or
 *                                   through L1 / L2 without LDS (A/B switch: which of two kernels sharing the chip yields is decided by the LDS
 *                                   a workgroup holds, profiles/r06_dwconv_shared.txt; bitwise identical)
 *   "dwconv_lds_pad"   bytes       experiment only: dynamic LDS the no-LDS form allocates without using it (default 0)
 * unknown name -> GDRNPP_EINVAL. */
int gdrnpp_set_option(const char* name, int value);

/* ------------------------------------------------------------------------ */
/* (1) reference-ABI symbols (host pointers)                                 */
/* ------------------------------------------------------------------------ */

/* core/csrc/fps/src/ext.h:1-6 — random start point (time-seeded like
 * farthest_point_sampling.cpp:93-94); pts f32[pn,3], idxs i32[sn]. */
void farthest_point_sampling(float* pts, int* idxs, int pn, int sn);
/* core/csrc/fps/src/ext.h:9-14 — start = farthest from bbox centre
 * (farthest_point_sampling.cpp:122-160). */
void farthest_point_sampling_init_center(float* pts, int* idxs, int pn, int sn);

/* core/csrc/uncertainty_pnp/src/ext.h:1-9 — covariance-weighted reprojection
 * LM (uncertainty_pnp.cpp:61-92).  pts2d f64[pn,2], pts3d f64[pn,3],
 * wgt2d f64[pn,3]=(wxx,wxy,wyy), K f64[9], init_rt/result_rt f64[6]
 * = (angle-axis, t). */
void uncertainty_pnp(double* pts2d, double* pts3d, double* wgt2d, double* K,
                     double* init_rt, double* result_rt, int pn);

/* ------------------------------------------------------------------------ */
/* (2) device-pointer entry points                                           */
/* ------------------------------------------------------------------------ */

/* ---- farthest point sampling (a9) ---------------------------------------
 * pts f32[b,pn,3]; idxs i32[b,sn]; start_idx i32[b] or NULL.
 * mode 0: start from start_idx[b] (farthest_point_sampling.cpp:76-105 with the
 *         random draw made explicit); mode 1: init-center (…:122-160).
 * workspace: gdrnpp_fps_workspace_bytes(b,pn) bytes (may be 0). */
size_t gdrnpp_fps_workspace_bytes(int b, int pn);
int gdrnpp_fps(const float* pts, int* idxs, const int* start_idx, int b, int pn,
               int sn, int mode, void* workspace, void* stream);

/* ---- bidirectional NN distance / chamfer (a12) ---------------------------
 * nnd_cuda.cpp:37-60 (forward), :63-84 (backward).
 * xyz1 f32[b,n,3], xyz2 f32[b,m,3] -> dist1 f32[b,n], idx1 i32[b,n],
 * dist2 f32[b,m], idx2 i32[b,m]; first minimum wins (nnd_cpu.cpp:3-25). */
int gdrnpp_nnd_forward(const float* xyz1, const float* xyz2, float* dist1,
                       float* dist2, int* idx1, int* idx2, int b, int n, int m,
                       void* stream);
/* grad buffers are overwritten (zeroed inside), like nnd_cuda_kernel.cu:185-222 */
int gdrnpp_nnd_backward(const float* xyz1, const float* xyz2, float* gradxyz1,
                        float* gradxyz2, const float* graddist1,
                        const float* graddist2, const int* idx1, const int* idx2,
                        int b, int n, int m, void* stream);

/* ---- PVNet-style RANSAC voting (a10) -------------------------------------
 * ransac_voting.cpp:30-41,51-65,74-85,95-109.
 * direct f32[tn,vn,2], coords f32[tn,2], idxs i32[hn,vn,2],
 * hypo_pts f32[hn,vn,2] (vanishing point: [hn,vn,3]); the kernels overwrite
 * the whole of hypo_pts (degenerate pairs -> 0, like at::zeros + skip);
 * inliers u8[hn,vn,tn] must be pre-zeroed by the caller exactly as in the
 * reference (ransac_voting_gpu.py:58) — only 1s are written. */
int gdrnpp_generate_hypothesis(const float* direct, const float* coords,
                               const int* idxs, float* hypo_pts, int tn, int vn,
                               int hn, void* stream);
int gdrnpp_voting_for_hypothesis(const float* direct, const float* coords,
                                 const float* hypo_pts, unsigned char* inliers,
                                 int tn, int vn, int hn, float inlier_thresh,
                                 void* stream);
int gdrnpp_generate_hypothesis_vanishing_point(const float* direct,
                                               const float* coords,
                                               const int* idxs, float* hypo_pts,
                                               int tn, int vn, int hn,
                                               void* stream);
int gdrnpp_voting_for_hypothesis_vanishing_point(
    const float* direct, const float* coords, const float* hypo_pts,
    unsigned char* inliers, int tn, int vn, int hn, float inlier_thresh,
    void* stream);
/* fused vote + count: counts i32[hn,vn] = sum_t inlier(hi,vi,ti) without
 * materialising the [hn,vn,tn] tensor (replaces voting + torch.sum(...,2),
 * ransac_voting_gpu.py:58-62).  homogeneous=1 -> vanishing-point variant. */
int gdrnpp_vote_count(const float* direct, const float* coords,
                      const float* hypo_pts, int* counts, int tn, int vn, int hn,
                      float inlier_thresh, int homogeneous, void* stream);

/* ---- uncertainty-PnP, batched (a11) --------------------------------------
 * One problem per workgroup; same cost and LM schedule as the host symbol.
 * pts2d f64[b,pn,2], pts3d f64[b,pn,3], wgt f64[b,pn,3], K f64[b,9],
 * init f64[b,6] -> result f64[b,6]; info i32[b,2] = (iterations, status) or NULL. */
int gdrnpp_uncertainty_pnp_batched(const double* pts2d, const double* pts3d,
                                   const double* wgt2d, const double* K,
                                   const double* init_rt, double* result_rt,
                                   int* info, int b, int pn, void* stream);

/* ---- EPnP + RANSAC on the decoded correspondences (a7: "ransac_pnp", "net_ransac_pnp", "net_ransac_pnp_rot") ------
 * cv2.solvePnPRansac(objectPoints, imagePoints, K, dist = 0, flags = SOLVEPNP_EPNP, reprojectionError, iterationsCount,
 * confidence = 0.99) as called at lib/pysixd/misc.py:182-193 (reprojErr 3, 100 iterations) and
 * gdrn_evaluator.py:319-330 (20 iterations), for every ROI of the batch.  OpenCV is a third-party dependency that is not
 * in the reference tree: its published algorithms are restated (calib3d epnp.cpp, ptsetreg.cpp, solvepnp.cpp) — 5-point
 * minimal sets, float32 squared reprojection error <= reprojErr^2, best = strictly more inliers, adaptive iteration
 * count, final EPnP over the inliers of the best hypothesis; count == 5: one EPnP over all five; count == 4: one P3P solve
 * (OpenCV switches to SOLVEPNP_P3P there: the poses consistent with the first three points, the fourth picks), all inliers.
 * img_pts f32[b,stride,2], mdl_pts f32[b,stride,3], count i32[b]: outputs of gdrnpp_decode_correspondences; K f32[b,9].
 * draws: NULL -> every ROI draws from cv::RNG(2^64-1) like OpenCV does; else u32[b,n_draws] words consumed in order
 * (index = word % count, redrawn while it repeats), n_draws >= 5*iters.
 * R_out f32[b,9], t_out f32[b,3], n_inliers i32[b], status i32[b] (1 = pose found; 0 = fewer than 4 points, no
 * hypothesis with at least 5 inliers, or a degenerate system: R = I, t = 0 and the caller applies the reference's
 * fallback), inlier_mask u8[b,stride].  workspace: gdrnpp_epnp_ransac_workspace_bytes(b, stride, iters). */
size_t gdrnpp_epnp_ransac_workspace_bytes(int b, int stride, int iters);
int gdrnpp_epnp_ransac(const float* img_pts, const float* mdl_pts, const int* count, int stride, const float* K,
                       const unsigned* draws, int n_draws, int iters, float reproj_err, double confidence,
                       float* R_out, float* t_out, int* n_inliers, int* status, unsigned char* inlier_mask,
                       int b, void* workspace, size_t workspace_bytes, void* stream);
/* plain EPnP on all n >= 4 points of each problem (cv2.solvePnP(flags=SOLVEPNP_EPNP); un_pnp_utils.py:27-44 runs it on the
 * four best-weighted keypoints to initialise uncertainty-PnP): img_pts f32[b,n,2], mdl_pts f32[b,n,3], K f32[b,9]. */
int gdrnpp_epnp_batched(const float* img_pts, const float* mdl_pts, int n, const float* K, float* R_out,
                        float* t_out, int* status, int b, void* stream);

/* ---- net-initialised iterative PnP on the decoded correspondences (a7, "net_iter_pnp") ------------------------
 * gdrn_evaluator.py:241-371 with pnp_type="iter": cv2.solvePnP(SOLVEPNP_ITERATIVE, useExtrinsicGuess=True) seeded
 * with the network pose = Levenberg-Marquardt on the plain reprojection error.  OpenCV's own LM is third-party and
 * not in the tree; the same restated LM as gdrnpp_uncertainty_pnp_batched is used with identity weights (both
 * converge to the minimiser of the same cost).  img_pts f32[b,stride,2], mdl_pts f32[b,stride,3] and count i32[b]
 * are the outputs of gdrnpp_decode_correspondences (stride = hw).  Reference fallbacks kept: count < 4 -> network
 * pose (:355-358); |t_est - t_net| > 1 m -> network translation (:347-351, info status += 100).
 * K f32[b,9], R_net f32[b,9], t_net f32[b,3] -> R_out f32[b,9], t_out f32[b,3], info i32[b,2] or NULL. */
int gdrnpp_pnp_iter_from_correspondences(const float* img_pts, const float* mdl_pts,
                                         const int* count, int stride, const float* K,
                                         const float* R_net, const float* t_net,
                                         float* R_out, float* t_out, int* info, int b,
                                         void* stream);

/* ---- map decoding + 2D-3D correspondences (a4 + a6) ----------------------
 * engine_utils.py:295-333 (regression xyz, mask_type: 0=L1 min-max, 1=sigmoid)
 * and gdrn_evaluator.py:115-153.  coor f32[b,3,64*64] (x,y,z planes, may be
 * three separate [b,1,h,w] tensors laid out contiguously or passed via
 * strides = plane stride in floats), mask_raw f32[b,hw], coord2d f32[b,2,hw],
 * extent f32[b,3], imwh f32[b,2]=(im_W, im_H).
 * outputs: out_mask f32[b,hw] (normalised mask, may be NULL), count i32[b],
 * sel_idx i32[b,hw] (row-major pixel index of every selected pixel, in
 * order), img_pts f32[b,hw,2], mdl_pts f32[b,hw,3]; rows >= count[b] are left
 * untouched. */
int gdrnpp_decode_correspondences(const float* coor_x, const float* coor_y,
                                  const float* coor_z, const float* mask_raw,
                                  const float* coord2d, const float* extent,
                                  const float* imwh, float* out_mask, int* count,
                                  int* sel_idx, float* img_pts, float* mdl_pts,
                                  int b, int hw, int mask_type, float mask_thr,
                                  void* stream);

/* ---- allocentric->egocentric pose from Patch-PnP output (a3.4 + a3.5) -----
 * rot_reps.py:34-55 + pose_from_pred_centroid_z.py:56-154 + utils.py:31-75.
 * rot6d f32[b,6] (allo), t_ f32[b,3] = (dx,dy,z_rel), cams f32[b,9],
 * centers f32[b,2], whs f32[b,2], resize_ratios f32[b] -> rot f32[b,9] (ego),
 * trans f32[b,3].  z_type 0=REL 1=ABS; is_allo 0/1. */
int gdrnpp_pose_from_pred_centroid_z(const float* rot6d, const float* t_,
                                     const float* cams, const float* centers,
                                     const float* whs, const float* resize_ratios,
                                     float* rot, float* trans, int b, int z_type,
                                     int is_allo, void* stream);
/* the other ROT_TYPE / TRANS_TYPE variants of GDRN_double_mask.py:162-200 through the same kernel:
 * rot_mode 0 = 6-d representation f32[b,6], 1 = quaternion (w,x,y,z) f32[b,4] (quat2mat_torch, pose_utils.py:349-400),
 *          2 = rotation matrix f32[b,9], 3 = log-quaternion f32[b,3] (quaternion_lf.qexp, core/utils/quaternion_lf.py:294-318,
 *          then quat2mat_torch), 4 = Lie vector / angle-axis f32[b,3] (lie_algebra.lie_vec_to_rot, core/utils/lie_algebra.py:7-77)
 *          — get_rot_mat, model_utils.py:347-359;
 * t_mode   0 = centroid_z with relative z, 1 = centroid_z with absolute z, 2 = centroid_z_abs (absolute 2-d centre and z,
 *          pose_from_pred_centroid_z_abs.py:44-76; centers / whs / resize_ratios unused), 3 = trans (t_ is the translation,
 *          pose_from_pred.py:25-27). */
int gdrnpp_pose_from_pred(const float* rot_in, int rot_mode, const float* t_, int t_mode, const float* cams,
                          const float* centers, const float* whs, const float* resize_ratios, float* rot, float* trans,
                          int b, int is_allo, void* stream);

/* Patch-PnP's two output layers in one launch (row a3.3): rot_ = fc_r(x), t_ = fc_t(x) of ConvPnPNet.forward,
 * core/gdrn_modeling/models/heads/conv_pnp_net.py:99-101,178-182 (two nn.Linear on the same [b,256] feature).
 * x f32[b,K] (K <= 1024), w_r f32[rot_dim,K] (rot_dim <= 9), b_r f32[rot_dim] | NULL, w_t f32[3,K], b_t f32[3] | NULL
 * -> rot_ f32[b,rot_dim], t_ f32[b,3]; fp32 fma chains, one wavefront per ROI, fixed summation order. */
int gdrnpp_pnp_fc_heads(const float* x, const float* w_r, const float* b_r, const float* w_t, const float* b_t, float* rot_,
                        float* t_, int b, int K, int rot_dim, void* stream);
/* the same launch followed, per ROI, by gdrnpp_pose_from_pred on the values just computed (GDRN_double_mask.py:162-200 behind
 * conv_pnp_net.py:178-182): the network's last kernel.  rot_dim follows from rot_mode (6 / 4 / 9 / 3 / 3); rot_ f32[b,rot_dim] and
 * t_ f32[b,3] are written as well (forward returns them to the losses / tests).  Pose arguments as gdrnpp_pose_from_pred. */
int gdrnpp_pnp_fc_heads_pose(const float* x, const float* w_r, const float* b_r, const float* w_t, const float* b_t, float* rot_,
                             float* t_, int b, int K, int rot_mode, int t_mode, const float* cams, const float* centers,
                             const float* whs, const float* resize_ratios, float* rot, float* trans, int is_allo, void* stream);

/* ---- crop-resize intrinsics (a8.2) — camera_geometry.py:6-21 --------------
 * K f32[b,9], centers f32[b,2], scales f32[b] -> K_crop f32[b,9],
 * with crop_xy = center - scale/2 and ratio = out_res/scale
 * (engine_utils.py:260-264). */
int gdrnpp_zoom_K(const float* K, const float* centers, const float* scales,
                  float* K_crop, int b, float out_res, void* stream);

/* ---- mesh set for the depth rasteriser -----------------------------------
 * A flat, caller-owned description of all object models resident in HBM:
 * verts f32[sum V,3], faces i32[sum F,3] (indices local to the object),
 * vert_off i32[n_obj+1], face_off i32[n_obj+1]. */
typedef struct gdrnpp_meshes {
  const float* verts;
  const int* faces;
  const int* vert_off;
  const int* face_off;
  int n_obj;
  int max_verts; /* host-side hint: max vertices of any object (0 = unknown -> no LDS staging) */
  int max_faces; /* host-side hint: max faces of any object (0 = unknown) */
} gdrnpp_meshes;

/* ---- depth render (a8.1 / a14) — lib/render_vispy/renderer.py:126-130,
 * 155-182,363-407,461-477 restated without GL: depth f32[b,res,res] in metres,
 * 0 = background; optional xyz f32[b,res,res,3] = object-space surface point
 * (the PCObject attachment of egl_renderer, egl_renderer_v3.py:1185-1225),
 * NULL to skip.  K f32[b,9], R f32[b,9], t f32[b,3], obj i32[b]. */
int gdrnpp_render_depth(const gdrnpp_meshes* meshes, const int* obj,
                        const float* K, const float* R, const float* t,
                        float* depth, float* xyz, int b, int res, float z_near,
                        float z_far, void* stream);

/* ---- fast depth refinement (a8) — gdrn_evaluator.py:461-573 ---------------
 * One workgroup per ROI runs all iterations (render -> query map -> threshold
 * -> median -> weighted centroid -> ray update) on chip.
 * coor_{x,y,z} f32[b,hw] normalised xyz maps; mask_raw f32[b,hw];
 * roi_depth f32[b,in_res,in_res] sensor depth crop, in_res must be 4*res (256x256 for res=64: the kernel reads the 2x2
 * centre of every 4x4 block like cv2.resize at scale 4; anything else is refused);
 * K_crop f32[b,9]; R f32[b,9]; t_in f32[b,3]; obj i32[b] -> t_out f64[b,3].  An obj id outside [0, n_obj) leaves t = t_in.
 * mask_type as in gdrnpp_decode_correspondences; use_coor_z: TEST.USE_COOR_Z_REFINE.
 * debug_depth (f32[b,iters,hw]) receives each iteration's render or NULL.
 * workspace: gdrnpp_depth_refine_workspace_bytes(meshes, b) — 0 unless a mesh has more than 4096 vertices (their
 * transformed vertices are then staged in this workspace instead of LDS; same kernel, same arithmetic). */
size_t gdrnpp_depth_refine_workspace_bytes(const gdrnpp_meshes* meshes, int b);
int gdrnpp_depth_refine(const gdrnpp_meshes* meshes, const int* obj,
                        const float* coor_x, const float* coor_y,
                        const float* coor_z, const float* mask_raw,
                        const float* roi_depth, const float* K_crop,
                        const float* R, const float* t_in, double* t_out,
                        float* debug_depth, int b, int res, int in_res, int iters,
                        float threshold, int mask_type, int use_coor_z,
                        float z_near, float z_far, void* workspace, size_t workspace_bytes, void* stream);
/* The post-processing tail of the refine configuration in ONE launch: get_K_crop_resize (camera_geometry.py:6-21, from
 * cam f32[b,9], center f32[b,2], scale f32[b] and OUTPUT_RES = res) -> the refinement above -> the f32[b,16] pose records
 * R(9) | t(3, metres) | score | obj | roi_id | valid of gdrnpp_pack_pose_records (score / roi_id nullable; valid = 0 for
 * an obj id outside the mesh set). */
int gdrnpp_refine_to_records(const gdrnpp_meshes* meshes, const int* obj, const float* coor_x, const float* coor_y,
                             const float* coor_z, const float* mask_raw, const float* roi_depth, const float* cam,
                             const float* center, const float* scale, const float* R, const float* t_in,
                             const float* score, const int* roi_id, float* rec, int b, int res, int in_res, int iters,
                             float threshold, int mask_type, int use_coor_z, float z_near, float z_far, void* workspace,
                             size_t workspace_bytes, void* stream);

/* device-to-device copy into a raw device pointer on `stream` — the transfer CppEGLRenderer::map_tensor performs with
 * cudaMemcpy2DFromArray in the reference (lib/egl_renderer/cpp/egl_renderer.cpp:262-298): attachment -> caller's tensor */
int gdrnpp_copy_d2d(void* dst, const void* src, size_t bytes, void* stream);

/* debug aid: PMC calibration stream — reads n floats with 4- or 16-byte lanes (a known byte count) */
int gdrnpp_debug_stream_read(const float* p, size_t n, int lane_bytes, float* out_blocks,
                             int blocks, void* stream);

/* debug aid: ONE wave busy-waits `micros` microseconds of the device's constant-rate wall clock on `stream` and computes nothing — the
 * probe engine.streams_overlap_ratio launches on two streams to learn whether they sit on different hardware queues (replaces
 * the private torch.cuda._sleep).  1 <= micros <= 1 000 000. */
int gdrnpp_debug_spin(int micros, void* stream);

/* debug aid: 16 s_memtime stamps written by workgroup 0 of the last gdrnpp_depth_refine launch (LDS-staged
 * kernel): [0] start, [1] prologue, then per iteration staged/rastered/reduced/median/updated. Host pointer. */
int gdrnpp_debug_refine_profile(long long* h_out16);

/* ---- network-side layers of GDRN_Net (a3), NHWC fp32 ----------------------------------------
 * Memory-bound layers that PyTorch-ROCm runs far from the HBM roofline: all tensors are
 * channels-last (N,H,W,C contiguous), C % 4 == 0.
 *  dwconv7x7_ln : timm ConvNeXtBlock.conv_dw (depthwise 7x7, pad 3) + bias, fused with the LayerNorm(C, eps)
 *                 that follows it when ln_w/ln_b are non-NULL.  w49c f32[49,C] = weight[C,1,7,7] tap-major.
 *  upsample_bilinear2x : nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True) of the geometry head
 *                 (top_down_doublemask_xyz_region_head.py:80).
 *  groupnorm_act : nn.GroupNorm(G, C, eps) [+ exact-erf GELU] of ConvModule / ConvPnPNet
 *                 (lib/torch_utils/layers/conv_module.py:222-236, conv_pnp_net.py:59-72); workspace sized by
 *                 gdrnpp_groupnorm_workspace_bytes.
 *  layernorm    : LayerNorm over C of n_pix NHWC pixels (timm LayerNorm2d of the ConvNeXt stem and downsample
 *                 layers): biased variance, eps inside the sqrt. */
int gdrnpp_layernorm_nhwc(const float* x, const float* weight, const float* bias, float* y,
                          long n_pix, int C, float eps, void* stream);
int gdrnpp_dwconv7x7_ln_nhwc(const float* x, const float* w49c, const float* bias,
                             const float* ln_w, const float* ln_b, float* y, int N,
                             int H, int W, int C, float eps, void* stream);
/* ... with y written as an "f16x2 rows" tensor when y_rows != 0 (C % 8 == 0; see gdrnpp_linear_f32_split2_rows): the block's
 * LayerNorm output then reaches fc1 already split. */
int gdrnpp_dwconv7x7_ln_nhwc_rows(const float* x, const float* w49c, const float* bias,
                                  const float* ln_w, const float* ln_b, float* y, int N,
                                  int H, int W, int C, float eps, int y_rows, void* stream);
int gdrnpp_upsample_bilinear2x_nhwc(const float* x, float* y, int N, int H, int W,
                                    int C, void* stream);
size_t gdrnpp_groupnorm_workspace_bytes(int N, int HW, int G);
int gdrnpp_groupnorm_act_nhwc(const float* x, const float* gamma, const float* beta,
                              float* y, void* workspace, int N, int HW, int C, int G,
                              float eps, int act_gelu, void* stream);

/* ---- ROI crop-resize (a1) — read_data_test, data_loader.py:754-797 ---------------------------------
 * cv2.warpAffine restated (data_utils.py:115-184): images u8[n_im,H,W,3] (BGR), depths f32[n_im,H,W] or NULL,
 * im_idx i32[b] (NULL = image 0), centers f64[b,2] = bbox centre, scales f64[b] (float64 like the reference's
 * call site) -> roi_img f32[b,3,out,out] = (bilinear u8 - mean[c]) / std[c] (float64 math, float32 store);
 * roi_depth f32[b,1,out,out] (nearest); roi_coord2d f32[b,2,out_small,out_small] = bilinear crop of the
 * analytic coord_2d map.  Any output pointer may be NULL.  h_mean3/h_std3 are HOST pointers to 3 doubles
 * (MODEL.PIXEL_MEAN / PIXEL_STD). */
int gdrnpp_crop_resize_roi(const unsigned char* images, const float* depths, int n_im,
                           int H, int W, const int* im_idx, const double* centers,
                           const double* scales, float* roi_img, float* roi_depth,
                           float* roi_coord2d, int b, int out_res, int out_res_small,
                           const double* h_mean3, const double* h_std3, void* stream);

/* ---- ROIAlign crop-resize (a1b) — detectron2 ROIAlign(output_size, spatial_scale, sampling_ratio, aligned)
 * as called at core/utils/data_utils.py:65-112 and core/utils/zoom_utils.py:80-96.
 * x f32[B,C,H,W] (NCHW), rois f32[N,5] = (batch idx, x1, y1, x2, y2) -> out f32[N,C,pooled_h,pooled_w]. */
int gdrnpp_roi_align(const float* x, const float* rois, float* out, int n_rois, int C,
                     int H, int W, int pooled_h, int pooled_w, float spatial_scale,
                     int sampling_ratio, int aligned, void* stream);

/* the "nearest" flavour of batch_crop_resize (core/utils/zoom_utils.py:92-93): torchvision RoIPool(output_size, spatial_scale)
 * forward — max over integer pixel bins, 0 for an empty bin.  Same tensor conventions as gdrnpp_roi_align. */
int gdrnpp_roi_pool(const float* x, const float* rois, float* out, int n_rois, int C, int H, int W, int pooled_h, int pooled_w,
                    float spatial_scale, void* stream);

/* ---- fp32 linear layer with fused epilogue (ConvNeXt Mlp of GDRN_Net, a3) --------------------------------------
 * C[M,N] = A[M,K] * W[N,K]^T + bias[N]; epilogue 0 = none, 1 = exact-erf GELU (timm Mlp.fc1 + act),
 * 2 = resid[M,N] + gamma[N] * C (Mlp.fc2 + layer scale + residual). */

/* ---- fp32-accurate linear layer on the bf16 matrix cores (a3) -------------------------------------------------
 * Every fp32 operand is split EXACTLY into three bf16 values (x = h + m + l);
 * six of the nine partial products (all terms above 2^-26 relative) are accumulated in fp32 by
 * v_mfma_f32_32x32x16_bf16.  Error against an fp64 product is below that of the fp32 fma chain.
 * gdrnpp_pack_weight_bf16x3: W f32[N][K] (nn.Linear weight) -> bf16[N/128][K/16][3][2][128][8]: the three splits
 * (h | m | l) of every 128x16 tile in the order the kernel stages them (split, k-block of 8, row); 6*N*K bytes,
 * done once per weight.  gdrnpp_linear_f32_split: A stays fp32 and is split on the way into LDS.
 * N multiple of 128, K multiple of 32; any M >= 1 (the last row tile is clamped on load and masked on store). */
int gdrnpp_pack_weight_bf16x3(const float* W, void* packed, int N, int K, void* stream);
/* Split-K form for problems with too few output tiles to fill the chip (the Patch-PnP fc layers: one row per ROI,
 * K = 8192; the stage-2/3 ConvNeXt MLPs at small ROI counts): K is cut into chunks so that every CU gets about three
 * workgroups, partial products go to the workspace (gdrnpp_linear_f32_splitk_workspace_bytes) and are summed in a
 * fixed order; bias and the epilogue (0 none, 1 GELU, 2 resid + gamma * y) are applied by the reduction. */
size_t gdrnpp_linear_f32_splitk_workspace_bytes(int M, int N, int K);
int gdrnpp_linear_f32_splitk(const float* A, const void* W_packed, const float* bias, const float* gamma,
                             const float* resid, float* C, int M, int N, int K, int epilogue, void* workspace,
                             size_t workspace_bytes, void* stream);
int gdrnpp_linear_f32_split(const float* A, const void* W_packed, const float* bias, const float* gamma,
                            const float* resid, float* C, int M, int N, int K, int epilogue,
                            void* stream);
/* ConvNeXt stem (timm stem_0 + stem_1): Conv2d(3 -> 128, kernel 4, stride 4) + bias + LayerNorm2d over the channels in one
 * pass.  x f32 NCHW [N,3,H,W] (H % 4 == 0, W % 16 == 0, W <= 1024), weight f32[128,3,4,4], bias f32[128] or NULL ->
 * y f32 NHWC [N,H/4,W/4,128].  fp32 fma chain in (ci, ky, kx) order; LayerNorm as gdrnpp_layernorm_nhwc. */
int gdrnpp_stem_conv4x4_ln(const float* x_nchw, const float* weight, const float* bias, const float* ln_weight,
                           const float* ln_bias, float* y_nhwc, int N, int H, int W, int Cout, float eps, void* stream);
/* Grouped form for the class-sliced 1x1 output layer of the geometry head (GDRN_double_mask.py:107-126 folded into the
 * weights): W_packed_stack holds one gdrnpp_pack_weight_bf16x3 image per group (all [N,K]), bias_stack f32[groups][N];
 * rows [g*rows_per_group, (g+1)*rows_per_group) of A use slice group_sel[g] (device i32[M / rows_per_group]); rows_per_group a
 * multiple of 256 dividing M.  C is f32[M][N]; columns >= n_store (multiple of 4) are not written.  Bias epilogue only.
 * n_groups = slices in the stack: rows whose selector lies outside [0, n_groups) are written as NaN (the reference's
 * t.view(B,C,k,h,w)[arange(B), roi_classes] raises an index error for such a label, GDRN_double_mask.py:107-126; a stream-ordered
 * launch cannot raise, so the result is poisoned instead of read out of bounds). */
int gdrnpp_linear_f32_split_grouped(const float* A, const void* W_packed_stack, const float* bias_stack, const int* group_sel,
                                    int n_groups, int rows_per_group, float* C, int M, int N, int K, int n_store, void* stream);
/* Tail of the geometry head on that NHWC result (GDRN_double_mask.py:128-160, conv_pnp_net.py:120-134): out_nhwc
 * f32[b*hw][pitch] with channels [vis | full (double_mask) | x | y | z | region bg, 1..64] ->
 *   pnp_in f32[b*hw][96] = [(xyz - 0.5) * extent | coord2d (f32[b,2,hw]) | softmax(region[1..64]) | 27 zeros]  (Patch-PnP input, NHWC,
 *          Cin padded to a multiple of 32), planes f32[3 + (double_mask ? 2 : 1)][b*hw] = vis, (full,) x, y, z of out_dict. */
int gdrnpp_head_tail_nhwc(const float* out_nhwc, int pitch, const float* coord2d, const float* extents, float* pnp_in,
                          float* planes, int b, int hw, int double_mask, void* stream);

/* 3x3 / stride 1 / zero-pad 1 convolution of the geometry head (a3) as an implicit GEMM on the same kernel:
 * x f32 NHWC [n_img,H,W,Cin] -> y f32 NHWC [n_img,H,W,Cout]; W_packed = gdrnpp_pack_weight_bf16x3 of the weight
 * reordered to [Cout][ky][kx][Cin] (N = Cout, K = 9*Cin); bias may be NULL; epilogue 0 = none, 1 = GELU.
 * Cout multiple of 128, Cin multiple of 32; any n_img*H*W. */
/* General form: KH x KW taps, stride, symmetric zero padding (pad < KH, KW); W_packed = gdrnpp_pack_weight_bf16x3 of the
 * weight reordered to [Cout][ky][kx][Cin].  Used for the 2x2/2 downsample convolutions of ConvNeXt and the 3x3/2
 * convolutions of Patch-PnP as well.  OH = (H + 2 pad - KH) / stride + 1 (likewise OW); y is [n_img,OH,OW,Cout]. */
int gdrnpp_conv2d_f32_split(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                            int n_img, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                            int epilogue, void* stream);
/* Split-K form for launches with too few output tiles for the chip (few ROIs x small maps, K = KH*KW*Cin long): K cut into
 * chunks, partial sums in `workspace` (gdrnpp_conv2d_f32_splitk_workspace_bytes; 0 = the shape runs as one launch and needs
 * none), fixed-order reduction with bias / GELU.  Same arguments and result as gdrnpp_conv2d_f32_split. */
size_t gdrnpp_conv2d_f32_splitk_workspace_bytes(int n_img, int OH, int OW, int Cin, int Cout, int KH, int KW);
int gdrnpp_conv2d_f32_splitk(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc, int n_img, int H, int W,
                             int Cin, int Cout, int KH, int KW, int stride, int pad, int epilogue, void* workspace,
                             size_t workspace_bytes, void* stream);
int gdrnpp_conv3x3_f32_split(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                             int n_img, int H, int W, int Cin, int Cout, int epilogue, void* stream);
/* The ConvModule pair conv3x3 -> GroupNorm of the geometry head (lib/torch_utils/layers/conv_module.py:222-236,
 * top_down_doublemask_xyz_region_head.py:84-107) with the GroupNorm statistics taken in the convolution's epilogue:
 * gn_partials f64[n_img, P, groups, 2] (sum, sum of squares; P = gdrnpp_conv3x3_gnstats_partials(H, W), 0 = shape not
 * supported) is fully written; gdrnpp_groupnorm_apply_nhwc then normalises y (+ affine, optional GELU) from them, the
 * same second pass gdrnpp_groupnorm_act_nhwc runs after its own statistics pass.  Needs H*W % 256 == 0 and
 * Cout == 8 * groups; no activation between convolution and norm. */
/* nn.ConvTranspose2d (the head's first upsampling step, top_down_doublemask_xyz_region_head.py:62-78) = one
 * gdrnpp_linear_f32_split of the NHWC input [n*H*W, Cin] with the weight reordered to [(ky, kx, co), ci]
 * (N = KS*KS*Cout columns) -> cols f32[n, H, W, KS*KS, Cout], then this gather: y f32 NHWC [n, OH, OW, Cout],
 * OH = (H-1)*stride - 2*pad + KS + out_pad; every output pixel adds its <= ceil(KS/stride)^2 taps in (ky, kx) order
 * (+ bias, may be NULL).  Cout % 4 == 0. */
/* y = act(x + bias[c] (+ resid)) over n_pix NHWC pixels of C channels (C % 4 == 0; resid may be NULL; relu 0 / 1; y may be
 * x): the [BatchNorm2d (inference), residual add, ReLU] tail of a ResNet BasicBlock (timm resnet.py BasicBlock.forward)
 * once the BatchNorm is folded into the convolution in front of it (hip_layers.conv_bn_act). */
int gdrnpp_bias_act_nhwc(const float* x, const float* bias, const float* resid, float* y, long n_pix, int C, int relu,
                         void* stream);
int gdrnpp_deconv_col2im_nhwc(const float* cols, const float* bias, float* y, int N, int H, int W, int C, int KS,
                              int stride, int pad, int out_pad, void* stream);
/* the same gather + the GroupNorm(G) statistics of its result (the head's ConvTranspose2d -> GroupNorm -> GELU,
 * top_down_doublemask_xyz_region_head.py:53-75): gn_partials f64[N, P, G, 2] with P = gdrnpp_groupnorm_workspace_bytes(N, OH*OW, G)
 * / (16 N G), bitwise what gdrnpp_groupnorm_act_nhwc's own statistics pass writes; follow with gdrnpp_groupnorm_apply_nhwc.
 * (C / G) % 4 == 0, C / 4 divides 256, G <= 64. */
int gdrnpp_deconv_col2im_gn_nhwc(const float* cols, const float* bias, float* y, double* gn_partials, int N, int H, int W, int C,
                                 int KS, int stride, int pad, int out_pad, int G, void* stream);
int gdrnpp_conv3x3_gnstats_partials(int H, int W);
int gdrnpp_conv3x3_f32_split_gnstats(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc,
                                     double* gn_partials, int n_img, int H, int W, int Cin, int Cout, int groups,
                                     void* stream);
int gdrnpp_groupnorm_apply_nhwc(const float* x, const double* partials, int P, const float* gamma, const float* beta,
                                float* y, int N, int HW, int C, int G, float eps, int act_gelu, void* stream);

/* ---- three-product form of the split GEMM ("fp16x2"): what hip_layers uses by default for large launches (a3) ----------------
 * Every fp32 operand is written as h + l with h = rn_f16(x), l = rn_f16(x - h): 22 significant bits; the products h*l, l*h, h*h
 * go through v_mfma_f32_32x32x16_f16 with fp32 accumulation — half the matrix work of the six-product form above.  The operand
 * representation costs 1.3e-7 .. 2.2e-7 of the output scale, below the fp32 accumulation error all three engines share on
 * realistic operands: against an fp64 product at K = 128 .. 4096 three products 3.9e-7 .. 3.0e-6, six products 4.8e-7 .. 3.0e-6,
 * hipBLASLt fp32 6.2e-7 .. 3.9e-6 (profiles/r03y_split2_accuracy.txt).  Weights are scaled into the fp16 range by an exact power
 * of two when they are packed
 * (gdrnpp_pack_weight_f16x2: W f32[N][K] -> fp16 [N/128][K/16][2][2][128][8] + a 16-byte trailer holding the scale,
 * gdrnpp_pack_weight_f16x2_bytes(N, K) bytes; conv weights reordered to [Cout][ky][kx][Cin] first, as for bf16x3), activations
 * are split as they are.  Both sides of the fp16 range are checked by every launch and reported in its RANGE WORD (range_flag
 * argument, or the library's own word behind gdrnpp_split2_range_word):
 *   GDRNPP_SPLIT2_NONFINITE   an activation beyond 65504 (or a non-finite input): a stored value or an A element was inf / NaN;
 *   GDRNPP_SPLIT2_SMALL_ROWS  an A row that is not all zero has rms below 2^-4 over its K elements: the low halves fall into the
 *                             fp16 subnormal spacing (absolute operand error 2^-25), i.e. more than 2^-21 of that row's output.
 * On either bit the caller repeats the work in the six-product form (engine.inference_step does, and keeps the flagged layer
 * there).  gdrnpp_pack_weight_f16x2 applies the small-rows test to the scaled weight rows and leaves the verdict in the trailer
 * (word 3 of the 16 bytes behind the tiles: 1 = some non-zero row is below range; such a weight belongs on the six-product form).
 * gdrnpp_linear_f32_split2: as gdrnpp_linear_f32_split (N % 128 == 0, K % 32 == 0, any M, M*K*4 < 4 GiB).
 * gdrnpp_conv3x3_f32_split2: 3x3 / stride 1 / pad 1 convolution over NHWC (Cin % 32 == 0, Cout % 128 == 0), epilogue 0 / 1
 * (bias / bias + GELU); gn_partials != NULL additionally writes the GroupNorm partials of gdrnpp_conv3x3_f32_split_gnstats
 * (epilogue 0, H*W % 256 == 0, Cout == 8 * groups). */
size_t gdrnpp_pack_weight_f16x2_bytes(int N, int K);
int gdrnpp_pack_weight_f16x2(const float* W, void* packed, int N, int K, void* stream);
int gdrnpp_linear_f32_split2(const float* A, const void* W_packed, const float* bias, const float* gamma, const float* resid,
                             float* C, int M, int N, int K, int epilogue, int* range_flag, void* stream);
/* "f16x2 rows": an activation tensor f32[M,K] in which every aligned group of 8 consecutive elements of a row has been replaced,
 * in the same 32 bytes, by its 8 fp16 h halves followed by its 8 fp16 l halves (h = rn_f16(x), l = rn_f16(x - h): the operand
 * split of the three-product kernels).  Same shape, strides and size as the fp32 tensor; only a consumer with the flag can read it.
 * A consumer takes its MFMA operands straight from it instead of splitting every k-tile again for every 128-column tile of the
 * output (K = 512, N = 2048: 16 times per element); a producer pays ~3 VALU operations per element once.  Results are bit-identical
 * to the fp32 hand-over (same conversions), range words unchanged (the consumer still sums the squares of the h halves per row).
 * rows: GDRNPP_A_F16X2_ROWS = A is one (epilogue 1 / 2: the two ConvNeXt MLP layers), GDRNPP_C_F16X2_ROWS = write C as one
 * (epilogue 0 / 1; N % 8 == 0 holds by N % 128 == 0).  Producers: this function, gdrnpp_dwconv7x7_ln_nhwc_rows. */
#define GDRNPP_A_F16X2_ROWS 1
#define GDRNPP_C_F16X2_ROWS 2
int gdrnpp_linear_f32_split2_rows(const float* A, const void* W_packed, const float* bias, const float* gamma, const float* resid,
                                  float* C, int M, int N, int K, int epilogue, int rows, int* range_flag, void* stream);
int gdrnpp_conv3x3_f32_split2(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc, double* gn_partials,
                              int n_img, int H, int W, int Cin, int Cout, int groups, int epilogue, int* range_flag,
                              void* stream);
/* any KH x KW / stride / zero-pad convolution with at most 32 taps (ConvNeXt's 2x2/2 downsamples, Patch-PnP's 3x3/2), epilogue 0 / 1 */
int gdrnpp_conv2d_f32_split2(const float* x_nhwc, const void* W_packed, const float* bias, float* y_nhwc, int n_img, int H, int W,
                             int Cin, int Cout, int KH, int KW, int stride, int pad, int epilogue, int* range_flag,
                             void* stream);
/* ConvNeXt block tail  y = resid + gamma * fc2(gelu(fc1(x)))  (timm ConvNeXtBlock.forward behind conv_dw + norm; the reference's
 * backbone module, core/utils/timm_utils.py:34) for the two shallow stages (C = 128, hidden = 512 and C = 256, hidden = 1024) in ONE
 * launch, three-product form:
 * the hidden tensor never reaches HBM (csrc/gemm_mlp_fused.hip: everything computed transposed, x rows split once and held in
 * registers, the GELU'd accumulator tile IS the second GEMM's operand, weights streamed through LDS once per 256 pixels).
 * W_packed: gdrnpp_pack_mlp_fused_f16x2(fc1.weight f32[hidden,C], fc2.weight f32[C,hidden]) — gdrnpp_pack_mlp_fused_f16x2_bytes(C,
 * hidden) bytes (0 = shape not supported); 32-byte trailer behind the tiles {amax1, 2^-e1, 2^e1, rows1, amax2, 2^-e2, 2^e2, rows2}
 * (rows* = 1: a non-zero weight row of that layer is below the range, the layer belongs on the six-product form).
 * x, resid, y f32[M,C] row-major (NHWC pixels); b1 f32[hidden], b2 / gamma f32[C].  range_flag_fc1 / _fc2: the range words of the
 * two layers, both required (x rows are judged into the first, hidden rows and stored values into the second).  Results equal
 * gdrnpp_linear_f32_split2(gelu) + gdrnpp_linear_f32_split2(scale_res) to fp32 rounding (other k order), not bit for bit. */
size_t gdrnpp_pack_mlp_fused_f16x2_bytes(int C, int hidden);
int gdrnpp_pack_mlp_fused_f16x2(const float* W1, const float* W2, void* packed, int C, int hidden, void* stream);
int gdrnpp_convnext_mlp_f32_fused(const float* x, const void* W_packed, const float* b1, const float* b2, const float* gamma,
                                  const float* resid, float* y, int M, int C, int hidden, int* range_flag_fc1, int* range_flag_fc2,
                                  void* stream);
/* range_flag: device int the launch ORs its range word into (the caller owns, zeroes and reads it — one per layer and stream
 * tells the caller WHICH layer left the range and keeps concurrent users apart); NULL = the library's own word, read (*word) and
 * optionally reset by gdrnpp_split2_range_word (synchronises the stream). */
#define GDRNPP_SPLIT2_NONFINITE 1
#define GDRNPP_SPLIT2_SMALL_ROWS 2
int gdrnpp_split2_range_word(int* word, int reset, void* stream);

/* ---- depth-to-flow (SURVEY §8b boundary "flow") — replaces the flow_cuda torch extension,
 * core/csrc/flow/src/flow_cuda.cpp:30-47 (kernel flow_cuda_kernel.cu:33-64).
 * depth_src, depth_tgt f32[B,1,H,W]; KT f32[B,3,4] = K [R|t] (source -> target); Kinv f32[B,3,3]
 * -> flow f32[B,2,H,W] (channel 0 = dh, 1 = dw), valid f32[B,1,H,W]; both fully written. */
int gdrnpp_flow_forward(const float* depth_src, const float* depth_tgt, const float* KT, const float* Kinv,
                        float* flow, float* valid, int B, int H, int W, void* stream);

/* ---- YOLOX detection post-processing — det/yolox/utils/boxes.py:34-74 (`postprocess`):
 * (cx,cy,w,h) -> corners, class max / first argmax, obj*class_conf >= conf_thre, then torchvision.ops.batched_nms
 * (class_agnostic = 0: boxes offset by class * (max_coordinate + 1)) or nms (class_agnostic = 1), restated.
 * det_preds f32[B,A,5+C] (read only; the reference overwrites its first four columns in place)
 * -> out_dets f32[B,max_det,7] = (x1,y1,x2,y2,obj_conf,class_conf,class) in keep order (descending score, ties by
 * anchor index), out_count i32[B] = number kept (may exceed max_det: only the first max_det rows are written).
 * A <= 16384.  workspace: gdrnpp_yolox_postprocess_workspace_bytes(B, A) bytes of device memory. */
size_t gdrnpp_yolox_postprocess_workspace_bytes(int B, int A);
int gdrnpp_yolox_postprocess(const float* det_preds, int B, int A, int C, float conf_thre, float nms_thre,
                             int class_agnostic, float* out_dets, int* out_count, int max_det,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- instance masks for SAVE_RESULTS_ONLY — gdrn_evaluator.py:914-945:
 * detectron2 paste_masks_in_image(mask_probs, boxes, (im_H, im_W), threshold) + the uncompressed COCO run-length
 * encoding of lib/utils/mask_utils.py:96-109, fused: the full-size masks are never materialised.
 * mask_probs f32[B,mask_h,mask_w]; boxes_xyxy f32[B,4] (x0,y0,x1,y1 = roi_center -/+ scale/2)
 * -> counts u32[B,max_runs] (column-major runs, the first run counts zeros), n_runs i32[B] (> max_runs = overflow:
 * re-run with a larger max_runs; im_H*im_W+1 always suffices). */
int gdrnpp_paste_masks_rle(const float* mask_probs, const float* boxes_xyxy, int B, int mask_h, int mask_w,
                           int im_H, int im_W, float threshold, unsigned* counts, int* n_runs, int max_runs,
                           void* stream);

/* ---- pose record packing for the RCCL all-gather (a13) -------------------
 * rec f32[b,16] = R(9) | t(3) | score | obj_id | roi_id | valid(1) */
int gdrnpp_pack_pose_records(const float* R, const double* t_refined,
                             const float* t_net, const float* score,
                             const int* obj_id, const int* roi_id, float* rec,
                             int b, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GDRNPP_HIP_H_ */
