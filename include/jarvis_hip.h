/* libjarvis_hip.so -- C ABI of the MI355X (gfx950) implementation of the
 * JARVIS-HybridNet multi-view inference hot path.
 *
 * This is the drop-in boundary: plain C, raw device pointers, explicit shapes,
 * an explicit hipStream_t (passed as void*), int status (0 = OK; the message of
 * the last failure on the calling thread is returned by jh_last_error()).
 * No forward entry point synchronises the device or allocates memory: networks and
 * predictors own their intermediates (allocated by *_create), the stand-alone
 * operators (jh_reproject_forward, jh_softargmax, jh_reconstruct_point) take a
 * caller-provided device workspace sized by the matching jh_*_workspace_bytes().
 * A whole forward may therefore be captured into a hipGraph
 * (tests/test_hip_stages.py::test_submodule_path_graph_capture).  The only
 * exceptions are the jh_op_* unit-test helpers at the end of this file, which say so.
 *
 * The library occupies the seam the reference itself uses for acceleration:
 * JarvisPredictor3D replaces three sub-networks by compiled modules after
 * torch.ops.load_library(<native .so>)  (jarvis/prediction/jarvis3D.py:50-69,
 * 72-125, binary converters under libs/).  Each entry point below cites the
 * reference function it replaces.
 *
 * Conventions: all tensors fp32 unless noted; "dev" = device pointer, "host" =
 * host pointer; NCHW / NCDHW = the reference's layouts; calibration in the
 * reference's transposed storage (jarvis/utils/reprojection.py:33-39):
 * cameraMatrices (C,4,3), intrinsicMatrices (C,3,3) with the principal point in
 * row 2, distortionCoefficients (C,1,5) with k1,k2 first.
 */
#ifndef JARVIS_HIP_H
#define JARVIS_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define JH_ABI_VERSION 4

const char* jh_last_error(void);
int jh_abi_version(void);

/* ---- precision mode.  A predictor carries its own (jh_predictor_config.precision, ABI v4): two
 * predictors of different precision coexist in one process and do not depend on call order.
 * jh_set_precision() only sets the process-wide DEFAULT: what the stand-alone networks
 * (jh_efftrack_create, jh_v2v_create, jh_op_*) CREATED from now on use, and what a predictor
 * configured with JH_PRECISION_DEFAULT picks up when it is created.
 * JH_PRECISION_F32 (default): fp32 products and accumulation everywhere, the mode every parity
 * figure of this library is quoted in.  JH_PRECISION_BF16X3: the 3x3x3 stride-1 convolutions of
 * V2V, its stride-2 front convolution and the keypoint head's ConvTranspose2d run on the bf16 matrix
 * cores with each fp32 operand split into two bf16 terms (three MFMAs per product, fp32 accumulation;
 * about 2^-16 relative per product) -- the labelled
 * reduced-precision mode that stands where the reference has its half-precision TensorRT
 * engines (jarvis/prediction/jarvis3D.py:93,107,122: enabled_precisions={torch.half}).
 * JH_PRECISION_BF16X3_WIDE additionally splits the dense k x k convolutions of the EfficientNet trunk
 * (experimental: up to 7.6e-4 mm on the fixture cases, no margin under the 1e-3 mm bar).
 * Environment JH_PRECISION=bf16x3 / bf16x3_wide sets the initial mode. */
#define JH_PRECISION_DEFAULT (-1) /* jh_predictor_config.precision only: follow jh_set_precision / JH_PRECISION */
#define JH_PRECISION_F32 0
#define JH_PRECISION_BF16X3 1
#define JH_PRECISION_BF16X3_WIDE 2
int jh_set_precision(int mode);
int jh_get_precision(void);

/* ---- parameters: a state dict in the reference's .pth key layout ----------
 * Replaces torch.load + load_state_dict of jarvis/hybridnet/hybridnet.py:84-97
 * and jarvis/efficienttrack/efficienttrack.py:90-113 on the native side. */
typedef struct jh_params jh_params;
int jh_params_create(jh_params** out);
int jh_params_set(jh_params* p, const char* key, const float* host, int64_t numel);
void jh_params_destroy(jh_params* p);

/* ---- EfficientTrackBackbone.forward  (jarvis/efficienttrack/model.py:114-130)
 * model_size: 0 small, 1 medium, 2 large.  Built for a fixed (N,3,H,W) input.
 * forward: x (N,3,H,W) NCHW dev -> res1 (N,J,H/4,W/4), res2 (N,J,H/2,W/2) NCHW dev:
 * the (res1, res2) tuple of model.py:126-130, i.e. the tensor contract of the
 * reference's trt_mode seam (jarvis3D.py:64-69).  res1 (final_conv1, model.py:128) is
 * never read on the inference path (hybridnet/model.py:57-58, jarvis3D.py:147): it is
 * computed only by networks created with want_res1 != 0, and res1_dev may be NULL. */
typedef struct jh_efftrack jh_efftrack;
int jh_efftrack_create(const jh_params* p, const char* prefix, int model_size, int joints, int n,
                       int h, int w, int want_res1, jh_efftrack** out);
int jh_efftrack_forward(jh_efftrack* net, const float* x_dev, float* res1_dev, float* res2_dev,
                        void* stream);
int64_t jh_efftrack_launches(const jh_efftrack* net);
void jh_efftrack_destroy(jh_efftrack* net);

/* ---- V2VNet.forward  (jarvis/hybridnet/v2vnet.py:98-102)
 * x (T,J,G,G,G) NCDHW dev -> y (T,J,G/2,G/2,G/2) NCDHW dev. */
typedef struct jh_v2v jh_v2v;
int jh_v2v_create(const jh_params* p, const char* prefix, int joints, int t, int g, jh_v2v** out);
int jh_v2v_forward(jh_v2v* net, const float* x_dev, float* y_dev, void* stream);
void jh_v2v_destroy(jh_v2v* net);

/* ---- ReprojectionLayer.forward  (jarvis/hybridnet/repro_layer.py:110-119)
 * heatmaps_padded (1,C,J,hs,hs) NCHW dev, center3d (3) int32 dev, center_hm
 * (C,2) int32 dev, calibration dev -> vol (1,J,G,G,G) NCDHW dev (NOT divided by
 * 255, like the reference layer).  idx_dev (C,G,G,G) int32 optional (may be
 * NULL): the reference's integer gather index (reprojectPoints, :40-85).
 * workspace_dev: >= jh_reproject_workspace_bytes(cams, joints, hs, grid_size) bytes of
 * device memory (256-byte aligned), owned by the caller, contents irrelevant. */
int64_t jh_reproject_workspace_bytes(int cams, int joints, int hs, int grid_size);
int jh_reproject_forward(const float* heatmaps_padded_dev, int cams, int joints, int hs,
                         const int32_t* center3d_dev, const int32_t* center_hm_dev,
                         const float* cam_dev, const float* intr_dev, const float* dist_dev,
                         int grid_size, float grid_spacing, float* vol_dev, int32_t* idx_dev,
                         void* workspace_dev, int64_t workspace_bytes, void* stream);

/* ---- soft-argmax tail of HybridNetBackbone.forward (hybridnet/model.py:73-88)
 * v2v_out (T,J,Gh,Gh,Gh) NCDHW dev, center3d (T,3) int32 dev -> points (T,J,3),
 * conf (T,J), heatmap_final (T,J,Gh,Gh,Gh) optional (NULL to skip).
 * workspace_dev: >= jh_softargmax_workspace_bytes(t, joints, gh) bytes, as above. */
int64_t jh_softargmax_workspace_bytes(int t, int joints, int gh);
int jh_softargmax(const float* v2v_out_dev, int t, int joints, int gh, float grid_spacing,
                  float roi_cube_size, const int32_t* center3d_dev, float* heatmap_final_dev,
                  float* points_dev, float* conf_dev, void* workspace_dev, int64_t workspace_bytes,
                  void* stream);

/* ---- ReprojectionTool  (jarvis/utils/reprojection.py:49-90)
 * reproject: points (P,3) dev -> uv (C,P,2) dev.
 * reconstruct: points2d (2,C) dev (pixels), maxvals (C) dev -> point3d (3) dev;
 * workspace_dev: >= jh_reconstruct_workspace_bytes(cams) bytes, as above. */
int jh_reproject_point(const float* points_dev, int npoints, int cams, const float* cam_dev,
                       const float* intr_dev, const float* dist_dev, float* uv_dev, void* stream);
int64_t jh_reconstruct_workspace_bytes(int cams);
int jh_reconstruct_point(const float* points2d_dev, const float* maxvals_dev, int cams,
                         const float* cam_dev, const float* intr_dev, const float* dist_dev,
                         float* point3d_dev, void* workspace_dev, int64_t workspace_bytes,
                         void* stream);

/* ---- JarvisPredictor3D  (jarvis/prediction/jarvis3D.py:20-46,129-190) and
 * HybridNetBackbone.forward (jarvis/hybridnet/model.py:53-90).
 * One object owns both 2D networks, the V2V network and every intermediate.
 * time_batch T independent multi-view frames are processed per call (the
 * reference's batch-1 call is T = 1).  cam_lo/cam_n select the cameras whose 2D
 * work this process owns (camera sharding across GPUs); the 3D stage always
 * sees all cameras. */
typedef struct jh_predictor jh_predictor;
typedef struct {
  int32_t num_cameras, num_joints;
  int32_t center_size;        /* CENTERDETECT.IMAGE_SIZE */
  int32_t bbox;               /* KEYPOINTDETECT.BOUNDING_BOX_SIZE */
  float roi_cube_size;        /* HYBRIDNET.ROI_CUBE_SIZE (mm) */
  float grid_spacing;         /* HYBRIDNET.GRID_SPACING (mm) */
  int32_t center_model, kp_model;   /* 0 small, 1 medium, 2 large */
  int32_t img_h, img_w;
  int32_t time_batch;         /* T: frames per call through the 2D stages */
  int32_t time_batch_3d;      /* frames per stage_3d call (<= T; 0 means T) */
  int32_t cam_lo, cam_n;
  float mean[3], std[3];      /* DATASET.MEAN / DATASET.STD */
  int32_t precision;          /* JH_PRECISION_F32 (0: what a zero-initialised struct gets), _BF16X3, _BF16X3_WIDE,
                               * or JH_PRECISION_DEFAULT = the process default at creation time.  The mode that
                               * stands where the reference has trt_mode != 'off' (jarvis3D.py:42-46,93,107,122) */
} jh_predictor_config;

/* center_params may be NULL (HybridNetBackbone-only use). */
int jh_predictor_create(const jh_params* center_params, const jh_params* hybrid_params,
                        const jh_predictor_config* cfg, jh_predictor** out);
void jh_predictor_destroy(jh_predictor* pr);
int64_t jh_predictor_launches(const jh_predictor* pr);
int64_t jh_predictor_device_bytes(const jh_predictor* pr);
int jh_predictor_precision(const jh_predictor* pr);    /* the resolved mode (never JH_PRECISION_DEFAULT) */

/* calibration of all cameras (device pointers, copied). */
int jh_predictor_set_calibration(jh_predictor* pr, const float* cam_dev, const float* intr_dev,
                                 const float* dist_dev, void* stream);

/* Stage 1 (jarvis3D.py:135-155): resize + normalise + CenterDetect + argmax for
 * the owned cameras.  frames (T,cam_n,3,H,W) dev -> det (T,cam_n,3) = (x, y,
 * raw maxval) dev. */
int jh_predictor_stage_center(jh_predictor* pr, const float* frames_dev, float* det_dev,
                              void* stream);
/* Stage 2 (jarvis3D.py:157-178 + model.py:55-63): det_all (T,C,3) of ALL cameras
 * -> triangulate, project, crop + normalise + KeypointDetect for the owned
 * cameras -> heat (T,cam_n,B/2,B/2,Jp) channel-last dev, Jp = joints rounded up
 * to 8. */
int jh_predictor_stage_keypoints(jh_predictor* pr, const float* frames_dev,
                                 const float* det_all_dev, float* heat_dev, void* stream);
/* Stage 3 (model.py:65-88) for the time_batch_3d frames t0 .. t0+T3-1 of the
 * batch: heat_all (T3,C,B/2,B/2,Jp) of ALL cameras for those frames -> points
 * (T3,J,3), conf (T3,J), valid (T3) int32 (0 = fewer than two cameras saw the
 * subject: the reference returns (None, None), jarvis3D.py:187-190).
 * Pipelining: a predictor keeps TWO sets of (crop centres, truncated 3D centre, validity).  Every
 * stage-2 call (jh_predictor_stage_keypoints*) writes the set the previous stage-2 call did not; a
 * stage-3 call reads the set of the stage-2 call that preceded it in HOST CALL ORDER.  Stage 3 of
 * time batch i may therefore run on a second stream concurrently with stage 2 of batch i+1 -- they
 * share no buffer -- provided the caller orders (events) stage 3 of batch i before stage 2 of batch
 * i+2 and keeps heat_all alive until stage 3 has read it (jarvis_hybridnet_amd/distributed.py). */
int jh_predictor_stage_3d(jh_predictor* pr, const float* heat_all_dev, int t0, float* points_dev,
                          float* conf_dev, int32_t* valid_dev, void* stream);
/* Stage 2 fed directly from the all-gather of the per-rank detections (new design, no
 * reference line): det_gathered is (n_blocks, T, C / n_blocks, 3), block b = the det output of
 * the rank that owns cameras [b*C/n_blocks, (b+1)*C/n_blocks).  frames_u8 != 0: uint8 BGR
 * frames as for the *_u8 entry points.  n_blocks = 1 is jh_predictor_stage_keypoints[_u8]. */
int jh_predictor_stage_keypoints_gathered(jh_predictor* pr, const void* frames_dev, int frames_u8,
                                          const float* det_gathered_dev, int n_blocks,
                                          float* heat_dev, void* stream);
/* Stage 3 reading the heatmaps IN PLACE from the receive buffer of the camera-sharded
 * exchange (jarvis_hybridnet_amd/distributed.py; new design, no reference line): heat_blocks
 * is (n_blocks, frames_per_block, C / n_blocks, B/2, B/2, Jp) -- block b holds cameras
 * [b*C/n_blocks, (b+1)*C/n_blocks) as source rank b produced them -- and the T3 frames
 * t_off .. t_off+T3-1 of every block are the frames t0 .. t0+T3-1 of the batch.
 * n_blocks = 1, frames_per_block = T3, t_off = 0 is jh_predictor_stage_3d. */
int jh_predictor_stage_3d_blocks(jh_predictor* pr, const float* heat_blocks_dev, int n_blocks,
                                 int frames_per_block, int t_off, int t0, float* points_dev,
                                 float* conf_dev, int32_t* valid_dev, void* stream);
/* All three stages for cam_lo = 0, cam_n = num_cameras. */
int jh_predictor_forward(jh_predictor* pr, const float* frames_dev, float* points_dev,
                         float* conf_dev, int32_t* valid_dev, void* stream);
/* jh_predictor_forward[_u8] as ONE hipGraph launch.  The reference driver calls the predictor
 * with one frame set at a time (jarvis/prediction/predict3D.py:82-85); at that size the ~150
 * launches of a forward are launch-bound, so a predictor with time_batch == 1 captures its
 * forward on first use and replays the graph afterwards (the call's frame pointer goes through a
 * device cell, results are copied out of the predictor's buffers: any pointers may be passed on
 * every call, results are bit-identical to the plain launches).  Default: on for time_batch
 * == 1, off otherwise (environment JH_GRAPH=0 / 1 overrides for the process); this switches it
 * per predictor.  Not used while jh_profile_begin() is active or while the caller's stream is
 * itself being captured. */
int jh_predictor_set_graph_replay(jh_predictor* pr, int on);
int jh_predictor_graph_replay(const jh_predictor* pr);
/* uint8 ingest (jarvis/prediction/predict3D.py:72-80): the
 * same three entry points for frames (T,cam_n,H,W,3) uint8 BGR exactly as the video
 * decoder delivers them.  The `.float().permute(0,3,1,2)[:, [2,1,0]] / 255.` of the
 * reference driver happens inside the resize / crop kernels, so the fp32 frame (4x the
 * bytes) is never materialised and only 1 byte per sample crosses PCIe. */
int jh_predictor_stage_center_u8(jh_predictor* pr, const uint8_t* frames_dev, float* det_dev,
                                 void* stream);
int jh_predictor_stage_keypoints_u8(jh_predictor* pr, const uint8_t* frames_dev,
                                    const float* det_all_dev, float* heat_dev, void* stream);
int jh_predictor_forward_u8(jh_predictor* pr, const uint8_t* frames_dev, float* points_dev,
                            float* conf_dev, int32_t* valid_dev, void* stream);

/* Integer path of the last call, for parity tests: center3d float (T,3),
 * center3d int (T,3), center_hm (T,C,2), det (T,C,3).  Any pointer may be NULL. */
int jh_predictor_debug(jh_predictor* pr, float* center3d_f_dev, int32_t* center3d_i_dev,
                       int32_t* center_hm_dev, float* det_dev, void* stream);

/* HybridNetBackbone.forward: crops (T,C,3,B,B) normalised NCHW dev, center_hm
 * (T,C,2) int32, center3d (T,3) int32 -> heatmap_final (T,J,Gh,Gh,Gh) optional,
 * heatmaps_padded (T,C,J,hs,hs) optional, points (T,J,3), conf (T,J). */
int jh_predictor_hybridnet_forward(jh_predictor* pr, const float* crops_dev,
                                   const int32_t* center_hm_dev, const int32_t* center3d_dev,
                                   float* heatmap_final_dev, float* heatmaps_padded_dev,
                                   float* points_dev, float* conf_dev, void* stream);

/* ---- JarvisPredictor2D  (jarvis/prediction/jarvis2D.py:20-44,102-155)
 * Single-camera 2D pose: resize -> CenterDetect -> argmax -> crop -> KeypointDetect ->
 * per-joint argmax.  `time_batch` independent images per call (the reference's call is
 * 1); num_cameras / roi / spacing / cam_* of the config are ignored.
 * frames (T,3,H,W) fp32 RGB [or (T,H,W,3) uint8 BGR] -> points2D (T,J,2) int32 full-frame
 * pixels, conf (T,J), valid (T) int32 (0 = centre maxval <= 40: the reference returns
 * (None, None), jarvis2D.py:121,150-153). */
typedef struct jh_predictor2d jh_predictor2d;
int jh_predictor2d_create(const jh_params* center_params, const jh_params* kp_params,
                          const jh_predictor_config* cfg, jh_predictor2d** out);
void jh_predictor2d_destroy(jh_predictor2d* pr);
int jh_predictor2d_forward(jh_predictor2d* pr, const float* frames_dev, int32_t* points_dev,
                           float* conf_dev, int32_t* valid_dev, void* stream);
int jh_predictor2d_forward_u8(jh_predictor2d* pr, const uint8_t* frames_dev, int32_t* points_dev,
                              float* conf_dev, int32_t* valid_dev, void* stream);

/* ---- per-launch timing (HIP events on the launch stream; used by bench.py for
 * the roofline figures).  begin() switches recording on for every kernel the
 * library launches from this process; end() synchronises and returns the number
 * of records; get() returns name, milliseconds, algorithmic FLOPs and bytes. */
int jh_profile_begin(void);
int jh_profile_end(int* n_records);
int jh_profile_get(int i, char* name, int name_cap, double* ms, double* flops, double* bytes);

/* ---- single-operator entry points (unit-test helpers, NOT part of the forward path:
 * they repack host weights, allocate scratch with hipMalloc and end in a stream
 * synchronisation, so they are neither asynchronous nor graph-capturable)
 * conv: x (N,Cin,[D,]H,W) -> y; weights/bias are HOST pointers in torch layout
 * ((Cout,Cin,k..) or, transposed, (Cin,Cout,k..)); kind 0 = conv (k, stride,
 * pad), 1 = ConvTranspose2d k4 s2 p1, 2 = ConvTranspose3d k2 s2.  When
 * norm_act >= 0 the InstanceNorm (+ activation 0 none / 1 relu / 2 silu) that
 * follows the conv in the networks is applied from the fused statistics. */
int jh_op_conv(int nd, int kind, int k, int stride, int pad, int cin, int cout,
               const float* w_host, const float* b_host, const float* x_dev, int n, int d, int h,
               int w, const float* gate_dev, int norm_act, float* y_dev, void* stream);
/* depthwise k x k stride 1: x (N,C,H,W), w_host (C,1,k,k) -> y. */
int jh_op_depthwise(int k, int c, const float* w_host, const float* x_dev, int n, int h, int w,
                    int norm_act, float* y_dev, void* stream);
/* ... with the squeeze-excite pooled sums of the same launch (MBConvBlock.forward,
 * jarvis/efficienttrack/efficientnet.py:100-107: _depthwise_conv -> _gn1 -> swish -> adaptive_avg_pool2d): h, w <= 16;
 * y_dev (N,C,h,w) RAW depthwise output, pool_dev (N,C) = sum over pixels of SiLU(InstanceNorm(y)). */
int jh_op_depthwise_pool(int k, int c, const float* w_host, const float* x_dev, int n, int h, int w,
                         float* y_dev, float* pool_dev, void* stream);

/* One fused BiFPN node (jarvis/efficienttrack/model.py:301-353 fusion expressions + :223-232
 * SeparableConvBlock.forward, without its trailing InstanceNorm):
 *   y = pointwise(depthwise3x3(act(sum_i weights[i] * resample_i(InstanceNorm(x_i))))) + bias
 * x_i (N,C,h_i,w_i) NCHW dev at the resolution its mode implies (0 same, 1 nearest x2 up from
 * (h/2,w/2), 2 nearest x4 up from (h/4,w/4), 3 2x2 max-pool from (2h,2w)); modes / weights host
 * arrays of 3; act 0 none / 2 SiLU; dw_host (C,1,3,3), pw_host (Cout,C), bias_host (Cout). */
int jh_op_bifpn_node(int n_in, const int* modes, const float* weights, int act, int n, int c, int cout,
                     int h, int w, const float* x0_dev, const float* x1_dev, const float* x2_dev,
                     const float* dw_host, const float* pw_host, const float* bias_host, float* y_dev,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif /* JARVIS_HIP_H */
