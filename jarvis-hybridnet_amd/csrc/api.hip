// C ABI of libjarvis_hip.so (declarations + reference citations: include/jarvis_hip.h)
#include <cstring>
#include <memory>
#include "../../include/jarvis_hip.h"
#include <cstdlib>
#include "nets.h"
#include "bifpn_node.h"

namespace jh {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace jh

using namespace jh;

struct jh_params { ParamMap map; };
struct jh_efftrack { EffTrackPlan plan; };
struct jh_v2v { V2VPlan plan; };

namespace {

// RAII device scratch for the non-pipelined (test / API-parity) entry points.
struct Scratch {
  std::vector<void*> ptrs;
  ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
  int get(void** p, size_t bytes) {
    JH_CHECK_HIP(hipMalloc(p, bytes));
    ptrs.push_back(*p);
    return 0;
  }
  int act(int N, int D, int H, int W, int C, Act* a) {
    a->N = N; a->D = D; a->H = H; a->W = W; a->C = C; a->Cp = cpad(C);
    return get(reinterpret_cast<void**>(&a->p), a->bytes());
  }
};

// Caller-provided workspace, carved into 256-byte aligned pieces.  With base == nullptr it
// only measures (jh_*_workspace_bytes).
struct Carver {
  char* base;
  size_t cap, off = 0;
  Carver(void* b, size_t c) : base(static_cast<char*>(b)), cap(c) {}
  template <typename T> T* take(size_t count) {
    const size_t at = off;
    off += (count * sizeof(T) + 255) / 256 * 256;
    return base ? reinterpret_cast<T*>(base + at) : nullptr;
  }
  void act(int N, int D, int H, int W, int C, Act* a) {
    a->N = N; a->D = D; a->H = H; a->W = W; a->C = C; a->Cp = cpad(C);
    a->p = take<float>(a->elems());
  }
  bool fits() const { return off <= cap; }
};

struct ReproWs { Act heat, vol; float2* coarse; };
static size_t carve_reproject(Carver& c, int cams, int joints, int hs, int g, ReproWs* w) {
  const int gh = g / 2;
  c.act(cams, 1, hs, hs, joints, &w->heat);
  c.act(1, g, g, g, joints, &w->vol);
  w->coarse = c.take<float2>((size_t)cams * gh * gh * gh);
  return c.off;
}
struct SoftWs { Act x; double* partial; int* pmax; };
static size_t carve_softargmax(Carver& c, int t, int joints, int gh, SoftWs* w) {
  c.act(t, gh, gh, gh, joints, &w->x);
  w->partial = c.take<double>((size_t)t * w->x.Cp * 4 * kLimbs);
  w->pmax = c.take<int>((size_t)t * w->x.Cp);
  return c.off;
}
struct ReconWs { float* det; int *c3i, *chm, *valid; };
static size_t carve_reconstruct(Carver& c, int cams, ReconWs* w) {
  w->det = c.take<float>((size_t)cams * 3);
  w->c3i = c.take<int>(3);
  w->chm = c.take<int>((size_t)cams * 2);
  w->valid = c.take<int>(1);
  return c.off;
}

// (T,C) channel-last heatmaps [N][Hh][Hh][Jp] -> padded NCHW (N,J,hs,hs)
__global__ void export_padded_kernel(const float* __restrict__ heat, float* __restrict__ out, int N,
                                     int J, int Jp, int Hh) {
  const int hs = Hh + 2;
  const size_t total = (size_t)N * J * hs * hs;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int x = (int)(i % hs), y = (int)((i / hs) % hs);
    const int j = (int)((i / ((size_t)hs * hs)) % J);
    const size_t n = i / ((size_t)hs * hs * J);
    float v = 0.f;
    if (x >= 1 && y >= 1 && x <= Hh && y <= Hh)
      v = heat[((n * Hh + (y - 1)) * Hh + (x - 1)) * Jp + j];
    out[i] = v;
  }
}

// all-gathered detections (blocks, T, C/blocks, 3) -> (T, C, 3), camera = block * C/blocks + local
__global__ void det_unblock_kernel(const float* __restrict__ src, float* __restrict__ dst, int T, int C,
                                   int cpb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * C * 3) return;
  const int k = i % 3, c = (i / 3) % C, t = i / (3 * C);
  dst[i] = src[(((size_t)(c / cpb) * T + t) * cpb + c % cpb) * 3 + k];
}

// graph replays: this call's frame pointer into the device cell the captured kernels read it from
__global__ void set_cell_kernel(const void** cell, const void* value) { *cell = value; }
// ... and the results out of the predictor's own buffers into this call's output tensors
__global__ void copy_out_kernel(const float* __restrict__ gp, const float* __restrict__ gc,
                                const int* __restrict__ gv, float* __restrict__ points,
                                float* __restrict__ conf, int* __restrict__ valid, int n_pts, int n_conf,
                                int n_valid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pts) points[i] = gp[i];
  if (i < n_conf) conf[i] = gc[i];
  if (valid && i < n_valid) valid[i] = gv[i];
}

__global__ void pack_det_kernel(const float* __restrict__ pts2d, const float* __restrict__ maxvals,
                                float* __restrict__ det, int C) {
  const int c = threadIdx.x;
  if (c < C) {
    det[c * 3 + 0] = pts2d[c];
    det[c * 3 + 1] = pts2d[C + c];
    det[c * 3 + 2] = maxvals[c];
  }
}

}  // namespace

extern "C" {

const char* jh_last_error(void) { return g_err.c_str(); }
int jh_abi_version(void) { return JH_ABI_VERSION; }

int jh_set_precision(int mode) {
  JH_REQUIRE(mode == JH_PRECISION_F32 || mode == JH_PRECISION_BF16X3 || mode == JH_PRECISION_BF16X3_WIDE,
             "unknown precision mode");
  set_precision_mode(mode);
  return 0;
}
int jh_get_precision(void) { return precision_mode(); }

int jh_params_create(jh_params** out) {
  *out = new jh_params();
  return 0;
}
int jh_params_set(jh_params* p, const char* key, const float* host, int64_t numel) {
  JH_REQUIRE(p && key && host && numel >= 0, "bad argument");
  p->map[key] = std::vector<float>(host, host + numel);
  return 0;
}
void jh_params_destroy(jh_params* p) { delete p; }

// ---------------------------------------------------------------- EfficientTrack
int jh_efftrack_create(const jh_params* p, const char* prefix, int model_size, int joints, int n,
                       int h, int w, int want_res1, jh_efftrack** out) {
  JH_REQUIRE(p && out, "bad argument");
  std::unique_ptr<jh_efftrack> net(new jh_efftrack());
  if (net->plan.build(p->map, prefix ? prefix : "", model_size, joints, n, h, w, want_res1 != 0)) return 1;
  *out = net.release();
  return 0;
}
int jh_efftrack_forward(jh_efftrack* net, const float* x_dev, float* res1_dev, float* res2_dev,
                        void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(net && x_dev && res2_dev, "bad argument");
  JH_REQUIRE(!res1_dev || net->plan.res1.p, "res1 requested from a network created with want_res1 = 0");
  if (launch_to_channel_last(x_dev, net->plan.input, s)) return 1;
  if (net->plan.run(s)) return 1;
  if (res1_dev && launch_from_channel_last(net->plan.res1, res1_dev, s)) return 1;
  return launch_from_channel_last(net->plan.heat, res2_dev, s);
}
int64_t jh_efftrack_launches(const jh_efftrack* net) { return (int64_t)net->plan.launches(); }
void jh_efftrack_destroy(jh_efftrack* net) { delete net; }

// --------------------------------------------------------------------------- V2V
int jh_v2v_create(const jh_params* p, const char* prefix, int joints, int t, int g, jh_v2v** out) {
  JH_REQUIRE(p && out, "bad argument");
  std::unique_ptr<jh_v2v> net(new jh_v2v());
  if (net->plan.build(p->map, prefix ? prefix : "", joints, t, g)) return 1;
  *out = net.release();
  return 0;
}
int jh_v2v_forward(jh_v2v* net, const float* x_dev, float* y_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (launch_to_channel_last(x_dev, net->plan.input, s)) return 1;
  if (net->plan.run(s)) return 1;
  return launch_from_channel_last(net->plan.output, y_dev, s);
}
void jh_v2v_destroy(jh_v2v* net) { delete net; }

// ------------------------------------------------------------------ reprojection
int64_t jh_reproject_workspace_bytes(int cams, int joints, int hs, int grid_size) {
  Carver c(nullptr, 0);
  ReproWs w;
  return (int64_t)carve_reproject(c, cams, joints, hs, grid_size, &w);
}

int jh_reproject_forward(const float* heatmaps_padded_dev, int cams, int joints, int hs,
                         const int32_t* center3d_dev, const int32_t* center_hm_dev,
                         const float* cam_dev, const float* intr_dev, const float* dist_dev,
                         int grid_size, float grid_spacing, float* vol_dev, int32_t* idx_dev,
                         void* workspace_dev, int64_t workspace_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(workspace_dev && workspace_bytes >= 0, "workspace (see jh_reproject_workspace_bytes)");
  JH_REQUIRE(grid_size > 0 && grid_size % 2 == 0, "grid size must be even");
  Carver c(workspace_dev, (size_t)workspace_bytes);
  ReproWs w;
  carve_reproject(c, cams, joints, hs, grid_size, &w);
  JH_REQUIRE(c.fits(), "workspace smaller than jh_reproject_workspace_bytes()");
  if (launch_to_channel_last(heatmaps_padded_dev, w.heat, s)) return 1;
  if (launch_reproject(cam_dev, intr_dev, dist_dev, center3d_dev, center_hm_dev, w.heat.p, w.coarse,
                       w.vol.p, idx_dev, 1, cams, grid_size, grid_spacing, hs, w.heat.Cp,
                       /*heat_pad=*/1, /*div255=*/0, s)) return 1;
  return launch_from_channel_last(w.vol, vol_dev, s);
}

int64_t jh_softargmax_workspace_bytes(int t, int joints, int gh) {
  Carver c(nullptr, 0);
  SoftWs w;
  return (int64_t)carve_softargmax(c, t, joints, gh, &w);
}

int jh_softargmax(const float* v2v_out_dev, int t, int joints, int gh, float grid_spacing,
                  float roi_cube_size, const int32_t* center3d_dev, float* heatmap_final_dev,
                  float* points_dev, float* conf_dev, void* workspace_dev, int64_t workspace_bytes,
                  void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(workspace_dev && workspace_bytes >= 0, "workspace (see jh_softargmax_workspace_bytes)");
  Carver c(workspace_dev, (size_t)workspace_bytes);
  SoftWs w;
  carve_softargmax(c, t, joints, gh, &w);
  JH_REQUIRE(c.fits(), "workspace smaller than jh_softargmax_workspace_bytes()");
  if (launch_to_channel_last(v2v_out_dev, w.x, s)) return 1;
  return launch_softargmax(w.x.p, center3d_dev, w.partial, w.pmax, points_dev, conf_dev,
                           heatmap_final_dev, t, joints, w.x.Cp, gh, grid_spacing, roi_cube_size, s);
}

int jh_reproject_point(const float* points_dev, int npoints, int cams, const float* cam_dev,
                       const float* intr_dev, const float* dist_dev, float* uv_dev, void* stream) {
  return launch_project_points(points_dev, cam_dev, intr_dev, dist_dev, uv_dev, npoints, cams,
                               static_cast<hipStream_t>(stream));
}

int64_t jh_reconstruct_workspace_bytes(int cams) {
  Carver c(nullptr, 0);
  ReconWs w;
  return (int64_t)carve_reconstruct(c, cams, &w);
}

int jh_reconstruct_point(const float* points2d_dev, const float* maxvals_dev, int cams,
                         const float* cam_dev, const float* intr_dev, const float* dist_dev,
                         float* point3d_dev, void* workspace_dev, int64_t workspace_bytes,
                         void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(cams >= 1 && cams <= 64, "at most 64 cameras");
  JH_REQUIRE(workspace_dev && workspace_bytes >= 0, "workspace (see jh_reconstruct_workspace_bytes)");
  Carver c(workspace_dev, (size_t)workspace_bytes);
  ReconWs w;
  carve_reconstruct(c, cams, &w);
  JH_REQUIRE(c.fits(), "workspace smaller than jh_reconstruct_workspace_bytes()");
  hipLaunchKernelGGL(pack_det_kernel, dim3(1), dim3(64), 0, s, points2d_dev, maxvals_dev, w.det, cams);
  JH_CHECK_HIP(hipGetLastError());
  return launch_triangulate(w.det, cam_dev, intr_dev, dist_dev, point3d_dev, w.c3i, w.chm, w.valid, 1,
                            cams, 1.f, 1.f, 1.f, 0, 1 << 20, 1 << 20, s);
}

}  // extern "C"

// ------------------------------------------------------------------- predictor
// time one glue launch when the profiler is on
#define JH_PROF(name, flops, bytes, call)                      \
  do {                                                         \
    Profiler& _pf = profiler();                                \
    if (_pf.on) _pf.begin(name, flops, bytes, s);              \
    if (call) return 1;                                        \
    if (_pf.on) _pf.end(s);                                    \
  } while (0)

struct jh_predictor {
  jh_predictor_config cfg{};
  int T3 = 0;
  int T = 0, C = 0, Cloc = 0, J = 0, Jp = 0, B = 0, Hh = 0, G = 0, Gh = 0, hs = 0;
  std::unique_ptr<EffTrackPlan> center, kp;
  std::unique_ptr<V2VPlan> v2v;
  Scratch mem;
  float *cam = nullptr, *intr = nullptr, *dist = nullptr;
  float *det_all = nullptr, *c3f = nullptr;
  // Crop centres, truncated 3D centre and validity of a time batch: written by stage 2 (triangulation), read by
  // stage 3.  TWO sets: the staged entry points alternate between them, so that stage 3 of time batch i may run
  // on a second stream while stage 2 of batch i+1 (which writes the other set) is under way -- a stage-3 call
  // reads the set of the stage-2 call that preceded it in host call order (distributed.py).  The whole-path
  // forward always uses set 0 (its captured graph bakes the pointers).
  int *c3i_[2] = {nullptr, nullptr}, *chm_[2] = {nullptr, nullptr}, *valid_[2] = {nullptr, nullptr};
  int slot = 0;
  int* c3i_cur() const { return c3i_[slot]; }
  int* chm_cur() const { return chm_[slot]; }
  int* valid_cur() const { return valid_[slot]; }
  float2* coarse = nullptr;
  double* sa_partial = nullptr;
  int* sa_max = nullptr;
  // hipGraph replay of the whole forward (the reference driver's call pattern: one frame set per
  // call, predict3D.py:82-85 -- 150+ launches of small kernels, launch-bound when issued one by one)
  int use_graph = 0;
  const void** frames_cell = nullptr;        // device: the frame pointer of the current call
  const void* const* cur_cell = nullptr;     // non-null only while the forward is being captured
  float *g_points = nullptr, *g_conf = nullptr;
  hipStream_t gstream = nullptr;             // capture stream (the caller's may be the null stream)
  hipGraphExec_t gexec[2] = {nullptr, nullptr};   // [fp32 frames, uint8 frames]
  ~jh_predictor() {
    for (auto& e : gexec) if (e) (void)hipGraphExecDestroy(e);
    if (gstream) (void)hipStreamDestroy(gstream);
  }

  // 3D stage for frames t0 .. t0+T3-1 of the batch (heat_all holds those frames)
  int run_3d(const float* heat_all, int t0, float* heatmap_final, float* points, float* conf,
             hipStream_t s, const HeatLayout* layout = nullptr) {
    const double g3 = (double)G * G * G;
    // algorithmic traffic of the gather: every heatmap byte once in, the volume once out
    JH_PROF("reproject_gather", 0.0, 4.0 * T3 * ((double)C * Hh * Hh * J + g3 * J),
            launch_reproject(cam, intr, dist, c3i_cur() + t0 * 3, chm_cur() + t0 * C * 2, heat_all, coarse,
                             v2v->input.p, nullptr, T3, C, G, cfg.grid_spacing, hs, Jp,
                             /*heat_pad=*/0, /*div255=*/1, s, layout));
    if (v2v->run(s)) return 1;
    JH_PROF("softargmax", 0.0, 4.0 * T3 * (g3 / 8) * J,
            launch_softargmax(v2v->output.p, c3i_cur() + t0 * 3, sa_partial, sa_max, points, conf,
                              heatmap_final, T3, J, Jp, Gh, cfg.grid_spacing, cfg.roi_cube_size, s));
    return 0;
  }
};

extern "C" {

int jh_predictor_create(const jh_params* center_params, const jh_params* hybrid_params,
                        const jh_predictor_config* cfg, jh_predictor** out) {
  JH_REQUIRE(hybrid_params && cfg && out, "bad argument");
  std::unique_ptr<jh_predictor> pr(new jh_predictor());
  pr->cfg = *cfg;
  pr->T = cfg->time_batch; pr->C = cfg->num_cameras; pr->Cloc = cfg->cam_n;
  pr->T3 = cfg->time_batch_3d > 0 ? cfg->time_batch_3d : cfg->time_batch;
  JH_REQUIRE(pr->T3 <= pr->T, "time_batch_3d must not exceed time_batch");
  pr->J = cfg->num_joints; pr->Jp = cpad(cfg->num_joints);
  pr->B = cfg->bbox; pr->Hh = cfg->bbox / 2; pr->hs = cfg->bbox / 2 + 2;
  pr->G = (int)(cfg->roi_cube_size / cfg->grid_spacing);
  pr->Gh = pr->G / 2;
  JH_REQUIRE(pr->T >= 1 && pr->C >= 1 && pr->C <= 64, "time batch / camera count");
  JH_REQUIRE(cfg->cam_lo >= 0 && cfg->cam_n >= 1 && cfg->cam_lo + cfg->cam_n <= pr->C, "camera range");
  JH_REQUIRE(pr->G % 4 == 0, "ROI_CUBE_SIZE / GRID_SPACING must be a multiple of 4");
  const int N = pr->T * pr->Cloc;
  // (form of the BiFPN nodes: by the time batch alone, see EffTrackPlan::node_rows)
  const int node_rows = pr->T >= 8 ? 1 : 0;
  // (... and with it the blocking of the InstanceNorm / pooled-sum passes, Plan::norm_block_kb; JH_NODE_ROWS=0 switches
  //  every class-dependent form off: one arithmetic for all time batches)
  const int norm_kb = node_rows && JH_ENV_KNOB("JH_NODE_ROWS") != 0 ? 64 : 0;
  // precision: the predictor's own setting; JH_PRECISION_DEFAULT follows the process-wide default
  JH_REQUIRE(cfg->precision >= JH_PRECISION_DEFAULT && cfg->precision <= JH_PRECISION_BF16X3_WIDE,
             "jh_predictor_config.precision: unknown mode");
  const int precision = cfg->precision == JH_PRECISION_DEFAULT ? precision_mode() : cfg->precision;
  pr->cfg.precision = precision;
  if (center_params) {
    pr->center.reset(new EffTrackPlan());
    pr->center->precision = precision;
    pr->center->node_rows = node_rows;
    pr->center->norm_block_kb = norm_kb;
    if (pr->center->build(center_params->map, "", cfg->center_model, 1, N, cfg->center_size,
                          cfg->center_size)) return 1;
  }
  pr->kp.reset(new EffTrackPlan());
  pr->kp->precision = precision;
  pr->kp->node_rows = node_rows;
  pr->kp->norm_block_kb = norm_kb;
  if (pr->kp->build(hybrid_params->map, "effTrack.", cfg->kp_model, pr->J, N, pr->B, pr->B)) return 1;
  pr->v2v.reset(new V2VPlan());
  pr->v2v->precision = precision;
  pr->v2v->norm_block_kb = norm_kb;
  if (pr->v2v->build(hybrid_params->map, "v2vNet.", pr->J, pr->T3, pr->G)) return 1;
  auto& m = pr->mem;
  const int T = pr->T, C = pr->C;
  if (m.get(reinterpret_cast<void**>(&pr->cam), (size_t)C * 12 * sizeof(float))) return 1;
  if (m.get(reinterpret_cast<void**>(&pr->intr), (size_t)C * 9 * sizeof(float))) return 1;
  if (m.get(reinterpret_cast<void**>(&pr->dist), (size_t)C * 5 * sizeof(float))) return 1;
  if (m.get(reinterpret_cast<void**>(&pr->det_all), (size_t)T * C * 3 * sizeof(float))) return 1;
  if (m.get(reinterpret_cast<void**>(&pr->c3f), (size_t)T * 3 * sizeof(float))) return 1;
  for (int k = 0; k < 2; ++k) {
    if (m.get(reinterpret_cast<void**>(&pr->c3i_[k]), (size_t)T * 3 * sizeof(int))) return 1;
    if (m.get(reinterpret_cast<void**>(&pr->chm_[k]), (size_t)T * C * 2 * sizeof(int))) return 1;
    if (m.get(reinterpret_cast<void**>(&pr->valid_[k]), (size_t)T * sizeof(int))) return 1;
  }
  if (m.get(reinterpret_cast<void**>(&pr->coarse),
            (size_t)pr->T3 * C * pr->Gh * pr->Gh * pr->Gh * sizeof(float2))) return 1;
  {   // soft-argmax accumulators and maxima in ONE allocation (zeroed by one launch per forward)
    const size_t pb = (size_t)T * pr->Jp * 4 * kLimbs * sizeof(double), mb = (size_t)T * pr->Jp * sizeof(int);
    if (m.get(reinterpret_cast<void**>(&pr->sa_partial), pb + mb)) return 1;
    pr->sa_max = reinterpret_cast<int*>(reinterpret_cast<char*>(pr->sa_partial) + pb);
  }
  for (int k = 0; k < 2; ++k) JH_CHECK_HIP(hipMemset(pr->valid_[k], 0, (size_t)T * sizeof(int)));
  // graph replay: by default for the single-frame-set call (T = 1), where the forward is
  // launch-bound; JH_GRAPH=1 / 0 forces it on / off for every time batch
  const int knob = JH_ENV_KNOB("JH_GRAPH");
  pr->use_graph = knob >= 0 ? (knob != 0) : (pr->T == 1);
  if (pr->Cloc == pr->C && pr->T3 == pr->T && pr->center) {
    if (m.get(reinterpret_cast<void**>(&pr->frames_cell), 256)) return 1;
    if (m.get(reinterpret_cast<void**>(&pr->g_points), (size_t)T * pr->J * 3 * sizeof(float))) return 1;
    if (m.get(reinterpret_cast<void**>(&pr->g_conf), (size_t)T * pr->J * sizeof(float))) return 1;
    // (the capture stream itself is created on first use: HIP multiplexes streams onto a few hardware
    // queues, and an idle extra stream per predictor was seen to put the caller's copy stream on the
    // compute stream's queue -- the overlapped uint8 upload of bench.py fell from 1135 to 705 frames/s)
  } else {
    pr->use_graph = 0;
  }
  JH_CHECK_HIP(hipDeviceSynchronize());
  *out = pr.release();
  return 0;
}

void jh_predictor_destroy(jh_predictor* pr) { delete pr; }

int jh_predictor_set_graph_replay(jh_predictor* pr, int on) {
  JH_REQUIRE(pr, "bad argument");
  JH_REQUIRE(!on || pr->frames_cell, "graph replay needs a predictor that owns all cameras and CenterDetect");
  pr->use_graph = on != 0;
  return 0;
}
int jh_predictor_graph_replay(const jh_predictor* pr) { return pr ? pr->use_graph : 0; }

int64_t jh_predictor_launches(const jh_predictor* pr) {
  return (int64_t)((pr->center ? pr->center->launches() : 0) + pr->kp->launches() +
                   pr->v2v->launches());
}
int jh_predictor_precision(const jh_predictor* pr) { return pr ? pr->cfg.precision : -1; }

int64_t jh_predictor_device_bytes(const jh_predictor* pr) {
  return (int64_t)((pr->center ? pr->center->device_bytes() : 0) + pr->kp->device_bytes() +
                   pr->v2v->device_bytes());
}

int jh_predictor_set_calibration(jh_predictor* pr, const float* cam_dev, const float* intr_dev,
                                 const float* dist_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_CHECK_HIP(hipMemcpyAsync(pr->cam, cam_dev, (size_t)pr->C * 12 * sizeof(float), hipMemcpyDeviceToDevice, s));
  JH_CHECK_HIP(hipMemcpyAsync(pr->intr, intr_dev, (size_t)pr->C * 9 * sizeof(float), hipMemcpyDeviceToDevice, s));
  JH_CHECK_HIP(hipMemcpyAsync(pr->dist, dist_dev, (size_t)pr->C * 5 * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

static int stage_center_impl(jh_predictor* pr, const void* frames_dev, int src_u8, float* det_dev,
                             void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(pr->center, "predictor was created without CenterDetect weights");
  const int N = pr->T * pr->Cloc, S = pr->cfg.center_size;
  if (pr->center->stem_fusable) {
    // resize + normalise happen inside the stem convolution's patch staging (csrc/stem.hip)
    StemSource& src = pr->center->stem_src;
    src.mode = 1; src.frames = frames_dev; src.frames_cell = pr->cur_cell; src.src_u8 = src_u8;
    src.H = pr->cfg.img_h; src.W = pr->cfg.img_w;
    for (int i = 0; i < 3; ++i) { src.mean[i] = pr->cfg.mean[i]; src.stdv[i] = pr->cfg.std[i]; }
    // algorithmic bytes of the fused launch: the four bilinear taps of every network-input pixel (3 channels,
    // 1 or 4 bytes each) + the stem's 16-channel output at half resolution (16 B per input pixel)
    // (stem output: 16 or 32 channels at half resolution = 16 or 32 B per input pixel)
    pr->center->set_stem_traffic((double)N * S * S * (12.0 * (src_u8 ? 1 : 4) + pr->center->stem_channels()));
  } else {
    JH_PROF("preprocess_resize", 0.0, (double)N * S * S * (12.0 * (src_u8 ? 1 : 4) + 3 * 4),
            launch_preprocess_resize(frames_dev, src_u8, pr->center->input.p, N, pr->cfg.img_h,
                                     pr->cfg.img_w, S, pr->cfg.mean, pr->cfg.std, s, pr->cur_cell));
  }
  if (pr->center->run(s)) return 1;
  const Act& h = pr->center->heat;
  JH_PROF("center_argmax", 0.0, 4.0 * N * h.H * h.W,
          launch_center_argmax(h.p, det_dev, N, h.H, h.W, h.Cp, s));
  return 0;
}

int jh_predictor_stage_center(jh_predictor* pr, const float* frames_dev, float* det_dev,
                              void* stream) {
  return stage_center_impl(pr, frames_dev, 0, det_dev, stream);
}
int jh_predictor_stage_center_u8(jh_predictor* pr, const uint8_t* frames_dev, float* det_dev,
                                 void* stream) {
  return stage_center_impl(pr, frames_dev, 1, det_dev, stream);
}

static int stage_keypoints_impl(jh_predictor* pr, const void* frames_dev, int src_u8,
                                const float* det_all_dev, float* heat_dev, void* stream,
                                int det_blocks = 1, bool next_centre_set = false) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const auto& c = pr->cfg;
  if (next_centre_set) pr->slot ^= 1;        // (the staged API: the previous batch's stage 3 may still read the other set)
  if (det_blocks > 1) {
    JH_REQUIRE(pr->C % det_blocks == 0, "cameras must divide evenly over the detection blocks");
    const int n = pr->T * pr->C * 3;
    hipLaunchKernelGGL(det_unblock_kernel, dim3((n + 255) / 256), dim3(256), 0, s, det_all_dev, pr->det_all,
                       pr->T, pr->C, pr->C / det_blocks);
    JH_CHECK_HIP(hipGetLastError());
    det_all_dev = pr->det_all;
  }
  // preds * (downsampling_scale * 2), jarvis3D.py:138-141,158-160
  const float sx2 = (float)((double)c.img_w / (double)c.center_size) * 2.f;
  const float sy2 = (float)((double)c.img_h / (double)c.center_size) * 2.f;
  JH_PROF("triangulate", 0.0, 0.0,
          launch_triangulate(det_all_dev, pr->cam, pr->intr, pr->dist, pr->c3f, pr->c3i_cur(), pr->chm_cur(),
                             pr->valid_cur(), pr->T, pr->C, sx2, sy2, 255.f, pr->B / 2, c.img_w, c.img_h, s));
  if (det_all_dev != pr->det_all)
    JH_CHECK_HIP(hipMemcpyAsync(pr->det_all, det_all_dev, (size_t)pr->T * pr->C * 3 * sizeof(float),
                                hipMemcpyDeviceToDevice, s));
  if (pr->kp->stem_fusable) {
    // crop + normalise happen inside the stem convolution's patch staging (csrc/stem.hip)
    StemSource& src = pr->kp->stem_src;
    src.mode = 2; src.frames = frames_dev; src.frames_cell = pr->cur_cell; src.src_u8 = src_u8;
    src.center_hm = pr->chm_cur(); src.Cloc = pr->Cloc; src.C = pr->C; src.cam0 = c.cam_lo;
    src.H = c.img_h; src.W = c.img_w;
    for (int i = 0; i < 3; ++i) { src.mean[i] = c.mean[i]; src.stdv[i] = c.std[i]; }
    pr->kp->set_stem_traffic((double)pr->T * pr->Cloc * pr->B * pr->B * (3.0 * (src_u8 ? 1 : 4) + pr->kp->stem_channels()));
  } else {
    JH_PROF("preprocess_crop", 0.0, (double)pr->T * pr->Cloc * pr->B * pr->B * (src_u8 ? 15.0 : 24.0),
            launch_preprocess_crop(frames_dev, src_u8, pr->chm_cur(), pr->kp->input.p, pr->T, pr->Cloc, pr->C,
                                   c.cam_lo, c.img_h, c.img_w, pr->B, c.mean, c.std, s, pr->cur_cell));
  }
  if (pr->kp->run(s)) return 1;
  if (heat_dev && heat_dev != pr->kp->heat.p)
    JH_CHECK_HIP(hipMemcpyAsync(heat_dev, pr->kp->heat.p, pr->kp->heat.bytes(),
                                hipMemcpyDeviceToDevice, s));
  return 0;
}

int jh_predictor_stage_keypoints(jh_predictor* pr, const float* frames_dev,
                                 const float* det_all_dev, float* heat_dev, void* stream) {
  return stage_keypoints_impl(pr, frames_dev, 0, det_all_dev, heat_dev, stream, 1, true);
}
int jh_predictor_stage_keypoints_u8(jh_predictor* pr, const uint8_t* frames_dev,
                                    const float* det_all_dev, float* heat_dev, void* stream) {
  return stage_keypoints_impl(pr, frames_dev, 1, det_all_dev, heat_dev, stream, 1, true);
}

int jh_predictor_stage_keypoints_gathered(jh_predictor* pr, const void* frames_dev, int frames_u8,
                                          const float* det_gathered_dev, int n_blocks, float* heat_dev,
                                          void* stream) {
  JH_REQUIRE(n_blocks >= 1, "block count");
  return stage_keypoints_impl(pr, frames_dev, frames_u8 != 0, det_gathered_dev, heat_dev, stream, n_blocks, true);
}

int jh_predictor_stage_3d(jh_predictor* pr, const float* heat_all_dev, int t0, float* points_dev,
                          float* conf_dev, int32_t* valid_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(t0 >= 0 && t0 + pr->T3 <= pr->T, "frame range of the 3D stage");
  if (pr->run_3d(heat_all_dev, t0, nullptr, points_dev, conf_dev, s)) return 1;
  if (valid_dev)
    JH_CHECK_HIP(hipMemcpyAsync(valid_dev, pr->valid_cur() + t0, (size_t)pr->T3 * sizeof(int),
                                hipMemcpyDeviceToDevice, s));
  return 0;
}

int jh_predictor_stage_3d_blocks(jh_predictor* pr, const float* heat_blocks_dev, int n_blocks,
                                 int frames_per_block, int t_off, int t0, float* points_dev,
                                 float* conf_dev, int32_t* valid_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(t0 >= 0 && t0 + pr->T3 <= pr->T, "frame range of the 3D stage");
  JH_REQUIRE(n_blocks >= 1 && pr->C % n_blocks == 0, "cameras must divide evenly over the blocks");
  JH_REQUIRE(t_off >= 0 && t_off + pr->T3 <= frames_per_block, "frame range inside a block");
  HeatLayout lay;
  const size_t plane = (size_t)pr->Hh * pr->Hh * pr->Jp;
  lay.cams_per_block = pr->C / n_blocks;
  lay.frame_stride = (size_t)lay.cams_per_block * plane;
  lay.block_stride = (size_t)frames_per_block * lay.frame_stride;
  if (pr->run_3d(heat_blocks_dev + (size_t)t_off * lay.frame_stride, t0, nullptr, points_dev, conf_dev, s,
                 &lay)) return 1;
  if (valid_dev)
    JH_CHECK_HIP(hipMemcpyAsync(valid_dev, pr->valid_cur() + t0, (size_t)pr->T3 * sizeof(int),
                                hipMemcpyDeviceToDevice, s));
  return 0;
}

static int forward_eager(jh_predictor* pr, const void* frames_dev, int src_u8, float* points_dev,
                         float* conf_dev, int32_t* valid_dev, void* stream) {
  if (stage_center_impl(pr, frames_dev, src_u8, pr->det_all, stream)) return 1;
  if (stage_keypoints_impl(pr, frames_dev, src_u8, pr->det_all, nullptr, stream)) return 1;
  return jh_predictor_stage_3d(pr, pr->kp->heat.p, 0, points_dev, conf_dev, valid_dev, stream);
}

// The forward as ONE graph launch.  Captured on first use (on the predictor's own stream: the
// caller's may be the null stream, which cannot capture) with the frame pointer read through
// `frames_cell` and the results written to the predictor's own buffers, so the same executable
// graph serves every later call: set the cell, launch the graph, copy the results out -- three
// submissions instead of ~150.  Calibration lives in the predictor's buffers (set_calibration
// copies into them), weights are immutable for the life of a predictor: nothing to invalidate.
static int forward_graph(jh_predictor* pr, const void* frames_dev, int src_u8, float* points_dev,
                         float* conf_dev, int32_t* valid_dev, hipStream_t s) {
  hipGraphExec_t& exec = pr->gexec[src_u8 ? 1 : 0];
  if (!exec) {
    hipGraph_t g = nullptr;
    if (!pr->gstream) JH_CHECK_HIP(hipStreamCreateWithFlags(&pr->gstream, hipStreamNonBlocking));
    JH_CHECK_HIP(hipStreamBeginCapture(pr->gstream, hipStreamCaptureModeRelaxed));
    pr->cur_cell = pr->frames_cell;
    const int rc = forward_eager(pr, nullptr, src_u8, pr->g_points, pr->g_conf, nullptr, pr->gstream);
    pr->cur_cell = nullptr;
    const hipError_t e = hipStreamEndCapture(pr->gstream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return 1; }
    JH_CHECK_HIP(e);
    const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) exec = nullptr;
    JH_CHECK_HIP(ei);
  }
  hipLaunchKernelGGL(set_cell_kernel, dim3(1), dim3(1), 0, s, pr->frames_cell, frames_dev);
  JH_CHECK_HIP(hipGetLastError());
  JH_CHECK_HIP(hipGraphLaunch(exec, s));
  const int n_pts = pr->T * pr->J * 3;
  hipLaunchKernelGGL(copy_out_kernel, dim3((n_pts + 255) / 256), dim3(256), 0, s, pr->g_points, pr->g_conf,
                     pr->valid_[0], points_dev, conf_dev, valid_dev, n_pts, pr->T * pr->J, pr->T);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

static int forward_impl(jh_predictor* pr, const void* frames_dev, int src_u8, float* points_dev,
                        float* conf_dev, int32_t* valid_dev, void* stream) {
  JH_REQUIRE(pr->Cloc == pr->C && pr->cfg.cam_lo == 0, "forward needs all cameras local");
  JH_REQUIRE(pr->T3 == pr->T, "forward needs time_batch_3d == time_batch");
  JH_REQUIRE(frames_dev && points_dev && conf_dev, "null frame / output pointer");
  pr->slot = 0;                               // (the whole-path forward and its captured graph: centre set 0)
  // per-launch profiling needs the launches one by one; a caller that is itself capturing this
  // stream gets the plain launches too (its graph then holds them)
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (pr->use_graph && !profiler().on) (void)hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs);
  if (!pr->use_graph || profiler().on || cs != hipStreamCaptureStatusNone)
    return forward_eager(pr, frames_dev, src_u8, points_dev, conf_dev, valid_dev, stream);
  return forward_graph(pr, frames_dev, src_u8, points_dev, conf_dev, valid_dev, static_cast<hipStream_t>(stream));
}
int jh_predictor_forward(jh_predictor* pr, const float* frames_dev, float* points_dev,
                         float* conf_dev, int32_t* valid_dev, void* stream) {
  return forward_impl(pr, frames_dev, 0, points_dev, conf_dev, valid_dev, stream);
}
int jh_predictor_forward_u8(jh_predictor* pr, const uint8_t* frames_dev, float* points_dev,
                            float* conf_dev, int32_t* valid_dev, void* stream) {
  return forward_impl(pr, frames_dev, 1, points_dev, conf_dev, valid_dev, stream);
}

int jh_predictor_debug(jh_predictor* pr, float* center3d_f_dev, int32_t* center3d_i_dev,
                       int32_t* center_hm_dev, float* det_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t T = pr->T, C = pr->C;
  if (center3d_f_dev) JH_CHECK_HIP(hipMemcpyAsync(center3d_f_dev, pr->c3f, T * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (center3d_i_dev) JH_CHECK_HIP(hipMemcpyAsync(center3d_i_dev, pr->c3i_cur(), T * 3 * sizeof(int), hipMemcpyDeviceToDevice, s));
  if (center_hm_dev) JH_CHECK_HIP(hipMemcpyAsync(center_hm_dev, pr->chm_cur(), T * C * 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
  if (det_dev) JH_CHECK_HIP(hipMemcpyAsync(det_dev, pr->det_all, T * C * 3 * sizeof(float), hipMemcpyDeviceToDevice, s));
  return 0;
}

int jh_predictor_hybridnet_forward(jh_predictor* pr, const float* crops_dev,
                                   const int32_t* center_hm_dev, const int32_t* center3d_dev,
                                   float* heatmap_final_dev, float* heatmaps_padded_dev,
                                   float* points_dev, float* conf_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(pr->Cloc == pr->C && pr->T3 == pr->T, "hybridnet_forward needs all cameras local");
  JH_CHECK_HIP(hipMemcpyAsync(pr->chm_cur(), center_hm_dev, (size_t)pr->T * pr->C * 2 * sizeof(int), hipMemcpyDeviceToDevice, s));
  JH_CHECK_HIP(hipMemcpyAsync(pr->c3i_cur(), center3d_dev, (size_t)pr->T * 3 * sizeof(int), hipMemcpyDeviceToDevice, s));
  if (launch_to_channel_last(crops_dev, pr->kp->input, s)) return 1;
  pr->kp->stem_src.mode = 0;                      // (the crops are given: the stem reads the plan's input)
  if (pr->kp->run(s)) return 1;
  if (heatmaps_padded_dev) {
    const Act& h = pr->kp->heat;
    const size_t total = (size_t)h.N * pr->J * pr->hs * pr->hs;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(export_padded_kernel, dim3(blocks), dim3(256), 0, s, h.p, heatmaps_padded_dev,
                       h.N, pr->J, h.Cp, pr->Hh);
    JH_CHECK_HIP(hipGetLastError());
  }
  return pr->run_3d(pr->kp->heat.p, 0, heatmap_final_dev, points_dev, conf_dev, s);
}

// ----------------------------------------------------------------- 2D predictor
}  // extern "C"

struct jh_predictor2d {
  jh_predictor_config cfg{};
  int T = 0, J = 0, B = 0;
  std::unique_ptr<EffTrackPlan> center, kp;
  Scratch mem;
  float* det = nullptr;
  int *chm = nullptr, *valid = nullptr;
};

extern "C" {

int jh_predictor2d_create(const jh_params* center_params, const jh_params* kp_params,
                          const jh_predictor_config* cfg, jh_predictor2d** out) {
  JH_REQUIRE(center_params && kp_params && cfg && out, "bad argument");
  std::unique_ptr<jh_predictor2d> pr(new jh_predictor2d());
  pr->cfg = *cfg;
  pr->T = cfg->time_batch; pr->J = cfg->num_joints; pr->B = cfg->bbox;
  JH_REQUIRE(pr->T >= 1, "batch");
  JH_REQUIRE(cfg->img_w >= pr->B + 1 && cfg->img_h >= pr->B + 1, "image smaller than the bounding box");
  JH_REQUIRE(cfg->precision >= JH_PRECISION_DEFAULT && cfg->precision <= JH_PRECISION_BF16X3_WIDE,
             "jh_predictor_config.precision: unknown mode");
  const int precision = cfg->precision == JH_PRECISION_DEFAULT ? precision_mode() : cfg->precision;
  pr->cfg.precision = precision;
  pr->center.reset(new EffTrackPlan());
  pr->center->precision = precision;
  if (pr->center->build(center_params->map, "", cfg->center_model, 1, pr->T, cfg->center_size,
                        cfg->center_size)) return 1;
  pr->kp.reset(new EffTrackPlan());
  pr->kp->precision = precision;
  if (pr->kp->build(kp_params->map, "", cfg->kp_model, pr->J, pr->T, pr->B, pr->B)) return 1;
  if (pr->mem.get(reinterpret_cast<void**>(&pr->det), (size_t)pr->T * 3 * sizeof(float))) return 1;
  if (pr->mem.get(reinterpret_cast<void**>(&pr->chm), (size_t)pr->T * 2 * sizeof(int))) return 1;
  if (pr->mem.get(reinterpret_cast<void**>(&pr->valid), (size_t)pr->T * sizeof(int))) return 1;
  JH_CHECK_HIP(hipDeviceSynchronize());
  *out = pr.release();
  return 0;
}

void jh_predictor2d_destroy(jh_predictor2d* pr) { delete pr; }

static int forward2d_impl(jh_predictor2d* pr, const void* frames, int src_u8, int32_t* points_dev,
                          float* conf_dev, int32_t* valid_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  const auto& c = pr->cfg;
  const int S = c.center_size;
  JH_PROF("preprocess_resize", 0.0, (double)pr->T * S * S * 24.0,
          launch_preprocess_resize(frames, src_u8, pr->center->input.p, pr->T, c.img_h, c.img_w, S,
                                   c.mean, c.std, s));
  if (pr->center->run(s)) return 1;
  const Act& h = pr->center->heat;
  JH_PROF("center_argmax", 0.0, 4.0 * pr->T * h.H * h.W,
          launch_center_argmax(h.p, pr->det, pr->T, h.H, h.W, h.Cp, s));
  // img_size / float(IMAGE_SIZE) in fp32 (jarvis2D.py:106-109)
  const float sx = (float)c.img_w / (float)S, sy = (float)c.img_h / (float)S;
  if (launch_center2d(pr->det, pr->chm, pr->valid, pr->T, sx, sy, pr->B / 2, c.img_w, c.img_h, s))
    return 1;
  JH_PROF("preprocess_crop", 0.0, (double)pr->T * pr->B * pr->B * 24.0,
          launch_preprocess_crop(frames, src_u8, pr->chm, pr->kp->input.p, pr->T, 1, 1, 0, c.img_h,
                                 c.img_w, pr->B, c.mean, c.std, s));
  if (pr->kp->run(s)) return 1;
  const Act& k = pr->kp->heat;
  JH_PROF("joint_argmax", 0.0, 4.0 * pr->T * k.H * k.W * pr->J,
          launch_joint_argmax(k.p, pr->chm, points_dev, conf_dev, pr->T, pr->J, k.Cp, k.H, k.W,
                              pr->B / 2, s));
  if (valid_dev)
    JH_CHECK_HIP(hipMemcpyAsync(valid_dev, pr->valid, (size_t)pr->T * sizeof(int),
                                hipMemcpyDeviceToDevice, s));
  return 0;
}

int jh_predictor2d_forward(jh_predictor2d* pr, const float* frames_dev, int32_t* points_dev,
                           float* conf_dev, int32_t* valid_dev, void* stream) {
  return forward2d_impl(pr, frames_dev, 0, points_dev, conf_dev, valid_dev, stream);
}
int jh_predictor2d_forward_u8(jh_predictor2d* pr, const uint8_t* frames_dev, int32_t* points_dev,
                              float* conf_dev, int32_t* valid_dev, void* stream) {
  return forward2d_impl(pr, frames_dev, 1, points_dev, conf_dev, valid_dev, stream);
}

// ------------------------------------------------------------------- profiling
int jh_profile_begin(void) {
  Profiler& pf = profiler();
  pf.recs.clear();
  pf.on = true;
  return 0;
}
int jh_profile_end(int* n_records) {
  Profiler& pf = profiler();
  pf.on = false;
  if (pf.finish()) return 1;
  if (n_records) *n_records = (int)pf.recs.size();
  return 0;
}
int jh_profile_get(int i, char* name, int name_cap, double* ms, double* flops, double* bytes) {
  Profiler& pf = profiler();
  JH_REQUIRE(i >= 0 && i < (int)pf.recs.size(), "profile record index");
  const ProfRec& r = pf.recs[i];
  if (name && name_cap > 0) {
    strncpy(name, r.name.c_str(), name_cap - 1);
    name[name_cap - 1] = 0;
  }
  if (ms) *ms = r.ms;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return 0;
}

// -------------------------------------------------------- single-operator tests
int jh_op_conv(int nd, int kind, int k, int stride, int pad, int cin, int cout,
               const float* w_host, const float* b_host, const float* x_dev, int n, int d, int h,
               int w, const float* gate_dev, int norm_act, float* y_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  ConvDesc desc = kind == 0 ? conv_desc(nd, k, stride, pad, cin, cout)
                            : (kind == 1 ? deconv2d_k4s2p1_desc(cin, cout) : deconv3d_k2s2_desc(cin, cout));
  Scratch sc;
  Act x, y;
  if (nd == 2) d = 1;
  if (sc.act(n, d, h, w, cin, &x)) return 1;
  int Do, Ho, Wo;
  conv_out_shape(desc, d, h, w, &Do, &Ho, &Wo);
  if (sc.act(n, Do, Ho, Wo, cout, &y)) return 1;
  JH_CHECK_HIP(hipMemsetAsync(y.p, 0, y.bytes(), s));
  // the same choice the V2V plan makes: 3x3x3 stride-1 convs run as Winograd (JH_WINO=0: direct)
  bool wino = kind == 0 && nd == 3 && k == 3 && stride == 1 && pad == 1 && !gate_dev;
  if (const char* e = getenv("JH_WINO")) wino = wino && atoi(e) != 0;
  const bool b3 = wino && precision_mode() >= 1;         // the same choices the plans make
  const bool d4b = kind == 1 && !b_host && norm_act < 0 && !gate_dev && precision_mode() >= 1 &&
                   deconv4_bf16x3_eligible(cout);
  const bool xb = kind == 0 && !wino && !gate_dev && conv_bf16x3_eligible(desc) && x.Cp == cpad(cin) &&
                  (precision_mode() == 2 || (precision_mode() == 1 && nd == 3));
  ConvWeights cw;
  if (xb) {
    if (pack_conv_bf16x3_weights(desc, w_host, b_host, &cw)) return 1;
  } else if (d4b) {
    if (pack_deconv4_bf16x3_weights(cin, cout, w_host, &cw)) return 1;
  } else if (b3) {
    if (pack_bf16x3_weights(cin, cout, w_host, b_host, &cw)) return 1;
  } else if (wino) {
    if (pack_wino_weights(cin, cout, w_host, b_host, &cw)) return 1;
  } else {
    if (pack_conv_weights(desc, w_host, b_host, kind != 0, &cw)) return 1;
  }
  double* stats = nullptr;
  float* gate_p = nullptr;
  int rc = 0;
  do {
    if (norm_act >= 0) {
      if ((rc = sc.get(reinterpret_cast<void**>(&stats), (size_t)n * y.Cp * kStatW * sizeof(double)))) break;
      if (hipMemsetAsync(stats, 0, (size_t)n * y.Cp * kStatW * sizeof(double), s) != hipSuccess) { rc = 1; break; }
    }
    if (gate_dev) {   // (N,Cin) -> padded (N,Cin_p)
      if ((rc = sc.get(reinterpret_cast<void**>(&gate_p), (size_t)n * x.Cp * sizeof(float)))) break;
      if (hipMemsetAsync(gate_p, 0, (size_t)n * x.Cp * sizeof(float), s) != hipSuccess) { rc = 1; break; }
      if (hipMemcpy2DAsync(gate_p, x.Cp * sizeof(float), gate_dev, cin * sizeof(float),
                           cin * sizeof(float), n, hipMemcpyDeviceToDevice, s) != hipSuccess) { rc = 1; break; }
    }
    if ((rc = launch_to_channel_last(x_dev, x, s))) break;
    if (xb) { if ((rc = launch_conv_bf16x3(desc, cw, x, y, stats, s, nullptr))) break; }
    else if (d4b) { if ((rc = launch_deconv4_bf16x3(cw, x, y, s, nullptr))) break; }
    else if (b3) { if ((rc = launch_conv3d_bf16x3(cw, x, y, stats, s, nullptr))) break; }
    else if (wino) { if ((rc = launch_conv3d_wino(cw, x, y, stats, s, nullptr, wino_variant_from_env()))) break; }
    else if ((rc = launch_conv(desc, cw, x, y, gate_p, stats, s))) break;
    if (norm_act >= 0 && (rc = launch_norm_apply(y, stats, 1e-5, norm_act, nullptr, nullptr, y.p, nullptr, s))) break;
    if ((rc = launch_from_channel_last(y, y_dev, s))) break;
    if (hipStreamSynchronize(s) != hipSuccess) { set_error("stream sync failed in jh_op_conv"); rc = 1; }
  } while (0);
  free_conv_weights(&cw);
  return rc;
}

int jh_op_depthwise(int k, int c, const float* w_host, const float* x_dev, int n, int h, int w,
                    int norm_act, float* y_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  Scratch sc;
  Act x, y;
  if (sc.act(n, 1, h, w, c, &x)) return 1;
  if (sc.act(n, 1, h, w, c, &y)) return 1;
  std::vector<float> wt((size_t)k * k * x.Cp, 0.f);
  for (int ch = 0; ch < c; ++ch)
    for (int t = 0; t < k * k; ++t) wt[(size_t)t * x.Cp + ch] = w_host[(size_t)ch * k * k + t];
  float* wd; double* stats = nullptr;
  if (sc.get(reinterpret_cast<void**>(&wd), wt.size() * sizeof(float))) return 1;
  JH_CHECK_HIP(hipMemcpyAsync(wd, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice, s));
  if (norm_act >= 0) {
    if (sc.get(reinterpret_cast<void**>(&stats), (size_t)n * x.Cp * kStatW * sizeof(double))) return 1;
    JH_CHECK_HIP(hipMemsetAsync(stats, 0, (size_t)n * x.Cp * kStatW * sizeof(double), s));
  }
  if (launch_to_channel_last(x_dev, x, s)) return 1;
  if (launch_depthwise(x, wd, k, y.p, stats, s)) return 1;
  if (norm_act >= 0 && launch_norm_apply(y, stats, 1e-5, norm_act, nullptr, nullptr, y.p, nullptr, s)) return 1;
  if (launch_from_channel_last(y, y_dev, s)) return 1;
  JH_CHECK_HIP(hipStreamSynchronize(s));
  return 0;
}

// Depthwise + the squeeze-excite pooled sums of SiLU(InstanceNorm(y)) in one launch (the form MBConv blocks with
// one-tile images use, efficientnet.py:100-107): y_dev (N,C,H,W) raw output, pool_dev (N,C) sums over the pixels.
int jh_op_depthwise_pool(int k, int c, const float* w_host, const float* x_dev, int n, int h, int w,
                         float* y_dev, float* pool_dev, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(depthwise_can_pool(h, w), "jh_op_depthwise_pool: the image must be one 16 x 16 tile");
  Scratch sc;
  Act x, y;
  if (sc.act(n, 1, h, w, c, &x)) return 1;
  if (sc.act(n, 1, h, w, c, &y)) return 1;
  std::vector<float> wt((size_t)k * k * x.Cp, 0.f);
  for (int ch = 0; ch < c; ++ch)
    for (int t = 0; t < k * k; ++t) wt[(size_t)t * x.Cp + ch] = w_host[(size_t)ch * k * k + t];
  float* wd; double *stats = nullptr, *pool = nullptr;
  if (sc.get(reinterpret_cast<void**>(&wd), wt.size() * sizeof(float))) return 1;
  JH_CHECK_HIP(hipMemcpyAsync(wd, wt.data(), wt.size() * sizeof(float), hipMemcpyHostToDevice, s));
  const size_t nst = (size_t)n * x.Cp * kStatW, npl = (size_t)n * x.Cp * kLimbs;
  if (sc.get(reinterpret_cast<void**>(&stats), nst * sizeof(double))) return 1;
  if (sc.get(reinterpret_cast<void**>(&pool), npl * sizeof(double))) return 1;
  JH_CHECK_HIP(hipMemsetAsync(stats, 0, nst * sizeof(double), s));
  JH_CHECK_HIP(hipMemsetAsync(pool, 0, npl * sizeof(double), s));
  if (launch_to_channel_last(x_dev, x, s)) return 1;
  if (launch_depthwise(x, wd, k, y.p, stats, s, pool)) return 1;
  if (launch_from_channel_last(y, y_dev, s)) return 1;
  std::vector<double> hp(npl);
  JH_CHECK_HIP(hipMemcpyAsync(hp.data(), pool, npl * sizeof(double), hipMemcpyDeviceToHost, s));
  JH_CHECK_HIP(hipStreamSynchronize(s));
  std::vector<float> out((size_t)n * c);
  for (int i = 0; i < n; ++i)
    for (int ch = 0; ch < c; ++ch) {
      const double* q = hp.data() + ((size_t)i * x.Cp + ch) * kLimbs;
      out[(size_t)i * c + ch] = (float)((q[0] + q[1]) + q[2]);
    }
  JH_CHECK_HIP(hipMemcpy(pool_dev, out.data(), out.size() * sizeof(float), hipMemcpyHostToDevice));
  return 0;
}


// One fused BiFPN node (csrc/bifpn_node.hip) as a unit-test entry point: every input is a RAW
// tensor that the node normalises on load with its own InstanceNorm statistics (computed here
// on the host), exactly as inside the network plan.
int jh_op_bifpn_node(int n_in, const int* modes, const float* weights, int act, int n, int c, int cout,
                     int h, int w, const float* x0_dev, const float* x1_dev, const float* x2_dev,
                     const float* dw_host, const float* pw_host, const float* bias_host, float* y_dev,
                     void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  JH_REQUIRE(n_in >= 2 && n_in <= 3 && modes && weights && y_dev, "bad argument");
  const float* xs[3] = {x0_dev, x1_dev, x2_dev};
  Scratch sc;
  Act in[3], y;
  NodeArgs a{};
  a.n_in = n_in; a.act = act;
  for (int i = 0; i < n_in; ++i) {
    int hi = h, wi = w;
    if (modes[i] == FUSE_UP2) { hi = h / 2; wi = w / 2; }
    else if (modes[i] == FUSE_UP4) { hi = h / 4; wi = w / 4; }
    else if (modes[i] == FUSE_POOL2) { hi = h * 2; wi = w * 2; }
    if (sc.act(n, 1, hi, wi, c, &in[i])) return 1;
    if (launch_to_channel_last(xs[i], in[i], s)) return 1;
    // statistics of the raw input (sum, sum of squares per (n, channel)), whole value in limb 1
    const size_t px = (size_t)hi * wi;
    std::vector<float> host((size_t)n * c * px);
    JH_CHECK_HIP(hipMemcpyAsync(host.data(), xs[i], host.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    JH_CHECK_HIP(hipStreamSynchronize(s));
    std::vector<double> st((size_t)n * in[i].Cp * kStatW, 0.0);
    for (int b = 0; b < n; ++b)
      for (int ch = 0; ch < c; ++ch) {
        double s1 = 0.0, s2 = 0.0;
        const float* p = host.data() + ((size_t)b * c + ch) * px;
        for (size_t k = 0; k < px; ++k) { s1 += p[k]; s2 += (double)p[k] * p[k]; }
        st[((size_t)b * in[i].Cp + ch) * kStatW + 1] = s1;
        st[((size_t)b * in[i].Cp + ch) * kStatW + kLimbs + 1] = s2;
      }
    double* std_dev;
    if (sc.get(reinterpret_cast<void**>(&std_dev), st.size() * sizeof(double))) return 1;
    JH_CHECK_HIP(hipMemcpyAsync(std_dev, st.data(), st.size() * sizeof(double), hipMemcpyHostToDevice, s));
    JH_CHECK_HIP(hipStreamSynchronize(s));                       // (st is a local)
    a.in[i] = in[i].p; a.st[i] = std_dev; a.inv_cnt[i] = 1.f / (float)px; a.mode[i] = modes[i];
    a.w[i] = weights[i];
  }
  if (sc.act(n, 1, h, w, cout, &y)) return 1;
  JH_CHECK_HIP(hipMemsetAsync(y.p, 0, y.bytes(), s));
  const int Cp = in[0].Cp;
  std::vector<float> dwt((size_t)9 * Cp, 0.f);
  for (int ch = 0; ch < c; ++ch)
    for (int t = 0; t < 9; ++t) dwt[(size_t)t * Cp + ch] = dw_host[(size_t)ch * 9 + t];
  float* dwd;
  if (sc.get(reinterpret_cast<void**>(&dwd), dwt.size() * sizeof(float))) return 1;
  JH_CHECK_HIP(hipMemcpyAsync(dwd, dwt.data(), dwt.size() * sizeof(float), hipMemcpyHostToDevice, s));
  ConvWeights cw;
  if (pack_conv_weights(conv_desc(2, 1, 1, 0, c, cout), pw_host, bias_host, false, &cw)) return 1;
  double* ost;
  int rc = 0;
  do {
    if ((rc = sc.get(reinterpret_cast<void**>(&ost), (size_t)n * y.Cp * kStatW * sizeof(double)))) break;
    if (hipMemsetAsync(ost, 0, (size_t)n * y.Cp * kStatW * sizeof(double), s) != hipSuccess) { rc = 1; break; }
    a.dw = dwd; a.pw = cw.w; a.bias = cw.bias; a.y = y.p; a.stats = ost;
    a.N = n; a.H = h; a.W = w; a.Cp = Cp; a.cout_p = y.Cp; a.cout_p16 = cw.cout_p16;
    if ((rc = launch_bifpn_node(a, s))) break;
    if ((rc = launch_from_channel_last(y, y_dev, s))) break;
    if (hipStreamSynchronize(s) != hipSuccess) { set_error("stream sync failed in jh_op_bifpn_node"); rc = 1; }
  } while (0);
  free_conv_weights(&cw);
  return rc;
}

}  // extern "C"
