// Launch plans of the two networks on the hot path.
//
// A plan is built once per (weights, input shape): activations are allocated,
// weights are repacked into MFMA operand order, and the forward pass becomes a
// flat list of kernel launches on one stream with no host synchronisation and
// no allocation, so a caller may capture it into a hipGraph.
#pragma once
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "jh_common.h"
#include "preprocess.h"

namespace jh {

// kernels implemented in other translation units
int launch_reproject(const float* cam, const float* intr, const float* dist, const int* center3d,
                     const int* center_hm, const float* heat, float2* coarse, float* vol,
                     int* idx_out, int T, int C, int G, float spacing, int hs, int Jp,
                     int heat_pad, int div255, hipStream_t s, const HeatLayout* layout = nullptr);
int launch_preprocess_resize(const void* frames, int src_u8, float* out, int N, int H, int W, int S,
                             const float* mean, const float* stdv, hipStream_t s,
                             const void* const* frames_cell = nullptr);
int launch_preprocess_crop(const void* frames, int src_u8, const int* center_hm, float* out, int T,
                           int Cloc, int C, int cam0, int H, int W, int B, const float* mean,
                           const float* stdv, hipStream_t s, const void* const* frames_cell = nullptr);
int launch_center_argmax(const float* heat, float* det, int N, int Hh, int Wh, int Cp,
                         hipStream_t s);
int launch_center2d(const float* det, int* center_hm, int* valid, int T, float sx, float sy, int hw,
                    int W, int H, hipStream_t s);
int launch_joint_argmax(const float* heat, const int* center_hm, int* points, float* conf, int T,
                        int J, int Jp, int Hh, int Wh, int hw, hipStream_t s);
int launch_triangulate(const float* det, const float* cam, const float* intr, const float* dist,
                       float* center3d_f, int* center3d_i, int* center_hm, int* valid, int T, int C,
                       float sx2, float sy2, float wdiv, int hw, int W, int H, hipStream_t s);
int launch_project_points(const float* pts, const float* cam, const float* intr, const float* dist,
                          float* uv, int P, int C, hipStream_t s);
int launch_softargmax(const float* x, const int* center3d, double* partial, int* pmax,
                      float* points, float* conf, float* heatmap_final, int T, int J, int Jp,
                      int Gh, float spacing, float roi, hipStream_t s);

typedef std::map<std::string, std::vector<float>> ParamMap;

// Common machinery: owned device buffers, a zero-initialised scratch arena for
// InstanceNorm statistics / SE pools, and the recorded launch list.
class Plan {
 public:
  ~Plan();
  int run(hipStream_t s);
  size_t launches() const { return ops_.size(); }
  size_t device_bytes() const { return bytes_; }
  // Precision of the convolutions this plan builds (0 fp32, 1 bf16x3, 2 bf16x3_wide; include/jarvis_hip.h):
  // the process default (jh_set_precision / JH_PRECISION) at construction; an owner with its own setting
  // (jh_predictor_config::precision) overrides it before build().
  int precision = precision_mode();
  // Least tensor bytes per workgroup of the InstanceNorm / pooled-sum passes (launch_norm_apply): part of the pooled
  // sums' arithmetic, so -- like the BiFPN nodes' form -- a function of the predictor's time-batch class only (64 KB
  // for time batches >= 8, 0 = small blocks for single frame sets); set before build().
  int norm_block_kb = 0;

 protected:
  int alloc(void** p, size_t bytes);
  int new_act(int N, int D, int H, int W, int C, Act* out);
  size_t scratch(size_t doubles);           // offset into the zeroed arena
  int finish();                             // allocate the arena
  double* sc(size_t off) const { return arena_ + off; }
  int upload(const std::vector<float>& host, float** dev);
  int get(const ParamMap& pm, const std::string& key, size_t numel, const float** out);

  // lane 1: the op belongs to a SIDE BRANCH -- it depends on nothing the main lane launches between the branch's first
  // op and the next op marked `join` (which, and everything behind it, may read what the branch wrote).  run() forks a
  // second stream at the first op of a branch and joins it at the marked op: two independent chains of a latency-bound
  // forward (single frame sets) then run side by side -- inside a captured hipGraph as two branches.
  struct Op { std::function<int(hipStream_t)> fn; std::string name; double flops, bytes; int lane = 0; bool join = false; };
  std::vector<Op> ops_;
  int lane_ = 0;                             // lane of the ops pushed from here on (builders set it around a branch)
  bool join_next_ = false;                   // the next op pushed joins the side branch
  void push(const std::string& name, double flops, double bytes, std::function<int(hipStream_t)> fn) {
    ops_.push_back(Op{std::move(fn), name, flops, bytes, lane_, join_next_});
    join_next_ = false;
  }
  hipStream_t side_ = nullptr;               // created by finish() when the plan has a side branch
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  std::vector<void*> owned_;
  std::vector<ConvWeights> convs_;
  double* arena_ = nullptr;
  size_t arena_doubles_ = 0;
  size_t bytes_ = 0;

  // building blocks shared by both networks
  int add_conv(const ParamMap& pm, const ConvDesc& d, const std::string& wkey,
               const std::string& bkey, bool transposed, const Act& x, const Act& y,
               const float* gate, bool want_stats, size_t* stats_off, long in_stats_off = -1,
               float in_inv = 0.f, int in_act = 0, const SeGate* se = nullptr, long se_pool_off = -1);
  void add_norm(const Act& x, size_t stats_off, int act, const float* r1, const float* r2,
                float* y, long pool_off, long r1_stats_off = -1);
};

// A tensor as consumers see it: the raw conv output plus the statistics the
// InstanceNorm that follows it needs (st < 0: already normalised / no norm).
struct Ref {
  Act a;
  long st = -1;
  float inv = 0.f;       // 1 / pixels the statistics were accumulated over
  int act = 0;           // activation that follows that InstanceNorm (ACT_*)
};

class EffTrackPlan : public Plan {
 public:
  // size: 0 small, 1 medium, 2 large.  N images of H x W (multiples of 64).
  // want_res1: also compute res1 = final_conv1(first_conv(..)) (model.py:128), which the
  // inference path never reads (hybridnet/model.py:57-58, jarvis3D.py:147)
  int build(const ParamMap& pm, const std::string& prefix, int size, int J, int N, int H, int W,
            bool want_res1 = false);
  // Pre-processing fused into the stem convolution (small model's vector-ALU stem only): when
  // stem_fusable, a caller may set stem_src.mode to 1 (resize) / 2 (crop) before run(); `input` is then
  // never read.  mode 0 (default): the stem reads `input`.
  bool stem_fusable = false;
  StemSource stem_src;
  // algorithmic bytes of the (fused) stem launch for the profiler: taps read from the frames + stem output
  void set_stem_traffic(double bytes);
  int stem_channels() const { return stem_ch_; }
  // Which form the high-resolution BiFPN nodes take (NodeArgs::rows), set before build().  A predictor sets it
  // from its time batch ALONE (not from the number of cameras it owns), so that a camera-sharded rank and the
  // single-GPU run of the same time batch launch the same kernels and stay bit-equal; -1: by the
  // number of workgroups of the launch (stand-alone operator).
  int node_rows = -1;
  Act input;     // [N][H][W][8]   normalised image, channel-last
  Act heat;      // [N][H/2][W/2][Jp]  res2 (ConvTranspose output)
  Act res1;      // [N][H/4][W/4][Jp]  final_conv1 output (only with want_res1)
  int J = 0;
  int stem_op_ = -1, stem_ch_ = 16;

 private:
  int mbconv(const ParamMap& pm, const std::string& p, int stage, int k, int stride, int cin,
             int cout, int expand, Ref* x, Ref* out);
  void materialise(Ref* x);
  int lateral(const ParamMap& pm, const std::string& p, int cout, const Ref& x, Ref* out);
  int pool(const Ref& x, Ref* out);
  int node(const ParamMap& pm, const std::string& conv_prefix, int n_in, const Ref* ins,
           const int* modes, const float* w, int act, const Act& like, int cout, Ref* out,
           Ref* pooled = nullptr);   // pooled: also MaxPool2d(2,2) of the raw output, if the node's form can (else a.p = 0)
};

class V2VPlan : public Plan {
 public:
  int build(const ParamMap& pm, const std::string& prefix, int J, int T, int G);
  Act input;     // [T][G][G][G][Jp]   reprojected volume / 255
  Act output;    // [T][G/2]^3[Jp]

 private:
  // x_stats >= 0: x is a raw conv output whose InstanceNorm + ReLU the block applies on load
  int res_block(const ParamMap& pm, const std::string& p, int c, const Act& x, const float* extra,
                Act* out, long x_stats = -1);
};

}  // namespace jh
