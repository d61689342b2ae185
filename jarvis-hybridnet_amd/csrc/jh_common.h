// Shared declarations of the MI355X (gfx950) HIP implementation.
//
// Activation layout everywhere inside the library: channel-last fp32,
// [N][D][H][W][Cp] with Cp = channels rounded up to a multiple of 8 and the
// pad channels held at exactly 0.  2D tensors are D == 1.  This is the layout
// the MFMA implicit-GEMM convolution wants (K-contiguous operand rows), the
// layout that makes the reprojection gather a contiguous J-vector per tap,
// and the layout the V2V input is produced in.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

namespace jh {

constexpr int kWave = 64;

// Experiment knobs (JH_* environment variables) are launch-path constants: each is read ONCE per
// process (first use), not on every launch.  -1 = unset.
#define JH_ENV_KNOB(name) ([]() -> int { static const int v = [] { const char* e = getenv(name); return e ? atoi(e) : -1; }(); return v; }())

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
inline int cpad(int c) { return round_up(c, 8); }

// thread-local last error text, exported through jh_last_error()
void set_error(const std::string& msg);
#define JH_CHECK_HIP(expr)                                                     \
  do {                                                                         \
    hipError_t _e = (expr);                                                    \
    if (_e != hipSuccess) {                                                    \
      jh::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));        \
      return 1;                                                                \
    }                                                                          \
  } while (0)
#define JH_REQUIRE(cond, msg)                                                  \
  do {                                                                         \
    if (!(cond)) {                                                             \
      jh::set_error(std::string("requirement failed: ") + #cond + " (" + msg + \
                    ")");                                                      \
      return 1;                                                                \
    }                                                                          \
  } while (0)

// A channel-last activation tensor in device memory.
struct Act {
  float* p = nullptr;
  int N = 0, D = 1, H = 0, W = 0, C = 0, Cp = 0;
  size_t pixels() const { return (size_t)D * H * W; }
  size_t elems() const { return (size_t)N * pixels() * Cp; }
  size_t bytes() const { return elems() * sizeof(float); }
};

// Where camera c of frame t of the 3D stage lives inside the heatmap buffer (floats):
//   heat + (c / cams_per_block) * block_stride + t * frame_stride + (c % cams_per_block) * Hh*Hh*Jp
// Dense (T, C, Hh, Hh, Jp) = one block.  The camera-sharded exchange delivers
// (ranks, frames, cameras per rank, Hh, Hh, Jp): one block per source rank, read in place.
struct HeatLayout {
  int cams_per_block = 0;
  size_t block_stride = 0, frame_stride = 0;
};

enum ActKind { ACT_NONE = 0, ACT_RELU = 1, ACT_SILU = 2 };

// Optional per-launch timing with HIP events on the launch stream (bench.py's
// roofline numbers).  Off by default: the hot path records nothing.
struct ProfRec { std::string name; double flops, bytes; hipEvent_t e0, e1; float ms; };
struct Profiler {
  bool on = false;
  std::vector<ProfRec> recs;
  void begin(const std::string& name, double flops, double bytes, hipStream_t s);
  void end(hipStream_t s);
  int finish();            // synchronise, fill ms, destroy events
};
Profiler& profiler();

// ---------------------------------------------------------------- conv (MFMA)
// One "phase" of a (transposed) convolution: out[o*os + ooff] =
// sum_t in[o*stride - pad + t] * w[t].  An ordinary conv has one phase with
// os = 1; ConvTranspose k4 s2 p1 (2D) has 4 phases of 2x2 taps, ConvTranspose
// k2 s2 (3D) has 8 phases of a single tap.
struct ConvPhase {
  int pad[3];    // z, y, x
  int ooff[3];   // z, y, x
};

struct ConvDesc {
  int nd;        // 2 or 3 spatial dims
  int k;         // taps per dim of each phase
  int stride;    // input stride
  int ostride;   // output stride (phases interleave when > 1)
  int nphase;
  ConvPhase phase[8];
  int cin, cout;
  int latency_class;   // 1: the plan belongs to the single-frame-set class (time batch < 8): tile forms that trade
                       // workgroups for fill (the whole-image tile) stay off.  A function of the class, never of the batch.
};

// Packed weights of one conv (all phases), see pack_conv_weights().
struct ConvWeights {
  float* w = nullptr;      // [phase][tap][cin_p/8][cout_p16/16][64][2]; paired: [phase][tap][cin_p/16][cout_p16/16][64][4]
  int paired = 0;          // two 8-channel steps per 16-byte lane word (kernels that read operands as b128)
  float* bias = nullptr;   // [cout_p16] or nullptr
  size_t phase_stride = 0; // floats
  int cin_p = 0, cout_p16 = 0;
};

// Division of block indices by run-time extents (grid dimensions, tiles per row): the compiler's 32-bit
// unsigned division is ~28 instructions (float reciprocal + corrections, through v_readfirstlane) and a
// tile prologue has five of them.  Host-computed multiply-shift instead: q = (mulhi(x, mul) + x) >> shift,
// exact for x < 2^31 (Granlund-Montgomery; mul = floor(2^32 (2^l - d) / d) + 1, l = ceil(log2 d)).
struct FastDiv {
  unsigned mul = 1, shift = 0, d = 1;
};
inline FastDiv make_fastdiv(unsigned d) {
  FastDiv f;
  f.d = d;
  unsigned l = 0;
  while ((1ull << l) < d) ++l;
  f.shift = l;
  f.mul = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  return f;
}

// Squeeze-excite gate of an MBConv block computed by the CONSUMER (the project convolution) from the pooled
// sums, instead of by a launch of its own: gate[c] = sigmoid(We silu(Wr mean + br) + be)
// (jarvis/efficienttrack/efficientnet.py:107-112).  pool == nullptr: not used.
struct SeGate {
  const double* pool = nullptr;   // [N][cin_p][kLimbs] pooled sums of the activated tensor
  const float *wr = nullptr, *br = nullptr, *we = nullptr, *be = nullptr;
  int C = 0, S = 0;               // channels, squeeze width
  float inv_hw = 0.f;
};

struct ConvArgs {
  const float* x;        // input activation
  float* y;              // output activation (raw, pre-norm)
  const float* w;        // packed weights
  const float* bias;     // [cout_p16] or nullptr
  const float* gate;     // [N][cin_p] multiplicative gate on the input or nullptr
  SeGate se;             // ... or the recipe to compute it in the kernel's prologue
  const double* in_stats;  // [N][cin_p][2]: InstanceNorm (+ in_act) applied to the input on load
  float in_inv;            // 1 / pixels the input statistics were taken over
  int in_act;
  int nrm_floats;          // LDS floats reserved in front of the patch for mean / rstd
  double* stats;         // [N][cout_p][2] (sum, sumsq) accumulated, or nullptr
  int N, Din, Hin, Win, cin_p;
  int Dout, Hout, Wout;  // logical conv output extent of ONE phase
  int Dy, Hy, Wy, cout_p;  // physical output tensor extent
  int cout_p16;
  int kc;                // input channels staged per LDS pass (multiple of 8)
  int in_px;             // floats per input pixel in memory: cin_p, or 4 for the 3-channel network
                         // input (one float4 per pixel; channels 4.. of the K padding read 0)
  int ostride, nphase;
  int paired = 0;        // weights in the paired layout (ConvWeights::paired)
  FastDiv fgx, fgy, ftx, fty;   // divisors: grid x / y, tiles per row / column (set by the launcher)
  size_t phase_stride;
  ConvPhase phase[8];
};

// host-side repack: torch layout (cout, cin, k..) [transposed: (cin, cout, k..)]
// -> ConvWeights device buffers.  Returns 0 on success.
int pack_conv_weights(const ConvDesc& d, const float* w_host, const float* b_host,
                      bool transposed, ConvWeights* out);
void free_conv_weights(ConvWeights* w);

// InstanceNorm (+ activation) of the conv INPUT, applied while the patch is staged: the
// producer's raw output and fused statistics are consumed directly, no normalised copy.
struct InNorm {
  const double* stats = nullptr;
  float inv = 0.f;
  int act = 0;
};
// which few-channel 1 x 1 layers the direct register kernel takes, by shape (csrc/conv_pw_direct.hip)
bool conv_pw_direct_shape_ok(int cin_p, int cout_p16, int pixels);
int launch_conv(const ConvDesc& d, const ConvWeights& w, const Act& x, const Act& y,
                const float* gate, double* stats, hipStream_t s, const InNorm* in = nullptr,
                const SeGate* se = nullptr);
// output extent of a conv described by d for an input of extent (D,H,W)
void conv_out_shape(const ConvDesc& d, int D, int H, int W, int* Do, int* Ho, int* Wo);

ConvDesc conv_desc(int nd, int k, int stride, int pad, int cin, int cout);
ConvDesc deconv2d_k4s2p1_desc(int cin, int cout);
ConvDesc deconv3d_k2s2_desc(int cin, int cout);

// ---------------------------------------------------------------- elementwise
// y = act((x - mean) * rstd + r1) + r2  with per-(n,c) statistics from `stats`
// (biased variance, eps); optional pooled sum of y per (n,c) into `pool`.
// r1_stats: r1 is a RAW tensor; its InstanceNorm + ReLU is applied on load (act must be ACT_RELU)
int launch_norm_apply(const Act& x, const double* stats, double eps, int act,
                      const float* r1, const float* r2, float* y, double* pool,
                      hipStream_t s, const double* r1_stats = nullptr, int min_block_kb = 0);
// squeeze-excite gate from pooled sums: gate[n][c] = sigmoid(We silu(Wr mean + br) + be)
int launch_se_gate(const double* pool, int N, int C, int Cp, int S, float inv_hw,
                   const float* wr, const float* br, const float* we, const float* be,
                   float* gate, hipStream_t s);
// depthwise k x k stride-1 conv, weights [k*k][Cp]; optional IN statistics.
// pool != nullptr: squeeze-excite pooled sums of SiLU(InstanceNorm(y)) in the same launch (one-tile images only)
bool depthwise_can_pool(int H, int W);
int launch_depthwise(const Act& x, const float* w, int k, float* y, double* stats,
                     hipStream_t s, double* pool = nullptr);
// weighted fusion node of the BiFPN: y = act(sum_i w[i] * resample_i(in_i))
enum FuseMode { FUSE_SAME = 0, FUSE_UP2 = 1, FUSE_UP4 = 2, FUSE_POOL2 = 3 };
struct FuseArgs {
  const float* in[3];
  int mode[3];
  float w[3];
  int n_in;
  int act;
};
int launch_fuse(const FuseArgs& f, const Act& out, hipStream_t s);
int launch_maxpool2(const Act& x, float* y, hipStream_t s);

// fused BiFPN node (csrc/bifpn_node.hip)
// XCD-aware block order.  The dispatcher places linear block b on XCD b % 8 (observed, not
// a contract: used for speed only), so neighbouring blocks -- which share halo pixels --
// land on 8 different, mutually non-coherent L2s and every halo is fetched from the fabric
// again.  This bijection of [0, total) hands each XCD one CONTIGUOUS range of logical
// blocks instead: XCD x runs logical blocks [start(x), start(x) + count(x)).
struct BlockId { unsigned x, y, z; };

// ---- order-independent statistics -------------------------------------------------------------
// InstanceNorm statistics, squeeze-excite pools and soft-argmax sums are accumulated across
// workgroups with fp64 atomics, whose order varies from run to run.  To make the result
// independent of that order every addend is split into three limbs on fixed power-of-two grids
// (2^16, 2^-20, 2^-56; what lies below 2^-56 is dropped, a function of the addend alone) and each
// limb goes to its own accumulator: all addends of one accumulator are multiples of its grid
// and the running sums stay far below 2^53 grid steps (<= 2^13 addends per accumulator), so EVERY
// addition is exact and the sum does not depend on the order.  The three limb sums are added in
// a fixed order by the reader.  A value therefore occupies kLimbs doubles; InstanceNorm
// statistics are (sum, sum of squares) = kStatW doubles per (n, c).  An fp32 partial has at most
// two non-zero limbs, usually one: the atomic count is what it was.
constexpr int kLimbs = 3;
constexpr int kStatW = 2 * kLimbs;
#if defined(__HIPCC__)
__device__ __forceinline__ void exact_add(double* dst, double p) {
  const double h = rint(p * 0x1p-16) * 0x1p16;
  const double r = p - h;
  const double m = rint(r * 0x1p20) * 0x1p-20;
  const double l = rint((r - m) * 0x1p56) * 0x1p-56;
  if (h != 0.0) unsafeAtomicAdd(dst + 0, h);
  if (m != 0.0) unsafeAtomicAdd(dst + 1, m);
  if (l != 0.0) unsafeAtomicAdd(dst + 2, l);
}
__device__ __forceinline__ double exact_read(const double* src) { return (src[0] + src[1]) + src[2]; }
// (sum, sum of squares) of one (n, c): st points at its kStatW doubles
__device__ __forceinline__ void stat_add(double* st, float s1, float s2) {
  exact_add(st, (double)s1);
  exact_add(st + kLimbs, (double)s2);
}
// The 2D networks' producers hand over DOUBLE partials (round 6).  variance = E[x^2] - mean^2 cancels: with fp32
// partial sums a channel whose variance is a small fraction r of its mean square -- a constant frame from a dead or
// saturated camera, where only the zero padding of each layer varies: r ~ 1e-2..1e-3 -- gets rstd to ~1e-7 / r, layer
// after layer (fixture cfg3_cam_black: 0.012 mm on the 3D keypoints; the reference accumulates in double,
// at::acc_type<float>).  Squares of fp32 values are exact in fp64 and a workgroup's few thousand addends lose nothing.
// A double partial through exact_add would light up two or three limbs -- two or three atomics per value where an fp32
// partial needs one, and the few-channel high-resolution layers are bound by exactly those atomics (64 workgroups per
// image adding to the same 16 addresses: conv2d_k1s1_16x8@128 0.146 -> 0.182 ms).  So a double partial is ROUNDED to one
// limb, chosen by its magnitude (a function of the addend alone: still order-independent): windows of ten binary
// orders with at least 30 bits below the window's lower end -- relative precision 2^-31 or better for |p| >= 2^-5
// (fp32: 2^-24), absolute 2^-36 below that (far under eps = 1e-5 per pixel).  Every limb's addends are multiples of its
// grid and below 2^40 grid steps, so every addition stays exact for up to 2^13 addends per (n, c) as before.
__device__ __forceinline__ void exact_add_rounded(double* dst, double p) {
  const double a = fabs(p);
  double q;
  int slot;
  if (a < 0x1p5) { q = rint(p * 0x1p35) * 0x1p-35; slot = 2; }
  else if (a < 0x1p15) { q = rint(p * 0x1p25) * 0x1p-25; slot = 1; }
  else { q = rint(p * 0x1p15) * 0x1p-15; slot = 0; }
  if (q != 0.0) unsafeAtomicAdd(dst + slot, q);
}
__device__ __forceinline__ void stat_add(double* st, double s1, double s2) {
  exact_add_rounded(st, s1);
  exact_add_rounded(st + kLimbs, s2);
}
// one value's contribution to a lane's (sum, sum of squares): cvt + add + fma, all full-rate fp64
__device__ __forceinline__ void stat_acc(double& s1, double& s2, float v) {
  const double d = (double)v;
  s1 += d;
  s2 = fma(d, d, s2);
}
#endif

#if defined(__HIPCC__)
__device__ __forceinline__ unsigned xcd_linear(unsigned L, unsigned total) {
  const unsigned q = total >> 3, r = total & 7u, x = L & 7u, s = L >> 3;
  return x < r ? x * (q + 1) + s : r * (q + 1) + (x - r) * q + s;
}
__device__ __forceinline__ unsigned fd_div(unsigned x, const FastDiv& f) { return (__umulhi(x, f.mul) + x) >> f.shift; }
// the same bijection with host-prepared divisors of gridDim.x / gridDim.y
__device__ __forceinline__ BlockId xcd_block(const FastDiv& fgx, const FastDiv& fgy) {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  const unsigned l = xcd_linear(L, gx * gy * gz);
  BlockId b;
  const unsigned l1 = fd_div(l, fgx);
  b.x = l - l1 * gx;
  b.z = fd_div(l1, fgy);
  b.y = l1 - b.z * gy;
  return b;
}
__device__ __forceinline__ BlockId xcd_block() {
  const unsigned gx = gridDim.x, gy = gridDim.y, gz = gridDim.z;
  const unsigned L = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
  unsigned l = xcd_linear(L, gx * gy * gz);
  BlockId b;
  b.x = l % gx; l /= gx;
  b.y = l % gy;
  b.z = l / gy;
  return b;
}
#endif

#if defined(__HIPCC__)
// s + (s of the lane 16 / 32 away): the two cross-row steps of a wave-wide butterfly sum through
// v_permlane16_swap / v_permlane32_swap (gfx950) instead of ds_bpermute (an LDS instruction each).
// Same two addends as `s + __shfl_xor(s, 16)`: bit-identical.
__device__ __forceinline__ float sum_xor16(float s) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float sum_xor32(float s) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// ... and the same steps for fp64 partials (two 32-bit swaps per step)
__device__ __forceinline__ double sum_xor16(double s) {
  const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
  const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
__device__ __forceinline__ double sum_xor32(double s) {
  const unsigned lo = (unsigned)__double2loint(s), hi = (unsigned)__double2hiint(s);
  const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
// x * sigmoid(x) with the hardware exp2 and reciprocal (about 1e-7 relative error).  `__fdividef` is a
// full IEEE division under this build's flags (v_div_scale / v_div_fmas / v_div_fixup: ten instructions).
// x * sigmoid(x) with the hardware exp2 / reciprocal (about 1e-7 relative error).  (Round 6: v / (1 + expf(-v)) in every
// kernel instead moves the 3D keypoints of ill-conditioned joints as much as any other reordering of the float32
// arithmetic does -- not closer to the float64 answer: HISTORY round-6 item 31.)
__device__ __forceinline__ float silu_fast(float v) { return v * __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
#endif

struct NodeArgs;
int launch_bifpn_node(const NodeArgs& a, hipStream_t s);
int pack_wino_weights(int cin, int cout, const float* w, const float* b, ConvWeights* out);
int wino_variant_from_env();
// tables: the persistent kernel's tables for this launch shape (wino_tables, uploaded by the plan) or nullptr
int launch_conv3d_wino(const ConvWeights& w, const Act& x, const Act& y, double* stats, hipStream_t s,
                       const InNorm* in, int variant = 4, const int* tables = nullptr);
// split-bf16 form of the same convolution (csrc/conv3d_bf16x3.hip; precision mode bf16x3)
int pack_bf16x3_weights(int cin, int cout, const float* w, const float* b, ConvWeights* out);
int launch_conv3d_bf16x3(const ConvWeights& w, const Act& x, const Act& y, double* stats, hipStream_t s,
                         const InNorm* in);
// ... and of the keypoint head's ConvTranspose2d k4 s2 p1 (csrc/deconv4_bf16x3.hip; no bias, no statistics)
int pack_deconv4_bf16x3_weights(int cin, int cout, const float* w, ConvWeights* out);
bool deconv4_bf16x3_eligible(int cout);
int launch_deconv4_bf16x3(const ConvWeights& w, const Act& x, const Act& y, hipStream_t s, const InNorm* in);
// ... and of the dense k x k convolutions with a generic split-bf16 kernel (csrc/conv_bf16x3.h: 3D k3 s2,
// 2D k3 s1 / k3 s2 / k5 s2; no gate)
bool conv_bf16x3_eligible(const ConvDesc& d);
int pack_conv_bf16x3_weights(const ConvDesc& d, const float* w, const float* b, ConvWeights* out);
int launch_conv_bf16x3(const ConvDesc& d, const ConvWeights& w, const Act& x, const Act& y, double* stats,
                       hipStream_t s, const InNorm* in);
// Precision mode of plans BUILT from now on: 0 = fp32 everywhere (default, the parity mode),
// 1 = bf16x3 for the layers that have a split-bf16 kernel.  Set by jh_set_precision().
int precision_mode();
void set_precision_mode(int m);
// vector-ALU stem convolution 3 -> 16, k3 s2 p1 (csrc/stem.hip)
void pack_stem_weights(const float* w_host, float* packed, int cout);
int launch_stem_conv(const Act& x, const float* w_dev, const Act& y, double* stats, hipStream_t s);
struct StemSource;
int launch_stem_conv_src(const StemSource& src, const Act& x, const float* w_dev, const Act& y, double* stats,
                         hipStream_t s);
int launch_deconv_c1(const Act& x, const double* stats, float inv_cnt, int in_act, const float* w,
                     const Act& y, hipStream_t s);

// Zero `bytes` (a multiple of 4) of device memory with a KERNEL.  The forward path uses this
// instead of hipMemsetAsync: inside a captured hipGraph a small memset node was observed not to
// take effect before the kernels that follow it (stale soft-argmax maxima of an earlier call
// survived a replay, tests/test_hip_predictor.py::test_multi_stream_set_calibration_sees_new_values),
// while kernel nodes of a linear chain are ordered reliably.
int launch_zero(void* p, size_t bytes, hipStream_t s);

// ---------------------------------------------------------------- layout moves
// NCHW/NCDHW fp32 -> channel-last padded (pad channels zeroed) and back.
int launch_to_channel_last(const float* src, const Act& dst, hipStream_t s);
int launch_from_channel_last(const Act& src, float* dst, hipStream_t s);

}  // namespace jh
