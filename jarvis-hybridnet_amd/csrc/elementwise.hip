// HBM-bound channel-last kernels around the MFMA convolutions.
//
//  norm_apply  InstanceNorm (affine-free, biased variance) + activation +
//              residual adds, optional pooled sums for squeeze-excite
//              (efficientnet.py:102-112,115-122; model.py:223-232;
//               v2vnet.py:17-19,32-43,54-55)
//  se_gate     squeeze-excite bottleneck MLP -> per-(n,c) sigmoid gate
//              (efficientnet.py:107-112)
//  depthwise   k x k depthwise convolution, stride 1, LDS-staged halo tiles
//              (efficientnet.py:68-71; the BiFPN sep-convs run fused, bifpn_node.hip)
//  fuse        BiFPN fast-normalised fusion node incl. nearest upsample /
//              2x2 max-pool of the neighbour level (model.py:309-353,119-125)
//  maxpool2    2x2/2 max pool (model.py:414-419)
//
// All of them stream float4 channel vectors; one thread owns a fixed channel
// quad and walks pixels, so per-channel reductions stay in registers.
#include "jh_common.h"

namespace jh {

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == ACT_RELU) return fmaxf(v, 0.f);
  if (act == ACT_SILU) return v / (1.f + expf(-v));
  return v;
}

// Block-level reduction of per-thread float4 partials that belong to channel
// quad (tid % q), followed by order-independent fp64 accumulation (exact_add: kLimbs
// doubles per value).  `nvals` = 1 (sum) or 2 (sum, sumsq); dst_stride = values per channel.
__device__ __forceinline__ void block_channel_reduce(float4 s1, float4 s2, int nvals, int q,
                                                     int rows, int tid, bool active,
                                                     double* dst, int dst_stride, float* sm) {
  // sm: [rows][q][nvals][4]
  if (active) {
    const int c4 = tid % q, r = tid / q;
    float4* p = reinterpret_cast<float4*>(sm) + ((size_t)r * q + c4) * nvals;
    p[0] = s1;
    if (nvals == 2) p[1] = s2;
  }
  __syncthreads();
  for (int i = tid; i < q * 4 * nvals; i += blockDim.x) {
    const int comp = i & 3, v = (i >> 2) % nvals, c4 = (i >> 2) / nvals;
    float acc = 0.f;
    for (int r = 0; r < rows; ++r) acc += sm[(((size_t)r * q + c4) * nvals + v) * 4 + comp];
    exact_add(dst + ((size_t)(c4 * 4 + comp) * dst_stride + v) * kLimbs, (double)acc);
  }
}

// ------------------------------------------------------------------ norm_apply
// ACT / R1 / R2 / Y are template parameters (the run-time form spent more scalar than vector
// instructions on its per-element branches) and four pixels per thread are in flight at a time.
// SiLU through v_exp_f32 / v_rcp_f32, as the convolutions apply it when they stage the same tensor.
// R1N: the residual operand r1 is itself a raw tensor whose InstanceNorm + ReLU (statistics `stats1`) is
// applied here, on load -- the V2V stage-entry tensors are then never materialised in normalised form.
// Block size 256, 512 or 1024 (launcher: whichever leaves the fewest lanes without a channel quad -- a 528-channel
// tensor has 132 quads: one pixel row per 256 threads used 132 of them).
template <int ACT, bool R1, bool R2, bool Y, bool R1N = false>
__global__ __launch_bounds__(1024) void norm_apply_kernel(
    const float* __restrict__ x, const double* __restrict__ stats, double eps,
    const float* __restrict__ r1, const float* __restrict__ r2, float* __restrict__ y,
    double* __restrict__ pool, int P, int Cp, int ppb, const double* __restrict__ stats1 = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = Cp >> 2;
  const int BT = blockDim.x;
  const int rows = BT / q;
  const int tid = threadIdx.x;
  const bool active = tid < rows * q;
  const int c4 = tid % q, row = tid / q;
  const int n = blockIdx.y;
  // mean / rstd: one channel per thread (the fp64 divisions and square root are ~80 instructions; every
  // thread doing its own four channels made this prologue longer than the 8 pixels a thread then
  // processes), shared through LDS behind the pooled-sum scratch
  float4 mean = make_float4(0, 0, 0, 0), rstd = make_float4(1, 1, 1, 1);
  if (stats) {
    float* mr = sm + (pool ? rows * q * 4 : 0);          // [Cp] mean, [Cp] rstd
    for (int c = tid; c < Cp; c += BT) {
      const double* st = stats + ((size_t)n * Cp + c) * kStatW;
      const double mu = exact_read(st) / (double)P;
      double var = exact_read(st + kLimbs) / (double)P - mu * mu;
      if (var < 0.0) var = 0.0;
      mr[c] = (float)mu;
      mr[Cp + c] = (float)(1.0 / sqrt(var + eps));
    }
    __syncthreads();
    if (active) {
      mean = *reinterpret_cast<const float4*>(mr + c4 * 4);
      rstd = *reinterpret_cast<const float4*>(mr + Cp + c4 * 4);
    }
    __syncthreads();                                     // (the pooled-sum reduce reuses nothing of mr, but
  }                                                      //  keeps the scratch layout simple)
  float4 mean1 = make_float4(0, 0, 0, 0), rstd1 = make_float4(1, 1, 1, 1);
  if (R1N) {
    float* mr = sm + (pool ? rows * q * 4 : 0);
    for (int c = tid; c < Cp; c += BT) {
      const double* st = stats1 + ((size_t)n * Cp + c) * kStatW;
      const double mu = exact_read(st) / (double)P;
      double var = exact_read(st + kLimbs) / (double)P - mu * mu;
      if (var < 0.0) var = 0.0;
      mr[c] = (float)mu;
      mr[Cp + c] = (float)(1.0 / sqrt(var + eps));
    }
    __syncthreads();
    if (active) {
      mean1 = *reinterpret_cast<const float4*>(mr + c4 * 4);
      rstd1 = *reinterpret_cast<const float4*>(mr + Cp + c4 * 4);
    }
    __syncthreads();
  }
  const size_t base = (size_t)n * P * Cp + c4 * 4;
  const int p0 = blockIdx.x * ppb;
  const int p1 = min(P, p0 + ppb);
  float4 ps = make_float4(0, 0, 0, 0);
  auto act1 = [](float v) __attribute__((always_inline)) {
    if (ACT == ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == ACT_SILU) return silu_fast(v);
    return v;
  };
  auto finish = [&](float4 v, float4 a1, float4 a2, size_t off) __attribute__((always_inline)) {
    v.x = (v.x - mean.x) * rstd.x; v.y = (v.y - mean.y) * rstd.y;
    v.z = (v.z - mean.z) * rstd.z; v.w = (v.w - mean.w) * rstd.w;
    if (R1N) {       // relu((r1 - mean1) * rstd1): the expression the stand-alone pass evaluates (ACT_RELU form above)
      a1.x = fmaxf((a1.x - mean1.x) * rstd1.x, 0.f); a1.y = fmaxf((a1.y - mean1.y) * rstd1.y, 0.f);
      a1.z = fmaxf((a1.z - mean1.z) * rstd1.z, 0.f); a1.w = fmaxf((a1.w - mean1.w) * rstd1.w, 0.f);
    }
    if (R1) { v.x += a1.x; v.y += a1.y; v.z += a1.z; v.w += a1.w; }
    v.x = act1(v.x); v.y = act1(v.y); v.z = act1(v.z); v.w = act1(v.w);
    if (R2) { v.x += a2.x; v.y += a2.y; v.z += a2.z; v.w += a2.w; }
    if (Y) *reinterpret_cast<float4*>(y + off) = v;         // !Y: pooled sums only
    ps.x += v.x; ps.y += v.y; ps.z += v.z; ps.w += v.w;
  };
  if (active) {
    constexpr int UN = 4;
    int p = p0 + row;
    for (; p + (UN - 1) * rows < p1; p += UN * rows) {
      float4 v[UN], a1[UN], a2[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const size_t off = base + (size_t)(p + u * rows) * Cp;
        v[u] = *reinterpret_cast<const float4*>(x + off);
        if (R1) a1[u] = *reinterpret_cast<const float4*>(r1 + off);
        if (R2) a2[u] = *reinterpret_cast<const float4*>(r2 + off);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) finish(v[u], a1[u], a2[u], base + (size_t)(p + u * rows) * Cp);
    }
    for (; p < p1; p += rows) {
      const size_t off = base + (size_t)p * Cp;
      float4 a1 = make_float4(0, 0, 0, 0), a2 = a1;
      if (R1) a1 = *reinterpret_cast<const float4*>(r1 + off);
      if (R2) a2 = *reinterpret_cast<const float4*>(r2 + off);
      finish(*reinterpret_cast<const float4*>(x + off), a1, a2, off);
    }
  }
  if (pool)
    block_channel_reduce(ps, ps, 1, q, rows, tid, active, pool + (size_t)n * Cp * kLimbs, 1, sm);
}

int launch_norm_apply(const Act& x, const double* stats, double eps, int act, const float* r1,
                      const float* r2, float* y, double* pool, hipStream_t s, const double* r1_stats,
                      int min_block_kb) {
  const int P = (int)x.pixels();
  const int q = x.Cp / 4;
  JH_REQUIRE(q >= 1 && q <= 256, "channel count out of range for norm_apply");
  // threads per block: a function of the channel count alone (part of the pooled sums' arithmetic, see below)
  int bt = 256;
  for (int cand = 512; cand <= 1024; cand *= 2)
    if ((cand / q) * q * bt > (bt / q) * q * cand + cand * bt / 16) bt = cand;      // > 1/16 better lane use
  const int rows = bt / q;
  // 8..64 pixels per row-slot, at most 64 blocks per image.  Chosen from the image size alone:
  // the pooled sums are sums of per-block fp32 partials, so the blocking is part of the
  // arithmetic and must not depend on the batch size.
  int iters = 8;
  while ((P + rows * iters - 1) / (rows * iters) > 64 && iters < 64) iters *= 2;
  // ... and at least `min_block_kb` of the tensor per block (the caller's choice -- plans of the time-batch class >= 8
  // ask for 64 KB; JH_NORM_MINKB overrides): a block's prologue turns the statistics of ALL its channels into mean / rstd
  // with fp64 arithmetic; with 480 channels and 16 pixels per block that was most of its life (small/small kernel time
  // 16.46 -> 16.31 ms per batch, medium 38.69 -> 38.19).  Single frame sets keep the small blocks: 12 images have to
  // spread over 256 CUs (2.059 against 2.090 ms per forward).
  const int min_kb = JH_ENV_KNOB("JH_NORM_MINKB") >= 0 ? JH_ENV_KNOB("JH_NORM_MINKB") : min_block_kb;
  while ((long)rows * iters * x.Cp * 4 < (long)min_kb * 1024 && iters < 64 && rows * iters < P) iters *= 2;
  const int ppb = rows * iters;
  dim3 grid((P + ppb - 1) / ppb, x.N);
  const size_t sm = ((pool ? (size_t)rows * q * 4 : 0) + (stats ? (size_t)2 * x.Cp : 0)) * sizeof(float);
  if (r1_stats) {     // residual operand normalised (+ReLU) on load: V2V residual blocks only
    JH_REQUIRE(stats && r1 && y && act == ACT_RELU && !pool, "norm_apply: normalised residual operand");
    if (r2)
      hipLaunchKernelGGL((norm_apply_kernel<ACT_RELU, true, true, true, true>), grid, dim3(bt), sm, s, x.p, stats, eps,
                         r1, r2, y, pool, P, x.Cp, ppb, r1_stats);
    else
      hipLaunchKernelGGL((norm_apply_kernel<ACT_RELU, true, false, true, true>), grid, dim3(bt), sm, s, x.p, stats, eps,
                         r1, r2, y, pool, P, x.Cp, ppb, r1_stats);
    JH_CHECK_HIP(hipGetLastError());
    return 0;
  }
#define JH_NA(A, B1, B2, BY)                                                                              \
  if (act == A && (r1 != nullptr) == B1 && (r2 != nullptr) == B2 && (y != nullptr) == BY) {             \
    hipLaunchKernelGGL((norm_apply_kernel<A, B1, B2, BY>), grid, dim3(bt), sm, s, x.p, stats, eps, r1, r2, y, \
                       pool, P, x.Cp, ppb, nullptr);                                                               \
    JH_CHECK_HIP(hipGetLastError());                                                                      \
    return 0;                                                                                             \
  }
#define JH_NA_ACT(A) JH_NA(A, false, false, false) JH_NA(A, false, false, true) JH_NA(A, true, false, true) \
                     JH_NA(A, true, true, true) JH_NA(A, false, true, true)
  JH_NA_ACT(ACT_NONE) JH_NA_ACT(ACT_RELU) JH_NA_ACT(ACT_SILU)
#undef JH_NA_ACT
#undef JH_NA
  JH_REQUIRE(false, "norm_apply: residual inputs without an output tensor");
}

// --------------------------------------------------------------------- se_gate
__global__ __launch_bounds__(256) void se_gate_kernel(
    const double* __restrict__ pool, int C, int Cp, int S, float inv_hw,
    const float* __restrict__ wr, const float* __restrict__ br, const float* __restrict__ we,
    const float* __restrict__ be, float* __restrict__ gate) {
  extern __shared__ __attribute__((aligned(16))) float sm[];   // [C] means, [S] hidden
  float* mean = sm;
  float* hid = sm + C;
  const int n = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    mean[c] = (float)(exact_read(pool + ((size_t)n * Cp + c) * kLimbs) * (double)inv_hw);
  __syncthreads();
  for (int j = threadIdx.x; j < S; j += blockDim.x) {
    float acc = br[j];
    for (int c = 0; c < C; ++c) acc = fmaf(wr[j * C + c], mean[c], acc);
    hid[j] = acc / (1.f + expf(-acc));
  }
  __syncthreads();
  for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
    float g = 0.f;
    if (c < C) {
      float acc = be[c];
      for (int j = 0; j < S; ++j) acc = fmaf(we[c * S + j], hid[j], acc);
      g = 1.f / (1.f + expf(-acc));
    }
    gate[(size_t)n * Cp + c] = g;
  }
}

int launch_se_gate(const double* pool, int N, int C, int Cp, int S, float inv_hw, const float* wr,
                   const float* br, const float* we, const float* be, float* gate,
                   hipStream_t s) {
  hipLaunchKernelGGL(se_gate_kernel, dim3(N), dim3(256), (C + S) * sizeof(float), s, pool, C, Cp,
                     S, inv_hw, wr, br, we, be, gate);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------- depthwise
// LDS-staged depthwise k x k (stride 1): a workgroup owns a 16 x 16 pixel tile of one
// image and a chunk of 32 channels.  The (16+k-1)^2 halo patch is staged once in LDS
// ([pixel][32 + 4] floats); every thread keeps the k*k weights of its channel quad in
// registers and produces 1 x 4 output strips, so each staged value is read from LDS
// (k+3)/4k times per tap instead of once.  Optional InstanceNorm statistics of the
// output are reduced in the block and added with fp64 atomics.
// POOL (images of one tile, i.e. H, W <= 16, channel chunk 16): the workgroup holds every output of its 16 channels
// of the image, so it also finishes InstanceNorm + SiLU for squeeze-excite in the same launch -- mean / rstd from its
// own (complete) statistics with norm_apply's expressions, then sum_p SiLU((y - mean) rstd) per channel from the
// outputs still in registers -> `pool`.  The second pass over the 6 x expanded tensor (norm_apply in pooled-sums-only
// form, efficientnet.py:102-106) is not launched for these blocks.
// TT: tile side.  16, or 20 for images 17 .. 20 pixels wide / high (the stride-16 level of the reference's DEFAULT
// 320-pixel geometry: 20 x 20 filled 39 % of its four 16 x 16 tiles and, not being one tile, could not fuse the pooled sums)
template <int K, int CC, bool POOL = false, int TT = 16>
__global__ __launch_bounds__(256) void depthwise_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
    double* __restrict__ stats, int H, int W, int Cp, double* __restrict__ pool = nullptr) {
  constexpr int T = TT, HT = T + K - 1, SP = CC + 4;
  static_assert(T % 4 == 0, "tiles are made of 4-pixel strips");
  constexpr int QN = CC / 4, SL = 256 / QN;         // channel quads of a chunk, pixel slots (threads per quad)
  constexpr int QS = QN == 8 ? 3 : 2;               // log2(QN)
  static_assert(CC == 32 || CC == 16, "channel chunk");
  extern __shared__ __attribute__((aligned(16))) float sm[];     // [HT*HT][SP], then [K*K][CC] weights
  float* wl = sm + HT * HT * SP;
  const int tid = threadIdx.x;
  const int tiles_x = (W + T - 1) / T;
  const BlockId bid = xcd_block();
  const int ox0 = (bid.x % tiles_x) * T, oy0 = (bid.x / tiles_x) * T;
  const int c0 = bid.y * CC;
  const int n = bid.z;
  const int q = min(CC, Cp - c0) >> 2;             // channel quads in this chunk (<= QN)
  typedef float df2 __attribute__((ext_vector_type(2)));
  typedef float df4 __attribute__((ext_vector_type(4)));
  // The kernel is latency-bound (two workgroups per CU, ~900 instructions per wave): everything it
  // needs from memory is requested in ONE round trip -- the weights of the chunk first, then the
  // whole halo patch (thread -> channel quad tid % 8, pixel slot tid / 8; pixels outside the image
  // and quads past the tensor are out-of-range buffer loads, i.e. the zero padding, no branches).
  const int lc4 = tid & (QN - 1), slot = tid >> QS;
  df4 wv = (df4){0.f, 0.f, 0.f, 0.f};
  if (tid < K * K * QN && lc4 < q)
    wv = *reinterpret_cast<const df4*>(w + (size_t)(tid >> QS) * Cp + c0 + lc4 * 4);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (size_t)n * H * W * Cp), 0, (int)((size_t)H * W * Cp * 4), 0x00020000);
  constexpr int NPX = HT * HT, ITERS = (NPX + SL - 1) / SL;
  df4 v[ITERS];
#pragma unroll
  for (int u = 0; u < ITERS; ++u) {
    const int pix = u * SL + slot;
    const int px = pix % HT, py = pix / HT;
    const int iy = oy0 + py - K / 2, ix = ox0 + px - K / 2;
    const bool ok = pix < NPX && lc4 < q && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const int off = ok ? ((iy * W + ix) * Cp + c0 + lc4 * 4) * 4 : (int)0x80000000;
    v[u] = __builtin_bit_cast(df4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
  }
  if (tid < K * K * QN) *reinterpret_cast<df4*>(wl + (tid >> QS) * CC + lc4 * 4) = wv;
#pragma unroll
  for (int u = 0; u < ITERS; ++u) {
    const int pix = u * SL + slot;
    if (pix < NPX) *reinterpret_cast<df4*>(sm + pix * SP + lc4 * 4) = v[u];
  }
  const int c4 = tid & (QN - 1);                   // fixed channel quad of this thread
  const bool active = c4 < q;
  __syncthreads();
  // statistics partials: packed fp32 sums of (v - pivot), (v - pivot)^2 around the thread's first output, un-shifted in
  // fp64 at the end (jh_common.h, stat_add(double); conv_mfma.h's epilogue does the same)
  df2 s1l = (df2){0.f, 0.f}, s1h = s1l, s2l = s1l, s2h = s1l, pvl = s1l, pvh = s1l;
  int cnt = 0;
  // strips of 4 pixels per tile row / per tile, strips per thread; the fused pooled sums keep a thread's strips in registers
  constexpr int SPR = T / 4, NSTR = T * SPR, NJ = (NSTR + SL - 1) / SL, NK = POOL ? NJ : 1, UNR = POOL ? NJ : 1;
  static_assert(!POOL || NJ <= 2, "registers for the strips of a thread");
  df2 alk[NK][4], ahk[NK][4];                      // packed fp32 FMAs: two channels per instruction
  unsigned okmk[NK];                               // (POOL) which of a strip's four pixels are inside the image
#pragma unroll
  for (int jj = 0; jj < NK; ++jj) okmk[jj] = 0;
  if (active) {
#pragma unroll UNR
    for (int j = 0; j < NJ; ++j) {
      const int g = (tid >> QS) + SL * j;          // strip g: row g / SPR, pixels 4 (g % SPR) .. + 3
      if (g >= NSTR) break;
      df2 (&al)[4] = alk[POOL ? j : 0], (&ah)[4] = ahk[POOL ? j : 0];
      unsigned& okm = okmk[POOL ? j : 0];
      const int ty = g / SPR, tx0 = (g % SPR) * 4;
#pragma unroll
      for (int o = 0; o < 4; ++o) { al[o] = (df2){0.f, 0.f}; ah[o] = al[o]; }
#pragma unroll 1
      for (int dy = 0; dy < K; ++dy) {
        df4 in[K + 3], kw[K];
#pragma unroll
        for (int dx = 0; dx < K; ++dx) kw[dx] = *reinterpret_cast<const df4*>(wl + (dy * K + dx) * CC + c4 * 4);
#pragma unroll
        for (int i = 0; i < K + 3; ++i)
          in[i] = *reinterpret_cast<const df4*>(sm + ((ty + dy) * HT + tx0 + i) * SP + c4 * 4);
#pragma unroll
        for (int dx = 0; dx < K; ++dx) {
          const df2 kl = (df2){kw[dx][0], kw[dx][1]}, kh = (df2){kw[dx][2], kw[dx][3]};
#pragma unroll
          for (int o = 0; o < 4; ++o) {
            al[o] = __builtin_elementwise_fma((df2){in[o + dx][0], in[o + dx][1]}, kl, al[o]);
            ah[o] = __builtin_elementwise_fma((df2){in[o + dx][2], in[o + dx][3]}, kh, ah[o]);
          }
        }
      }
      const int oy = oy0 + ty;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        const int ox = ox0 + tx0 + o;
        if (oy < H && ox < W) {
          *reinterpret_cast<df4*>(y + (((size_t)n * H + oy) * W + ox) * Cp + c0 + c4 * 4) =
              (df4){al[o][0], al[o][1], ah[o][0], ah[o][1]};
          if (cnt == 0) { pvl = al[o]; pvh = ah[o]; }
          const df2 dl = al[o] - pvl, dh = ah[o] - pvh;
          s1l += dl; s1h += dh;
          s2l = __builtin_elementwise_fma(dl, dl, s2l); s2h = __builtin_elementwise_fma(dh, dh, s2h);
          ++cnt;
          okm |= 1u << o;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();                               // the patch is dead: reuse LDS for the reduce
    // [SL rows][QN quads][2][4] doubles; fixed-order two-level sum (8 threads x SL / 8 rows, then 8 partials)
    double* smd = reinterpret_cast<double*>(sm);
    double* p = smd + ((size_t)(tid >> QS) * QN + c4) * 8;
    {
      const float pv[4] = {pvl[0], pvl[1], pvh[0], pvh[1]}, t1[4] = {s1l[0], s1l[1], s1h[0], s1h[1]},
                  t2[4] = {s2l[0], s2l[1], s2h[0], s2h[1]};
      const double nd = (double)cnt;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const double md = (double)pv[k], a1 = (double)t1[k];
        p[k] = fma(nd, md, a1);
        p[4 + k] = fma(md, fma(nd, md, 2.0 * a1), (double)t2[k]);
      }
    }
    double* red2 = smd + SL * QN * 8;              // [QN * 8 values][8 parts]
    __syncthreads();
    for (int i = tid; i < q * 8 * 8; i += 256) {
      const int part = i & 7, val = i >> 3;        // val = (quad, sum / sum of squares, component)
      const int comp = val & 3, sq = (val >> 2) & 1, cq = val >> 3;
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < SL / 8; ++r) acc += smd[(((size_t)(part * (SL / 8) + r) * QN + cq) * 2 + sq) * 4 + comp];
      red2[val * 8 + part] = acc;
    }
    __syncthreads();
    double* tot = red2 + QN * 8 * 8;               // (POOL) [QN * 8] totals of the image, then [CC] mean, [CC] rstd
    for (int i = tid; i < q * 8; i += 256) {
      const int comp = i & 3, sq = (i >> 2) & 1, cq = i >> 3;
      double acc = 0.0;
#pragma unroll
      for (int r = 0; r < 8; ++r) acc += red2[i * 8 + r];
      exact_add_rounded(stats + (((size_t)n * Cp + c0 + cq * 4 + comp) * 2 + sq) * kLimbs, acc);
      if (POOL) tot[i] = acc;
    }
    if (POOL) {
      // the image is one tile: `tot` ARE its statistics (the only contribution to stats[n][c]); mean / rstd exactly
      // as norm_apply_kernel derives them from the accumulators
      float* mr = reinterpret_cast<float*>(tot + QN * 8);
      float* red2f = reinterpret_cast<float*>(red2);     // the pooled sums (plain sums of bounded terms) stay fp32
      __syncthreads();
      if (tid < q * 4) {
        const int cq = tid >> 2, comp = tid & 3;
        const double P = (double)(H * W);
        const double mu = tot[(cq * 2 + 0) * 4 + comp] / P;
        double var = tot[(cq * 2 + 1) * 4 + comp] / P - mu * mu;
        if (var < 0.0) var = 0.0;
        mr[tid] = (float)mu;
        mr[CC + tid] = (float)(1.0 / sqrt(var + 1e-5));
      }
      __syncthreads();
      df4 ps = (df4){0.f, 0.f, 0.f, 0.f};
      if (active) {
        const df4 mean = *reinterpret_cast<const df4*>(mr + c4 * 4);
        const df4 rstd = *reinterpret_cast<const df4*>(mr + CC + c4 * 4);
#pragma unroll
        for (int jj = 0; jj < NK; ++jj)
#pragma unroll
          for (int o = 0; o < 4; ++o)
            if (okmk[jj] >> o & 1) {
              ps[0] += silu_fast((alk[jj][o][0] - mean[0]) * rstd[0]);
              ps[1] += silu_fast((alk[jj][o][1] - mean[1]) * rstd[1]);
              ps[2] += silu_fast((ahk[jj][o][0] - mean[2]) * rstd[2]);
              ps[3] += silu_fast((ahk[jj][o][1] - mean[3]) * rstd[3]);
            }
      }
      __syncthreads();                             // (mr has been read; the reduce below reuses the front of sm)
      reinterpret_cast<df4*>(sm)[(size_t)(tid >> QS) * QN + c4] = ps;        // [SL rows][QN quads][4]
      __syncthreads();
      for (int i = tid; i < q * 4 * 8; i += 256) {
        const int part = i & 7, val = i >> 3;      // val = (quad, component)
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < SL / 8; ++r) acc += sm[((size_t)(part * (SL / 8) + r) * QN + (val >> 2)) * 4 + (val & 3)];
        red2f[val * 8 + part] = acc;
      }
      __syncthreads();
      for (int i = tid; i < q * 4; i += 256) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) acc += red2f[i * 8 + r];
        exact_add(pool + ((size_t)n * Cp + c0 + i) * kLimbs, (double)acc);
      }
    }
  }
}

bool depthwise_can_pool(int H, int W) {
  return (H <= 16 && W <= 16) || (H <= 20 && W <= 20 && JH_ENV_KNOB("JH_DW_T20") != 0);
}

int launch_depthwise(const Act& x, const float* w, int k, float* y, double* stats, hipStream_t s, double* pool) {
  JH_REQUIRE(x.D == 1, "depthwise is 2D only");
  JH_REQUIRE(k == 3 || k == 5, "depthwise kernel size must be 3 or 5");
  // tile side: a function of the image only (16, or 20 for images 17 .. 20 pixels wide / high: one tile instead of four)
  const int tt = (std::max(x.H, x.W) > 16 && std::max(x.H, x.W) <= 20 && JH_ENV_KNOB("JH_DW_T20") != 0) ? 20 : 16;
  const int tiles = ((x.H + tt - 1) / tt) * ((x.W + tt - 1) / tt);
  if (pool) {
    // fused squeeze-excite pooled sums: one tile per image, channel chunk 16 (the form does not depend on the batch)
    JH_REQUIRE(depthwise_can_pool(x.H, x.W) && tiles == 1 && stats,
               "fused depthwise pooled sums need a one-tile image and statistics");
    dim3 grid(1, (x.Cp + 15) / 16, x.N);
    const int ht = tt + k - 1;
    size_t lds = (size_t)(ht * ht * (16 + 4) + k * k * 16) * sizeof(float);
    const size_t red = (size_t)(64 * 4 * 8 + 4 * 8 * 8 + 4 * 8) * sizeof(double) + 2 * 16 * sizeof(float);
    if (lds < red) lds = red;
#define JH_DWP(K, TTV) hipLaunchKernelGGL((depthwise_lds_kernel<K, 16, true, TTV>), grid, dim3(256), lds, s, x.p, w, y, stats, x.H, x.W, x.Cp, pool)
    if (k == 3) { if (tt == 20) JH_DWP(3, 20); else JH_DWP(3, 16); }
    else { if (tt == 20) JH_DWP(5, 20); else JH_DWP(5, 16); }
#undef JH_DWP
    JH_CHECK_HIP(hipGetLastError());
    return 0;
  }
  // Channel chunk per workgroup: 16 (33.6 KB of LDS at k = 5: four workgroups per CU) unless JH_DW_CC=32 (60.8 KB,
  // two per CU: the round-2 form).  The kernel is latency-bound; the chunk is part of no sum (statistics are
  // per channel), so the choice does not depend on anything but the knob.
  const int cc = (JH_ENV_KNOB("JH_DW_CC") == 32 && tt == 16) ? 32 : 16;
  dim3 grid(tiles, (x.Cp + cc - 1) / cc, x.N);
  const int ht = tt + k - 1;
  size_t lds = (size_t)(ht * ht * (cc + 4) + k * k * cc) * sizeof(float);
  const size_t red = (size_t)((256 / (cc / 4)) * (cc / 4) * 8 + (cc / 4) * 8 * 8) * sizeof(double);
  if (lds < red) lds = red;
#define JH_DW(K, CC, TTV) hipLaunchKernelGGL((depthwise_lds_kernel<K, CC, false, TTV>), grid, dim3(256), lds, s, x.p, w, y, stats, x.H, x.W, x.Cp)
  if (tt == 20) { if (k == 3) JH_DW(3, 16, 20); else JH_DW(5, 16, 20); }
  else if (k == 3) { if (cc == 32) JH_DW(3, 32, 16); else JH_DW(3, 16, 16); }
  else { if (cc == 32) JH_DW(5, 32, 16); else JH_DW(5, 16, 16); }
#undef JH_DW
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------------ fuse
__device__ __forceinline__ float4 fuse_fetch(const float* in, int mode, int n, int oy, int ox,
                                             int H, int W, int Cp, int c) {
  if (mode == FUSE_SAME)
    return *reinterpret_cast<const float4*>(in + (((size_t)n * H + oy) * W + ox) * Cp + c);
  if (mode == FUSE_UP2) {
    const int h = H >> 1, w = W >> 1;
    return *reinterpret_cast<const float4*>(in + (((size_t)n * h + (oy >> 1)) * w + (ox >> 1)) * Cp + c);
  }
  if (mode == FUSE_UP4) {
    const int h = H >> 2, w = W >> 2;
    return *reinterpret_cast<const float4*>(in + (((size_t)n * h + (oy >> 2)) * w + (ox >> 2)) * Cp + c);
  }
  // FUSE_POOL2: source is twice the size
  const int h = H * 2, w = W * 2;
  const float* b = in + (((size_t)n * h + oy * 2) * w + ox * 2) * Cp + c;
  const float4 a0 = *reinterpret_cast<const float4*>(b);
  const float4 a1 = *reinterpret_cast<const float4*>(b + Cp);
  const float4 a2 = *reinterpret_cast<const float4*>(b + (size_t)w * Cp);
  const float4 a3 = *reinterpret_cast<const float4*>(b + (size_t)w * Cp + Cp);
  return make_float4(fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x)),
                     fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y)),
                     fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z)),
                     fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w)));
}

__global__ __launch_bounds__(256) void fuse_kernel(FuseArgs f, float* __restrict__ out, int N,
                                                   int H, int W, int Cp) {
  const int q = Cp >> 2;
  const size_t total = (size_t)N * H * W * q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % q) * 4;
    size_t p = i / q;
    const int ox = (int)(p % W); p /= W;
    const int oy = (int)(p % H);
    const int n = (int)(p / H);
    float4 v = fuse_fetch(f.in[0], f.mode[0], n, oy, ox, H, W, Cp, c);
    // same rounding sequence as the reference expression w0*a + w1*b (+ w2*c)
    float4 acc = make_float4(__fmul_rn(f.w[0], v.x), __fmul_rn(f.w[0], v.y),
                             __fmul_rn(f.w[0], v.z), __fmul_rn(f.w[0], v.w));
    for (int k = 1; k < f.n_in; ++k) {
      v = fuse_fetch(f.in[k], f.mode[k], n, oy, ox, H, W, Cp, c);
      acc.x = __fadd_rn(acc.x, __fmul_rn(f.w[k], v.x));
      acc.y = __fadd_rn(acc.y, __fmul_rn(f.w[k], v.y));
      acc.z = __fadd_rn(acc.z, __fmul_rn(f.w[k], v.z));
      acc.w = __fadd_rn(acc.w, __fmul_rn(f.w[k], v.w));
    }
    acc.x = act_apply(acc.x, f.act); acc.y = act_apply(acc.y, f.act);
    acc.z = act_apply(acc.z, f.act); acc.w = act_apply(acc.w, f.act);
    *reinterpret_cast<float4*>(out + i * 4) = acc;
  }
}

int launch_fuse(const FuseArgs& f, const Act& out, hipStream_t s) {
  const size_t total = (size_t)out.N * out.H * out.W * (out.Cp / 4);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fuse_kernel, dim3(blocks), dim3(256), 0, s, f, out.p, out.N, out.H, out.W, out.Cp);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_maxpool2(const Act& x, float* y, hipStream_t s) {
  FuseArgs f{};
  f.in[0] = x.p; f.mode[0] = FUSE_POOL2; f.w[0] = 1.f; f.n_in = 1; f.act = ACT_NONE;
  Act o = x;
  o.p = y; o.H = x.H / 2; o.W = x.W / 2;
  return launch_fuse(f, o, s);
}

// ---------------------------------------------------------------- layout moves
__global__ __launch_bounds__(256) void to_channel_last_kernel(const float* __restrict__ src,
                                                              float* __restrict__ dst, int N, int C,
                                                              int Cp, size_t P) {
  const size_t total = (size_t)N * P * Cp;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    const size_t p = (i / Cp) % P;
    const size_t n = i / (Cp * P);
    dst[i] = (c < C) ? src[(n * C + c) * P + p] : 0.f;
  }
}

__global__ __launch_bounds__(256) void from_channel_last_kernel(const float* __restrict__ src,
                                                                float* __restrict__ dst, int N,
                                                                int C, int Cp, size_t P) {
  const size_t total = (size_t)N * C * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % P;
    const int c = (int)((i / P) % C);
    const size_t n = i / (P * C);
    dst[i] = src[(n * P + p) * Cp + c];
  }
}

int launch_to_channel_last(const float* src, const Act& dst, hipStream_t s) {
  const size_t total = dst.elems();
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(to_channel_last_kernel, dim3(blocks), dim3(256), 0, s, src, dst.p, dst.N,
                     dst.C, dst.Cp, dst.pixels());
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_from_channel_last(const Act& src, float* dst, hipStream_t s) {
  const size_t total = (size_t)src.N * src.C * src.pixels();
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(from_channel_last_kernel, dim3(blocks), dim3(256), 0, s, src.p, dst, src.N,
                     src.C, src.Cp, src.pixels());
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

__global__ __launch_bounds__(256) void zero_kernel(uint4* __restrict__ p16, size_t n16, unsigned* __restrict__ tail,
                                                   int ntail) {
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (size_t i = i0; i < n16; i += (size_t)gridDim.x * blockDim.x) p16[i] = make_uint4(0, 0, 0, 0);
  if (i0 < (size_t)ntail) tail[i0] = 0u;
}

int launch_zero(void* p, size_t bytes, hipStream_t s) {
  JH_REQUIRE(bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0, "zeroed range: 16-byte aligned, whole words");
  if (bytes == 0) return 0;
  const size_t n16 = bytes / 16;
  const int ntail = (int)((bytes - n16 * 16) / 4);
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint4*>(p), n16,
                     reinterpret_cast<unsigned*>(static_cast<char*>(p) + n16 * 16), ntail);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace jh
