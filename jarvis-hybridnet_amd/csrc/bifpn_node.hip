// One BiFPN node as ONE kernel:
//
//   out_raw = pointwise( depthwise3x3( act( sum_i w_i * resample_i( IN_i(in_i) ) ) ) ) + bias
//
// i.e. the fast-normalised fusion (incl. nearest upsampling / 2x2 max-pooling of the
// neighbour level and the InstanceNorm of every input, applied on load from the
// producer's fused statistics), the SiLU, the SeparableConvBlock's depthwise 3x3 and
// its 1x1 pointwise convolution with bias, plus the statistics of the output for the
// InstanceNorm that follows.  Replaces, per node, jarvis/efficienttrack/model.py:
// 309-353 (fusion expressions) + :223-232 (SeparableConvBlock.forward) and the head
// expression :119-126.  Unfused this was fuse + depthwise + pointwise + norm_apply =
// 9 passes over a P3-sized tensor; fused it is (n_in reads + 1 write).
//
// Tile: 8 x 16 output pixels per workgroup (4 waves x 2 row blocks of 16 pixels).
//   1. mean / rstd of every normalised input -> LDS
//   2. per channel chunk: fused + activated halo tile (10 x 18 pixels) -> LDS F,
//      depthwise 3x3 from F -> LDS operand tile A[128][Cp + 4]
//   3. MFMA 16x16x4 fp32 over K = Cp for 4 output-channel blocks at a time
//      (weights streamed from L2 in packed B-operand order), shared epilogue.
#include <algorithm>
#include <type_traits>
#include <cstdlib>
#include "conv_mfma.h"
#include "bifpn_node.h"

namespace jh {


__device__ __forceinline__ float4 node_fetch(const float* in, int mode, int n, int oy, int ox, int H,
                                             int W, int Cp, int c) {
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
  const int h = H * 2, w = W * 2;     // FUSE_POOL2 (max commutes with the monotone IN map)
  const float* b = in + (((size_t)n * h + oy * 2) * w + ox * 2) * Cp + c;
  const float4 a0 = *reinterpret_cast<const float4*>(b);
  const float4 a1 = *reinterpret_cast<const float4*>(b + Cp);
  const float4 a2 = *reinterpret_cast<const float4*>(b + (size_t)w * Cp);
  const float4 a3 = *reinterpret_cast<const float4*>(b + (size_t)w * Cp + Cp);
  return make_float4(fmaxf(fmaxf(a0.x, a1.x), fmaxf(a2.x, a3.x)), fmaxf(fmaxf(a0.y, a1.y), fmaxf(a2.y, a3.y)),
                     fmaxf(fmaxf(a0.z, a1.z), fmaxf(a2.z, a3.z)), fmaxf(fmaxf(a0.w, a1.w), fmaxf(a2.w, a3.w)));
}

// The resampling modes are template parameters: with them known at compile time the
// halo loads of several items can be issued back to back (no data-dependent branches),
// which is what hides the HBM latency of this otherwise latency-bound prologue.
// CP: channel count known at compile time (0 = run-time a.Cp).  PMC on the run-time form: the 56 MFMAs of
// a wave came with 147 other vector and 102 scalar instructions of address arithmetic and loop control
// (operand offsets are multiples of Cp); with CP they are immediates of fully unrolled loops.
template <int NIN, int M0, int M1, int M2, bool ONE, int CP = 0>
__global__ __launch_bounds__(512, 6) void bifpn_node_kernel(const NodeArgs a) {
  constexpr int kModes[3] = {M0, M1, M2};
  // items in flight per thread: 512 x 5 covers a whole 56-channel halo tile in one round
  // trip; with three inputs that costs > 80 VGPRs, i.e. the third workgroup per CU, which
  // is worth more than the single round trip
  constexpr int U = ONE ? ((NIN == 3 || M1 == FUSE_POOL2) ? 2 : 6) : (NIN == 3 ? 3 : 5);
  constexpr int NT = 512;                        // threads (8 waves: latency-bound prologue)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int Cp = CP ? CP : a.Cp;
  const int SA = Cp + 4;                         // operand tile stride (floats): b64 reads conflict-free
  const int SF = CP ? CP : a.cf;                 // halo tile stride: unpadded, so that consecutive
                                                 // lanes touch consecutive 16-byte words
  float* mr_ = lds;                              // [3][Cp] mean, then [3][Cp] rstd
  float* dwl = mr_ + 3 * Cp * 2;                 // [9][Cp] depthwise weights
  float* Ft = dwl + 9 * Cp;                      // [180][SF]
  // With a single channel chunk the operand tile is built in registers and then written
  // OVER the halo tile (a.alias): 44 KB instead of 77 KB of LDS, 3 workgroups per CU.
  float* At = a.alias ? Ft : Ft + kNodePY * kNodePX * SF;      // [128][SA]
  // a.blds: the packed pointwise weights (<= 16 KB) sit behind the operand tile in LDS, so
  // the MFMA loop has no L2 round trips (it was L2-latency-bound: 8 MFMAs per iteration
  // cannot cover a weight fetch).  The epilogue scratch then reuses the dead operand tile.
  float* Bl = At + 128 * SA;                     // [Cp/8][cout_p16/16][64][2]
  float* red = a.alias ? (a.blds ? At : At + 128 * SA) : Ft;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, mrow = lane & 15, kq = lane >> 4;
  const int tiles_x = (a.W + kNodeTX - 1) / kNodeTX;
  const BlockId bid = xcd_block();
  const int tile_x = bid.x % tiles_x, tile_y = bid.x / tiles_x;
  const int n = bid.y;
  const int oy0 = tile_y * kNodeTY, ox0 = tile_x * kNodeTX;

  const int nk8 = Cp >> 3, nb = CP ? kNodeNRG : (a.cout_p16 >> 4);      // (CP: the launcher checks cout_p16)
  const float2* wl = reinterpret_cast<const float2*>(a.pw) + lane;
  if (a.abl & 16) {             // experiment: de-phase the first generation of workgroups
    const unsigned L = blockIdx.x + gridDim.x * blockIdx.y;
    if (L < 768u) {
      const int k = (L >> 3) % 3;
      for (int i = 0; i < k * (a.abl >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }
  }

  // 1 + 2. statistics -> mean / rstd, fused halo tile -> depthwise -> operand tile
  if constexpr (ONE) {
    // Round 2: this kernel is bound by instruction issue (about 950 non-MFMA instructions per
    // wave and tile at ~9 SIMD cycles each), the halo phase being the
    // largest part, so it is written for instruction COUNT:
    //  * InstanceNorm and the fusion weights are folded per channel into one multiply-add per
    //    input, fused = sum_k x_k * a_k + B with a_k = w_k rstd_k and B = -sum_k w_k mean_k rstd_k
    //    (computed once per workgroup by the statistics threads, held in registers by the
    //    consumers: no per-item LDS reads, packed fp32 FMAs);
    //  * halo pixels outside the image (the depthwise convolution's zero padding) are buffer
    //    loads with bit 31 set in the offset -- out of range, they return 0 without a branch --
    //    and one multiplication by 0 / 1 after the activation.
    // The statistics (and depthwise weights) are requested FIRST and consumed after the first
    // batch of halo loads has been issued (loads return in order).
    double sv0 = 0.0, sv1 = 0.0;
    bool has_st = false;
    const int sk = tid / Cp, scn = tid - sk * Cp;        // launcher: n_in * Cp <= 512
    if (tid < a.n_in * Cp && a.st[sk]) {
      const double* st = a.st[sk] + ((size_t)n * Cp + scn) * kStatW;
      sv0 = exact_read(st);
      sv1 = exact_read(st + kLimbs);
      has_st = true;
    }
    float dwv[2];                                        // launcher: 9 * Cp <= 1024
#pragma unroll
    for (int j = 0; j < 2; ++j) dwv[j] = tid + j * NT < 9 * Cp ? a.dw[tid + j * NT] : 0.f;
    bool first = true;
    // thread -> (channel quad c4 = tid % 16, pixel slot tid / 16); 16 - q of every 16 lanes idle
    // (q = 14 for the 56-channel pyramid)
    const int q = Cp >> 2;
    const int c4 = tid & 15, slot = tid >> 4;
    // lanes past the last channel quad (2 of every 16 for the 56-channel pyramid) repeat the last quad:
    // identical values to identical addresses, and no per-item exec masking
    constexpr bool cact = true;
    const int c = min(c4, q - 1) * 4;
    constexpr int NPX = kNodePY * kNodePX;
    typedef float nf2 __attribute__((ext_vector_type(2)));
    typedef float nf4 __attribute__((ext_vector_type(4)));
    __amdgpu_buffer_rsrc_t rs[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const size_t plane = node_plane(kModes[k], a.H, a.W) * Cp;
      rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in[k] + (size_t)n * plane), 0,
                                                (int)(plane * 4), 0x00020000);
    }
    nf2 ak[NIN][2], bb[2];                               // folded norm + fusion weights of this quad
    int py = slot >= kNodePX ? 1 : 0;
    int px = slot - py * kNodePX;
    int pix = slot;
    // one round = UU halo pixels per thread (32 pixel slots x UU); the 180 pixels of the tile are 4 + 2
    // rounds of 32 for the U = 4 variants (a second round of 4 would issue two empty items)
    auto round = [&](auto uu_c) __attribute__((always_inline)) {
      constexpr int UU = decltype(uu_c)::value;
      nf4 v[UU][NIN];
      float msk[UU];
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
        // (unsigned compares: one test per axis; only the last item of a round can run past the tile)
        const bool ok = cact && (u * 32 + 31 < NPX || pix + u * 32 < NPX) && (unsigned)iy < (unsigned)a.H &&
                        (unsigned)ix < (unsigned)a.W;
        msk[u] = ok ? 1.f : 0.f;
        const unsigned oob = ok ? 0u : 0x80000000u;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
          if (kModes[k] == FUSE_POOL2) {                 // max commutes with the monotone IN map
            const int w2 = a.W * 2;
            const unsigned o = (unsigned)(((iy * 2 * w2 + ix * 2) * Cp + c) * 4) | oob;
            const nf4 a0 = __builtin_bit_cast(nf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], o, 0, 0));
            const nf4 a1 = __builtin_bit_cast(nf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], o, Cp * 4, 0));
            const nf4 a2 = __builtin_bit_cast(nf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], o, w2 * Cp * 4, 0));
            const nf4 a3 = __builtin_bit_cast(nf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], o, (w2 + 1) * Cp * 4, 0));
            v[u][k] = __builtin_elementwise_max(__builtin_elementwise_max(a0, a1), __builtin_elementwise_max(a2, a3));
          } else {
            const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
            const unsigned o = (unsigned)((((iy >> sh) * (a.W >> sh) + (ix >> sh)) * Cp + c) * 4) | oob;
            v[u][k] = __builtin_bit_cast(nf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], o, 0, 0));
          }
        }
        px += 32 - kNodePX; py += 1;
        if (px >= kNodePX) { px -= kNodePX; py += 1; }
      }
      if (first) {                               // (uniform)
        first = false;
        if (tid < a.n_in * Cp) {
          float mean = 0.f, rstd = 1.f;
          if (has_st) {                          // biased variance, eps 1e-5
            const double mu = sv0 * (double)a.inv_cnt[sk];
            double var = sv1 * (double)a.inv_cnt[sk] - mu * mu;
            if (var < 0.0) var = 0.0;
            mean = (float)mu;
            rstd = (float)(1.0 / sqrt(var + 1e-5));
          }
          const float ak1 = a.w[sk] * rstd;
          mr_[sk * Cp + scn] = ak1;                      // a_k
          mr_[3 * Cp + sk * Cp + scn] = -mean * ak1;     // b_k
        }
#pragma unroll
        for (int j = 0; j < 2; ++j)
          if (tid + j * NT < 9 * Cp) dwl[tid + j * NT] = dwv[j];
        __syncthreads();
        bb[0] = (nf2){0.f, 0.f}; bb[1] = bb[0];
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
          const float4 av = cact ? *reinterpret_cast<const float4*>(mr_ + k * Cp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 bv = cact ? *reinterpret_cast<const float4*>(mr_ + 3 * Cp + k * Cp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
          ak[k][0] = (nf2){av.x, av.y}; ak[k][1] = (nf2){av.z, av.w};
          bb[0] += (nf2){bv.x, bv.y}; bb[1] += (nf2){bv.z, bv.w};
        }
      }
#pragma unroll
      for (int u = 0; u < UU; ++u) {
        const int pu = pix + u * 32;
        if (u * 32 + 31 < NPX || pu < NPX) {
          nf2 lo = bb[0], hi = bb[1];
#pragma unroll
          for (int k = 0; k < NIN; ++k) {
            lo = __builtin_elementwise_fma((nf2){v[u][k][0], v[u][k][1]}, ak[k][0], lo);
            hi = __builtin_elementwise_fma((nf2){v[u][k][2], v[u][k][3]}, ak[k][1], hi);
          }
          if (!(a.abl & 1) && a.act == ACT_SILU) {
            // x * sigmoid(x) with the hardware exp2 / reciprocal (about 1e-7 relative error)
            lo = (nf2){silu_fast(lo.x), silu_fast(lo.y)};
            hi = (nf2){silu_fast(hi.x), silu_fast(hi.y)};
          } else if (!(a.abl & 1) && a.act == ACT_RELU) {
            lo = __builtin_elementwise_max(lo, (nf2){0.f, 0.f});
            hi = __builtin_elementwise_max(hi, (nf2){0.f, 0.f});
          }
          const nf2 mm = (nf2){msk[u], msk[u]};
          lo *= mm;
          hi *= mm;
          *reinterpret_cast<float4*>(Ft + pu * SF + c) = make_float4(lo.x, lo.y, hi.x, hi.y);
        }
      }
      pix += 32 * UU;
    };
    if constexpr (U == 4) {
      round(std::integral_constant<int, 4>{});
      round(std::integral_constant<int, 2>{});
    } else {
      for (int p0 = 0; p0 < NPX; p0 += 32 * U) round(std::integral_constant<int, U>{});
    }
    __syncthreads();
    // depthwise 3x3: the kernel's on-chip floor is LDS bandwidth, and the depthwise reads were
    // two thirds of it (9 data + 9 weight quads per output quad).  Each thread therefore owns
    // a 1 x 4 strip of outputs of its channel quad (32 strips x 16 lanes = the whole tile in
    // one round) and walks it tap row by tap row: 3 weight + 6 data quads per row feed 12
    // FMAs x 4 channels, i.e. 6.75 instead of 18 LDS quads per output quad.  All reads of the
    // halo tile precede the barrier, all writes of the aliased operand tile follow it.
    const bool dact = cact && !(a.abl & 2);
    const bool bact = a.blds && tid * 8 < nk8 * nb * 128;
    const int sty = slot >> 2, stx = (slot & 3) * 4;
    nf2 dlo[4], dhi[4];                          // packed fp32 FMAs: two channels per instruction
#pragma unroll
    for (int i = 0; i < 4; ++i) { dlo[i] = (nf2){0.f, 0.f}; dhi[i] = dlo[i]; }
    if (dact) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        nf2 wl2[3], wh2[3], xl[6], xh[6];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float4 t = *reinterpret_cast<const float4*>(dwl + (dy * 3 + dx) * Cp + c);
          wl2[dx] = (nf2){t.x, t.y}; wh2[dx] = (nf2){t.z, t.w};
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const float4 t = *reinterpret_cast<const float4*>(Ft + ((sty + dy) * kNodePX + stx + j) * SF + c);
          xl[j] = (nf2){t.x, t.y}; xh[j] = (nf2){t.z, t.w};
        }
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dlo[i] = __builtin_elementwise_fma(xl[i + dx], wl2[dx], dlo[i]);
            dhi[i] = __builtin_elementwise_fma(xh[i + dx], wh2[dx], dhi[i]);
          }
      }
    }
    __syncthreads();
    if (dact) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(At + (sty * kNodeTX + stx + i) * SA + c) =
            make_float4(dlo[i].x, dlo[i].y, dhi[i].x, dhi[i].y);
    }
    if (bact) {                                  // (the barrier has passed: the halo tile is dead)
      const float4 b0 = *reinterpret_cast<const float4*>(a.pw + tid * 8);
      const float4 b1 = *reinterpret_cast<const float4*>(a.pw + tid * 8 + 4);
      *reinterpret_cast<float4*>(Bl + tid * 8) = b0;
      *reinterpret_cast<float4*>(Bl + tid * 8 + 4) = b1;
    }
    __syncthreads();
  } else {
    // 1. depthwise weights and statistics -> mean / rstd (biased variance, eps 1e-5)
    for (int i = tid; i < 9 * Cp; i += NT) dwl[i] = a.dw[i];
    for (int i = tid; i < a.n_in * Cp; i += NT) {
      const int k = i / Cp, c = i % Cp;
      float mean = 0.f, rstd = 1.f;
      if (a.st[k]) {
        const double* st = a.st[k] + ((size_t)n * Cp + c) * kStatW;
        const double mu = exact_read(st) * (double)a.inv_cnt[k];
        double var = exact_read(st + kLimbs) * (double)a.inv_cnt[k] - mu * mu;
        if (var < 0.0) var = 0.0;
        mean = (float)mu;
        rstd = (float)(1.0 / sqrt(var + 1e-5));
      }
      mr_[k * Cp + c] = mean;
      mr_[3 * Cp + k * Cp + c] = rstd;
    }
    __syncthreads();

    // general form: one channel chunk at a time (models with wider pyramids)
    for (int cf0 = 0; cf0 < Cp; cf0 += a.cf) {
      const int cw = min(a.cf, Cp - cf0);          // multiple of 4
      const int q = cw >> 2;
      const int total = kNodePY * kNodePX * q;
      for (int b0 = 0; b0 < total; b0 += NT * U) {
        const int base = b0 + tid;
        float4 v[U][NIN];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * NT;
          const int c4 = idx % q, pix = idx / q;
          const int px = pix % kNodePX, py = pix / kNodePX;
          const int iy = oy0 - 1 + py, ix = ox0 - 1 + px;
          ok[u] = idx < total && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
#pragma unroll
          for (int k = 0; k < NIN; ++k)
            v[u][k] = ok[u] ? node_fetch(a.in[k], kModes[k], n, iy, ix, a.H, a.W, Cp, cf0 + c4 * 4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int idx = base + u * NT;
          if (idx >= total) break;
          const int c4 = idx % q, pix = idx / q;
          const int c = cf0 + c4 * 4;
          float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
          if (ok[u]) {
#pragma unroll
            for (int k = 0; k < NIN; ++k) {
              float4 x = v[u][k];
              const float4 mu = *reinterpret_cast<const float4*>(mr_ + k * Cp + c);
              const float4 rs = *reinterpret_cast<const float4*>(mr_ + 3 * Cp + k * Cp + c);
              x.x = (x.x - mu.x) * rs.x; x.y = (x.y - mu.y) * rs.y;
              x.z = (x.z - mu.z) * rs.z; x.w = (x.w - mu.w) * rs.w;
              const float wk = a.w[k];
              if (k == 0) {
                r = make_float4(__fmul_rn(wk, x.x), __fmul_rn(wk, x.y), __fmul_rn(wk, x.z), __fmul_rn(wk, x.w));
              } else {
                r.x = __fadd_rn(r.x, __fmul_rn(wk, x.x)); r.y = __fadd_rn(r.y, __fmul_rn(wk, x.y));
                r.z = __fadd_rn(r.z, __fmul_rn(wk, x.z)); r.w = __fadd_rn(r.w, __fmul_rn(wk, x.w));
              }
            }
            if (!(a.abl & 1)) {
              r.x = node_act(r.x, a.act); r.y = node_act(r.y, a.act);
              r.z = node_act(r.z, a.act); r.w = node_act(r.w, a.act);
            }
          }
          *reinterpret_cast<float4*>(Ft + pix * SF + c4 * 4) = r;
        }
      }
      __syncthreads();
      // depthwise 3x3, two output rows (32 pixels x q channel quads <= 512 items) per round.
      // When the operand tile aliases the halo tile the rounds are what makes that legal:
      // operand rows <= 2r+1 (stride SA <= 18 * SF / 16 floats) only overwrite halo rows
      // <= 2r+1, which no later round reads; the barrier orders this round's reads
      // before its writes.
      const int ditems = (a.abl & 2) ? 0 : 32 * q;
      const bool bact = a.blds && tid * 8 < nk8 * nb * 128;
#pragma unroll 1
      for (int r = 0; r < kNodeTY / 2; ++r) {
        float4 dacc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c4 = tid % q, p = r * 32 + tid / q;
        const int tx = p % kNodeTX, ty = p / kNodeTX;
        if (tid < ditems) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
              const float4 v = *reinterpret_cast<const float4*>(Ft + ((ty + dy) * kNodePX + tx + dx) * SF + c4 * 4);
              const float4 k = *reinterpret_cast<const float4*>(dwl + (dy * 3 + dx) * Cp + cf0 + c4 * 4);
              dacc.x = fmaf(v.x, k.x, dacc.x); dacc.y = fmaf(v.y, k.y, dacc.y);
              dacc.z = fmaf(v.z, k.z, dacc.z); dacc.w = fmaf(v.w, k.w, dacc.w);
            }
        }
        if (a.alias) __syncthreads();
        if (tid < ditems) *reinterpret_cast<float4*>(At + p * SA + cf0 + c4 * 4) = dacc;
      }
      if (bact) {                                  // (the last round's barrier has passed: the
        const float4 b0 = *reinterpret_cast<const float4*>(a.pw + tid * 8);     // halo tile is dead)
        const float4 b1 = *reinterpret_cast<const float4*>(a.pw + tid * 8 + 4);
        *reinterpret_cast<float4*>(Bl + tid * 8) = b0;
        *reinterpret_cast<float4*>(Bl + tid * 8 + 4) = b1;
      }
      __syncthreads();
    }
  }

  // 3. pointwise convolution on the matrix cores: wave w owns pixels 16w .. 16w+15
  if (a.abl & 4) return;
  const float2* A2 = reinterpret_cast<const float2*>(At);
  const int SA2 = SA >> 1;
  const int abase = (wave * 16 + mrow) * SA2 + kq;
  EpilogueArgs e;
  e.y = a.y + (size_t)n * a.H * a.W * a.cout_p;
  e.bias = a.bias;
  e.stats = a.stats ? a.stats + (size_t)n * a.cout_p * kStatW : nullptr;
  e.Dout = 1; e.Hout = a.H; e.Wout = a.W; e.Hy = a.H; e.Wy = a.W;
  const bool full_tile = oy0 + kNodeTY <= a.H && ox0 + kNodeTX <= a.W;      // (uniform)
  e.cout_p = a.cout_p; e.cout_p16 = a.cout_p16; e.os = 1; e.osz = 1; e.offz = e.offy = e.offx = 0;
  for (int nb0 = 0; nb0 < nb; nb0 += kNodeNRG) {
    f32x4 acc[1][kNodeNRG];
#pragma unroll
    for (int nr = 0; nr < kNodeNRG; ++nr) acc[0][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int boff[kNodeNRG];
#pragma unroll
    for (int nr = 0; nr < kNodeNRG; ++nr) boff[nr] = min(nb0 + nr, nb - 1) * 64;
    if (CP != 0) {                                // (implies a.blds)
      const float2* B2 = reinterpret_cast<const float2*>(Bl) + lane;
#pragma unroll
      for (int k8 = 0; k8 < (CP >> 3); ++k8) {
        const float2 ac = A2[abase + k8 * 4];
        float2 bc[kNodeNRG];
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr) bc[nr] = B2[(k8 * kNodeNRG + nr) * 64];
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.x, bc[nr].x, acc[0][nr], 0, 0, 0);
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.y, bc[nr].y, acc[0][nr], 0, 0, 0);
      }
    } else if (a.blds) {
      const float2* B2 = reinterpret_cast<const float2*>(Bl) + lane;
#pragma unroll 1
      for (int k8 = 0; k8 < nk8; ++k8) {
        const float2 ac = A2[abase + k8 * 4];
        float2 bc[kNodeNRG];
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr) bc[nr] = B2[k8 * nb * 64 + boff[nr]];
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.x, bc[nr].x, acc[0][nr], 0, 0, 0);
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.y, bc[nr].y, acc[0][nr], 0, 0, 0);
      }
    } else if (!a.alias && kNodePY * kNodePX * SF >= nk8 * kNodeNRG * 128 && nk8 * kNodeNRG * 32 <= 6 * NT &&
               !(a.abl & 32)) {
      // Wide pyramids (88 / 160 channels; round 4): the weights of this group of column blocks go through the dead
      // halo tile in ONE cooperative copy (all its loads in flight at once) instead of one L2 round trip per
      // channel step -- with 8 MFMAs per step and two waves per SIMD the two-deep register prefetch below covers
      // about half of an L2 latency, and a tile of the 160-channel pyramid spent 60 such steps (the nodes of the
      // 8 x 8 and smaller levels: 130-220 us per launch whatever their size).
      const int gcb = min(kNodeNRG, nb - nb0);
      const int per_k8 = gcb * 32, total = nk8 * per_k8;                 // float4 items
      const float4* src = reinterpret_cast<const float4*>(a.pw);
      float4 t[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int idx = tid + u * NT;
        if (idx < total) {
          const int k8 = idx / per_k8, j = idx - k8 * per_k8;
          t[u] = src[(size_t)(k8 * nb + nb0) * 32 + j];
        }
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int idx = tid + u * NT;
        if (idx < total) {
          const int k8 = idx / per_k8, j = idx - k8 * per_k8;
          reinterpret_cast<float4*>(Ft)[k8 * (kNodeNRG * 32) + j] = t[u];
        }
      }
      __syncthreads();
      const float2* B2 = reinterpret_cast<const float2*>(Ft) + lane;
      int lo[kNodeNRG];
#pragma unroll
      for (int nr = 0; nr < kNodeNRG; ++nr) lo[nr] = min(nr, gcb - 1) * 64;
#pragma unroll 1
      for (int k8 = 0; k8 < nk8; ++k8) {
        const float2 ac = A2[abase + k8 * 4];
        float2 bc[kNodeNRG];
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr) bc[nr] = B2[k8 * (kNodeNRG * 64) + lo[nr]];
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.x, bc[nr].x, acc[0][nr], 0, 0, 0);
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.y, bc[nr].y, acc[0][nr], 0, 0, 0);
      }
    } else {
      float2 bn[kNodeNRG], bnn[kNodeNRG];
#pragma unroll
      for (int nr = 0; nr < kNodeNRG; ++nr) {
        bn[nr] = wl[boff[nr]];
        bnn[nr] = wl[(size_t)min(1, nk8 - 1) * nb * 64 + boff[nr]];
      }
      for (int k8 = 0; k8 < nk8; ++k8) {
        float2 bc[kNodeNRG];
        const float2 ac = A2[abase + k8 * 4];
        const float2* wn = wl + (size_t)min(k8 + 2, nk8 - 1) * nb * 64;
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr) { bc[nr] = bn[nr]; bn[nr] = bnn[nr]; bnn[nr] = wn[boff[nr]]; }
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.x, bc[nr].x, acc[0][nr], 0, 0, 0);
#pragma unroll
        for (int nr = 0; nr < kNodeNRG; ++nr)
          acc[0][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac.y, bc[nr].y, acc[0][nr], 0, 0, 0);
      }
    }
    __syncthreads();          // the scratch may be the (dead) halo tile
    if (!(a.abl & 8)) {
      if (full_tile) conv_epilogue<1, kNodeNRG, kNodeTY, kNodeTX, 8, true>(acc, e, red, nb0, 0, oy0, ox0, tid);
      else conv_epilogue<1, kNodeNRG, kNodeTY, kNodeTX, 8>(acc, e, red, nb0, 0, oy0, ox0, tid);
    }
    __syncthreads();
  }
}

template <int NIN, int M0, int M1, int M2, bool ONE, int CP = 0>
static int launch_node_one(const NodeArgs& a, size_t lds, hipStream_t s) {
  auto kern = bifpn_node_kernel<NIN, M0, M1, M2, ONE, CP>;
  static bool big = false;
  if (lds > 64 * 1024 && !big) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  const int tiles = ((a.H + kNodeTY - 1) / kNodeTY) * ((a.W + kNodeTX - 1) / kNodeTX);
  hipLaunchKernelGGL(kern, dim3(tiles, a.N), dim3(512), lds, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int NIN, int M0, int M1, int M2>
static int launch_node_variant(const NodeArgs& a, size_t lds, hipStream_t s) {
  // the small model's pyramid (56 channels, one MFMA group of 4 column blocks, weights in LDS)
  if (a.alias && a.blds && a.Cp == 56 && a.cf == 56 && a.cout_p16 == 16 * kNodeNRG && JH_ENV_KNOB("JH_NODE_CP") != 0)
    return launch_node_one<NIN, M0, M1, M2, true, 56>(a, lds, s);
  if (a.alias) return launch_node_one<NIN, M0, M1, M2, true>(a, lds, s);
  return launch_node_one<NIN, M0, M1, M2, false>(a, lds, s);
}

bool bifpn_rows_eligible(const NodeArgs& a);
int launch_bifpn_rows(const NodeArgs& a, hipStream_t s);

int launch_bifpn_node(const NodeArgs& args, hipStream_t s) {
  if (bifpn_rows_eligible(args)) return launch_bifpn_rows(args, s);      // csrc/bifpn_rows.hip
  NodeArgs a = args;
  if (JH_ENV_KNOB("JH_NODE_ABL") >= 0) a.abl = JH_ENV_KNOB("JH_NODE_ABL");   // timing experiments only
  const size_t head = ((size_t)3 * a.Cp * 2 + (size_t)9 * a.Cp) * sizeof(float);
  const size_t at_bytes = (size_t)128 * (a.Cp + 4) * sizeof(float);
  const size_t red = (size_t)8 * kNodeNRG * 16 * 2 * sizeof(double);   // conv_epilogue: fp64 partials
  const size_t halo_px = (size_t)kNodePY * kNodePX * sizeof(float);
  // preferred: the whole channel range in one halo chunk, operand tile aliased onto it
  // (3 workgroups per CU for the 56-channel pyramid of the small model)
  const size_t b_bytes = (size_t)(a.Cp / 8) * (a.cout_p16 / 16) * 128 * sizeof(float);
  a.blds = (b_bytes <= 16 * 1024) ? 1 : 0;        // 512 threads x 32 bytes
  if (JH_ENV_KNOB("JH_NODE_NOBLDS") > 0) a.blds = 0;
  size_t lds = head + std::max(halo_px * a.Cp, at_bytes + (a.blds ? b_bytes : red));
  a.cf = a.Cp;
  a.alias = 1;
  const bool na = JH_ENV_KNOB("JH_NODE_NOALIAS") > 0;
  if (a.Cp > 64 || a.Cp < 32 || lds > 54 * 1024 || na) {
    // fallback: separate operand tile, halo chunked to the LDS budget
    a.alias = 0;
    a.blds = 0;
    size_t budget = 78 * 1024;
    if (JH_ENV_KNOB("JH_NODE_LDS_KB") > 0) budget = (size_t)JH_ENV_KNOB("JH_NODE_LDS_KB") * 1024;
    const size_t fixed = head + at_bytes;
    if (fixed + halo_px * 12 > budget) budget = 156 * 1024;
    int cf = std::min(a.Cp, 64);          // the depthwise rounds handle <= 16 channel quads
    while (cf > 8 && fixed + halo_px * cf > budget) cf -= 4;
    a.cf = cf;
    lds = fixed + halo_px * cf;
    JH_REQUIRE(halo_px * cf >= red, "halo chunk too small");
  }
  JH_REQUIRE(lds <= 160 * 1024, "BiFPN node does not fit LDS");
  const int m0 = a.mode[0], m1 = a.mode[1], m2 = a.mode[2];
  JH_REQUIRE(m0 == FUSE_SAME, "first input of a node is at the node's own level");
  if (a.n_in == 2 && m1 == FUSE_UP2) return launch_node_variant<2, FUSE_SAME, FUSE_UP2, 0>(a, lds, s);
  if (a.n_in == 2 && m1 == FUSE_POOL2) return launch_node_variant<2, FUSE_SAME, FUSE_POOL2, 0>(a, lds, s);
  if (a.n_in == 3 && m1 == FUSE_SAME && m2 == FUSE_POOL2)
    return launch_node_variant<3, FUSE_SAME, FUSE_SAME, FUSE_POOL2>(a, lds, s);
  if (a.n_in == 3 && m1 == FUSE_UP2 && m2 == FUSE_UP4)
    return launch_node_variant<3, FUSE_SAME, FUSE_UP2, FUSE_UP4>(a, lds, s);
  // (bottom-up node whose finer-level input arrives pooled -- written by that level's row node -- but which is
  //  itself too small for the row form)
  if (a.n_in == 3 && m1 == FUSE_SAME && m2 == FUSE_SAME)
    return launch_node_variant<3, FUSE_SAME, FUSE_SAME, FUSE_SAME>(a, lds, s);
  // (... and its two-input sibling at the coarsest level: P7 from P7_in and the pooled P6 written by P6's row node)
  if (a.n_in == 2 && m1 == FUSE_SAME) return launch_node_variant<2, FUSE_SAME, FUSE_SAME, 0>(a, lds, s);
  JH_REQUIRE(false, "unsupported BiFPN node variant");
}

}  // namespace jh
