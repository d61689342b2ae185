// ConvTranspose2d(k=4, s=2, p=1, bias=False) with ONE output channel: the CenterDetect head
// (jarvis/efficienttrack/model.py:104-105,127 with num_joints = 1).
//
// On the MFMA path a single output channel is padded to a 16-wide column block, so 15/16 of
// the matrix work is thrown away (0.64 ms per step of 32 frames).  With one channel the op
// is a 4-tap x cin dot product per output pixel, i.e. HBM-bound (cin floats read per input
// pixel, 4 outputs written), so it runs on the vector ALUs here:
//
//   * 256 threads own an 8 x 16 tile of INPUT pixels; the 10 x 18 halo tile is staged once in
//     LDS with the producer's InstanceNorm(+act) applied on load (as conv_mfma.h does);
//   * wave pairs split by output-row parity (wave-uniform, so the weights are scalar loads);
//     a thread produces the two horizontally adjacent outputs (2m+py, 2n), (2m+py, 2n+1) from
//     2 rows x 3 columns of the tile;
//   * output in the library's channel-last layout [N][2H][2W][cout_p] (pad channels 0), one
//     64-byte store per thread, consecutive lanes contiguous.
//
// out(oy, ox) = sum_ci sum_ky,kx in(iy, ix, ci) * w(ci, 0, ky, kx),  oy = 2 iy - 1 + ky.
#include "jh_common.h"

namespace jh {

constexpr int kDcTY = 8, kDcTX = 16, kDcPY = 10, kDcPX = 18;

__global__ __launch_bounds__(256) void deconv_c1_kernel(
    const float* __restrict__ x, const double* __restrict__ st, float inv_cnt, int in_act,
    const float* __restrict__ w /* [16 taps][Cp] */, float* __restrict__ y, int H, int W, int Cp,
    int cout_p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int S = Cp + 4;                          // pixel stride: 16 lanes x b128 conflict-free
  float* nrm = sm;                               // [Cp] mean, [Cp] rstd
  float* tile = nrm + 2 * Cp;                    // [180][S]
  const int tid = threadIdx.x;
  const BlockId bid = xcd_block();
  const int tiles_x = (W + kDcTX - 1) / kDcTX;
  const int y0 = (bid.x / tiles_x) * kDcTY, x0 = (bid.x % tiles_x) * kDcTX;
  const int n = bid.y;
  const int q = Cp >> 2;

  for (int c = tid; c < Cp; c += 256) {
    float mean = 0.f, rstd = 1.f;
    if (st) {
      const double* s2 = st + ((size_t)n * Cp + c) * kStatW;
      const double mu = exact_read(s2) * (double)inv_cnt;
      double var = exact_read(s2 + kLimbs) * (double)inv_cnt - mu * mu;
      if (var < 0.0) var = 0.0;
      mean = (float)mu;
      rstd = (float)(1.0 / sqrt(var + 1e-5));
    }
    nrm[c] = mean;
    nrm[Cp + c] = rstd;
  }
  __syncthreads();
  const float* xin = x + (size_t)n * H * W * Cp;
  // halo tile: thread -> (channel quad tid % 16, pixel slot tid / 16), compile-time trip count,
  // the loads of a batch issued before the first LDS store
  {
    constexpr int NPX = kDcPY * kDcPX, ITERS = (NPX + 15) / 16, UB = ITERS;    // one round trip
    const int slot = tid >> 4;
   for (int cq0 = 0; cq0 < q; cq0 += 16) {       // 16 channel quads at a time (one for cin <= 64)
    const int lc4 = cq0 + (tid & 15);
    const bool cact = lc4 < q;
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
    if (cact) {
      mu = *reinterpret_cast<const float4*>(nrm + lc4 * 4);
      rs = *reinterpret_cast<const float4*>(nrm + Cp + lc4 * 4);
    }
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += UB) {
      float4 v[UB];
      bool ok[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int pix = (it0 + u) * 16 + slot;
        const int iy = y0 - 1 + pix / kDcPX, ix = x0 - 1 + pix % kDcPX;
        ok[u] = it0 + u < ITERS && pix < NPX && cact && iy >= 0 && iy < H && ix >= 0 && ix < W;
        v[u] = ok[u] ? *reinterpret_cast<const float4*>(xin + ((size_t)iy * W + ix) * Cp + lc4 * 4)
                     : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int pix = (it0 + u) * 16 + slot;
        if (it0 + u < ITERS && pix < NPX && cact) {
          float4 x4 = v[u];
          if (ok[u]) {
            x4.x = (x4.x - mu.x) * rs.x; x4.y = (x4.y - mu.y) * rs.y;
            x4.z = (x4.z - mu.z) * rs.z; x4.w = (x4.w - mu.w) * rs.w;
            if (in_act == ACT_RELU) {
              x4.x = fmaxf(x4.x, 0.f); x4.y = fmaxf(x4.y, 0.f); x4.z = fmaxf(x4.z, 0.f); x4.w = fmaxf(x4.w, 0.f);
            } else if (in_act == ACT_SILU) {
              x4.x = silu_fast(x4.x); x4.y = silu_fast(x4.y);
              x4.z = silu_fast(x4.z); x4.w = silu_fast(x4.w);
            }
          }
          *reinterpret_cast<float4*>(tile + pix * S + lc4 * 4) = x4;
        }
      }
    }
   }
  }
  __syncthreads();

  // thread -> (input pixel m, n of the tile; output-row parity py), py uniform per wave
  const int py = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int p = tid & 127;
  const int tm = p / kDcTX, tn = p % kDcTX;
  // py = 0: rows (m, ky = 1), (m - 1, ky = 3);  py = 1: rows (m + 1, ky = 0), (m, ky = 2)
  const int ra = py ? tm + 2 : tm + 1, ka = py ? 0 : 1;      // tile row (halo offset 1), ky
  const int rb = py ? tm + 1 : tm, kb = py ? 2 : 3;
  const float* ta = tile + (ra * kDcPX + tn) * S;            // columns n-1, n, n+1 at +0, +S, +2S
  const float* tb = tile + (rb * kDcPX + tn) * S;
  // The weights are wave-uniform (py is): they are read through the scalar cache into SGPRs
  // instead of as LDS broadcasts -- the kernel was LDS-bandwidth-bound with 8 weight + 6 data
  // quads per 4 channels.
  const float* __restrict__ wa = w + ka * 4 * Cp;            // [kx][Cp]
  const float* __restrict__ wb = w + kb * 4 * Cp;
  float o0 = 0.f, o1 = 0.f;
  for (int c = 0; c < Cp; c += 4) {
    const float4 a0 = *reinterpret_cast<const float4*>(ta + c);
    const float4 a1 = *reinterpret_cast<const float4*>(ta + S + c);
    const float4 a2 = *reinterpret_cast<const float4*>(ta + 2 * S + c);
    const float4 b0 = *reinterpret_cast<const float4*>(tb + c);
    const float4 b1 = *reinterpret_cast<const float4*>(tb + S + c);
    const float4 b2 = *reinterpret_cast<const float4*>(tb + 2 * S + c);
    // px = 0: columns (n, kx = 1), (n - 1, kx = 3);  px = 1: (n + 1, kx = 0), (n, kx = 2)
    const float4 wa0 = *reinterpret_cast<const float4*>(wa + 0 * Cp + c);
    const float4 wa1 = *reinterpret_cast<const float4*>(wa + 1 * Cp + c);
    const float4 wa2 = *reinterpret_cast<const float4*>(wa + 2 * Cp + c);
    const float4 wa3 = *reinterpret_cast<const float4*>(wa + 3 * Cp + c);
    const float4 wb0 = *reinterpret_cast<const float4*>(wb + 0 * Cp + c);
    const float4 wb1 = *reinterpret_cast<const float4*>(wb + 1 * Cp + c);
    const float4 wb2 = *reinterpret_cast<const float4*>(wb + 2 * Cp + c);
    const float4 wb3 = *reinterpret_cast<const float4*>(wb + 3 * Cp + c);
#define JH_DOT4(acc, u, v) acc = fmaf(u.x, v.x, acc); acc = fmaf(u.y, v.y, acc); \
                           acc = fmaf(u.z, v.z, acc); acc = fmaf(u.w, v.w, acc);
    JH_DOT4(o0, a1, wa1) JH_DOT4(o0, a0, wa3) JH_DOT4(o0, b1, wb1) JH_DOT4(o0, b0, wb3)
    JH_DOT4(o1, a2, wa0) JH_DOT4(o1, a1, wa2) JH_DOT4(o1, b2, wb0) JH_DOT4(o1, b1, wb2)
#undef JH_DOT4
  }
  const int m = y0 + tm, nn = x0 + tn;
  if (m < H && nn < W) {
    float* dst = y + (((size_t)n * 2 * H + 2 * m + py) * 2 * W + 2 * nn) * cout_p;
    for (int o = 0; o < 2; ++o) {
      float* d = dst + o * cout_p;
      *reinterpret_cast<float4*>(d) = make_float4(o ? o1 : o0, 0.f, 0.f, 0.f);
      for (int c = 4; c < cout_p; c += 4) *reinterpret_cast<float4*>(d + c) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// x: [N][H][W][Cp] (raw + optional statistics), w: [16][Cp] tap-major, y: [N][2H][2W][cout_p]
int launch_deconv_c1(const Act& x, const double* stats, float inv_cnt, int in_act, const float* w,
                     const Act& y, hipStream_t s) {
  JH_REQUIRE(y.H == 2 * x.H && y.W == 2 * x.W && y.N == x.N && y.C == 1, "deconv_c1 shapes");
  const size_t lds = ((size_t)2 * x.Cp + (size_t)kDcPY * kDcPX * (x.Cp + 4)) * sizeof(float);
  JH_REQUIRE(lds <= 160 * 1024, "deconv_c1 tile does not fit LDS");
  static bool big = false;
  if (lds > 64 * 1024 && !big) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(deconv_c1_kernel),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  const int tiles = ((x.H + kDcTY - 1) / kDcTY) * ((x.W + kDcTX - 1) / kDcTX);
  hipLaunchKernelGGL(deconv_c1_kernel, dim3(tiles, x.N), dim3(256), lds, s, x.p, stats, inv_cnt,
                     in_act, w, y.p, x.H, x.W, x.Cp, y.Cp);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace jh
