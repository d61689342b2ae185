// fp32 MFMA implicit-GEMM convolution for gfx950 (2D and 3D, channel-last).
//
// Replaces (all on the inference hot path of the reference):
//   nn.Conv2d dense k x k / 1x1      jarvis/efficienttrack/efficientnet.py:47-88,
//                                     jarvis/efficienttrack/model.py:203-215,404-425
//   nn.ConvTranspose2d k4 s2 p1      jarvis/efficienttrack/model.py:90-96
//   nn.Conv3d 3^3 s1/s2, 2^3 s2, 1^3 jarvis/hybridnet/v2vnet.py:15-16,30-37,94-95
//   nn.ConvTranspose3d k2 s2         jarvis/hybridnet/v2vnet.py:52-53
//
// Mapping: GEMM M = output pixels of a (TZ,TY,TX) tile, N = output channels in
// blocks of 16, K = taps x input channels.  The input halo patch of the tile
// is staged once per channel chunk in LDS ([pixel][kc] with a stride chosen so
// that the 16 pixel rows of an MFMA operand fall on distinct banks for
// ds_read_b64); weights are pre-packed on the host into the exact per-lane
// MFMA B-operand order and streamed from L2 with one coalesced 512-byte read
// per (tap, 8 channels, 16 couts).  v_mfma_f32_16x16x4_f32 keeps exact fp32
// products with fp32 accumulation (needed for the 1e-3 mm parity bar).
//
// Epilogue: + bias, store raw output, and accumulate the per-(n, channel)
// sum / sum of squares that the following InstanceNorm needs (fp64 atomics),
// so no extra pass over the activation is required for the statistics.
#pragma once
#include "jh_common.h"

namespace jh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int ND, int K, int STRIDE, int TZ, int TY, int TX>
struct ConvGeom {
  static constexpr int KD = (ND == 3) ? K : 1;
  static constexpr int KH = K, KW = K;
  static constexpr int NT = KD * KH * KW;
  static constexpr int PZ = (TZ - 1) * STRIDE + KD;
  static constexpr int PY = (TY - 1) * STRIDE + KH;
  static constexpr int PX = (TX - 1) * STRIDE + KW;
  static constexpr int NPIX = PZ * PY * PX;
  static constexpr int TM = TZ * TY * TX;
  static constexpr int MR = TM / 64;          // 16-pixel blocks per wave (4 waves)
  static constexpr int SPAD = (STRIDE == 1) ? 4 : 2;
  static_assert(TM % 64 == 0, "tile must be a multiple of 64 pixels");
  static size_t lds_bytes(int kc) { return (size_t)NPIX * (kc + SPAD) * sizeof(float); }
};

template <int ND, int K, int STRIDE, int TZ, int TY, int TX, int NR>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  using G = ConvGeom<ND, K, STRIDE, TZ, TY, TX>;
  constexpr int MR = G::MR;
  extern __shared__ __attribute__((aligned(16))) float lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 15;
  const int kq = lane >> 4;

  // ---- block coordinates
  const int tiles_x = (a.Wout + TX - 1) / TX;
  const int tiles_y = (a.Hout + TY - 1) / TY;
  int t = blockIdx.x;
  const int tile_x = t % tiles_x; t /= tiles_x;
  const int tile_y = t % tiles_y; t /= tiles_y;
  const int tile_z = t;
  const int nb0 = blockIdx.y * NR;
  const int n = blockIdx.z / a.nphase;
  const int ph = blockIdx.z % a.nphase;
  const int oz0 = tile_z * TZ, oy0 = tile_y * TY, ox0 = tile_x * TX;
  const int iz0 = oz0 * STRIDE - a.phase[ph].pad[0];
  const int iy0 = oy0 * STRIDE - a.phase[ph].pad[1];
  const int ix0 = ox0 * STRIDE - a.phase[ph].pad[2];

  const int S = a.kc + G::SPAD;   // LDS pixel stride in floats

  // ---- per-lane LDS base of each of this wave's 16-pixel row blocks
  int abase[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int p = (wave * MR + mr) * 16 + mrow;
    const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
    abase[mr] = (((tz * STRIDE) * G::PY + ty * STRIDE) * G::PX + tx * STRIDE) * S + 2 * kq;
  }

  f32x4 acc[MR][NR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const float* __restrict__ xin = a.x + (size_t)n * a.Din * a.Hin * a.Win * a.cin_p;
  const int nkc8_total = a.cin_p >> 3;
  const float* __restrict__ wph = a.w + (size_t)ph * a.phase_stride;
  const int nb16_total = a.cout_p16 >> 4;

  for (int c0 = 0; c0 < a.cin_p; c0 += a.kc) {
    const int kcur = min(a.kc, a.cin_p - c0);
    const int q4 = kcur >> 2;                 // float4 per pixel in this chunk
    __syncthreads();
    // ---- stage the halo patch: [pixel][kcur] with stride S
    const int total = G::NPIX * q4;
    for (int idx = tid; idx < total; idx += 256) {
      const int c4 = idx % q4;
      const int pix = idx / q4;
      const int px = pix % G::PX;
      const int py = (pix / G::PX) % G::PY;
      const int pz = pix / (G::PX * G::PY);
      const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iz >= 0 && iz < a.Din && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
        v = *reinterpret_cast<const float4*>(
            xin + ((size_t)(iz * a.Hin + iy) * a.Win + ix) * a.cin_p + c0 + c4 * 4);
        if (a.gate) {
          const float4 g = *reinterpret_cast<const float4*>(
              a.gate + (size_t)n * a.cin_p + c0 + c4 * 4);
          v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
        }
      }
      float2* dst = reinterpret_cast<float2*>(lds + pix * S + c4 * 4);
      dst[0] = make_float2(v.x, v.y);
      dst[1] = make_float2(v.z, v.w);
    }
    __syncthreads();

    // ---- taps x 8-channel steps, operands prefetched one step ahead
    const int nk8 = kcur >> 3;
    const int steps = G::NT * nk8;
    const int kc8_0 = c0 >> 3;
    int tap = 0, k8 = 0, dz = 0, dy = 0, dx = 0;
    float2 an[MR], bn[NR];
    {
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
        an[mr] = *reinterpret_cast<const float2*>(lds + abase[mr]);
      const float* wp = wph + ((size_t)(0 * nkc8_total + kc8_0) * nb16_total + nb0) * 128 + lane * 2;
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
        bn[nr] = (nb0 + nr < nb16_total) ? *reinterpret_cast<const float2*>(wp + nr * 128)
                                         : make_float2(0.f, 0.f);
    }
    for (int it = 0; it < steps; ++it) {
      float2 ac[MR], bc[NR];
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) ac[mr] = an[mr];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) bc[nr] = bn[nr];
      // advance (tap, k8) and prefetch
      ++k8;
      if (k8 == nk8) {
        k8 = 0; ++tap; ++dx;
        if (dx == G::KW) { dx = 0; ++dy; if (dy == G::KH) { dy = 0; ++dz; } }
      }
      if (it + 1 < steps) {
        const int toff = ((dz * G::PY + dy) * G::PX + dx) * S + k8 * 8;
#pragma unroll
        for (int mr = 0; mr < MR; ++mr)
          an[mr] = *reinterpret_cast<const float2*>(lds + abase[mr] + toff);
        const float* wp = wph + ((size_t)(tap * nkc8_total + kc8_0 + k8) * nb16_total + nb0) * 128 + lane * 2;
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
          bn[nr] = (nb0 + nr < nb16_total) ? *reinterpret_cast<const float2*>(wp + nr * 128)
                                           : make_float2(0.f, 0.f);
      }
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
          acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[mr].x, bc[nr].x, acc[mr][nr], 0, 0, 0);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
          acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[mr].y, bc[nr].y, acc[mr][nr], 0, 0, 0);
    }
  }

  // ---- epilogue: bias, store, InstanceNorm statistics
  __syncthreads();                       // LDS is reused for the cross-wave reduce
  float* red = lds;                      // [4 waves][NR*16][2]
  const int os = a.ostride;
  const int offz = a.phase[ph].ooff[0], offy = a.phase[ph].ooff[1], offx = a.phase[ph].ooff[2];
  float* __restrict__ yout = a.y + (size_t)n * a.Dy * a.Hy * a.Wy * a.cout_p;
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int ch = (nb0 + nr) * 16 + mrow;
    const bool ch_ok = ch < a.cout_p;
    const float bv = (a.bias && ch < a.cout_p16) ? a.bias[ch] : 0.f;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = (wave * MR + mr) * 16 + kq * 4 + r;
        const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
        const int oz = oz0 + tz, oy = oy0 + ty, ox = ox0 + tx;
        const float v = acc[mr][nr][r] + bv;
        if (ch_ok && oz < a.Dout && oy < a.Hout && ox < a.Wout) {
          yout[((size_t)((oz * os + offz) * a.Hy + (oy * os + offy)) * a.Wy + (ox * os + offx)) * a.cout_p + ch] = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
    if (a.stats) {
      s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
      s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
      if (kq == 0) {
        red[(wave * NR * 16 + nr * 16 + mrow) * 2 + 0] = s1;
        red[(wave * NR * 16 + nr * 16 + mrow) * 2 + 1] = s2;
      }
    }
  }
  if (a.stats) {
    __syncthreads();
    if (tid < NR * 16) {
      const int ch = nb0 * 16 + tid;
      if (ch < a.cout_p) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          s1 += red[(w * NR * 16 + tid) * 2 + 0];
          s2 += red[(w * NR * 16 + tid) * 2 + 1];
        }
        double* st = a.stats + ((size_t)n * a.cout_p + ch) * 2;
        unsafeAtomicAdd(st + 0, (double)s1);
        unsafeAtomicAdd(st + 1, (double)s2);
      }
    }
  }
}

// Launch one instantiation.  Returns false if this (nd,k,stride) is not the
// instantiation's, so callers can chain.
template <int ND, int K, int STRIDE, int TZ, int TY, int TX>
int launch_conv_geom(const ConvArgs& a, int nr, size_t lds_budget, hipStream_t s) {
  using G = ConvGeom<ND, K, STRIDE, TZ, TY, TX>;
  ConvArgs b = a;
  // largest channel chunk (multiple of 8) whose patch fits the LDS budget
  int kc = b.cin_p;
  while (kc > 8 && G::lds_bytes(kc) > lds_budget) kc -= 8;
  const int nchunk = (b.cin_p + kc - 1) / kc;         // balance the chunks
  kc = round_up((b.cin_p + nchunk - 1) / nchunk, 8);
  b.kc = kc;
  size_t lds = G::lds_bytes(kc);
  const size_t red = (size_t)4 * 4 * 16 * 2 * sizeof(float);
  if (lds < red) lds = red;
  JH_REQUIRE(lds <= 160 * 1024, "conv patch does not fit LDS");
  const int tiles = ((b.Dout + TZ - 1) / TZ) * ((b.Hout + TY - 1) / TY) * ((b.Wout + TX - 1) / TX);
  const int nb = b.cout_p16 / 16;
  dim3 grid(tiles, (nb + nr - 1) / nr, b.N * b.nphase);
  dim3 block(256);
#define JH_CONV_LAUNCH(NRV)                                                                   \
  case NRV: {                                                                                 \
    auto kern = conv_mfma_kernel<ND, K, STRIDE, TZ, TY, TX, NRV>;                            \
    static bool big_lds_enabled = false;                                                      \
    if (lds > 64 * 1024 && !big_lds_enabled) {                                                \
      JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                   \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      big_lds_enabled = true;                                                                 \
    }                                                                                         \
    hipLaunchKernelGGL(kern, grid, block, lds, s, b);                                         \
  } break;
  switch (nr) {
    JH_CONV_LAUNCH(1)
    JH_CONV_LAUNCH(2)
    JH_CONV_LAUNCH(3)
    JH_CONV_LAUNCH(4)
    default:
      JH_REQUIRE(false, "bad NR");
  }
#undef JH_CONV_LAUNCH
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// per-translation-unit entry points (one .hip file per kernel family so the
// instantiations compile in parallel)
int conv_launch_2d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k3(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k5(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_3d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_3d_k2s2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_3d_k3(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s);

}  // namespace jh
