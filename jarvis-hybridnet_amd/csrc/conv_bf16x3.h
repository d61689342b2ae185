// Generic split-bf16 implicit-GEMM convolution (precision mode bf16x3; fp32 form: conv_mfma.h).
//
// Covers the dense k x k convolutions that carry most of the matrix work outside the V2V residual
// blocks: the stride-2 front convolution of V2V (jarvis/hybridnet/v2vnet.py:15-16,89-90) and the
// "fused" dense MBConv convolutions of the EfficientNet trunk (jarvis/efficienttrack/efficientnet.py:
// 57-61,93-94: k3 s1, k3 s2, k5 s2).  Arithmetic as in conv3d_bf16x3.hip: operands split into hi + lo
// bf16, three v_mfma_f32_16x16x32_bf16 per product, fp32 accumulation, InstanceNorm (+ activation) of
// the producer applied while the patch is staged, fused statistics in the epilogue.
//
// Mapping: output tile TZ x TY x 16 (a row block of an MFMA = 16 consecutive output x), 4 waves x MR
// row blocks x NCB blocks of 16 output channels.  K = (chunk of 8 CH8 input channels) x (slices of
// 4 / CH8 taps x 8 CH8 channels = 32).  The halo patch of a chunk lives in LDS as two planes (hi, lo)
// of [pixel][8 CH8 channels] bf16; for stride 2 the x axis is stored de-interleaved (even pixels, then
// odd pixels), so the 16 rows of an A operand -- input x = 2 ox + dx -- are 16 CONSECUTIVE slots and
// one ds_read_b128 per lane fetches its 8 channels without bank conflicts, exactly as for stride 1.
// The tap a K group reads is a per-lane offset per slice (toff[]); row block, plane and chunk offsets
// are immediates.  Weights: [chunk][slice][column block][hi, lo][lane][8], packed on the host.
#pragma once
#include <cstring>
#include "conv_mfma.h"

namespace jh {

typedef __bf16 xbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 xbf16x4 __attribute__((ext_vector_type(4)));

struct XArgs {
  const float* x;          // [N][D][H][W][cin_p]
  float* y;                // [N][Do][Ho][Wo][cout_p] raw output
  const uint4* w;
  const float* bias;
  const double* in_stats;
  float in_inv;
  int in_act;
  double* stats;
  int N, Din, Hin, Win, cin_p, Dout, Hout, Wout, cout_p, cout_p16, nchunk, ncbt, pad;
  int tiles_x, tiles_y, tiles_z;
};

template <int ND, int K, int S, int TZ, int TY, int CH8>
struct XGeom {
  static constexpr int KD = ND == 3 ? K : 1;
  static constexpr int NT = KD * K * K;
  static constexpr int TPS = 4 / CH8;                       // taps per 32-wide K slice
  static constexpr int NSL = (NT + TPS - 1) / TPS;
  static constexpr int PZ = (TZ - 1) * S + KD, PY = (TY - 1) * S + K, PX = 15 * S + K;
  static constexpr int PXH = (PX + 1) / 2;
  static constexpr int PXS = S == 1 ? PX : 2 * PXH;         // slots per patch row
  static constexpr int NSLOT = PZ * PY * PXS;
  static constexpr int PB = CH8 * 16;                       // bytes per pixel in one plane
  static constexpr int PLANE = NSLOT * PB;
  static constexpr int MR = TZ * TY / 4;
  static_assert(TZ * TY % 4 == 0 && (TY % MR == 0 || MR % TY == 0), "row blocks of a wave are a contiguous run");
  static constexpr int slot(int px) { return S == 1 ? px : (px & 1) * PXH + (px >> 1); }
  // slot offset of tap t relative to the lane's (tz S, ty S, x) origin
  static constexpr int tap_off(int t) {
    const int dx = t % K, dy = (t / K) % K, dz = t / (K * K);
    return ((dz * PY + dy) * PXS + (S == 1 ? dx : (dx & 1) * PXH + (dx >> 1)));
  }
  static constexpr int rb_off(int mr) {                     // slots, relative to the wave's first row block
    return ((mr / TY) * S * PY + (mr % TY) * S) * PXS;
  }
  static size_t lds_bytes(int cin_p, int ncb) {
    return (size_t)2 * PLANE + (size_t)2 * cin_p * sizeof(float) + (size_t)4 * ncb * 16 * 2 * sizeof(double);
  }
};

template <int ND, int K, int S, int TZ, int TY, int CH8, int NCB>
__global__ __launch_bounds__(256) void conv_bf16x3_kernel(XArgs a) {
  using G = XGeom<ND, K, S, TZ, TY, CH8>;
  constexpr int MR = G::MR, CHK = CH8 * 8, CQ = CH8 * 2;    // channels / channel quads per chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* nrm = reinterpret_cast<float*>(smem + 2 * G::PLANE);
  float* red = nrm + 2 * a.cin_p;
  const BlockId bid = xcd_block();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = bid.y, nb0 = bid.z * NCB;
  const int tile = (int)bid.x;
  const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, tz = tile / (a.tiles_x * a.tiles_y);
  const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * 16;
  const int iz0 = oz0 * S - (ND == 3 ? a.pad : 0), iy0 = oy0 * S - a.pad, ix0 = ox0 * S - a.pad;

  if (a.in_stats) {
    for (int c = tid; c < a.cin_p; c += 256) {
      const double* st = a.in_stats + ((size_t)n * a.cin_p + c) * kStatW;
      const double mu = exact_read(st) * (double)a.in_inv;
      double var = exact_read(st + kLimbs) * (double)a.in_inv - mu * mu;
      if (var < 0.0) var = 0.0;
      const float rsf = (float)(1.0 / sqrt(var + 1e-5));
      nrm[c] = -(float)mu * rsf;
      nrm[a.cin_p + c] = rsf;
    }
  }
  const float* __restrict__ xin = a.x + (size_t)n * a.Din * a.Hin * a.Win * a.cin_p;

  // this lane's A rows: x = lane & 15; K group g reads tap (g / CH8) of the slice, channel octet g % CH8
  const int g = lane >> 4, tg = g / CH8, oct = g % CH8;
  const int rbg0 = wave * MR;                                 // first row block of this wave
  const int abase = ((((rbg0 / TY) * S * G::PY + (rbg0 % TY) * S) * G::PXS) + (lane & 15)) * G::PB + oct * 16;
  int toff[G::NSL];
#pragma unroll
  for (int s = 0; s < G::NSL; ++s) {
    int v = 0;
#pragma unroll
    for (int j = 0; j < G::TPS; ++j) {
      const int t = s * G::TPS + j;
      const int o = t < G::NT ? G::tap_off(t) * G::PB : G::tap_off(0) * G::PB;   // phantom taps: finite data, zero weights
      v = (tg == j) ? o : v;
    }
    toff[s] = abase + v;
  }

  f32x4 acc[MR][NCB];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[mr][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(a.w), 0, a.nchunk * G::NSL * a.ncbt * 2 * 1024, 0x00020000);

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();
    // ---- stage the patch of this chunk: norm (+ activation) on load, split, two planes ---------------
    constexpr int NREAL = G::PZ * G::PY * G::PX;
    for (int idx = tid; idx < NREAL * CQ; idx += 256) {
      const int pix = idx / CQ, q = idx - pix * CQ;
      const int px = pix % G::PX, py = (pix / G::PX) % G::PY, pz = pix / (G::PX * G::PY);
      const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
      const int c0 = chunk * CHK + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iz >= 0 && iz < a.Din && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win && c0 < a.cin_p) {
        v = *reinterpret_cast<const float4*>(xin + ((size_t)(iz * a.Hin + iy) * a.Win + ix) * a.cin_p + c0);
        if (a.in_stats) {
          const float4 m = *reinterpret_cast<const float4*>(nrm + c0);
          const float4 r = *reinterpret_cast<const float4*>(nrm + a.cin_p + c0);
          v.x = __fmaf_rn(v.x, r.x, m.x); v.y = __fmaf_rn(v.y, r.y, m.y);
          v.z = __fmaf_rn(v.z, r.z, m.z); v.w = __fmaf_rn(v.w, r.w, m.w);
          if (a.in_act == ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          } else if (a.in_act == ACT_SILU) {
            v.x = silu_fast(v.x); v.y = silu_fast(v.y); v.z = silu_fast(v.z); v.w = silu_fast(v.w);
          }
        }
      }
      const xbf16x4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      const xbf16x4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]),
                          (__bf16)(v.z - (float)hi[2]), (__bf16)(v.w - (float)hi[3])};
      unsigned char* dst = smem + ((pz * G::PY + py) * G::PXS + G::slot(px)) * G::PB + q * 8;
      *reinterpret_cast<xbf16x4*>(dst) = hi;
      *reinterpret_cast<xbf16x4*>(dst + G::PLANE) = lo;
    }
    __syncthreads();
    // ---- slices: operands one step ahead ---------------------------------------------------------------
    const int wchunk = chunk * G::NSL * a.ncbt * 2 * 1024;
    xbf16x8 bh[2][NCB], bl[2][NCB], ah[2], al[2];
    auto load_b = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int so = wchunk + ((s * a.ncbt + nb0 + cb) * 2) * 1024;
        bh[buf][cb] = __builtin_bit_cast(xbf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so, 0));
        bl[buf][cb] = __builtin_bit_cast(xbf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so + 1024, 0));
      }
    };
    auto load_a = [&](int i, int buf) __attribute__((always_inline)) {
      const int s = i / MR, mr = i % MR;
      const unsigned char* p = smem + toff[s] + G::rb_off(mr) * G::PB;
      ah[buf] = *reinterpret_cast<const xbf16x8*>(p);
      al[buf] = *reinterpret_cast<const xbf16x8*>(p + G::PLANE);
    };
    load_b(0, 0);
    load_a(0, 0);
#pragma unroll
    for (int s = 0; s < G::NSL; ++s) {
      const int cur = s & 1;
      if (s + 1 < G::NSL) load_b(s + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        const int i = s * MR + mr;
        if (i + 1 < G::NSL * MR) load_a(i + 1, (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[mr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[cur][cb], acc[mr][cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[mr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[cur][cb], acc[mr][cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[mr][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[cur][cb], acc[mr][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  __syncthreads();
  EpilogueArgs e;
  e.y = a.y + (size_t)n * a.Dout * a.Hout * a.Wout * a.cout_p;
  e.bias = a.bias;
  e.stats = a.stats ? a.stats + (size_t)n * a.cout_p * kStatW : nullptr;
  e.Dout = a.Dout; e.Hout = a.Hout; e.Wout = a.Wout; e.Hy = a.Hout; e.Wy = a.Wout;
  e.cout_p = a.cout_p; e.cout_p16 = a.cout_p16; e.os = 1; e.offz = e.offy = e.offx = 0; e.osz = 1;
  conv_epilogue<MR, NCB, TY, 16, 4, false>(acc, e, red, nb0, oz0, oy0, ox0, tid);
}

template <int ND, int K, int S, int TZ, int TY, int CH8, int NCB>
int launch_xconv(const XArgs& a, int groups, hipStream_t s) {
  using G = XGeom<ND, K, S, TZ, TY, CH8>;
  auto kern = conv_bf16x3_kernel<ND, K, S, TZ, TY, CH8, NCB>;
  const size_t lds = G::lds_bytes(a.cin_p, NCB);
  static bool big = false;
  if (!big && lds > 64 * 1024) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  dim3 grid(a.tiles_x * a.tiles_y * a.tiles_z, a.N, groups);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// dispatch over the column-block count (explicitly instantiated in conv_bf16x3_*.hip)
template <int ND, int K, int S, int TZ, int TY, int CH8>
int launch_xconv_ncb(const XArgs& a, hipStream_t s);

#define JH_XCONV_DEFINE(ND, K, S, TZ, TY, CH8)                                                      \
  template <> int launch_xconv_ncb<ND, K, S, TZ, TY, CH8>(const XArgs& a, hipStream_t s) {          \
    const int t = a.ncbt;                                                                            \
    if (t % 4 == 0) return launch_xconv<ND, K, S, TZ, TY, CH8, 4>(a, t / 4, s);                      \
    if (t % 3 == 0) return launch_xconv<ND, K, S, TZ, TY, CH8, 3>(a, t / 3, s);                      \
    if (t % 2 == 0) return launch_xconv<ND, K, S, TZ, TY, CH8, 2>(a, t / 2, s);                      \
    return launch_xconv<ND, K, S, TZ, TY, CH8, 1>(a, t, s);                                          \
  }

}  // namespace jh
