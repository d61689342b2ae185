// ConvTranspose2d(k = 4, s = 2, p = 1) with all four output parities in ONE workgroup:
// nn.ConvTranspose2d of the keypoint head, jarvis/efficienttrack/model.py:90-96.
//
// Output (2y + py, 2x + px) is a 2 x 2-tap convolution of the input: even parities read inputs
// {y - 1, y}, odd ones {y, y + 1} (conv_host.hip: deconv2d_k4s2p1_desc).  The general kernel
// (conv_mfma.h) runs the four parities as four workgroups, each of which stages (nearly) the
// same input patch.  The layer is bound by the non-matrix instructions of staging, so here ONE
// workgroup stages the 10 x 18 halo patch of an 8 x 16 tile of INPUT pixels once and produces
// the 16 x 32 output pixels of all four parities from it:
//
//   for each of the 3 x 3 window positions (r, s):       A rows from LDS, once
//     for each parity (py, px) with r - py, s - px in {0, 1}:    (1, 2 or 4 of them: 16 in all)
//       MFMAs with that parity's packed weights of tap (r - py, s - px)
//
// i.e. a quarter of the staging work and 9 instead of 16 A-operand reads per channel step for
// the same 16 (parity, tap) MFMA groups.  Per parity the taps are visited in the order of the
// general kernel and the per-workgroup statistics cover the same pixels, so outputs and fused
// statistics are bit-identical to it (JH_DECONV_FUSED=0 selects the general kernel).
//
// Operand layout ("paired"): in an fp32 MFMA stream every LDS read costs ~24 cycles and every
// global load ~36 cycles of issue whatever its width (tools/mfma_valu_coissue.hip), against 32 per
// MFMA, so both operands are read 16 bytes at a time: the patch keeps the channels of two
// consecutive 8-channel steps interleaved per lane quarter ([kq][step parity][2]), the weights are
// packed as [tap][step pair][column block][lane][4] (pack_conv_weights, ConvWeights::paired) --
// one ds_read_b128 / global_load_dwordx4 feeds four MFMAs instead of two.
#include <type_traits>
#include "conv_mfma.h"

namespace jh {

namespace {
constexpr int kDTY = 8, kDTX = 16, kDPY = kDTY + 2, kDPX = kDTX + 2, kDNPIX = kDPY * kDPX;
constexpr int kDSPAD = 4;
}  // namespace

// TR: layers without fused statistics issue the MFMAs with swapped operands (conv_epilogue_tr)
template <int NRP, int KC8, bool TR>
__global__ __launch_bounds__(256) void deconv4_fused_kernel(const ConvArgs a) {
  constexpr int MR = 2, NR = 4 * NRP;
  constexpr int KC = KC8 * 8, S = KC + kDSPAD, S2 = S / 2, Q4 = KC / 4, KP = KC8 / 2;
  static_assert(KC8 % 2 == 0 && S2 % 2 == 0, "paired operand layout");
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float* nrm = lds_all;                         // [cin_p] mean, [cin_p] rstd (optional)
  float* lds = lds_all + a.nrm_floats;          // halo patch [180][S]
  float2* lds2 = reinterpret_cast<float2*>(lds);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 15, kq = lane >> 4;
  const int tiles_x = (a.Win + kDTX - 1) / kDTX;
  const BlockId bid = xcd_block();
  const int tile_x = bid.x % tiles_x, tile_y = bid.x / tiles_x;
  const int n = bid.z;
  const int oy0 = tile_y * kDTY, ox0 = tile_x * kDTX;          // in input pixels
  const int iy0 = oy0 - 1, ix0 = ox0 - 1;

  int abase[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int p = (wave * MR + mr) * 16 + mrow;
    abase[mr] = ((p / kDTX) * kDPX + p % kDTX) * S2 + kq * 2;          // (float2 units, 16-byte aligned)
  }
  f32x4 acc[MR][NR];                            // [row block][parity * NRP + column block]
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (a.in_stats) {
    for (int c = tid; c < a.cin_p; c += 256) {
      const double* st = a.in_stats + ((size_t)n * a.cin_p + c) * kStatW;
      const double mu = exact_read(st) * (double)a.in_inv;
      double var = exact_read(st + kLimbs) * (double)a.in_inv - mu * mu;
      if (var < 0.0) var = 0.0;
      nrm[c] = (float)mu;
      nrm[a.cin_p + c] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }
  const float* __restrict__ xin = a.x + (size_t)n * a.Hin * a.Win * a.in_px;
  const int nkc8_total = a.cin_p >> 3;
  const float4* __restrict__ wbase = reinterpret_cast<const float4*>(a.w);
  const unsigned ulane = lane;
  const int tap_stride = (nkc8_total >> 1) * NRP * 64;     // float4 units (launcher: cout_p16 = 16 NRP)
  const int phase_stride4 = (int)(a.phase_stride >> 2);

  // staging: as the PF kernels of conv_mfma.h (items addressed once, zero padding = out-of-range
  // buffer loads, the next channel pass's loads in flight under this pass's MFMAs)
  constexpr int ITER = (kDNPIX * Q4 + 255) / 256;
  static_assert(256 % Q4 == 0, "one channel quad per thread");
  typedef float cf4 __attribute__((ext_vector_type(4)));
  float4 pf[ITER];
  int pvo[ITER];
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xin), 0, (int)((size_t)a.Hin * a.Win * a.in_px * 4), 0x00020000);
  const int c4 = tid % Q4;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int pix = tid / Q4 + it * (256 / Q4);
    const int px = pix % kDPX, py = pix / kDPX;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool ok = pix < kDNPIX && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
    pvo[it] = ok ? ((iy * a.Win + ix) * a.in_px + c4 * 4) * 4 : (int)0x80000000;
  }
  auto issue_pf = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int off = (c0 + c4 * 4 < a.in_px) ? pvo[it] : (int)0x80000000;
      const cf4 v = __builtin_bit_cast(cf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, c0 * 4, 0));
      pf[it] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  issue_pf(0);

  for (int c0 = 0; c0 < a.cin_p; c0 += KC) {
    __syncthreads();
    {
      const int mode = !a.in_stats ? 0 : (a.in_act == ACT_RELU ? 2 : (a.in_act == ACT_SILU ? 3 : 1));
      auto commit_pf = [&](auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = make_float4(1.f, 1.f, 1.f, 1.f);
        const int cc = min(c0 + c4 * 4, a.cin_p - 4);      // (clamped: such items are 0 anyway)
        if (MODE != 0) {
          mu = *reinterpret_cast<const float4*>(nrm + cc);
          rs = *reinterpret_cast<const float4*>(nrm + a.cin_p + cc);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int pix = tid / Q4 + it * (256 / Q4);
          if (pix < kDNPIX) {
            float4 v = pf[it];
            if (MODE != 0) {
              const float m = (pvo[it] < 0 || c0 + c4 * 4 >= a.in_px) ? 0.f : 1.f;
              v.x = (v.x - mu.x) * rs.x; v.y = (v.y - mu.y) * rs.y;
              v.z = (v.z - mu.z) * rs.z; v.w = (v.w - mu.w) * rs.w;
              if (MODE == 2) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
              } else if (MODE == 3) {
                v.x = silu_fast(v.x); v.y = silu_fast(v.y);
                v.z = silu_fast(v.z); v.w = silu_fast(v.w);
              }
              v.x *= m; v.y *= m; v.z *= m; v.w *= m;
            }
            // quad c4 = channels of step c4 / 2, lane quarters 2 (c4 & 1) and + 1; slot of
            // (step, quarter) inside its 16-channel pair: quarter * 2 + step parity
            float2* dst = lds2 + pix * S2 + (c4 >> 2) * 8 + (c4 & 1) * 4 + ((c4 >> 1) & 1);
            dst[0] = make_float2(v.x, v.y);
            dst[2] = make_float2(v.z, v.w);
          }
        }
      };
      if (mode == 0) commit_pf(std::integral_constant<int, 0>{});
      else if (mode == 3) commit_pf(std::integral_constant<int, 3>{});
      else if (mode == 2) commit_pf(std::integral_constant<int, 2>{});
      else commit_pf(std::integral_constant<int, 1>{});
    }
    __syncthreads();
    if (c0 + KC < a.cin_p) issue_pf(c0 + KC);

    int koff[KP];
#pragma unroll
    for (int k8 = 0; k8 < KP; ++k8) koff[k8] = min((c0 >> 4) + k8, (nkc8_total >> 1) - 1) * NRP * 64;
    // the 16 (window position, parity) groups in window order; the weights of group g + 1 are
    // requested before the MFMAs of group g
    auto wptr = [&](int g) __attribute__((always_inline)) -> const float4* {
      // g -> (r, s, py, px) by enumeration (compile-time after unrolling)
      int idx = 0;
      for (int r = 0; r < 3; ++r)
        for (int s = 0; s < 3; ++s)
          for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
              const int ty = r - py, tx = s - px;
              if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
              if (idx == g) return wbase + (size_t)(py * 2 + px) * phase_stride4 + (ty * 2 + tx) * tap_stride;
              ++idx;
            }
      return wbase;
    };
    // Pinned software pipeline (left alone the compiler sinks every weight load to a few MFMAs
    // before its use, far less than an L2 round trip): the weights of group g + 2 and the A rows of
    // the next window position are requested BEFORE the 32 MFMAs of group g, and the scheduling
    // barriers keep them there.
    float4 bq[3][KP][NRP];
#pragma unroll
    for (int g0 = 0; g0 < 2; ++g0) {
      const float4* w0 = wptr(g0) + ulane;
#pragma unroll
      for (int k8 = 0; k8 < KP; ++k8)
#pragma unroll
        for (int nr = 0; nr < NRP; ++nr) bq[g0][k8][nr] = w0[koff[k8] + nr * 64];
    }
    float4 an[KP][MR];
#pragma unroll
    for (int k8 = 0; k8 < KP; ++k8)
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) an[k8][mr] = *reinterpret_cast<const float4*>(lds2 + abase[mr] + k8 * 8);
    int g = 0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        float4 ac[KP][MR];
#pragma unroll
        for (int k8 = 0; k8 < KP; ++k8)
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) ac[k8][mr] = an[k8][mr];
        if (r * 3 + s + 1 < 9) {
          const int r1 = (r * 3 + s + 1) / 3, s1 = (r * 3 + s + 1) % 3;
#pragma unroll
          for (int k8 = 0; k8 < KP; ++k8)
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
              an[k8][mr] = *reinterpret_cast<const float4*>(lds2 + abase[mr] + (r1 * kDPX + s1) * S2 + k8 * 8);
        }
#pragma unroll
        for (int py = 0; py < 2; ++py)
#pragma unroll
          for (int px = 0; px < 2; ++px) {
            const int ty = r - py, tx = s - px;
            if (ty < 0 || ty > 1 || tx < 0 || tx > 1) continue;
            if (g + 2 < 16) {
              const float4* wn = wptr(g + 2) + ulane;
#pragma unroll
              for (int k8 = 0; k8 < KP; ++k8)
#pragma unroll
                for (int nr = 0; nr < NRP; ++nr) bq[(g + 2) % 3][k8][nr] = wn[koff[k8] + nr * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            const int ph = py * 2 + px;
#pragma unroll
            for (int k8 = 0; k8 < KP; ++k8) {
              // x, y: the even 8-channel step; z, w: the odd one (the order of the unpaired kernels)
#define JH_D4_STEP(C)                                                                                          \
  _Pragma("unroll") for (int mr = 0; mr < MR; ++mr) _Pragma("unroll") for (int nr = 0; nr < NRP; ++nr)       \
    acc[mr][ph * NRP + nr] = TR ? __builtin_amdgcn_mfma_f32_16x16x4f32(bq[g % 3][k8][nr].C, ac[k8][mr].C,    \
                                                                       acc[mr][ph * NRP + nr], 0, 0, 0)       \
                                : __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].C, bq[g % 3][k8][nr].C,    \
                                                                       acc[mr][ph * NRP + nr], 0, 0, 0);
              JH_D4_STEP(x) JH_D4_STEP(y) JH_D4_STEP(z) JH_D4_STEP(w)
#undef JH_D4_STEP
            }
            __builtin_amdgcn_sched_barrier(0);
            ++g;
          }
      }
  }

  // ---- epilogue, one parity at a time
  __syncthreads();
  EpilogueArgs e;
  e.y = a.y + (size_t)n * a.Hy * a.Wy * a.cout_p;
  e.bias = a.bias;
  e.stats = a.stats ? a.stats + (size_t)n * a.cout_p * kStatW : nullptr;
  e.Dout = 1; e.Hout = a.Hin; e.Wout = a.Win; e.Hy = a.Hy; e.Wy = a.Wy;
  e.cout_p = a.cout_p; e.cout_p16 = a.cout_p16; e.os = 2; e.osz = 2; e.offz = 0;
  const bool full = oy0 + kDTY <= a.Hin && ox0 + kDTX <= a.Win;
#pragma unroll
  for (int ph = 0; ph < 4; ++ph) {
    f32x4 pa[MR][NRP];
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int nr = 0; nr < NRP; ++nr) pa[mr][nr] = acc[mr][ph * NRP + nr];
    e.offy = ph >> 1; e.offx = ph & 1;
    if constexpr (TR) {
      if (full) conv_epilogue_tr<MR, NRP, kDTY, kDTX, true>(pa, e, 0, 0, oy0, ox0, tid);
      else conv_epilogue_tr<MR, NRP, kDTY, kDTX, false>(pa, e, 0, 0, oy0, ox0, tid);
    } else {
      if (full) conv_epilogue<MR, NRP, kDTY, kDTX, 4, true>(pa, e, lds, 0, 0, oy0, ox0, tid);
      else conv_epilogue<MR, NRP, kDTY, kDTX>(pa, e, lds, 0, 0, oy0, ox0, tid);
      if (e.stats && ph < 3) __syncthreads();   // the reduction scratch is reused
    }
  }
}

template <int NRP, int KC8, bool TR>
static int launch_deconv4_tr(const ConvArgs& a, hipStream_t s) {
  size_t lds = (size_t)kDNPIX * (KC8 * 8 + kDSPAD) * sizeof(float);
  const size_t red = (size_t)4 * 4 * 16 * 2 * sizeof(double);
  if (lds < red) lds = red;
  lds += (size_t)a.nrm_floats * sizeof(float);
  auto kern = deconv4_fused_kernel<NRP, KC8, TR>;
  const int tiles = ((a.Hin + kDTY - 1) / kDTY) * ((a.Win + kDTX - 1) / kDTX);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, a.N), dim3(256), lds, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int NRP, int KC8>
static int launch_deconv4_inst(const ConvArgs& a, hipStream_t s) {
  if (!a.stats && JH_ENV_KNOB("JH_CONV_TR") != 0) return launch_deconv4_tr<NRP, KC8, true>(a, s);
  return launch_deconv4_tr<NRP, KC8, false>(a, s);
}

// Layers this kernel takes (their weights are then packed in the paired layout)
bool deconv4_eligible(int cin_p, int cout_p16) {
  return cout_p16 <= 32 && cin_p % 16 == 0 && JH_ENV_KNOB("JH_DECONV_FUSED") != 0;
}

// Returns -1 when the layer is not this kernel's (the caller then takes the general path).
int launch_deconv4_fused(const ConvArgs& a, hipStream_t s) {
  const int nrp = a.cout_p16 / 16;
  if (a.gate || nrp > 2 || a.nphase != 4) return -1;
  if (!a.paired) return -1;                        // (pack_conv_weights decided with deconv4_eligible)
  const int kc8 = a.cin_p % 32 == 0 ? 4 : 2;
  if (nrp == 2 && kc8 == 4) return launch_deconv4_inst<2, 4>(a, s);
  if (nrp == 2 && kc8 == 2) return launch_deconv4_inst<2, 2>(a, s);
  if (nrp == 1 && kc8 == 4) return launch_deconv4_inst<1, 4>(a, s);
  return launch_deconv4_inst<1, 2>(a, s);
}

}  // namespace jh
