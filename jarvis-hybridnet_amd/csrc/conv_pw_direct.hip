// Pointwise (1 x 1) convolution of a FEW-channel input, operands straight from registers (round 6).
//
// The project convolutions of the first MBConv stages and the BiFPN's lateral convolutions (16 -> 8 @ 128^2,
// 48 -> 16 @ 64^2, 16 -> 56 @ 64^2, 24 -> 56 @ 32^2 in the small model; jarvis/efficienttrack/efficientnet.py:114,
// model.py:404-425) read and write far more bytes than they multiply: they are HBM-bound, and in the general MFMA
// kernel (conv_mfma.h: global -> registers -> LDS -> registers, two barriers per pass, a cross-wave reduction of the
// statistics) they ran at about half of the memory rate.  With K <= 48 nothing needs to be shared:
//   * a wave owns 16-pixel row blocks; lane (p = lane & 15, kq = lane >> 4) loads the 16-byte words u * 4 + kq of pixel
//     p (channels 16 u + 4 kq .. + 3) -- every byte of the input exactly once per wave -- and applies InstanceNorm /
//     activation / squeeze-excite gate to its four values in registers;
//   * the MFMAs take the WEIGHTS as A and the pixels as B with K permuted to match (step 4 u + i multiplies channel
//     16 u + 4 kq + i: any bijection of K is a valid contraction order, the weights are fetched in the same order),
//     so a lane ends up with channels 4 kq .. + 3 of column block cb for its pixel: one 16-byte store, no transposes;
//   * the weights of all steps and column blocks stay in registers for the wave's life (<= 72), gathered once from
//     the packed operand of the general kernel (no second weight layout);
//   * statistics: packed fp32 sums around a pivot shared by the 16 pixel lanes of a row (ds_bpermute of the first
//     value), reduced over the row by DPP, un-shifted in fp64 and added with ONE atomic per value and wave
//     (jh_common.h: stat_add(double)).
// No LDS traffic beyond 128 floats of mean / rstd per wave, no barrier.  Row blocks per wave: a function of the image
// size only (the per-wave partials are part of the statistics' arithmetic).
#include "conv_mfma.h"

namespace jh {

namespace {
typedef float pf4 __attribute__((ext_vector_type(4)));
typedef unsigned pu4 __attribute__((ext_vector_type(4)));

template <int KW, int NCB, int MRU>
__global__ __launch_bounds__(256) void conv_pw_direct_kernel(const ConvArgs a, int rb_per_wave) {
  __shared__ float nrm_s[4][2][64];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int px = lane & 15, kq = lane >> 4;
  const int n = blockIdx.y;
  const int P = a.Hin * a.Win;                     // pixels per image (a multiple of 16: the launcher)
  const int nrb = P >> 4;
  const int rb0 = (blockIdx.x * 4 + wave) * rb_per_wave;
  const int rb1 = min(nrb, rb0 + rb_per_wave);
  if (rb0 >= nrb) return;                          // (whole waves: the kernel has no barrier)

  // ---- weights: A operand of step (u, i) and column block cb = W[16 cb + px][16 u + 4 kq + i], gathered from the packed
  // B-operand layout of conv_mfma.h (pack_conv_weights: [cin_p / 8][cout_p16 / 16][64 lanes][2], lane (n, k) holds the
  // channels 8 k8 + 2 k and 8 k8 + 2 k + 1 of output channel 16 cb + n)
  float aw[KW][4][NCB];
#pragma unroll
  for (int u = 0; u < KW; ++u)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = 16 * u + 4 * kq + i;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int idx = (((c >> 3) * NCB + cb) * 64 + ((c & 7) >> 1) * 16 + px) * 2 + (c & 1);
        aw[u][i][cb] = c < a.cin_p ? a.w[idx] : 0.f;
      }
    }
  // ---- InstanceNorm constants of the input: lane c computes channel c (fp64), shared through this wave's LDS row
  const bool normed = a.in_stats != nullptr;
  pf4 mu4[KW], rs4[KW], g4[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u) { mu4[u] = (pf4){0.f, 0.f, 0.f, 0.f}; rs4[u] = (pf4){1.f, 1.f, 1.f, 1.f}; g4[u] = rs4[u]; }
  if (normed) {
    float m = 0.f, r = 1.f;
    if (lane < a.cin_p) {
      const double* st = a.in_stats + ((size_t)n * a.cin_p + lane) * kStatW;
      const double mu = exact_read(st) * (double)a.in_inv;
      double var = exact_read(st + kLimbs) * (double)a.in_inv - mu * mu;
      if (var < 0.0) var = 0.0;
      m = (float)mu;
      r = (float)(1.0 / sqrt(var + 1e-5));
    }
    nrm_s[wave][0][lane] = m;
    nrm_s[wave][1][lane] = r;
    __builtin_amdgcn_wave_barrier();               // this wave wrote, this wave reads (LDS operations complete in order)
#pragma unroll
    for (int u = 0; u < KW; ++u)
      if (16 * u + 4 * kq < a.cin_p) {
        mu4[u] = *reinterpret_cast<const pf4*>(&nrm_s[wave][0][16 * u + 4 * kq]);
        rs4[u] = *reinterpret_cast<const pf4*>(&nrm_s[wave][1][16 * u + 4 * kq]);
      }
  }
  const bool gated = a.gate != nullptr || a.se.pool != nullptr;
  if (a.gate) {
#pragma unroll
    for (int u = 0; u < KW; ++u)
      if (16 * u + 4 * kq < a.cin_p) g4[u] = *reinterpret_cast<const pf4*>(a.gate + (size_t)n * a.cin_p + 16 * u + 4 * kq);
  } else if (a.se.pool) {
    // the squeeze-excite gate of image n from the pooled sums, per wave (C <= 48 channels, S <= 16 hidden units: lane c /
    // lane j each own one): the arithmetic of se_gate_kernel (elementwise.hip) and of the general kernel's prologue, term
    // for term -- whichever form the plan picks (by batch size), the gate has the same bits
    __shared__ float se_s[4][2][64];
    const int C = a.se.C, S = a.se.S;
    se_s[wave][0][lane] = lane < C ? (float)(exact_read(a.se.pool + ((size_t)n * a.cin_p + lane) * kLimbs) * (double)a.se.inv_hw) : 0.f;
    __builtin_amdgcn_wave_barrier();
    float hid = 0.f;
    if (lane < S) {
      float acc = a.se.br[lane];
      for (int c = 0; c < C; ++c) acc = fmaf(a.se.wr[lane * C + c], se_s[wave][0][c], acc);
      hid = acc / (1.f + expf(-acc));
    }
    se_s[wave][1][lane] = hid;
    __builtin_amdgcn_wave_barrier();
    float g = 0.f;
    if (lane < C) {
      float acc = a.se.be[lane];
      for (int j = 0; j < S; ++j) acc = fmaf(a.se.we[lane * S + j], se_s[wave][1][j], acc);
      g = 1.f / (1.f + expf(-acc));
    }
    __builtin_amdgcn_wave_barrier();                // (every lane has read the means before they are overwritten)
    se_s[wave][0][lane] = g;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < KW; ++u)
      if (16 * u + 4 * kq < a.cin_p) g4[u] = *reinterpret_cast<const pf4*>(&se_s[wave][0][16 * u + 4 * kq]);
  }
  const int act = a.in_act;

  // ---- addressing: the image as a buffer; a lane's word u of pixel p of row block rb
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.x + (size_t)n * P * a.cin_p), 0, (int)((size_t)P * a.cin_p * 4), 0x00020000);
  int xoff[KW];
#pragma unroll
  for (int u = 0; u < KW; ++u)
    xoff[u] = 16 * u + 4 * kq < a.cin_p ? (px * a.cin_p + 16 * u + 4 * kq) * 4 : (int)0x80000000;   // K padding reads 0
  const int xrb = 16 * a.cin_p * 4;                 // bytes per row block
  const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
      a.y + (size_t)n * P * a.cout_p, 0, (int)((size_t)P * a.cout_p * 4), 0x00020000);
  int yoff[NCB];
  pf4 b4[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) {
    const int c = 16 * cb + 4 * kq;
    yoff[cb] = c < a.cout_p ? (px * a.cout_p + c) * 4 : (int)0x80000000;
    b4[cb] = (a.bias && c < a.cout_p16) ? *reinterpret_cast<const pf4*>(a.bias + c) : (pf4){0.f, 0.f, 0.f, 0.f};
  }
  const int yrb = 16 * a.cout_p * 4;

  pf4 t1[NCB], t2[NCB], pv[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) { t1[cb] = (pf4){0.f, 0.f, 0.f, 0.f}; t2[cb] = t1[cb]; pv[cb] = t1[cb]; }
  bool first = true;

  auto load = [&](int rb, pf4 (&x)[KW]) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < KW; ++u)
      x[u] = __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, xoff[u], rb * xrb, 0));
  };
  auto compute = [&](int rb, pf4 (&x)[KW]) __attribute__((always_inline)) {
    // InstanceNorm + activation + gate on the lane's own values (K padding: words past cin_p are 0 and have zero weights)
#pragma unroll
    for (int u = 0; u < KW; ++u) {
      pf4 v = x[u];
      if (normed) {
        v = (v - mu4[u]) * rs4[u];
        if (act == ACT_SILU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = silu_fast(v[j]);
        } else if (act == ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
      }
      if (gated) v *= g4[u];
      x[u] = v;
    }
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < KW; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[u][i][cb], x[u][i], acc[cb], 0, 0, 0);
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const pf4 v = (pf4){acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]} + b4[cb];
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu4, v), yrs, yoff[cb], rb * yrb, 0);
      if (a.stats) {
        if (first) {
          // pivot = the value of the row's first pixel lane (lane & 48), the same for the 16 lanes that share channels
#pragma unroll
          for (int j = 0; j < 4; ++j)
            pv[cb][j] = __int_as_float(__builtin_amdgcn_ds_bpermute((lane & 48) << 2, __float_as_int(v[j])));
        }
        const pf4 d = v - pv[cb];
        t1[cb] += d;
        t2[cb] = __builtin_elementwise_fma(d, d, t2[cb]);
      }
    }
    first = false;
  };

  int rb = rb0;
  for (; rb + MRU <= rb1; rb += MRU) {
    pf4 x[MRU][KW];
#pragma unroll
    for (int m = 0; m < MRU; ++m) load(rb + m, x[m]);
#pragma unroll
    for (int m = 0; m < MRU; ++m) compute(rb + m, x[m]);
  }
  for (; rb < rb1; ++rb) {
    pf4 x[KW];
    load(rb, x);
    compute(rb, x);
  }

  if (a.stats) {
    // sum over the 16 pixel lanes of a DPP row: xor 1, xor 2 (quad permutes), half-row mirror, row mirror
    auto row_sum = [](float x) __attribute__((always_inline)) {
      x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true));
      x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true));
      x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true));
      x += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x140, 0xF, 0xF, true));
      return x;
    };
    const double cnt = (double)(16 * (rb1 - rb0));
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float r1 = row_sum(t1[cb][j]), r2 = row_sum(t2[cb][j]);
        const int ch = 16 * cb + 4 * kq + j;
        if (px == 0 && ch < a.cout_p) {
          const double md = (double)pv[cb][j], d1 = (double)r1;
          stat_add(a.stats + ((size_t)n * a.cout_p + ch) * kStatW, fma(cnt, md, d1),
                   fma(md, fma(cnt, md, 2.0 * d1), (double)r2));
        }
      }
  }
}
}  // namespace

// Layers this kernel takes: plain 1 x 1 convolutions with at most 48 input channels and 1 or 4 column blocks of output
// channels on images of a multiple of 16 pixels, gate (if any) given as a tensor.  JH_CONV_PW_DIRECT=0: never.
// The shape part of the decision.  Which kernel runs must not depend on the batch (the two kernels group the statistics'
// partial sums differently): this kernel therefore takes the squeeze-excite gate in BOTH forms the plan uses -- a tensor,
// or the recipe the general kernel evaluates in its prologue for small batches.
bool conv_pw_direct_shape_ok(int cin_p, int cout_p16, int pixels) {
  if (JH_ENV_KNOB("JH_CONV_PW_DIRECT") == 0) return false;
  const int nb = cout_p16 / 16;
  if (cin_p > 48 || (nb != 1 && nb != 4)) return false;
  return pixels % 16 == 0 && pixels >= (nb == 1 ? 1024 : 4096);
}

bool conv_pw_direct_eligible(const ConvDesc& d, const ConvArgs& a) {
  if (JH_ENV_KNOB("JH_CONV_PW_DIRECT") == 0) return false;
  if (d.nd != 2 || d.k != 1 || d.stride != 1 || d.ostride != 1 || d.nphase != 1 || a.paired) return false;
  if (a.se.pool && (a.se.C > 64 || a.se.S > 64)) return false;
  if (a.cin_p > 48 || a.in_px != a.cin_p) return false;
  const int nb = a.cout_p16 / 16;
  // (six column blocks -- the 88-channel laterals of the medium model -- need 256 registers: one wave per SIMD, measured
  //  24 -> 88 @ 64^2 0.19 -> 0.28 ms: not instantiated)
  if (nb != 1 && nb != 4) return false;
  const int P = a.Hin * a.Win;
  // (measured at 384 images: one column block 16 -> 8 @ 128^2 0.132 -> 0.116 ms = 0.65 of HBM, 48 -> 16 @ 64^2 0.095 ->
  //  0.080; four column blocks 16 -> 56 @ 64^2 0.125 -> 0.115, but 24 -> 56 @ 32^2 0.050 -> 0.061: with 64 row blocks per
  //  image the waves' prologue -- weight gather, fp64 mean / rstd -- and statistics epilogue are not amortised)
  const int min_px = nb == 1 ? 1024 : 4096;
  return P % 16 == 0 && P >= min_px && a.Hout == a.Hin && a.Wout == a.Win && a.Hy == a.Hin && a.Wy == a.Win;
}

int launch_conv_pw_direct(const ConvArgs& a, hipStream_t s) {
  const int P = a.Hin * a.Win, nrb = P / 16;
  // row blocks per wave: a function of the image size only
  const int rbw = std::max(8, std::min(32, nrb / 16));
  const dim3 grid((nrb + 4 * rbw - 1) / (4 * rbw), a.N);
  const int kw = (a.cin_p + 15) / 16, nb = a.cout_p16 / 16;
#define JH_PWD(KWV, NCBV, MRUV)                                                                                 \
  if (kw == KWV && nb == NCBV) {                                                                                \
    hipLaunchKernelGGL((conv_pw_direct_kernel<KWV, NCBV, MRUV>), grid, dim3(256), 0, s, a, rbw);                 \
    JH_CHECK_HIP(hipGetLastError());                                                                            \
    return 0;                                                                                                   \
  }
  JH_PWD(1, 1, 4) JH_PWD(2, 1, 4) JH_PWD(3, 1, 4)
  JH_PWD(1, 4, 4) JH_PWD(2, 4, 2) JH_PWD(3, 4, 2)
#undef JH_PWD
  JH_REQUIRE(false, "no direct pointwise kernel for this shape");
}

}  // namespace jh
