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
#include <cstdlib>
#include <type_traits>
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
  // TM is a multiple of 64, or -- the whole-image geometries, e.g. 23 x 20 for images up to 22 x 20 -- the pixel slots
  // past 64 MR are never computed: the launcher uses such a geometry only where they lie outside the output
  static_assert(TM % 64 == 0 || (TZ == 1 && TM / 64 >= 1), "tile must cover at least 64 pixels");
  static size_t lds_bytes(int kc) { return (size_t)NPIX * (kc + SPAD) * sizeof(float); }
};

struct EpilogueArgs {
  float* y;              // output of image n
  const float* bias;
  double* stats;         // statistics row of image n, or nullptr
  int Dout, Hout, Wout, Hy, Wy, cout_p, cout_p16, os, offz, offy, offx;
  int osz;               // output stride along z (= os except for the (y, x)-only phases of conv3d_wino)
};

// Shared epilogue of the MFMA kernels: bias, store of the raw output, and the
// per-channel sum / sum of squares of the tile (registers -> wave shuffles -> LDS ->
// one fp64 atomic pair per channel and workgroup).  `red` is >= NW*NR*16*2 DOUBLES of LDS, 8-byte aligned
// (NW = waves per workgroup).
// exchange with the lane whose index differs in bit 0 / bit 1 (DPP quad permutes)
__device__ __forceinline__ float quad_xor1(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

// FULL: the caller guarantees that the whole tile lies inside the output (no bounds checks).
template <int MR, int NR, int TY, int TX, int NW = 4, bool FULL = false>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[MR][NR], const EpilogueArgs& e,
                                              float* red, int nb0, int oz0, int oy0, int ox0,
                                              int tid) {
  const int lane = tid & 63, wave = tid >> 6, mrow = lane & 15, kq = lane >> 4;
  const int j = lane & 3;                  // position inside the lane quad
  double* redd = reinterpret_cast<double*>(red);   // NW * NR * 16 * 2 doubles (callers size and 8-byte-align `red` for it)
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int ch = (nb0 + nr) * 16 + mrow;
    const bool ch_ok = ch < e.cout_p;
    const float bv = (e.bias && ch < e.cout_p16) ? e.bias[ch] : 0.f;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[mr][nr][r] + bv;
      // The MFMA result has 4 pixels x 1 channel per lane.  Transpose 4x4 inside each
      // lane quad so a lane holds 1 pixel x 4 consecutive channels: one 16-byte store
      // instead of four 4-byte stores (the epilogue is store-issue bound).
      {
        float x, y;
        x = (j & 1) ? v[0] : v[1]; y = quad_xor1(x); if (j & 1) v[0] = y; else v[1] = y;
        x = (j & 1) ? v[2] : v[3]; y = quad_xor1(x); if (j & 1) v[2] = y; else v[3] = y;
        x = (j & 2) ? v[0] : v[2]; y = quad_xor2(x); if (j & 2) v[0] = y; else v[2] = y;
        x = (j & 2) ? v[1] : v[3]; y = quad_xor2(x); if (j & 2) v[1] = y; else v[3] = y;
      }
      const int p = (wave * MR + mr) * 16 + kq * 4 + j;
      const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
      const int oz = oz0 + tz, oy = oy0 + ty, ox = ox0 + tx;
      const int c0 = (nb0 + nr) * 16 + (mrow & ~3);
      if (c0 < e.cout_p && (FULL || (oz < e.Dout && oy < e.Hout && ox < e.Wout)))
        *reinterpret_cast<float4*>(
            e.y + ((size_t)((oz * e.osz + e.offz) * e.Hy + (oy * e.os + e.offy)) * e.Wy + (ox * e.os + e.offx)) * e.cout_p + c0) =
            make_float4(v[0], v[1], v[2], v[3]);
    }
    // Statistics partials of this lane, handed on as DOUBLES (jh_common.h: stat_add(double)): accumulated in packed
    // fp32 around a per-lane PIVOT (the lane's first value: sums of v - m and (v - m)^2, whose rounding errors scale
    // with the spread of the data instead of its mean), un-shifted in fp64 once per lane and column block:
    //   sum v = n m + sum d,   sum v^2 = n m^2 + 2 m sum d + sum d^2.
    // 1.5 packed instructions per value + 8 fp64 ones per lane (all-fp64 accumulation measured +20 % on the few-channel
    // layers: conv2d_k1s1_16x8@128 0.146 -> 0.182 ms).  Behind the stores: they drain while this runs.
    double s1 = 0.0, s2 = 0.0;
    if (e.stats) {
      typedef float sf2 __attribute__((ext_vector_type(2)));
      // (any finite value works as the pivot; pixel slots outside the output hold sums over the zero-filled halo)
      const float m = acc[0][nr][0] + bv;
      const sf2 m2 = (sf2){m, m};
      sf2 t1a = (sf2){0.f, 0.f}, t1b = t1a, t2a = t1a, t2b = t1a;
      int cnt = FULL ? MR * 4 : 0;
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) {
        sf2 da = (sf2){acc[mr][nr][0] + bv, acc[mr][nr][1] + bv} - m2;
        sf2 db = (sf2){acc[mr][nr][2] + bv, acc[mr][nr][3] + bv} - m2;
        if (!FULL) {
          bool ok[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int p = (wave * MR + mr) * 16 + kq * 4 + r;
            const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
            ok[r] = ch_ok && oz0 + tz < e.Dout && oy0 + ty < e.Hout && ox0 + tx < e.Wout;
            cnt += ok[r] ? 1 : 0;
          }
          da = (sf2){ok[0] ? da[0] : 0.f, ok[1] ? da[1] : 0.f};
          db = (sf2){ok[2] ? db[0] : 0.f, ok[3] ? db[1] : 0.f};
        }
        // (FULL: no test at all -- channels past cout have zero weights and zero bias, their values are 0)
        t1a += da; t1b += db;
        t2a = __builtin_elementwise_fma(da, da, t2a);
        t2b = __builtin_elementwise_fma(db, db, t2b);
      }
      const sf2 t1 = t1a + t1b, t2 = t2a + t2b;
      const double md = (double)m, nd = (double)cnt, d1 = (double)(t1[0] + t1[1]), d2 = (double)(t2[0] + t2[1]);
      s1 = fma(nd, md, d1);
      s2 = fma(md, fma(nd, md, 2.0 * d1), d2);
    }
    if (e.stats) {
      s1 = sum_xor16(s1); s2 = sum_xor16(s2);
      s1 = sum_xor32(s1); s2 = sum_xor32(s2);
      if (kq == 0) {
        redd[(wave * NR * 16 + nr * 16 + mrow) * 2 + 0] = s1;
        redd[(wave * NR * 16 + nr * 16 + mrow) * 2 + 1] = s2;
      }
    }
  }
  if (e.stats) {
    __syncthreads();
    if (tid < NR * 16) {
      const int ch = nb0 * 16 + tid;
      if (ch < e.cout_p) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          s1 += redd[(w * NR * 16 + tid) * 2 + 0];
          s2 += redd[(w * NR * 16 + tid) * 2 + 1];
        }
        stat_add(e.stats + (size_t)ch * kStatW, s1, s2);
      }
    }
  }
}

// Epilogue of kernels that issue their MFMAs with the operands SWAPPED (weights as A, pixels as
// B): the accumulator is then the transposed tile -- a lane holds, for pixel lane & 15 of the
// block, the four consecutive channels 4 (lane >> 4) .. + 3 -- which is exactly one 16-byte
// channel-last store: no 4 x 4 lane-quad transposes (16 of the ~34 instructions per block of
// the epilogue above).  The per-channel statistics would need a 16-lane reduction in this
// layout, so it is used for layers WITHOUT fused statistics only (the heads' last layers).
template <int MR, int NR, int TY, int TX, bool FULL>
__device__ __forceinline__ void conv_epilogue_tr(f32x4 (&acc)[MR][NR], const EpilogueArgs& e, int nb0,
                                                 int oz0, int oy0, int ox0, int tid) {
  const int lane = tid & 63, wave = tid >> 6, mrow = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int c0 = (nb0 + nr) * 16 + kq * 4;
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (e.bias && c0 < e.cout_p16) bv = *reinterpret_cast<const f32x4*>(e.bias + c0);
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
      const int p = (wave * MR + mr) * 16 + mrow;
      const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
      const int oz = oz0 + tz, oy = oy0 + ty, ox = ox0 + tx;
      const f32x4 v = acc[mr][nr] + bv;
      if (c0 < e.cout_p && (FULL || (oz < e.Dout && oy < e.Hout && ox < e.Wout)))
        *reinterpret_cast<f32x4*>(
            e.y + ((size_t)((oz * e.osz + e.offz) * e.Hy + (oy * e.os + e.offy)) * e.Wy + (ox * e.os + e.offx)) * e.cout_p + c0) = v;
    }
  }
}

template <int ND, int K, int STRIDE, int TZ, int TY, int TX, int NR, int KC8>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
  using G = ConvGeom<ND, K, STRIDE, TZ, TY, TX>;
  constexpr int MR = G::MR;
  constexpr int KC = KC8 * 8;                 // input channels staged per LDS pass
  constexpr int S = KC + G::SPAD;             // LDS pixel stride in floats (even)
  constexpr int S2 = S / 2;                   // ... in float2 units
  constexpr int Q4 = KC / 4;                  // float4 per pixel and pass
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  float* nrm = lds_all;                         // [cin_p] mean, [cin_p] rstd (optional)
  float* lds = lds_all + a.nrm_floats;          // halo patch
  float2* lds2 = reinterpret_cast<float2*>(lds);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 15;
  const int kq = lane >> 4;

  // ---- block coordinates
  const int tiles_x = (a.Wout + TX - 1) / TX;
  const int tiles_y = (a.Hout + TY - 1) / TY;
  const BlockId bid = xcd_block(a.fgx, a.fgy);  // neighbouring tiles share an XCD's L2
  const int t1 = (int)fd_div(bid.x, a.ftx);
  const int tile_x = (int)bid.x - t1 * tiles_x;
  const int tile_z = (int)fd_div((unsigned)t1, a.fty);
  const int tile_y = t1 - tile_z * tiles_y;
  const int nb0 = bid.y * NR;
  int n = bid.z, ph = 0;
  if (a.nphase > 1) {                          // (power of two: 4 or 8 phases)
    ph = bid.z & (a.nphase - 1);
    n = bid.z >> (a.nphase == 4 ? 2 : 3);
  }
  const int oz0 = tile_z * TZ, oy0 = tile_y * TY, ox0 = tile_x * TX;
  const int iz0 = oz0 * STRIDE - a.phase[ph].pad[0];
  const int iy0 = oy0 * STRIDE - a.phase[ph].pad[1];
  const int ix0 = ox0 * STRIDE - a.phase[ph].pad[2];

  // ---- per-lane LDS base (float2 units) of each of this wave's 16-pixel row blocks
  int abase[MR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr) {
    const int p = (wave * MR + mr) * 16 + mrow;
    const int tx = p % TX, ty = (p / TX) % TY, tz = p / (TX * TY);
    abase[mr] = (((tz * STRIDE) * G::PY + ty * STRIDE) * G::PX + tx * STRIDE) * S2 + kq;
  }

  f32x4 acc[MR][NR];
#pragma unroll
  for (int mr = 0; mr < MR; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (a.in_stats) {
    // InstanceNorm of the input from the producer's fused statistics (biased variance)
    for (int c = tid; c < a.cin_p; c += 256) {
      const double* st = a.in_stats + ((size_t)n * a.cin_p + c) * kStatW;
      const double mu = exact_read(st) * (double)a.in_inv;
      double var = exact_read(st + kLimbs) * (double)a.in_inv - mu * mu;
      if (var < 0.0) var = 0.0;
      nrm[c] = (float)mu;
      nrm[a.cin_p + c] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }

  // the squeeze-excite gate of image n, behind mean / rstd (read while the patch is staged)
  float* gate_l = nrm + (a.in_stats ? 2 * a.cin_p : 0);
  const bool gated = a.gate || a.se.pool;
  if (a.gate)
    for (int c = tid; c < a.cin_p; c += 256) gate_l[c] = a.gate[(size_t)n * a.cin_p + c];
  if (a.se.pool) {
    // the gate of image n from the pooled sums: the arithmetic of se_gate_kernel (elementwise.hip), term
    // for term, so the fused and the stand-alone form give the same bits
    const int C = a.se.C, S = a.se.S;
    float* mean = gate_l + a.cin_p;
    float* hid = mean + ((C + 3) & ~3);
    for (int c = tid; c < C; c += 256)
      mean[c] = (float)(exact_read(a.se.pool + ((size_t)n * a.cin_p + c) * kLimbs) * (double)a.se.inv_hw);
    __syncthreads();
    for (int j = tid; j < S; j += 256) {
      float acc = a.se.br[j];
      for (int c = 0; c < C; ++c) acc = fmaf(a.se.wr[j * C + c], mean[c], acc);
      hid[j] = acc / (1.f + expf(-acc));
    }
    __syncthreads();
    for (int c = tid; c < a.cin_p; c += 256) {
      float g = 0.f;
      if (c < C) {
        float acc = a.se.be[c];
        for (int j = 0; j < S; ++j) acc = fmaf(a.se.we[c * S + j], hid[j], acc);
        g = 1.f / (1.f + expf(-acc));
      }
      gate_l[c] = g;
    }
  }

  const float* __restrict__ xin = a.x + (size_t)n * a.Din * a.Hin * a.Win * a.in_px;
  const int nkc8_total = a.cin_p >> 3;
  const int nb16_total = a.cout_p16 >> 4;
  // per-lane weight base; column blocks past the end are clamped (their results are
  // never stored), so the main loop has no conditional loads
  // (uniform base + 32-bit lane offset: the loads take the SGPR-base addressing form and
  // their addresses are scalar arithmetic)
  int boff[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) boff[nr] = min(nb0 + nr, nb16_total - 1) * 64;
  const int tap_stride = nkc8_total * nb16_total * 64;     // float2 units
  // Weight loads are buffer loads: per-lane byte offset (column block, lane) in a register, the
  // (tap, channel step) part as the SCALAR offset -- no per-load vector address arithmetic (the
  // pointer form cost two 64-bit vector adds per load).
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.w + (size_t)ph * a.phase_stride), 0, (int)(a.phase_stride * 4), 0x00020000);
  typedef unsigned int wu2 __attribute__((ext_vector_type(2)));
  int wvo[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) wvo[nr] = (boff[nr] + lane) * 8;
  auto wload = [&](int idx2, int nr) __attribute__((always_inline)) -> float2 {   // idx2: uniform float2 index
    const wu2 v = __builtin_amdgcn_raw_buffer_load_b64(wrs, wvo[nr], idx2 * 8, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
  };

  // Patch staging in two halves -- issue(c0): raw global loads into registers; commit(c0):
  // InstanceNorm / activation / gate, then LDS.  Kernels with small patches (PF: 1x1 convs and
  // the 2D layers with few loads per thread) issue the loads of the NEXT channel pass before
  // the MFMAs of the current one: those layers run with one to three workgroups per CU and
  // many short passes, so nothing else hides that round trip.
  constexpr int ITER = (G::NPIX * Q4 + 255) / 256;
  constexpr bool PF = ITER <= 8;
  float4 pf[PF ? ITER : 1];
  // PF kernels: the patch items of a thread are addressed ONCE -- byte offsets relative to the
  // image, with bit 31 set for pixels outside it: a buffer load of such an offset is out of range
  // and returns 0, which is the zero padding, without a branch; the channel pass is the scalar
  // offset of the load.  (PMC on the 16-channel 128^2 layers: 8.5 vector-ALU instructions per
  // MFMA, most of them the index arithmetic of staging, which ran three times per item and pass.)
  // CQ: 256 is a multiple of the channel quads per pixel, so all items of a thread share one
  // channel quad and its mean / rstd / gate are loaded once per pass.
  constexpr bool CQ = (256 % Q4) == 0;
  typedef float cf4 __attribute__((ext_vector_type(4)));
  int pvo[PF ? ITER : 1];
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xin), 0, (int)((size_t)a.Din * a.Hin * a.Win * a.in_px * 4), 0x00020000);
  if constexpr (PF) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      const int c4 = idx % Q4, pix = idx / Q4;
      const int px = pix % G::PX;
      const int py = (pix / G::PX) % G::PY;
      const int pz = pix / (G::PX * G::PY);
      const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
      const bool ok = idx < G::NPIX * Q4 && iz >= 0 && iz < a.Din && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
      pvo[it] = ok ? (((iz * a.Hin + iy) * a.Win + ix) * a.in_px + c4 * 4) * 4 : (int)0x80000000;
    }
  }
  auto issue_pf = [&](int c0) __attribute__((always_inline)) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int c4 = (tid + it * 256) % Q4;
      // channels past the tensor (K padding of the last pass, the 3-channel network input) read 0
      const int off = (c0 + c4 * 4 < a.in_px) ? pvo[it] : (int)0x80000000;
      const cf4 v = __builtin_bit_cast(cf4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off, c0 * 4, 0));
      pf[it] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto patch_ok = [&](int idx, int c0, const float** src) -> bool {
    const int c4 = idx % Q4;
    const int pix = idx / Q4;
    const int px = pix % G::PX;
    const int py = (pix / G::PX) % G::PY;
    const int pz = pix / (G::PX * G::PY);
    const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
    const int c = c0 + c4 * 4;
    *src = xin + ((size_t)(iz * a.Hin + iy) * a.Win + ix) * a.in_px + c;
    return idx < G::NPIX * Q4 && c < a.in_px && iz >= 0 && iz < a.Din && iy >= 0 && iy < a.Hin &&
           ix >= 0 && ix < a.Win;
  };
  auto finish = [&](float4 v, int c) -> float4 {       // in-range pixels only
    if (a.in_stats) {
      const float4 mu = *reinterpret_cast<const float4*>(nrm + c);
      const float4 rs = *reinterpret_cast<const float4*>(nrm + a.cin_p + c);
      v.x = (v.x - mu.x) * rs.x; v.y = (v.y - mu.y) * rs.y;
      v.z = (v.z - mu.z) * rs.z; v.w = (v.w - mu.w) * rs.w;
      if (a.in_act == ACT_RELU) {
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      } else if (a.in_act == ACT_SILU) {
        v.x = silu_fast(v.x); v.y = silu_fast(v.y);
        v.z = silu_fast(v.z); v.w = silu_fast(v.w);
      }
    }
    if (gated) {
      const float4 g = *reinterpret_cast<const float4*>(gate_l + c);
      v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w;
    }
    return v;
  };
  if constexpr (PF) issue_pf(0);

  // BREG: kernels whose whole weight set of a channel pass is small (1x1 convs, narrow 3x3
  // layers) fetch it into registers at the top of the pass, so the round trip to L2 runs under
  // the patch commit and the barriers instead of in front of every tap.
  // (the stride-2 3D family always takes the plain loop: it has a tap-paired weight form there)
  constexpr bool TAPPAIR = ND == 3 && K == 3 && STRIDE == 2;
  constexpr bool PIPE = (G::NT > 1) && (MR * NR * KC8 >= 12) && !TAPPAIR;
  constexpr bool BREG = !PIPE && (G::NT == 1 ? KC8 * NR <= 16 : G::NT * KC8 * NR <= 9);   // (measured)
  for (int c0 = 0; c0 < a.cin_p; c0 += KC) {
    float2 breg[BREG ? G::NT : 1][KC8][NR];
    if constexpr (BREG) {
#pragma unroll
      for (int tp = 0; tp < G::NT; ++tp)
#pragma unroll
        for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            breg[tp][k8][nr] = wload(tp * tap_stride + min((c0 >> 3) + k8, nkc8_total - 1) * nb16_total * 64, nr);
    }
    __syncthreads();
    // ---- stage the halo patch: [pixel][KC] with stride S; channels past cin_p read 0
    if constexpr (PF) {
      // mode: 0 copy, 1 norm, 2 norm + relu, 3 norm + silu (dispatched once per pass, not per
      // item); the gate multiplies in every mode
      const int mode = !a.in_stats ? 0 : (a.in_act == ACT_RELU ? 2 : (a.in_act == ACT_SILU ? 3 : 1));
      auto commit_pf = [&](auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        float4 mu0 = make_float4(0.f, 0.f, 0.f, 0.f), rs0 = make_float4(1.f, 1.f, 1.f, 1.f), gt0 = rs0;
        if (CQ) {
          const int cc = min(c0 + (tid % Q4) * 4, a.cin_p - 4);      // (clamped: such items are 0 anyway)
          if (MODE != 0) {
            mu0 = *reinterpret_cast<const float4*>(nrm + cc);
            rs0 = *reinterpret_cast<const float4*>(nrm + a.cin_p + cc);
          }
          if (gated) gt0 = *reinterpret_cast<const float4*>(gate_l + cc);
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
          const int idx = tid + it * 256;
          if (idx < G::NPIX * Q4) {
            const int c4 = idx % Q4, pix = idx / Q4;
            float4 mu = mu0, rs = rs0, gt = gt0;
            if (!CQ) {
              const int cc = min(c0 + c4 * 4, a.cin_p - 4);
              if (MODE != 0) {
                mu = *reinterpret_cast<const float4*>(nrm + cc);
                rs = *reinterpret_cast<const float4*>(nrm + a.cin_p + cc);
              }
              if (gated) gt = *reinterpret_cast<const float4*>(gate_l + cc);
            }
            float4 v = pf[it];
            if (MODE != 0) {
              // pixels outside the image and channels past the tensor (loaded as 0) stay 0
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
            if (gated) { v.x *= gt.x; v.y *= gt.y; v.z *= gt.z; v.w *= gt.w; }
            float2* dst = lds2 + pix * S2 + c4 * 2;
            dst[0] = make_float2(v.x, v.y);
            dst[1] = make_float2(v.z, v.w);
          }
        }
      };
      if (mode == 0) commit_pf(std::integral_constant<int, 0>{});
      else if (mode == 3) commit_pf(std::integral_constant<int, 3>{});
      else if (mode == 2) commit_pf(std::integral_constant<int, 2>{});
      else commit_pf(std::integral_constant<int, 1>{});
    } else {
      for (int idx = tid; idx < G::NPIX * Q4; idx += 256) {
        const float* src;
        const bool ok = patch_ok(idx, c0, &src);
        const int c4 = idx % Q4, pix = idx / Q4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = finish(*reinterpret_cast<const float4*>(src), c0 + c4 * 4);
        float2* dst = lds2 + pix * S2 + c4 * 2;
        dst[0] = make_float2(v.x, v.y);
        dst[1] = make_float2(v.z, v.w);
      }
    }
    __syncthreads();
    if constexpr (PF) {
      if (c0 + KC < a.cin_p) issue_pf(c0 + KC);  // next pass's loads fly under this pass's MFMAs
    }

    // Two forms of the tap loop.  PIPE (kernels with enough matrix work per tap): explicit
    // one-tap-ahead operand prefetch with a prescribed issue order.  Otherwise (1x1 convs,
    // narrow layers) the plain loop, which the compiler schedules better on its own.
    if constexpr (PIPE) {
      // ---- taps: all KC8 8-channel steps of a tap are unrolled.  BOTH operands of tap t+1
      // (A rows from LDS, B columns from L2) are requested while the MFMAs of tap t issue,
      // spread evenly between them by the sched_group_barrier sequence below; left alone
      // the compiler sinks the weight loads to ~10 MFMAs before their use.
      int koff[KC8];
#pragma unroll
      for (int k8 = 0; k8 < KC8; ++k8)
        koff[k8] = min((c0 >> 3) + k8, nkc8_total - 1) * nb16_total * 64;
      float2 bn[KC8][NR], an[KC8][MR];
#pragma unroll
      for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bn[k8][nr] = wload(koff[k8], nr);
#pragma unroll
        for (int mr = 0; mr < MR; ++mr) an[k8][mr] = lds2[abase[mr] + k8 * 4];
      }
      for (int row = 0; row < G::KD * G::KH; ++row) {
        const int dz = row / G::KH, dy = row % G::KH;
        const int row_off = (dz * G::PY + dy) * G::PX * S2;
        const int nrow = min(row + 1, G::KD * G::KH - 1);
        const int next_row_off = ((nrow / G::KH) * G::PY + nrow % G::KH) * G::PX * S2;
#pragma unroll
        for (int dx = 0; dx < G::KW; ++dx) {
          const int tap = row * G::KW + dx;
          float2 bc[KC8][NR], ac[KC8][MR];
#pragma unroll
          for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) ac[k8][mr] = an[k8][mr];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) bc[k8][nr] = bn[k8][nr];
          }
          // operands of the next tap (the last tap re-requests itself: results unused)
          const int wn = min(tap + 1, G::NT - 1) * tap_stride;
          const int noff = (dx + 1 < G::KW) ? row_off + (dx + 1) * S2 : next_row_off;
#pragma unroll
          for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) bn[k8][nr] = wload(wn + koff[k8], nr);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr) an[k8][mr] = lds2[abase[mr] + noff + k8 * 4];
          }
#pragma unroll
          for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
              for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].x, bc[k8][nr].x, acc[mr][nr], 0, 0, 0);
#pragma unroll
            for (int mr = 0; mr < MR; ++mr)
#pragma unroll
              for (int nr = 0; nr < NR; ++nr)
                acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].y, bc[k8][nr].y, acc[mr][nr], 0, 0, 0);
          }
          // issue order of this tap: its loads (which feed the NEXT tap) spread evenly
          // between its MFMAs, weights first
          {
            constexpr int MPK = MR * NR * 2, LPK = NR + MR, EACH = MPK / LPK, REM = MPK - EACH * LPK;
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
              for (int j = 0; j < NR; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x008, EACH, 0);   // MFMA
              }
#pragma unroll
              for (int j = 0; j < MR; ++j) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, EACH, 0);
              }
              if (REM > 0) __builtin_amdgcn_sched_group_barrier(0x008, REM, 0);
            }
          }
        }
      }
    } else if constexpr (BREG) {
      // ---- taps with register-resident weights: only LDS reads and MFMAs in the loop
#pragma unroll
      for (int tp = 0; tp < G::NT; ++tp) {
        const int dz = tp / (G::KH * G::KW), dy = (tp / G::KW) % G::KH, dx = tp % G::KW;
        const int toff = ((dz * G::PY + dy) * G::PX + dx) * S2;
        float2 ac[KC8][MR];
#pragma unroll
        for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
          for (int mr = 0; mr < MR; ++mr) ac[k8][mr] = lds2[abase[mr] + toff + k8 * 4];
#pragma unroll
        for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
          for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
              acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].x, breg[tp][k8][nr].x, acc[mr][nr], 0, 0, 0);
#pragma unroll
          for (int mr = 0; mr < MR; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
              acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].y, breg[tp][k8][nr].y, acc[mr][nr], 0, 0, 0);
        }
      }
    } else {
      // ---- taps: all KC8 8-channel steps of a tap are unrolled; the weights of the
      // next tap are fetched (L2) while the current tap's MFMAs issue
      int koff[KC8];
#pragma unroll
      for (int k8 = 0; k8 < KC8; ++k8)
        koff[k8] = min((c0 >> 3) + k8, nkc8_total - 1) * nb16_total * 64;
      bool done = false;
      if constexpr (TAPPAIR) {
        if (a.paired == 2) {
          // Tap-paired weights ([tap pair][channel step][column block][lane][4]: x, y = even tap, z, w =
          // odd tap; the 28th tap is zero): one 16-byte load per (pair, step, column block) instead of
          // two 8-byte ones.  This layer issues 3 weight loads + 1 LDS read per 6 MFMAs, and a memory
          // instruction costs about one MFMA whatever its width (tools/mfma_valu_coissue.hip).
          // Per accumulator the order of the products is the unpaired loop's: bit-identical.
          done = true;
          constexpr int NP = (G::NT + 1) / 2;
          typedef unsigned int wu4 __attribute__((ext_vector_type(4)));
          auto wload4 = [&](int idx4, int nr) __attribute__((always_inline)) -> float4 {
            const wu4 v = __builtin_amdgcn_raw_buffer_load_b128(wrs, wvo[nr] * 2, idx4 * 16, 0);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
          };
          float4 b4[KC8][NR];
#pragma unroll
          for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) b4[k8][nr] = wload4(koff[k8], nr);
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            constexpr int KHW = G::KH * G::KW;
            const int t0 = 2 * p, t1 = min(2 * p + 1, G::NT - 1);       // (tap 27 re-reads tap 26: weights 0)
            const int o0 = (((t0 / KHW) * G::PY + (t0 / G::KW) % G::KH) * G::PX + t0 % G::KW) * S2;
            const int o1 = (((t1 / KHW) * G::PY + (t1 / G::KW) % G::KH) * G::PX + t1 % G::KW) * S2;
            float2 a0[KC8][MR], a1[KC8][MR];
            float4 bc[KC8][NR];
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
              for (int mr = 0; mr < MR; ++mr) {
                a0[k8][mr] = lds2[abase[mr] + o0 + k8 * 4];
                a1[k8][mr] = lds2[abase[mr] + o1 + k8 * 4];
              }
            const int wn = min(p + 1, NP - 1) * tap_stride;
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) {
                bc[k8][nr] = b4[k8][nr];
                b4[k8][nr] = wload4(wn + koff[k8], nr);
              }
#define JH_TP_STEP(AV, AC, BC)                                                                              \
  _Pragma("unroll") for (int mr = 0; mr < MR; ++mr) _Pragma("unroll") for (int nr = 0; nr < NR; ++nr)     \
    acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(AV[k8][mr].AC, bc[k8][nr].BC, acc[mr][nr], 0, 0, 0);
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8) { JH_TP_STEP(a0, x, x) JH_TP_STEP(a0, y, y) }
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8) { JH_TP_STEP(a1, x, z) JH_TP_STEP(a1, y, w) }
#undef JH_TP_STEP
          }
        }
      }
      if (!done) {
      float2 bn[KC8][NR];
#pragma unroll
      for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr) bn[k8][nr] = wload(koff[k8], nr);
      int tap = 0;
      for (int dz = 0; dz < G::KD; ++dz)
        for (int dy = 0; dy < G::KH; ++dy) {
          const int row_off = (dz * G::PY + dy) * G::PX * S2;
#pragma unroll
          for (int dx = 0; dx < G::KW; ++dx, ++tap) {
            float2 bc[KC8][NR], ac[KC8][MR];
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
              for (int mr = 0; mr < MR; ++mr) ac[k8][mr] = lds2[abase[mr] + row_off + dx * S2 + k8 * 4];
            const int wn = min(tap + 1, G::NT - 1) * tap_stride;
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8)
#pragma unroll
              for (int nr = 0; nr < NR; ++nr) {
                bc[k8][nr] = bn[k8][nr];
                bn[k8][nr] = wload(wn + koff[k8], nr);
              }
#pragma unroll
            for (int k8 = 0; k8 < KC8; ++k8) {
#pragma unroll
              for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                  acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].x, bc[k8][nr].x, acc[mr][nr], 0, 0, 0);
#pragma unroll
              for (int mr = 0; mr < MR; ++mr)
#pragma unroll
                for (int nr = 0; nr < NR; ++nr)
                  acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[k8][mr].y, bc[k8][nr].y, acc[mr][nr], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  // ---- epilogue: bias, store, InstanceNorm statistics
  __syncthreads();                       // LDS is reused for the cross-wave reduce
  EpilogueArgs e;
  e.y = a.y + (size_t)n * a.Dy * a.Hy * a.Wy * a.cout_p;
  e.bias = a.bias;
  e.stats = a.stats ? a.stats + (size_t)n * a.cout_p * kStatW : nullptr;
  e.Dout = a.Dout; e.Hout = a.Hout; e.Wout = a.Wout; e.Hy = a.Hy; e.Wy = a.Wy;
  e.cout_p = a.cout_p; e.cout_p16 = a.cout_p16; e.os = a.ostride; e.osz = a.ostride;
  e.offz = a.phase[ph].ooff[0]; e.offy = a.phase[ph].ooff[1]; e.offx = a.phase[ph].ooff[2];
  // tiles entirely inside the output (all of them when the extents are multiples of the tile)
  // take the epilogue without per-value bounds checks: with them every one of the 4 MR NR values
  // of a lane costs an exec-masked block of a dozen instructions
  if (oz0 + TZ <= a.Dout && oy0 + TY <= a.Hout && ox0 + TX <= a.Wout)
    conv_epilogue<MR, NR, TY, TX, 4, true>(acc, e, lds, nb0, oz0, oy0, ox0, tid);
  else
    conv_epilogue<MR, NR, TY, TX>(acc, e, lds, nb0, oz0, oy0, ox0, tid);
}

// Channels per LDS pass: the candidate (8..32) that wastes the fewest zero-padded
// channels; ties go to the larger pass (fewer staging rounds).
inline int pick_kc8(int cin_p, size_t (*lds_bytes)(int), size_t budget) {
  int best = 1;
  double best_cost = 1e30;
  for (int k = 1; k <= 4; ++k) {
    if (lds_bytes(k * 8) > budget && k > 1) continue;
    const int chunks = (cin_p + k * 8 - 1) / (k * 8);
    const double cost = (double)chunks * k * 8 / cin_p + 0.02 * chunks;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = k; }
  }
  return best;
}

// Launch one instantiation.  Returns false if this (nd,k,stride) is not the
// instantiation's, so callers can chain.
template <int ND, int K, int STRIDE, int TZ, int TY, int TX, int NRV, int KC8V>
int launch_conv_inst(const ConvArgs& b, dim3 grid, hipStream_t s) {
  using G = ConvGeom<ND, K, STRIDE, TZ, TY, TX>;
  size_t lds = G::lds_bytes(KC8V * 8);
  const size_t red = (size_t)4 * 4 * 16 * 2 * sizeof(double);       // conv_epilogue: fp64 partials
  if (lds < red) lds = red;
  lds += (size_t)b.nrm_floats * sizeof(float);
  JH_REQUIRE(lds <= 160 * 1024, "conv patch does not fit LDS");
  auto kern = conv_mfma_kernel<ND, K, STRIDE, TZ, TY, TX, NRV, KC8V>;
  static bool big_lds_enabled = false;
  if (lds > 64 * 1024 && !big_lds_enabled) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big_lds_enabled = true;
  }
  ConvArgs c = b;
  c.fgx = make_fastdiv(grid.x); c.fgy = make_fastdiv(grid.y);
  c.ftx = make_fastdiv((b.Wout + TX - 1) / TX); c.fty = make_fastdiv((b.Hout + TY - 1) / TY);
  JH_REQUIRE(b.nphase == 1 || b.nphase == 4 || b.nphase == 8, "phase count");
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, c);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

template <int ND, int K, int STRIDE, int TZ, int TY, int TX>
int launch_conv_geom(const ConvArgs& a, int nr, size_t lds_budget, hipStream_t s) {
  using G = ConvGeom<ND, K, STRIDE, TZ, TY, TX>;
  ConvArgs b = a;
  if (JH_ENV_KNOB("JH_CONV_LDS_KB") > 0) lds_budget = (size_t)JH_ENV_KNOB("JH_CONV_LDS_KB") * 1024;
  int kc8 = pick_kc8(b.cin_p, &G::lds_bytes, lds_budget);
  if (JH_ENV_KNOB("JH_CONV_KC8") > 0) kc8 = JH_ENV_KNOB("JH_CONV_KC8");
  b.kc = kc8 * 8;
  const int tiles = ((b.Dout + TZ - 1) / TZ) * ((b.Hout + TY - 1) / TY) * ((b.Wout + TX - 1) / TX);
  const int nb = b.cout_p16 / 16;
  // Launches that cannot fill the chip (the single-frame-set call: 24 tiles at 16 x 16 x 12 images): fewer column
  // blocks per workgroup = more workgroups, each staging the same operand but running a fraction of the MFMAs.
  // nr only partitions the output channels, so outputs and the per-channel fp32 partials of the fused statistics
  // are the same bits for any nr (test_time_batch_at_bench_scale_tile_nodes_bit_equal compares a 12-image call,
  // which takes this path, with a 192-image one, which does not).
  if (JH_ENV_KNOB("JH_CONV_NR_SPLIT") != 0)
    while (nr > 1 && (long)tiles * ((nb + nr - 1) / nr) * b.N * b.nphase <= 128) --nr;
  dim3 grid(tiles, (nb + nr - 1) / nr, b.N * b.nphase);
#define JH_CONV_CASE(NRV, KV) \
  if (nr == NRV && kc8 == KV) return launch_conv_inst<ND, K, STRIDE, TZ, TY, TX, NRV, KV>(b, grid, s);
#define JH_CONV_ROW(NRV) JH_CONV_CASE(NRV, 1) JH_CONV_CASE(NRV, 2) JH_CONV_CASE(NRV, 3) JH_CONV_CASE(NRV, 4)
  JH_CONV_ROW(1) JH_CONV_ROW(2) JH_CONV_ROW(3) JH_CONV_ROW(4)
#undef JH_CONV_ROW
#undef JH_CONV_CASE
  JH_REQUIRE(false, "bad (NR, KC8)");
}

// few-channel pointwise layers straight from registers (csrc/conv_pw_direct.hip)
bool conv_pw_direct_eligible(const ConvDesc& d, const ConvArgs& a);
bool conv_pw_direct_shape_ok(int cin_p, int cout_p16, int pixels);
int launch_conv_pw_direct(const ConvArgs& a, hipStream_t s);

// ConvTranspose2d k4 s2 p1 with the four parities in one workgroup (csrc/deconv4.hip); -1: not its layer
int launch_deconv4_fused(const ConvArgs& a, hipStream_t s);
bool deconv4_eligible(int cin_p, int cout_p16);

// per-translation-unit entry points (one .hip file per kernel family so the
// instantiations compile in parallel)
int conv_launch_2d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k1_flat(const ConvArgs& a, int nr, size_t budget, hipStream_t s);
int conv_launch_2d_k2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k3(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k3_w20(const ConvArgs& a, int nr, size_t budget, hipStream_t s);
int conv_launch_2d_k5(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_2d_k5_w40(const ConvArgs& a, int stride, int nr, size_t budget, hipStream_t s);
int conv_launch_2d_k3s2_w20(const ConvArgs& a, int nr, size_t budget, hipStream_t s);
int conv_launch_2d_big(const ConvArgs& a, int k, int stride, int nr, size_t budget, hipStream_t s);
int conv_launch_3d_k1(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_3d_k2s2(const ConvArgs& a, int nr, int small, size_t budget, hipStream_t s);
int conv_launch_3d_k3(const ConvArgs& a, int stride, int nr, int small, size_t budget, hipStream_t s);

}  // namespace jh
