// Fused BiFPN node, row-streaming form with PRODUCER / CONSUMER waves (round 5) for the 88-channel pyramid (the
// reference's default model size, jarvis/config/config.py:37,49) at bench scale.
//
// Same function, the same walk and the same arithmetic -- instruction for instruction per output value -- as
// bifpn_rows.hip (fusion of 2-3 inputs with InstanceNorm on load, activation, depthwise 3x3, pointwise 1x1 + bias,
// statistics of the output; jarvis/efficienttrack/model.py:309-353 + :223-232).  There ONE wave owns a strip and runs
// its phases back to back: at 88 channels the pointwise weights alone take 132 registers, so a SIMD holds one wave and
// nothing hides that wave's memory and LDS latencies: 11 600 cycles per row measured against 4 224 cycles of MFMAs +
// ~6 000 of issue for everything else (35 % of HBM at P3).  Here a strip belongs to a PAIR of waves:
//   * the producer wave fuses row y + 1 into the three-row ring, requests row y + 2 and runs the depthwise of row y
//     into one of two 16 x RC operand blocks in LDS (by row parity) -- vector ALU, LDS and memory instructions only,
//   * the consumer wave keeps the pointwise weights in registers for the whole strip and does nothing but the MFMAs of
//     the operand block the producer finished one row ago, bias, statistics and the 16-byte stores (+ the 2 x 2
//     max-pooled output: the even row's maxima wait in LDS for the odd row),
//   * ONE workgroup barrier per row (LDS traffic only: global loads and stores stay in flight) hands a block over.
// A workgroup is FOUR pairs (8 waves, one workgroup per CU): waves 0-3 are the consumers, 4-7 the producers, so every
// SIMD carries one of each (the wave-role placement the persistent Winograd kernel, conv3d_wino_pw.hip, relies on).
// The four pairs own four consecutive (strip, segment) items -- the four strips of a 64-pixel row at P3 -- and share
// nothing but the barrier.
// Round 6: the producer of the two-input nodes walks its items in two passes (16 quads x 4 pixel slots + 6 quads x 10)
// instead of one of 22 quads x 2 slots with 20 idle lanes: 7 fusion rounds instead of 9, 30 ring reads instead of 36.
// What it buys (measured, 384 images): P3 node 0.476 -> 0.430 ms, P4 0.207 -> 0.190, head 0.549 -> 0.447.  NOT the
// 2 x a concurrent matrix and vector pipe would give: a SIMD issues ONE instruction stream -- while the consumer streams
// MFMAs the producer of the same SIMD is starved (tools/mfma_valu_coissue.hip: times add, whatever the instruction
// type or priority), so a row still costs MFMA cycles + the issue cost of everything else; the pair only removes the
// exposed latencies of the one-wave form (each role runs while the other waits at the barrier or for memory).
#include <algorithm>
#include <type_traits>

#include "conv_mfma.h"
#include "bifpn_node.h"

namespace jh {

namespace {
constexpr int kPsPX = 18;                          // ring row width: 16 pixels + the depthwise halo
constexpr int kPsPairs = 4;
template <int RC>
struct PsGeo {
  static constexpr int RQ = RC / 4;                // channel quads
  static constexpr int NSUB = 64 / RQ;             // pixel slots of the producer wave
  static constexpr int NIT = (kPsPX + NSUB - 1) / NSUB;
  static constexpr int PPL = 16 / NSUB;            // depthwise output pixels per producer lane
  static constexpr int RSA = RC + 4;               // operand-block row stride (floats): 16 rows on distinct banks
  static constexpr int NCB = (RC + 15) / 16;
  static constexpr int K8 = RC / 8;
  static constexpr int ROWB = kPsPX * RC * 4;      // bytes per ring row
  static constexpr int OPB = 16 * RSA * 4;         // bytes per operand block
  static constexpr int POOLB = 8 * RC * 4;         // even row's horizontal maxima of the pooled output (8 pixels)
  static constexpr int PAIRB = 3 * ROWB + 2 * OPB + NCB * 16 * 4 + POOLB;
  static constexpr size_t lds_bytes() { return (size_t)kPsPairs * PAIRB; }
  static_assert(RC % 8 == 0 && NSUB >= 1 && 16 % NSUB == 0, "channel count of the producer / consumer node");
};
typedef float pf2 __attribute__((ext_vector_type(2)));
typedef float pf4 __attribute__((ext_vector_type(4)));
typedef unsigned pu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ps_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

template <int RC, int NIN, int M1, int M2, int ACT, bool POOL, int REPACK>
__global__ __launch_bounds__(kPsPairs * 128) void bifpn_rows_ps_kernel(const NodeArgs a, int seg_rows, int strips,
                                                                        int segs) {
  using GEO = PsGeo<RC>;
  constexpr int kRQ = GEO::RQ, kRSA = GEO::RSA;
  constexpr int NCB = GEO::NCB, K8 = GEO::K8, kRowB = GEO::ROWB, kOpB = GEO::OPB;
  constexpr int kAtOff = 3 * kRowB, kBiasOff = kAtOff + 2 * kOpB, kPoolOff = kBiasOff + NCB * 16 * 4;
  constexpr int kModes[3] = {FUSE_SAME, M1, M2};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int pair = wave & (kPsPairs - 1);
  unsigned char* smem = smem_all + pair * GEO::PAIRB;
  // this pair's (image, segment, strip): four consecutive items per workgroup
  const int item = blockIdx.x * kPsPairs + pair;
  const int per_img = strips * segs;
  const int n = item / per_img, rem = item - n * per_img;
  const int sx = rem % strips, seg = rem / strips;
  const int ox0 = sx * 16, y_begin = seg * seg_rows, y_end = min(a.H, y_begin + seg_rows);

  if (wave >= kPsPairs) {
    // ========================================================================================== producer wave
    __amdgpu_buffer_rsrc_t rs[NIN];
    int rowstep[NIN];                              // bytes per source row of input k
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
      const size_t plane = node_plane(kModes[k], a.H, a.W) * RC;
      rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in[k] + (size_t)n * plane), 0, (int)(plane * 4),
                                                0x00020000);
      rowstep[k] = (a.W >> sh) * RC * 4;
    }
    // The 18 x RQ (pixel, channel quad) items of a fused row and the 16 x RQ of a depthwise row are walked in PASSES of
    // NQ quads x 64 / NQ pixel slots.  One pass of all 22 quads leaves 20 lanes idle (2 slots: 9 fusion rounds, 8
    // depthwise pixels per lane); REPACK (round 6) walks quads 0..15 on 4 slots (5 rounds, 4 pixels) and quads 16..21 on
    // 10 slots (2 rounds; depthwise on 8 of them, 2 pixels): 7 rounds of loads / FMAs / SiLU instead of 9, 30 ring reads
    // instead of 36.  Every item is computed exactly as before: same bits.
    auto pass = [&](auto qb_c, auto nq_c) __attribute__((always_inline)) {
      constexpr int QB = decltype(qb_c)::value, NQ = decltype(nq_c)::value;
      constexpr int NSL = 64 / NQ;                               // pixel slots of the fusion
      constexpr int NITP = (kPsPX + NSL - 1) / NSL;              // fusion items per lane
      constexpr int DSL = NSL >= 8 ? 8 : (NSL >= 4 ? 4 : (NSL >= 2 ? 2 : 1));   // depthwise slots: a divisor of 16
      constexpr int PPLP = 16 / DSL;                             // depthwise output pixels per lane
      struct P {
        pf4 ak[NIN], bb, dwr[9], raw[NIN][NITP];
        int voff[NIN][NITP], fdst, dsrc, adst, sub;
        float msk[NITP];
        bool act, dact;
      } st;
      const int q = lane % NQ;
      st.sub = lane / NQ;                          // pixel slot (NSL and up: idle lanes)
      st.act = st.sub < NSL;
      st.dact = st.sub < DSL;
      const int c = (QB + q) * 4;
      // folded norm + fusion weights of this lane's channel quad: fused = sum_k x_k a_k + B
      st.bb = (pf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        float m4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {1.f, 1.f, 1.f, 1.f};
        if (a.st[k]) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const double* sp = a.st[k] + ((size_t)n * RC + c + j) * kStatW;
            const double mu = exact_read(sp) * (double)a.inv_cnt[k];
            double var = exact_read(sp + kLimbs) * (double)a.inv_cnt[k] - mu * mu;
            if (var < 0.0) var = 0.0;
            m4[j] = (float)mu;
            r4[j] = (float)(1.0 / sqrt(var + 1e-5));
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float ak1 = a.w[k] * r4[j];
          st.ak[k][j] = ak1;
          st.bb[j] += -m4[j] * ak1;
        }
      }
#pragma unroll
      for (int t = 0; t < 9; ++t) st.dwr[t] = *reinterpret_cast<const pf4*>(a.dw + t * RC + c);   // depthwise weights
      // the items of this lane in a fused row (pixel it * NSL + sub of the 18): load offsets inside a source row
      // (bit 31 = outside the image: the buffer load returns 0) and the 0 / 1 mask of the depthwise zero padding
#pragma unroll
      for (int it = 0; it < NITP; ++it) {
        const int px = it * NSL + st.sub, ix = ox0 - 1 + px;
        const bool ok = st.act && px < kPsPX && (unsigned)ix < (unsigned)a.W;
        st.msk[it] = ok ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < NIN; ++k) {
          const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
          st.voff[k][it] = ok ? ((ix >> sh) * RC + c) * 4 : (int)0x80000000;
        }
      }
      st.fdst = (st.sub * RC + c) * 4;                            // + it * NSL pixels (imm) + ring slot
      st.dsrc = (st.sub * PPLP * RC + c) * 4;                     // + ring slot + tap (imm)
      st.adst = kAtOff + (st.sub * PPLP * kRSA + c) * 4;          // + operand buffer + pixel (imm)
      return st;
    };
    // (see bifpn_rows.hip: an up-sampled input changes its source row only every 2nd output row; `all_c` is a
    //  compile-time flag because loads under a run-time branch make the compiler's in-order wait counts pessimistic)
    auto issue = [&](auto& st, int yf, auto all_c) __attribute__((always_inline)) {       // (yf inside the image)
      constexpr bool all = decltype(all_c)::value;
      constexpr int NITP = sizeof(st.msk) / sizeof(float);
      int srow[NIN];
#pragma unroll
      for (int k = 0; k < NIN; ++k) {
        const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
        srow[k] = (yf >> sh) * rowstep[k];
      }
#pragma unroll
      for (int it = 0; it < NITP; ++it)
#pragma unroll
        for (int k = 0; k < NIN; ++k)
          if (all || kModes[k] == FUSE_SAME)
            st.raw[k][it] =
                __builtin_bit_cast(pf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], st.voff[k][it], srow[k], 0));
    };
    // fused + activated row yf -> ring slot (yf + 1) % 3 (zeros outside the image: the depthwise padding)
    auto fuse = [&](auto& st, auto nsl_c, int yf, int slot) __attribute__((always_inline)) {
      constexpr int NSL = decltype(nsl_c)::value;
      constexpr int NITP = sizeof(st.msk) / sizeof(float);
      unsigned char* dst = smem + slot * kRowB + st.fdst;
      if ((unsigned)yf >= (unsigned)a.H) {                             // (uniform) padding row
#pragma unroll
        for (int it = 0; it < NITP; ++it)
          if (st.act && it * NSL + st.sub < kPsPX)
            *reinterpret_cast<pf4*>(dst + it * NSL * RC * 4) = (pf4){0.f, 0.f, 0.f, 0.f};
        return;
      }
#pragma unroll
      for (int it = 0; it < NITP; ++it) {
        pf4 v = st.bb;
#pragma unroll
        for (int k = 0; k < NIN; ++k) v = __builtin_elementwise_fma(st.raw[k][it], st.ak[k], v);
        if (ACT == ACT_SILU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = silu_fast(v[j]);
        } else if (ACT == ACT_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        // (pixels 0 and 17 of a strip at the image border live in the first and the last item)
        if (it == 0 || it == NITP - 1) v *= (pf4){st.msk[it], st.msk[it], st.msk[it], st.msk[it]};
        if (st.act && it * NSL + st.sub < kPsPX) *reinterpret_cast<pf4*>(dst + it * NSL * RC * 4) = v;
      }
    };
    // depthwise 3x3 of output row y (ring slot of row y - 1: s_top) -> operand block kBuf
    auto depthwise = [&](auto& st, auto ppl_c, int s_top, int kBuf) __attribute__((always_inline)) {
      constexpr int PPLP = decltype(ppl_c)::value;
      constexpr int PXN = PPLP >= 4 ? 4 : PPLP;
      if (st.dact) {
        int rs_ = s_top;                                               // ring slot of row y - 1 + dy
        const unsigned char* src[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          src[dy] = smem + rs_ * kRowB + st.dsrc;
          rs_ = rs_ == 2 ? 0 : rs_ + 1;
        }
#pragma unroll
        for (int part = 0; part < PPLP / PXN; ++part) {
          pf4 d[PXN];
#pragma unroll
          for (int i = 0; i < PXN; ++i) d[i] = (pf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            pf4 x[PXN + 2];
#pragma unroll
            for (int j = 0; j < PXN + 2; ++j)
              x[j] = *reinterpret_cast<const pf4*>(src[dy] + (part * PXN + j) * RC * 4);
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
              for (int i = 0; i < PXN; ++i) d[i] = __builtin_elementwise_fma(x[i + dx], st.dwr[dy * 3 + dx], d[i]);
          }
#pragma unroll
          for (int i = 0; i < PXN; ++i)
            *reinterpret_cast<pf4*>(smem + st.adst + kBuf + (part * PXN + i) * kRSA * 4) = d[i];
        }
      }
    };
    // REPACK 1: both phases in two passes; 2: the fusion in two passes, the depthwise in one (three inputs: the second
    // pass's depthwise weights do not fit the registers next to 21 loads in flight)
    constexpr int QA = REPACK ? 16 : kRQ;          // quads of the first pass; the second takes the rest
    constexpr int QT = REPACK ? kRQ - 16 : kRQ;    // (not used without REPACK)
    constexpr int kDSL1 = 64 / kRQ >= 8 ? 8 : (64 / kRQ >= 4 ? 4 : (64 / kRQ >= 2 ? 2 : 1));
    constexpr int NSLA = 64 / QA, NSLB = 64 / QT;
    constexpr int DSLA = NSLA >= 8 ? 8 : (NSLA >= 4 ? 4 : (NSLA >= 2 ? 2 : 1));
    constexpr int DSLB = NSLB >= 8 ? 8 : (NSLB >= 4 ? 4 : (NSLB >= 2 ? 2 : 1));
    auto pa = pass(std::integral_constant<int, 0>{}, std::integral_constant<int, QA>{});
    auto pb = pass(std::integral_constant<int, REPACK ? 16 : 0>{}, std::integral_constant<int, QT>{});
    auto pd = pass(std::integral_constant<int, 0>{}, std::integral_constant<int, kRQ>{});   // (REPACK 2: its depthwise)
    __builtin_amdgcn_s_waitcnt(0);                 // (the preamble's loads: see bifpn_rows.hip)
    int slot = (y_begin + 3) % 3;                  // slot of row yf = y_begin - 1: (yf + 1) % 3
    auto row = [&](int yf, auto next_all_c, auto out_c) __attribute__((always_inline)) {
      fuse(pa, std::integral_constant<int, NSLA>{}, yf, slot);
      if (REPACK) fuse(pb, std::integral_constant<int, NSLB>{}, yf, slot);
      if (yf + 1 <= y_end && yf + 1 < a.H) {
        issue(pa, yf + 1, next_all_c);
        if (REPACK) issue(pb, yf + 1, next_all_c);
      }
      const int s_top = slot == 0 ? 1 : (slot == 1 ? 2 : 0);          // slot of row y - 1 = (slot + 1) % 3
      slot = s_top;
      if (!decltype(out_c)::value) return;           // (the two rows above the segment's first output row)
      // operand block of this output row: by row parity (next_all_c is true in the half of the unrolled loop that
      // produces the EVEN output rows)
      constexpr int kBuf = decltype(next_all_c)::value ? 0 : kOpB;
      if (REPACK == 2) {
        depthwise(pd, std::integral_constant<int, 16 / kDSL1>{}, s_top, kBuf);
      } else {
        depthwise(pa, std::integral_constant<int, 16 / DSLA>{}, s_top, kBuf);
        if (REPACK) depthwise(pb, std::integral_constant<int, 16 / DSLB>{}, s_top, kBuf);
      }
      ps_barrier();                                  // this row's operand block is complete; the consumer has left
    };                                               // the other block (it read it before arriving here)
    if (y_begin - 1 >= 0) {
      issue(pa, y_begin - 1, std::true_type{});
      if (REPACK) issue(pb, y_begin - 1, std::true_type{});
    }
    row(y_begin - 1, std::true_type{}, std::false_type{});
    row(y_begin, std::false_type{}, std::false_type{});
    for (int yf = y_begin + 1; yf <= y_end; yf += 2) {
      row(yf, std::true_type{}, std::true_type{});
      if (yf + 1 <= y_end) row(yf + 1, std::false_type{}, std::true_type{});
    }
    return;
  }

  // ============================================================================================ consumer wave
  const int mrow = lane & 15, kq = lane >> 4;
  // pointwise weights of this lane for all channel steps x column blocks (registers for the whole strip); the MFMAs
  // take the weights as A and the pixels as B, so the accumulator of column block cb holds, for pixel lane & 15, the
  // four channels 16 cb + 4 (lane >> 4) .. + 3: one 16-byte store, no transposes
  pf2 bw[K8][NCB];
#pragma unroll
  for (int k8 = 0; k8 < K8; ++k8)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
      bw[k8][cb] = *reinterpret_cast<const pf2*>(a.pw + ((size_t)(k8 * NCB + cb) * 64 + lane) * 2);
  for (int i = lane; i < NCB * 16; i += 64)         // (bias: cout_p16 = 16 NCB floats, padded; read back by this wave only)
    reinterpret_cast<float*>(smem + kBiasOff)[i] = a.bias ? a.bias[i] : 0.f;
  const int ard = kAtOff + (mrow * kRSA) * 4 + kq * 8;            // + operand buffer + channel step (imm)
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      a.y + (size_t)n * a.H * a.W * a.cout_p, 0, (int)((size_t)a.H * a.W * a.cout_p * 4), 0x00020000);
  int yoff[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
    yoff[cb] = cb * 16 + kq * 4 < a.cout_p ? ((ox0 + mrow) * a.cout_p + cb * 16 + kq * 4) * 4 : (int)0x80000000;
  // POOL: the node also writes MaxPool2d(2, 2) of its raw output (max commutes with the monotone InstanceNorm map).
  // Horizontal max: the neighbour pixel is lane ^ 1 (one DPP move); vertical: the even row's maxima wait in LDS for
  // the odd row (this wave writes and reads them: no barrier; in registers they cost 24 and the kernel spills).
  // Lanes of even pixels own the pooled pixel.
  __amdgpu_buffer_rsrc_t rp = ry;
  int pbase = (int)0x80000000;
  if (POOL) {
    rp = __builtin_amdgcn_make_buffer_rsrc(a.y_pool + (size_t)n * (a.H >> 1) * (a.W >> 1) * a.cout_p, 0,
                                           (int)((size_t)(a.H >> 1) * (a.W >> 1) * a.cout_p * 4), 0x00020000);
    if (!(mrow & 1)) pbase = (((ox0 + mrow) >> 1) * a.cout_p + kq * 4) * 4;
  }
  const int pool_lds = kPoolOff + ((mrow >> 1) * RC + kq * 4) * 4;          // + 64 * cb (channels < RC only)
  pf4 s1[NCB], s2[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) { s1[cb] = (pf4){0.f, 0.f, 0.f, 0.f}; s2[cb] = s1[cb]; }
  __builtin_amdgcn_s_waitcnt(0);

  auto out_row = [&](int y, auto even_c) __attribute__((always_inline)) {
    constexpr bool even = decltype(even_c)::value;
    constexpr int kBuf = even ? 0 : kOpB;
    ps_barrier();                                    // the producer has finished this row's operand block
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k8 = 0; k8 < K8; ++k8) {
      const float2 xc = *reinterpret_cast<const float2*>(smem + ard + kBuf + k8 * 32);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[k8][cb][0], xc.x, acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[k8][cb][1], xc.y, acc[cb], 0, 0, 0);
    }
    const int yrow = y * a.W * a.cout_p * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const pf4 b4 = *reinterpret_cast<const pf4*>(smem + kBiasOff + (cb * 16 + kq * 4) * 4);
      const pf4 v = (pf4){acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]} + b4;
      s1[cb] += v;
      s2[cb] = __builtin_elementwise_fma(v, v, s2[cb]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu4, v), ry, yoff[cb] + yrow, 0, 0);
      if (POOL) {
        pf4 hm;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          hm[j] = fmaxf(v[j], __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[j]), 0xB1, 0xF, 0xF, true)));
        const bool own = !(mrow & 1) && cb * 16 + kq * 4 < RC;              // (cout_p == RC: checked by the launcher)
        if (even) {
          if (own) *reinterpret_cast<pf4*>(smem + pool_lds + cb * 64) = hm;
        } else {
          pf4 pv = hm;
          if (own) pv = *reinterpret_cast<const pf4*>(smem + pool_lds + cb * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) pv[j] = fmaxf(pv[j], hm[j]);
          // (cout_p == RC: the last column block holds 8 channels, its lanes kq >= 2 store nothing)
          const int po = cb * 16 + kq * 4 >= RC ? (int)0x80000000 : pbase + cb * 64;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(pu4, pv), rp,
                                                 po + (y >> 1) * (a.W >> 1) * a.cout_p * 4, 0, 0);
        }
      }
    }
  };
  // (segments start on even rows: the launcher)
  for (int y = y_begin; y < y_end; y += 2) {
    out_row(y, std::true_type{});
    if (y + 1 < y_end) out_row(y + 1, std::false_type{});
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
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float t1 = row_sum(s1[cb][i]), t2 = row_sum(s2[cb][i]);
        const int ch = cb * 16 + kq * 4 + i;
        if (mrow == 0 && ch < a.cout_p) stat_add(a.stats + ((size_t)n * a.cout_p + ch) * kStatW, t1, t2);
      }
  }
}

// Is this launch one for the producer / consumer form?  (Decided by the caller's segmentation: every pair of a
// workgroup must walk the same number of rows, and the items must fill whole workgroups.)
bool bifpn_rows_ps_ok(const NodeArgs& a, int seg_rows, int strips, int segs) {
  if (JH_ENV_KNOB("JH_NODE_PS") == 0) return false;
  if (a.Cp != 88 || a.cout_p != 88) return false;
  if (a.H % seg_rows != 0 || seg_rows % 2 != 0) return false;
  return ((long)strips * segs * a.N) % kPsPairs == 0;
}

int launch_bifpn_rows_ps(const NodeArgs& a, int seg_rows, int strips, int segs, hipStream_t s) {
  constexpr int RC = 88;
  const size_t lds = PsGeo<RC>::lds_bytes();
  const dim3 grid((unsigned)((long)strips * segs * a.N / kPsPairs)), block(kPsPairs * 128);
#define JH_PS(NIN, M1, M2, ACT, POOL)                                                                            \
  do {                                                                                                           \
    /* (two-input nodes, 384 images: P3 0.432 -> 0.397 ms, def320's 80-pixel level 0.350 -> 0.322.  Three inputs  \
       stay on the one pass: fully repacked the producer holds 21 loads of a row in flight -- 256 registers +   \
       72..128 B of scratch; with only the fusion repacked (2) the eight P4 nodes of medium take 2.418 against   \
       2.378 ms.  JH_NODE_PS_REPACK=0: off, 2 / 3: those two forms for three inputs) */                          \
    if (JH_ENV_KNOB("JH_NODE_PS_REPACK") == 0 || (NIN == 3 && JH_ENV_KNOB("JH_NODE_PS_REPACK") < 2))             \
      JH_PS_(NIN, M1, M2, ACT, POOL, 0);                                                                         \
    else if (NIN == 3 && JH_ENV_KNOB("JH_NODE_PS_REPACK") == 2) JH_PS_(NIN, M1, M2, ACT, POOL, 2);               \
    else JH_PS_(NIN, M1, M2, ACT, POOL, 1);                                                                      \
  } while (0)
#define JH_PS_(NIN, M1, M2, ACT, POOL, REPACK)                                                                   \
  do {                                                                                                           \
    auto kern = bifpn_rows_ps_kernel<RC, NIN, M1, M2, ACT, POOL, REPACK>;                                        \
    static bool big = false;                                                                                     \
    if (!big) {                                                                                                  \
      JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                      \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
      big = true;                                                                                                \
    }                                                                                                            \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a, seg_rows, strips, segs);                                    \
  } while (0)
  JH_REQUIRE(!a.y_pool || (a.n_in == 2 && a.act == ACT_SILU && a.cout_p == RC && a.H % 2 == 0),
             "producer / consumer node: pooled output");
  if (a.n_in == 2) {
    if (a.act == ACT_SILU && a.y_pool) JH_PS(2, FUSE_UP2, 0, ACT_SILU, true);
    else if (a.act == ACT_SILU) JH_PS(2, FUSE_UP2, 0, ACT_SILU, false);
    else if (a.act == ACT_NONE) JH_PS(2, FUSE_UP2, 0, ACT_NONE, false);
    else JH_REQUIRE(false, "producer / consumer node: activation");
  } else if (a.mode[1] == FUSE_SAME) {
    if (a.act == ACT_SILU) JH_PS(3, FUSE_SAME, FUSE_SAME, ACT_SILU, false);
    else JH_REQUIRE(false, "producer / consumer node: activation");
  } else {
    if (a.act == ACT_SILU) JH_PS(3, FUSE_UP2, FUSE_UP4, ACT_SILU, false);
    else if (a.act == ACT_NONE) JH_PS(3, FUSE_UP2, FUSE_UP4, ACT_NONE, false);
    else JH_REQUIRE(false, "producer / consumer node: activation");
  }
#undef JH_PS
#undef JH_PS_
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace jh
