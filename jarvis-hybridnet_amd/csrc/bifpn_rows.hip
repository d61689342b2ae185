// Fused BiFPN node, row-streaming form (round 3) for the high-resolution levels of the 56-channel pyramid.
//
// Same function as bifpn_node.hip (fusion of 2-3 inputs with InstanceNorm on load, activation, depthwise
// 3x3, pointwise 1x1 + bias, statistics of the output; jarvis/efficienttrack/model.py:309-353 + :223-232)
// for the top-down nodes and the head (inputs: same level, x2, x4 up-sampled -- no max-pooled input).
//
// The tile kernel owns 8 x 16 pixels per 512-thread workgroup: 44-48 KB of LDS, three workgroups per CU, six
// barriers per tile, 1.41 x halo overhead on the fusion work, and it is occupancy-bound (two workgroups per
// CU: 505 us, three: 272 us at P3).  Here ONE WAVE owns a strip 16 pixels wide and walks it row by row:
//   * a ring of three fused rows (18 pixels x 56 channels) in LDS is all the halo there is (1.125 x),
//   * per output row: depthwise from the three ring rows -> 16 x 56 operand block -> 56 fp32 MFMAs against the
//     pointwise weights, which stay in REGISTERS for the whole strip (56 VGPRs); the MFMAs take the weights as A
//     and the pixels as B, so a lane ends up with four consecutive channels of one pixel: bias, statistics
//     (per lane across the strip, reduced over the 16 pixel lanes once per strip) and one 16-byte channel-last
//     store per column block, no transposes,
//   * the raw inputs of row y+1 are requested as soon as row y is fused: a wave hides its own memory latency,
//   * a workgroup is a single wave: no barriers at all; 18 KB of LDS, eight waves per CU (248 VGPRs),
//   * everything that can be a per-lane constant is one (load offsets + scalar row offset, LDS addresses with
//     immediate offsets, padding masks): the row loop is 480 instructions, 56 of them MFMAs.
// The kernel is bound by instruction issue, not by memory (three inputs run as fast as two): per row and wave
// 56 MFMAs x 32 cycles + ~100 packed FMAs + 40 exp / rcp + ~40 LDS and 14 memory instructions ~ 5 000 cycles,
// which is what it measures (a SIMD issues one stream, times add).  P3 node of the bench
// (384 images 64 x 64): 273.6 -> 194 us (219 us when it also writes its 2 x 2 max-pooled output, which lets the P4
// bottom-up node run in this form with three same-level inputs: 151 -> 99 us), head 304 -> 221 us.
#include <algorithm>
#include <type_traits>

#include "conv_mfma.h"
#include "bifpn_node.h"

namespace jh {

namespace {
constexpr int kRPX = 18;                           // ring row width: 16 pixels + the depthwise halo
// Geometry of the kernel for an RC-channel pyramid (56: the small model; 88: medium, round 4).  A lane owns one
// channel quad and one of NSUB pixel slots: 14 quads x 4 slots = 56 active lanes at 56 channels, 22 x 2 = 44 at 88.
template <int RC>
struct RowGeo {
  static constexpr int RQ = RC / 4;                // channel quads
  static constexpr int NSUB = 64 / RQ;             // pixel slots per wave
  static constexpr int NIT = (kRPX + NSUB - 1) / NSUB;   // items (pixels) of a fused row per lane
  static constexpr int PPL = 16 / NSUB;            // depthwise output pixels per lane
  static constexpr int RSA = RC + 4;               // operand-block row stride (floats): 16 rows on distinct banks
  static constexpr int AFLOATS = 16 * RSA;
  static constexpr int NCB = (RC + 15) / 16;       // 16-channel column blocks of the pointwise output
  static constexpr int K8 = RC / 8;
  // registers: 2 K8 NCB for the pointwise weights alone (56 at 56 channels, 132 at 88): two waves per SIMD fit
  // at 56 channels, one at 88 (the kernel is bound by instruction issue, not by latency: DESIGN section 3)
  // (56 channels with ONE wave per SIMD and the freed registers spent on depthwise weights in registers / four
  //  pixels per pass in every variant: 257 instead of 219 us at P3 -- the second wave's latency hiding is worth more)
  static constexpr int WAVES = RC <= 56 ? 2 : 1;
  static constexpr size_t lds_bytes() { return (size_t)(3 * kRPX * RC + AFLOATS + 9 * RC + NCB * 16) * sizeof(float); }
  static_assert(RC % 8 == 0 && NSUB >= 1 && 16 % NSUB == 0, "channel count of the row-streaming node");
};
typedef float rf2 __attribute__((ext_vector_type(2)));
typedef float rf4 __attribute__((ext_vector_type(4)));
typedef unsigned ru4 __attribute__((ext_vector_type(4)));
}  // namespace

template <int RC, int NIN, int M1, int M2, int ACT, bool POOL = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RowGeo<RC>::WAVES, RowGeo<RC>::WAVES)))
void bifpn_rows_kernel(const NodeArgs a, int seg_rows, int strips) {
  static_assert(!POOL || NIN == 2, "the pooled row is parked in the (unused) depthwise-weight region of the LDS");
  using GEO = RowGeo<RC>;
  constexpr int kRC = RC, kRQ = GEO::RQ, NSUB = GEO::NSUB, NIT = GEO::NIT, PPL = GEO::PPL, kRSA = GEO::RSA;
  constexpr int NCB = GEO::NCB, K8 = GEO::K8, kAFloats = GEO::AFLOATS;
  constexpr int kModes[3] = {FUSE_SAME, M1, M2};
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // byte offsets inside the wave's LDS: ring of three fused rows, operand block, depthwise weights
  constexpr int kRowB = kRPX * kRC * 4;           // 4032 at 56 channels
  constexpr int kAtOff = 3 * kRowB, kDwOff = kAtOff + kAFloats * 4, kBiasOff = kDwOff + 9 * kRC * 4;
  const int lane = threadIdx.x;
  const int q = lane % kRQ, sub = lane / kRQ;     // channel quad, pixel slot (0..NSUB-1; NSUB = the idle lanes)
  const bool act_lane = sub < NSUB;
  const int c = q * 4;
  const int sx = blockIdx.x % strips, seg = blockIdx.x / strips, n = blockIdx.y;
  const int ox0 = sx * 16, y_begin = seg * seg_rows, y_end = min(a.H, y_begin + seg_rows);
  const int mrow = lane & 15, kq = lane >> 4;

  // ---- per-lane constants --------------------------------------------------------------------------------
  // folded norm + fusion weights of this lane's channel quad: fused = sum_k x_k a_k + B
  rf4 ak[NIN], bb = (rf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < NIN; ++k) {
    float m4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {1.f, 1.f, 1.f, 1.f};
    if (a.st[k]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double* st = a.st[k] + ((size_t)n * kRC + c + j) * kStatW;
        const double mu = exact_read(st) * (double)a.inv_cnt[k];
        double var = exact_read(st + kLimbs) * (double)a.inv_cnt[k] - mu * mu;
        if (var < 0.0) var = 0.0;
        m4[j] = (float)mu;
        r4[j] = (float)(1.0 / sqrt(var + 1e-5));
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float ak1 = a.w[k] * r4[j];
      ak[k][j] = ak1;
      bb[j] += -m4[j] * ak1;
    }
  }
  // depthwise weights of this lane's channel quad: registers when the budget allows (two inputs), else LDS --
  // every LDS instruction costs the SIMD about as much issue time as an MFMA
  constexpr bool kDwReg = NIN == 2;
  rf4 dwr[kDwReg ? 9 : 1];
  if (kDwReg) {
#pragma unroll
    for (int t = 0; t < 9; ++t) dwr[t] = *reinterpret_cast<const rf4*>(a.dw + t * kRC + (act_lane ? c : 0));
  } else {
    for (int i = lane; i < 9 * kRC; i += 64) reinterpret_cast<float*>(smem + kDwOff)[i] = a.dw[i];
  }
  // pointwise weights of this lane for all 7 channel steps x 4 column blocks (registers for the whole strip);
  // the MFMAs run with the operands swapped (weights as A, pixels as B), so the accumulator of column block cb
  // holds, for pixel lane & 15, the four channels 16 cb + 4 (lane >> 4) .. + 3: one 16-byte store, no transposes
  rf2 bw[K8][NCB];
#pragma unroll
  for (int k8 = 0; k8 < K8; ++k8)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
      bw[k8][cb] = *reinterpret_cast<const rf2*>(a.pw + ((size_t)(k8 * NCB + cb) * 64 + lane) * 2);
  for (int i = lane; i < NCB * 16; i += 64)       // (bias: cout_p16 = 16 NCB floats, padded)
    reinterpret_cast<float*>(smem + kBiasOff)[i] = a.bias ? a.bias[i] : 0.f;

  __amdgpu_buffer_rsrc_t rs[NIN];
  int rowstep[NIN];                                // bytes per source row of input k
#pragma unroll
  for (int k = 0; k < NIN; ++k) {
    const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
    const size_t plane = node_plane(kModes[k], a.H, a.W) * kRC;
    rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in[k] + (size_t)n * plane), 0, (int)(plane * 4),
                                              0x00020000);
    rowstep[k] = (a.W >> sh) * kRC * 4;
  }
  // the NIT items of this lane in a fused row (pixel it * NSUB + sub of the 18): load offsets inside a source row
  // (bit 31 = outside the image: the buffer load returns 0) and the 0 / 1 mask of the depthwise zero padding
  int voff[NIN][NIT];
  float msk[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int px = it * NSUB + sub, ix = ox0 - 1 + px;
    const bool ok = act_lane && px < kRPX && (unsigned)ix < (unsigned)a.W;
    msk[it] = ok ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
      voff[k][it] = ok ? ((ix >> sh) * kRC + c) * 4 : (int)0x80000000;
    }
  }
  const int fdst = (sub * kRC + c) * 4;                         // + it * NSUB pixels (imm) + ring slot
  const int dsrc = (sub * PPL * kRC + c) * 4;                   // + ring slot + tap (imm)
  const int adst = kAtOff + (sub * PPL * kRSA + c) * 4;         // + pixel (imm)
  const int ard = kAtOff + (mrow * kRSA) * 4 + kq * 8;          // + channel step (imm)
  // Output through a buffer descriptor of image n: lanes whose four channels lie past cout_p (the padding of the
  // last column block) store with bit 31 set in the offset -- dropped by the range check -- instead of under a
  // branch: a store the compiler cannot count makes every later wait for loads a wait for ALL memory operations.
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      a.y + (size_t)n * a.H * a.W * a.cout_p, 0, (int)((size_t)a.H * a.W * a.cout_p * 4), 0x00020000);
  int yoff[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb)
    yoff[cb] = cb * 16 + kq * 4 < a.cout_p ? ((ox0 + mrow) * a.cout_p + cb * 16 + kq * 4) * 4 : (int)0x80000000;

  // (one raw row in flight per wave: a second register set -- loads of row yf + 2 behind the fusion of row yf -- was
  //  slower, 274 us, the kernel is bound by instruction issue and the registers it costs forced re-loads)
  // POOL: the node also writes MaxPool2d(2, 2) of its raw output (the bottom-up node of the next level then reads a
  // same-resolution tensor instead of four pixels per tap; max commutes with the monotone InstanceNorm map, the
  // pooled tensor keeps this node's statistics).  Horizontal max: the neighbour pixel is lane ^ 1 (one DPP
  // move); vertical: the even row's maxima wait in LDS (the depthwise-weight region, free with two inputs) for
  // the odd row.  Lanes of even pixels own the pooled pixel.
  __amdgpu_buffer_rsrc_t rp = ry;
  int pbase = (int)0x80000000;                    // offset of this lane's pooled pixel (bit 31: not an owner)
  if (POOL) {
    rp = __builtin_amdgcn_make_buffer_rsrc(a.y_pool + (size_t)n * (a.H >> 1) * (a.W >> 1) * a.cout_p, 0,
                                           (int)((size_t)(a.H >> 1) * (a.W >> 1) * a.cout_p * 4), 0x00020000);
    if (!(mrow & 1)) pbase = (((ox0 + mrow) >> 1) * a.cout_p + kq * 4) * 4;
  }
  const int pool_lds = kDwOff + ((mrow >> 1) * kRC + kq * 4) * 4;          // + 64 * cb (channels < RC only)
  rf4 raw[NIN][NIT];
  // An up-sampled input changes its source row only every 2nd output row: when the requested row is ODD its
  // registers are simply kept (5 load instructions less).  `all_c` is a compile-time flag -- the row loop is
  // unrolled by two -- because a load under a run-time branch makes the compiler's in-order wait counts
  // pessimistic (the fusion then also waits for the previous row's stores: 216 -> 326 us).
  auto issue = [&](int yf, auto all_c) __attribute__((always_inline)) {       // (yf inside the image)
    constexpr bool all = decltype(all_c)::value;
    int srow[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
      srow[k] = (yf >> sh) * rowstep[k];
    }
    // (item-major: the fusion consumes item 0 of every input first, the in-order counter then releases it
    //  after the first NIN loads instead of after most of the row)
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
      for (int k = 0; k < NIN; ++k)
        if (all || kModes[k] == FUSE_SAME)
          raw[k][it] =
              __builtin_bit_cast(rf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], voff[k][it], srow[k], 0));
  };
  // fused + activated row yf -> ring slot (yf + 1) % 3 (zeros outside the image: the depthwise padding)
  auto fuse = [&](int yf, int slot) __attribute__((always_inline)) {
    unsigned char* dst = smem + slot * kRowB + fdst;
    if ((unsigned)yf >= (unsigned)a.H) {                             // (uniform) padding row
#pragma unroll
      for (int it = 0; it < NIT; ++it)
        if (act_lane && it * NSUB + sub < kRPX)
          *reinterpret_cast<rf4*>(dst + it * NSUB * kRC * 4) = (rf4){0.f, 0.f, 0.f, 0.f};
      return;
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      rf4 v = bb;
#pragma unroll
      for (int k = 0; k < NIN; ++k) v = __builtin_elementwise_fma(raw[k][it], ak[k], v);
      if (ACT == ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_fast(v[j]);
      } else if (ACT == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      // (the items in between are always inside the image: W >= 32, only the strip's first and last columns --
      //  pixels 0 and 17, i.e. the first and the last item -- can be padding)
      if (it == 0 || it == NIT - 1) v *= (rf4){msk[it], msk[it], msk[it], msk[it]};
      if (act_lane && it * NSUB + sub < kRPX) *reinterpret_cast<rf4*>(dst + it * NSUB * kRC * 4) = v;
    }
  };

  rf4 s1[NCB], s2[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; ++cb) { s1[cb] = (rf4){0.f, 0.f, 0.f, 0.f}; s2[cb] = s1[cb]; }

  // (everything the preamble loaded is waited for HERE: left pending, the compiler puts the wait at the first use
  //  inside the loop, where the in-order counter then also waits for the rows just requested)
  __builtin_amdgcn_s_waitcnt(0);
  int slot = (y_begin + 3) % 3;                    // slot of row yf = y_begin - 1: (yf + 1) % 3

  auto row = [&](int yf, auto next_all_c, auto out_c) __attribute__((always_inline)) {
    fuse(yf, slot);
    // (the raw registers are free again: request the next row)
    if (yf + 1 <= y_end && yf + 1 < a.H) issue(yf + 1, next_all_c);
    const int y = yf - 1;                               // output row whose three ring rows are now complete
    const int s_top = slot == 0 ? 1 : (slot == 1 ? 2 : 0);          // slot of row y - 1 = (slot + 1) % 3
    slot = s_top;                                 // (the next fused row replaces row y - 1 after this iteration)
    if (!decltype(out_c)::value) return;             // (the two rows above the segment's first output row)
    // ---- depthwise 3x3: lane = (strip of PPL pixels, channel quad) -> operand block --------------------
    if (act_lane) {
      int rs_ = s_top;                                               // ring slot of row y - 1 + dy
      const unsigned char* src[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        src[dy] = smem + rs_ * kRowB + dsrc;
        rs_ = rs_ == 2 ? 0 : rs_ + 1;
      }
      // (two inputs: all four pixels at once, 18 row reads; three inputs: two pixels at a time, 24 reads but half
      //  the live registers)
      constexpr int PXN = (kDwReg && !POOL) ? 4 : 2;        // (the pooled-output variant has no registers to spare)
#pragma unroll
      for (int part = 0; part < PPL / PXN; ++part) {
        rf4 d[PXN];
#pragma unroll
        for (int i = 0; i < PXN; ++i) d[i] = (rf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          rf4 x[PXN + 2];
#pragma unroll
          for (int j = 0; j < PXN + 2; ++j)
            x[j] = *reinterpret_cast<const rf4*>(src[dy] + (part * PXN + j) * kRC * 4);
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const rf4 wv = kDwReg ? dwr[kDwReg ? dy * 3 + dx : 0]
                                  : *reinterpret_cast<const rf4*>(smem + kDwOff + ((dy * 3 + dx) * kRC + c) * 4);
#pragma unroll
            for (int i = 0; i < PXN; ++i) d[i] = __builtin_elementwise_fma(x[i + dx], wv, d[i]);
          }
        }
#pragma unroll
        for (int i = 0; i < PXN; ++i) *reinterpret_cast<rf4*>(smem + adst + (part * PXN + i) * kRSA * 4) = d[i];
      }
    }
    // ---- pointwise 1x1: 64 output channels x 16 pixels x 56 channels on the matrix cores -------------------
    f32x4 acc[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k8 = 0; k8 < K8; ++k8) {
      const float2 xc = *reinterpret_cast<const float2*>(smem + ard + k8 * 32);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[k8][cb][0], xc.x, acc[cb], 0, 0, 0);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[k8][cb][1], xc.y, acc[cb], 0, 0, 0);
    }
    // ---- bias, statistics (in registers across the strip), one 16-byte store per column block ---------------
    const int yrow = y * a.W * a.cout_p * 4;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const rf4 b4 = *reinterpret_cast<const rf4*>(smem + kBiasOff + (cb * 16 + kq * 4) * 4);
      const rf4 v = (rf4){acc[cb][0], acc[cb][1], acc[cb][2], acc[cb][3]} + b4;
      s1[cb] += v;
      s2[cb] = __builtin_elementwise_fma(v, v, s2[cb]);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ru4, v), ry, yoff[cb] + yrow, 0, 0);
      if (POOL) {
        rf4 hm;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          hm[j] = fmaxf(v[j], __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[j]), 0xB1, 0xF, 0xF, true)));
        const bool own = !(mrow & 1) && cb * 16 + kq * 4 < kRC;             // (cout_p == RC: checked by the launcher)
        if (decltype(next_all_c)::value) {                                  // even output row: park
          if (own) *reinterpret_cast<rf4*>(smem + pool_lds + cb * 64) = hm;
        } else {                                                            // odd row: combine, store
          rf4 pv = hm;
          if (own) pv = *reinterpret_cast<const rf4*>(smem + pool_lds + cb * 64);
#pragma unroll
          for (int j = 0; j < 4; ++j) pv[j] = fmaxf(pv[j], hm[j]);
          // (cout_p == RC: the last column block holds 8 channels, its lanes kq >= 2 store nothing)
          const int po = cb * 16 + kq * 4 >= kRC ? (int)0x80000000 : pbase + cb * 64;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ru4, pv), rp,
                                                 po + (y >> 1) * (a.W >> 1) * a.cout_p * 4, 0, 0);
        }
      }
    }
  };
  // (segments start on even rows -- launch_bifpn_rows -- so row y_begin - 1 is odd and the rows requested from the
  //  first half of the unrolled body are even: all inputs; from the second half odd: same-level inputs only)
  if (y_begin - 1 >= 0) issue(y_begin - 1, std::true_type{});
  // The first two fused rows produce no output yet: peeled, so that the loop body has no path without stores --
  // with one, the compiler's in-order wait counts for the loads must assume the stores were never issued and
  // every wait for a load becomes a wait for (nearly) all of them.
  row(y_begin - 1, std::true_type{}, std::false_type{});
  row(y_begin, std::false_type{}, std::false_type{});
  for (int yf = y_begin + 1; yf <= y_end; yf += 2) {
    row(yf, std::true_type{}, std::true_type{});
    if (yf + 1 <= y_end) row(yf + 1, std::false_type{}, std::true_type{});
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

// Which nodes take the row-streaming form: the 56-channel (small model) or 88-channel (medium) pyramid with as many
// output column blocks as the input has, no max-pooled input, level at least 32 pixels wide and a multiple of 16
// (JH_NODE_ROWS=0: never; JH_NODE_ROWS88=0: not at 88 channels).
bool bifpn_rows_wg_shape_ok(const NodeArgs& a);                 // csrc/bifpn_rows_wg.hip
int launch_bifpn_rows_wg(const NodeArgs& a, hipStream_t s);
bool bifpn_rows_ps_ok(const NodeArgs& a, int seg_rows, int strips, int segs);      // csrc/bifpn_rows_ps.hip
int launch_bifpn_rows_ps(const NodeArgs& a, int seg_rows, int strips, int segs, hipStream_t s);

// 88 channels, time-batch class >= 8, a level whose width is no multiple of 16 (or below 16): the one-wave / pair forms
// need whole 16-pixel strips, the tile kernel's path for more than 64 channels is slow whatever the level -- the
// workgroup form masks the last strip's pixels past the row end.  The reference's DEFAULT geometry (320-pixel images:
// levels 80 / 40 / 20 / 10 / 5) has four of its five levels here.  A function of the node only.  JH_NODE_ROWS88_RAGGED=0: off.
bool bifpn_rows_ragged88(const NodeArgs& a) {
  return a.Cp == 88 && a.rows == 1 && (a.W % 16 != 0 || a.W < 16) && JH_ENV_KNOB("JH_NODE_ROWS88_RAGGED") != 0;
}

bool bifpn_rows_eligible(const NodeArgs& a) {
  if (JH_ENV_KNOB("JH_NODE_ROWS") == 0 || a.rows == 0) return false;
  // (160 channels, the large model: the workgroup form, csrc/bifpn_rows_wg.hip; JH_NODE_ROWS160=0: tile form.
  //  a.rows == 2 -- the wide pyramids at time batches below 8 -- is the same form with 8-row segments, also at 88
  //  channels, where one wave per strip is the faster form at bench scale but far too slow per row for a few images)
  if (a.rows == 2 && JH_ENV_KNOB("JH_NODE_ROWS_LAT") == 0) return false;
  const bool wg = (a.Cp == 160 || (a.Cp == 88 && (a.rows == 2 || bifpn_rows_ragged88(a)))) &&
                  JH_ENV_KNOB("JH_NODE_ROWS160") != 0 && bifpn_rows_wg_shape_ok(a);
  if (a.rows == 2 && !wg) return false;
  if (a.Cp != 56 && !(a.Cp == 88 && JH_ENV_KNOB("JH_NODE_ROWS88") != 0) && !wg) return false;
  // (88 / 160 channels: also the 16-pixel-wide level -- one strip per image -- because the tile kernel's path for more
  //  than 64 channels is slow there, 0.10 ms per launch against 0.05; at 56 channels the tile kernel wins below 32 pixels)
  if (a.cout_p16 != (a.Cp + 15) / 16 * 16 || a.mode[0] != FUSE_SAME) return false;
  if (wg) {
    // the workgroup form masks the pixels past the row end of its last strip: every level down to 2 x 2 (the tile
    // kernel's path for more than 64 channels costs 35-130 us per launch whatever the level)
    if (a.W < 2 || a.H < 2) return false;
  } else if (a.W % 16 != 0 || a.W < (a.Cp >= 88 ? 16 : 32) || a.H < 16) {
    return false;
  }
  if (a.rows < 0) {
    // (one wave per workgroup walking >= 10 rows: below ~2048 strips the chip is not filled and the tile form wins)
    const int min_wg = JH_ENV_KNOB("JH_NODE_ROWS_MINWG") > 0 ? JH_ENV_KNOB("JH_NODE_ROWS_MINWG") : 2048;
    if (((a.W + 15) / 16) * ((a.H + 7) / 8) * a.N < min_wg) return false;
  }
  const bool up2_ok = a.W % 2 == 0 && a.H % 2 == 0, up4_ok = a.W % 4 == 0 && a.H % 4 == 0;
  if (a.n_in == 2) return (a.mode[1] == FUSE_UP2 && up2_ok) || (wg && a.mode[1] == FUSE_SAME);
  return a.n_in == 3 && ((a.mode[1] == FUSE_UP2 && a.mode[2] == FUSE_UP4 && up4_ok) ||
                         (a.mode[1] == FUSE_SAME && a.mode[2] == FUSE_SAME));
}

template <int RC>
static int launch_rows_rc(const NodeArgs& a, hipStream_t s) {
  const int strips = a.W / 16;
  // Rows per workgroup: a function of the node (image size, number of inputs) ONLY -- the float partial sums of the
  // statistics are taken per strip segment, so the segmentation must not depend on how many images a launch
  // carries (bit-equal results for any number of cameras per rank).  Measured at 384 images (P3 two-input node:
  // 8 rows 240 us, 16: 205, 32: 191; P4: 8: 74, 16: 66, 32: 71; three-input head: 16: 221, 32: 223; three
  // same-level inputs at P4: 8: 116, 16: 99): half the image height, 16 / 8 rows for the head.
  int seg_rows = JH_ENV_KNOB("JH_NODE_SEG") > 0 ? JH_ENV_KNOB("JH_NODE_SEG")
                 : (a.n_in == 2 || a.mode[1] == FUSE_SAME ? std::max(8, a.H / 2) : (a.H >= 64 ? 16 : 8));
  // (88 channels, producer / consumer pairs: four items per 512-thread workgroup, so fewer and longer segments fill the
  //  chip better -- measured at 384 images: 32-row segments at the 32-pixel levels 0.155 -> 0.140 / 0.188 -> 0.167 ms,
  //  the three-input head at 64 pixels 0.445 -> 0.397 with 32 rows instead of 16; the 16-pixel levels keep 8 rows: 0.051
  //  against 0.078 with one segment per image)
  if (RC == 88 && JH_ENV_KNOB("JH_NODE_SEG") <= 0 && a.H >= 32) {
    seg_rows = 32;
    // (the pairs of a workgroup must walk the same number of rows: a level that is no multiple of 32 -- 80 x 80 in the
    //  reference's DEFAULT 320-pixel geometry -- takes its largest even divisor up to 40 that leaves two segments; with
    //  32 it fell back to the one-wave form at 1.4x the time per pixel)
    if (a.H % 32 != 0)
      for (int d = std::min(40, a.H / 2) & ~1; d >= 8; d -= 2)
        if (a.H % d == 0) { seg_rows = d; break; }
  }
  seg_rows = (seg_rows + 1) & ~1;                // (even: the kernel's row loop is unrolled by two on row parity)
  if (seg_rows > a.H) seg_rows = a.H;
  const int segs = (a.H + seg_rows - 1) / seg_rows;
  // 88 channels at bench scale: producer / consumer wave pairs (csrc/bifpn_rows_ps.hip) -- the same arithmetic per
  // output value and the same segmentation, so the results are bit-equal to the kernel below
  if (RC == 88 && bifpn_rows_ps_ok(a, seg_rows, strips, segs)) return launch_bifpn_rows_ps(a, seg_rows, strips, segs, s);
  const size_t lds = RowGeo<RC>::lds_bytes();
  const dim3 grid(strips * segs, a.N);
#define JH_ROWS(NIN, M1, M2, ACT, POOL) \
  hipLaunchKernelGGL((bifpn_rows_kernel<RC, NIN, M1, M2, ACT, POOL>), grid, dim3(64), lds, s, a, seg_rows, strips)
  JH_REQUIRE(!a.y_pool || (a.n_in == 2 && a.act == ACT_SILU && a.cout_p == RC && a.H % 2 == 0),
             "row-streaming node: pooled output");
  if (a.n_in == 2) {
    if (a.act == ACT_SILU && a.y_pool) JH_ROWS(2, FUSE_UP2, 0, ACT_SILU, true);
    else if (a.act == ACT_SILU) JH_ROWS(2, FUSE_UP2, 0, ACT_SILU, false);
    else if (a.act == ACT_NONE) JH_ROWS(2, FUSE_UP2, 0, ACT_NONE, false);
    else JH_REQUIRE(false, "row-streaming node: activation");
  } else if (a.mode[1] == FUSE_SAME) {
    if (a.act == ACT_SILU) JH_ROWS(3, FUSE_SAME, FUSE_SAME, ACT_SILU, false);
    else JH_REQUIRE(false, "row-streaming node: activation");
  } else {
    if (a.act == ACT_SILU) JH_ROWS(3, FUSE_UP2, FUSE_UP4, ACT_SILU, false);
    else if (a.act == ACT_NONE) JH_ROWS(3, FUSE_UP2, FUSE_UP4, ACT_NONE, false);
    else JH_REQUIRE(false, "row-streaming node: activation");
  }
#undef JH_ROWS
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_bifpn_rows(const NodeArgs& a, hipStream_t s) {
  if (a.Cp == 160 || a.rows == 2 || bifpn_rows_ragged88(a)) return launch_bifpn_rows_wg(a, s);
  if (a.Cp == 88) return launch_rows_rc<88>(a, s);
  return launch_rows_rc<56>(a, s);
}

}  // namespace jh
