// Fused BiFPN node, row-streaming form for the WIDE pyramids (160 channels: the large model; 88: medium; round 4).
//
// Same function and the same walk as bifpn_rows.hip (fusion of 2-3 inputs with InstanceNorm on load, activation,
// depthwise 3x3, pointwise 1x1 + bias, statistics of the output; jarvis/efficienttrack/model.py:309-353 +
// :223-232), but a strip 16 pixels wide belongs to a WORKGROUP of RC / 16 waves instead of one wave: at 160
// channels the pointwise weights (160 x 160 fp32 = 102 KB) fit neither one wave's registers (400 per lane) nor,
// next to anything else, the LDS -- the tile kernel re-streams them per tile and runs at 8-15 % of HBM there
// (3.6 ms per P3 node of the large model at 384 images).  Here
//   * wave w owns the 16 channels [16 w, 16 w + 16) on BOTH sides of the depthwise: it fuses them into its own
//     channel slice of the row ring in LDS (no synchronisation: the depthwise reads per channel), and it owns
//     output column block w of the pointwise: its 16 x RC weight slice stays in RC / 4 = 40 registers for the strip,
//   * the 16 x RC operand block (depthwise output of one row) is the only thing the waves share: double-buffered by
//     row parity, ONE workgroup barrier per row (LDS traffic only: the next row's global loads stay in flight),
//   * per row and wave: one fused item per lane (16 pixels x 4 quads; pixels 16, 17 of the 18: the helper waves,
//     see the kernel), one depthwise pixel per lane
//     (9 ring reads, 9 packed FMAs), RC / 8 operand reads + RC / 4 MFMAs (two accumulators: even / odd channel
//     pairs), bias + statistics + one 16-byte store; the 2 x 2 max-pooled output is carried in registers between the
//     two halves of the unrolled row loop, so every variant can write it.
// One workgroup (10 + 2 waves at 160 channels, <= 168 registers) per CU; 71 KB of LDS (four ring rows, two operand blocks,
// the helpers' partial sums).
//
// Also (round 4, single-frame latency of the medium / large models):
//   * 88 channels = 5.5 channel groups: six waves, the last one's quads 2, 3 and output channels 88..95 masked.  At
//     bench scale one wave per strip (bifpn_rows.hip) is the faster form there (0.49 against 0.67 ms per P3 node);
//     this form serves the 88-channel pyramid at time batches below 8 only (NodeArgs::rows == 2),
//   * the last strip of a row masks the pixels past the row end (no store, no statistics): every level down to
//     2 x 2 and widths that are not multiples of 16 qualify,
//   * rows == 2 picks short segments (a node is a latency chain of its rows when there are only a few images),
//   * a variant with two same-level inputs (P7's bottom-up node fed by P6's pooled output).
#include <algorithm>
#include <type_traits>

#include "conv_mfma.h"
#include "bifpn_node.h"

namespace jh {

namespace {
constexpr int kWPX18 = 18;                         // ring row width: 16 pixels + the depthwise halo
template <int RC>
struct RowWgGeo {
  static constexpr int NW = (RC + 15) / 16;        // waves per workgroup = channel groups = output column blocks
  static constexpr int K8 = RC / 8;
  static constexpr int RSA = RC + 4;               // operand-block row stride (floats): 16 rows on distinct banks
  static constexpr int ROWB = kWPX18 * RC * 4;     // bytes per ring row
  static constexpr int OPB = 16 * RSA * 4;         // bytes per operand block
  static constexpr int NH = 2;                     // helper waves (see the kernel)
  static constexpr int NSLOT = 4;                  // ring rows
  static constexpr size_t lds_bytes() { return (size_t)NSLOT * ROWB + 2 * OPB + 4 * 1024; }   // (+ the helpers' partial sums)
  static_assert(RC % 8 == 0 && NW <= 16, "channel count of the workgroup row-streaming node");
};
typedef float wf2 __attribute__((ext_vector_type(2)));
typedef float wf4 __attribute__((ext_vector_type(4)));
typedef unsigned wu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
}  // namespace

// Helper waves (round 6).  A ring row is 18 pixels x 4 quads = 72 fusion items per channel group: the second fusion
// round of a group wave carried 8 items on 64 lanes -- loads, FMAs and the two transcendentals of SiLU issued for an
// eighth of a wave -- and the NW group waves sit unevenly on the four SIMDs (160 channels: 3 / 3 / 2 / 2; 88: 2 / 2 / 1 / 1;
// waves w and w + 4 share a SIMD: tools/wave_simd_probe.hip).  So the workgroup has NH = 2 more waves, which land on the
// SIMDs with the fewest group waves:
//   * they fuse pixels 16, 17 of ALL channel groups (8 NW items over the two helpers), one row AHEAD of the group
//     waves, so that the row's one barrier orders their ring writes before the depthwise that reads them; that takes a
//     fourth ring slot (the slot of row r + 1 is still being read as row r - 2 otherwise).  Same bits as without them.
//   * KSPLIT: helper h also runs the UPPER half of the channel steps of column block NW - 2 + h -- the MFMAs of the
//     last group wave of each crowded SIMD (160 channels: 120 / 120 / 80 / 80 MFMAs per row and SIMD become 100 each; 88:
//     44 / 44 / 22 / 22 become 33 each).  It leaves its partial sums in LDS; the group wave (ROLE 1) adds them one row
//     later, behind the next row's barrier, and only then finishes that row (bias, statistics, stores).  The sum of the
//     two halves is not the one chain's sum: the last two column blocks differ in the last bits from the unsplit form
//     (a function of the node's shape alone, like every other choice of form).
template <int RC, int NIN, int M1, int M2, int ACT, bool POOL, bool KSPLIT>
__global__ __launch_bounds__((RowWgGeo<RC>::NW + RowWgGeo<RC>::NH) * 64)
void bifpn_rows_wg_kernel(const NodeArgs a, int seg_rows, int strips) {
  using GEO = RowWgGeo<RC>;
  constexpr int NSLOT = GEO::NSLOT, NW = GEO::NW, NH = GEO::NH;
  constexpr int K8 = GEO::K8, RSA = GEO::RSA, kRowB = GEO::ROWB, kOpOff = NSLOT * GEO::ROWB, kOpB = GEO::OPB;
  constexpr int kModes[3] = {FUSE_SAME, M1, M2};
  // channel steps of a split column block that stay with its group wave (the even split is the fastest: 160 channels,
  // P3 node at 384 images, 10 / 8 / 7 / 6 of 20 steps: 0.973 / 1.021 / 0.969 / 0.986 ms, P4 nodes 6.44 / 6.33 / 6.56 / 6.63 ms
  // per batch; 88 channels: 6, 4, 3, 2 of 11 within 1 %)
  constexpr int KS = KSPLIT ? (K8 + 1) / 2 : K8;
  constexpr int kPartOff = kOpOff + 2 * kOpB;      // + (helper, row parity) x 1 KB: the helpers' partial sums
  static_assert(NSLOT == 4 && NH == 2 && NW >= 4, "helper waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int sx = blockIdx.x % strips, seg = blockIdx.x / strips, n = blockIdx.y;
  const int ox0 = sx * 16, y_begin = seg * seg_rows, y_end = min(a.H, y_begin + seg_rows);
  // (the roles are instantiations of one body: the role must be a compile-time constant or the channel masks of the
  //  160-channel form stop folding and the group waves' row loop fills with exec-masked regions and spills)
  // ROLE 0: group wave; 1: group wave whose upper channel steps a helper runs (KSPLIT); 2: helper
  auto body = [&](auto role_c) __attribute__((always_inline)) {
  constexpr int ROLE = decltype(role_c)::value;
  constexpr int NI = 1, NIA = 1;                   // fusion items per lane and row
  const int mrow = lane & 15, kq = lane >> 4;

  // ---- fusion items of this lane: ring pixel, channel quad ---------------------------------------------------
  // (88 channels = 5.5 groups: quads 2, 3 of the last group and its output channels 88..95 do not exist)
  int ipx[NIA], ic[NIA];
  bool iok[NIA];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (ROLE != 2) {
      ipx[i] = lane >> 2;
      ic[i] = wave * 16 + (lane & 3) * 4;
      iok[i] = RC % 16 == 0 || ic[i] < RC;
    } else {
      constexpr int per = NW * 8 / NH;
      const int hitem = (wave - NW) * per + lane;                     // (channel group, pixel 16 / 17, quad)
      ipx[i] = 16 + ((hitem >> 2) & 1);
      ic[i] = (hitem >> 3) * 16 + (hitem & 3) * 4;
      iok[i] = lane < per && ic[i] < RC;
    }
  }
  // folded norm + fusion weights of an item's channel quad: fused = sum_k x_k a_k + B
  wf4 ak[NIA][NIN], bb[NIA];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int cs = iok[i] ? ic[i] : 0;               // a valid quad for the loads of per-channel constants
    bb[i] = (wf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      float m4[4] = {0.f, 0.f, 0.f, 0.f}, r4[4] = {1.f, 1.f, 1.f, 1.f};
      if (a.st[k]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const double* st = a.st[k] + ((size_t)n * RC + cs + j) * kStatW;
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
        ak[i][k][j] = ak1;
        bb[i][j] += -m4[j] * ak1;
      }
    }
  }
  __amdgpu_buffer_rsrc_t rs[NIN];
  int rowstep[NIN];                                // bytes per source row of input k
#pragma unroll
  for (int k = 0; k < NIN; ++k) {
    const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
    const size_t plane = node_plane(kModes[k], a.H, a.W) * RC;
    rs[k] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in[k] + (size_t)n * plane), 0, (int)(plane * 4),
                                              0x00020000);
    rowstep[k] = (a.W >> sh) * RC * 4;
  }
  // load offsets of the items inside a source row (bit 31 = outside the image or no item: the buffer load returns 0),
  // the 0 / 1 mask of the zero padding, the ring address
  int voff[NIA][NIN], fdst[NIA];
  float msk[NIA];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int ix = ox0 - 1 + ipx[i];
    const bool ok = iok[i] && (unsigned)ix < (unsigned)a.W;
    msk[i] = ok ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
      voff[i][k] = ok ? ((ix >> sh) * RC + ic[i]) * 4 : (int)0x80000000;
    }
    fdst[i] = (ipx[i] * RC + ic[i]) * 4;                        // + ring slot
  }
  // (see bifpn_rows.hip: an up-sampled input changes its source row only every 2nd output row; `all_c` is a
  //  compile-time flag because loads under a run-time branch make the compiler's in-order wait counts pessimistic)
  auto issue = [&](int yf, auto all_c, wf4 (&raw)[NIA][NIN]) __attribute__((always_inline)) {   // (yf inside the image)
    constexpr bool all = decltype(all_c)::value;
    int srow[NIN];
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      const int sh = kModes[k] == FUSE_UP2 ? 1 : (kModes[k] == FUSE_UP4 ? 2 : 0);
      srow[k] = (yf >> sh) * rowstep[k];
    }
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
      for (int k = 0; k < NIN; ++k)
        if (all || kModes[k] == FUSE_SAME)
          raw[i][k] =
              __builtin_bit_cast(wf4, __builtin_amdgcn_raw_buffer_load_b128(rs[k], voff[i][k], srow[k], 0));
  };
  // fused + activated row yf -> its ring slot (zeros outside the image: the depthwise padding)
  auto fuse = [&](int yf, int slot, const wf4 (&raw)[NIA][NIN]) __attribute__((always_inline)) {
    unsigned char* dst = smem + slot * kRowB;
    if ((unsigned)yf >= (unsigned)a.H) {                             // (uniform) padding row
#pragma unroll
      for (int i = 0; i < NI; ++i)
        if (iok[i]) *reinterpret_cast<wf4*>(dst + fdst[i]) = (wf4){0.f, 0.f, 0.f, 0.f};
      return;
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      wf4 v = bb[i];
#pragma unroll
      for (int k = 0; k < NIN; ++k) v = __builtin_elementwise_fma(raw[i][k], ak[i][k], v);
      if (ACT == ACT_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = silu_fast(v[j]);
      } else if (ACT == ACT_RELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
      }
      // (pixels 0 and 17 of a strip at the image border: the depthwise padding)
      v *= (wf4){msk[i], msk[i], msk[i], msk[i]};
      if (iok[i]) *reinterpret_cast<wf4*>(dst + fdst[i]) = v;
    }
  };

  if constexpr (ROLE == 2) {
    // rows y_begin - 1 .. y_begin + 1 before the group waves' first depthwise (the barrier in front of their row loop),
    // then row yf + 1 in front of the barrier of step yf.  Row r lives in slot (r + 1) & 3.
    wf4 ra[NIA][NIN], rb[NIA][NIN];
    // KSPLIT: the upper channel steps of column block NW - 2 + (wave - NW), weights in registers like a group wave's
    constexpr int KH = KSPLIT ? K8 - KS : 1;
    wf2 bwh[KH];
    if (KSPLIT) {
#pragma unroll
      for (int k8 = 0; k8 < KH; ++k8)
        bwh[k8] = *reinterpret_cast<const wf2*>(a.pw + ((size_t)((KS + k8) * NW + (wave - 2)) * 64 + lane) * 2);
    }
    const int ardh = kOpOff + (mrow * RSA) * 4 + kq * 8 + KS * 32;    // + operand buffer + channel step (imm)
    const int pdst = kPartOff + (wave - NW) * 2048 + lane * 16;       // + row parity x 1 KB
    auto partial = [&](int y) __attribute__((always_inline)) {       // (behind the barrier that completes row y's block)
      const int kBuf = (y & 1) ? kOpB : 0;
      f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
      for (int k8 = 0; k8 < KH; ++k8) {
        const float2 xc = *reinterpret_cast<const float2*>(smem + ardh + kBuf + k8 * 32);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bwh[k8][0], xc.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bwh[k8][1], xc.y, acc1, 0, 0, 0);
      }
      *reinterpret_cast<wf4*>(smem + pdst + (y & 1) * 1024) =
          (wf4){acc0[0], acc0[1], acc0[2], acc0[3]} + (wf4){acc1[0], acc1[1], acc1[2], acc1[3]};
    };
    __builtin_amdgcn_s_waitcnt(0);
    if (y_begin - 1 >= 0) issue(y_begin - 1, std::true_type{}, ra);
    issue(y_begin, std::true_type{}, rb);
    fuse(y_begin - 1, y_begin & 3, ra);
    if (y_begin + 1 < a.H) issue(y_begin + 1, std::true_type{}, ra);
    fuse(y_begin, (y_begin + 1) & 3, rb);
    fuse(y_begin + 1, (y_begin + 2) & 3, ra);
    if (y_begin + 2 <= y_end && y_begin + 2 < a.H) issue(y_begin + 2, std::true_type{}, ra);
    lds_barrier();
    for (int yf = y_begin + 1; yf <= y_end; ++yf) {
      if (yf + 1 <= y_end) {
        fuse(yf + 1, (yf + 2) & 3, ra);
        if (yf + 2 <= y_end && yf + 2 < a.H) issue(yf + 2, std::true_type{}, ra);
      }
      lds_barrier();
      if (KSPLIT) partial(yf - 1);
    }
    if (KSPLIT) lds_barrier();                         // the last row's partial sums are in LDS
    return;
  }

  // ---- group waves: depthwise + pointwise of channel group `wave` --------------------------------------------
  const int sub = lane >> 2, c = wave * 16 + (lane & 3) * 4;    // pixel slot, channel quad of the depthwise
  const bool cq_ok = RC % 16 == 0 || c < RC;
  const int cs = cq_ok ? c : 0;
  const int dsrc = (sub * RC + c) * 4;                          // + ring slot + tap pixel (imm)
  const int adst = kOpOff + (sub * RSA + c) * 4;                // + operand buffer
  const int ard = kOpOff + (mrow * RSA) * 4 + kq * 8;           // + operand buffer + channel step (imm)
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
      a.y + (size_t)n * a.H * a.W * a.cout_p, 0, (int)((size_t)a.H * a.W * a.cout_p * 4), 0x00020000);
  const bool co_ok = RC % 16 == 0 || wave * 16 + kq * 4 < RC;          // (cout_p == RC: checked by the launcher)
  // (ragged last strip -- levels narrower than 16 pixels or not a multiple of 16: pixels past the row end are computed
  //  from zero-padded input like any border pixel and dropped here: no store, no statistics)
  const bool pok = ox0 + mrow < a.W;
  const float pm = pok ? 1.f : 0.f;
  const int yoff = co_ok && pok ? ((ox0 + mrow) * a.cout_p + wave * 16 + kq * 4) * 4 : (int)0x80000000;
  // POOL: the node also writes MaxPool2d(2, 2) of its raw output (max commutes with the monotone InstanceNorm map,
  // the pooled tensor keeps this node's statistics).  Horizontal max: the neighbour pixel is lane ^ 1 (one DPP
  // move); vertical: the even row's maxima wait in registers for the odd row.  Lanes of even pixels own the pixel.
  __amdgpu_buffer_rsrc_t rp = ry;
  int pbase = (int)0x80000000;
  if (POOL) {
    rp = __builtin_amdgcn_make_buffer_rsrc(a.y_pool + (size_t)n * (a.H >> 1) * (a.W >> 1) * a.cout_p, 0,
                                           (int)((size_t)(a.H >> 1) * (a.W >> 1) * a.cout_p * 4), 0x00020000);
    if (!(mrow & 1) && co_ok && pok) pbase = (((ox0 + mrow) >> 1) * a.cout_p + wave * 16 + kq * 4) * 4;
  }
  wf4 park = (wf4){0.f, 0.f, 0.f, 0.f};
  wf4 raw[NIA][NIN];
  wf4 dwr[9];                                      // depthwise weights of this lane's channel quad
#pragma unroll
  for (int t = 0; t < 9; ++t) dwr[t] = *reinterpret_cast<const wf4*>(a.dw + t * RC + cs);
  // pointwise weights of output column block `wave` for all RC / 8 channel steps (packed as for bifpn_rows.hip:
  // weights are the A operand, pixels the B operand, so a lane's accumulator is four consecutive channels
  // 16 wave + 4 (lane >> 4) .. + 3 of pixel lane & 15)
  constexpr int KM = ROLE == 1 ? KS : K8;          // (ROLE 1: the upper channel steps are the helper's)
  wf2 bw[KM];
#pragma unroll
  for (int k8 = 0; k8 < KM; ++k8)
    bw[k8] = *reinterpret_cast<const wf2*>(a.pw + ((size_t)(k8 * NW + wave) * 64 + lane) * 2);
  const int psrc = kPartOff + (wave - (NW - 2)) * 2048 + lane * 16;     // + row parity x 1 KB (ROLE 1)
  wf4 plo = (wf4){0.f, 0.f, 0.f, 0.f};             // ROLE 1: this wave's half of the previous row's sums
  wf4 b4 = (wf4){0.f, 0.f, 0.f, 0.f};
  if (a.bias) b4 = *reinterpret_cast<const wf4*>(a.bias + wave * 16 + kq * 4);

  wf4 s1 = (wf4){0.f, 0.f, 0.f, 0.f}, s2 = s1;
  __builtin_amdgcn_s_waitcnt(0);                   // (the preamble's loads: see bifpn_rows.hip)
  int slot = y_begin & 3;                          // slot of row yf = y_begin - 1: (yf + 1) & 3

  // ---- bias, statistics (in registers across the strip), one 16-byte store: p = the row's sums, even_c = its parity
  auto finish = [&](const wf4 p, int y, auto even_c) __attribute__((always_inline)) {
    const wf4 v = p + b4;
    const wf4 vm = v * (wf4){pm, pm, pm, pm};
    s1 += vm;
    s2 = __builtin_elementwise_fma(vm, v, s2);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wu4, v), ry, yoff + y * a.W * a.cout_p * 4, 0, 0);
    if (POOL) {
      wf4 hm;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        hm[j] = fmaxf(v[j], __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[j]), 0xB1, 0xF, 0xF, true)));
      if (decltype(even_c)::value) {                                      // even output row: keep
        park = hm;
      } else {                                                            // odd row: combine, store
        wf4 pv;
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[j] = fmaxf(park[j], hm[j]);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wu4, pv), rp,
                                               pbase + (y >> 1) * (a.W >> 1) * a.cout_p * 4, 0, 0);
      }
    }
  };

  auto row = [&](int yf, auto next_all_c, auto out_c, auto fin_c) __attribute__((always_inline)) {
    if constexpr (NI > 0) {
      fuse(yf, slot, raw);
      if (yf + 1 <= y_end && yf + 1 < a.H) issue(yf + 1, next_all_c, raw);
    }
    const int y = yf - 1;                               // output row whose three ring rows are now complete
    const int s_top = (slot + 2) & 3;                   // slot of row y - 1 = yf - 2
    slot = (slot + 1) & 3;
    if (!decltype(out_c)::value) return;             // (the two rows above the segment's first output row)
    // the operand buffer of this row: by row parity (next_all_c is true in the half of the unrolled loop that
    // produces the EVEN output rows)
    constexpr int kBuf = decltype(next_all_c)::value ? 0 : kOpB;
    // ---- depthwise 3x3: lane = (pixel, channel quad of the wave's 16 channels) -> operand block ------------
    {
      int rs_ = s_top;
      wf4 d = (wf4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const unsigned char* src = smem + rs_ * kRowB + dsrc;
        rs_ = (rs_ + 1) & 3;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          d = __builtin_elementwise_fma(*reinterpret_cast<const wf4*>(src + dx * RC * 4), dwr[dy * 3 + dx], d);
      }
      if (cq_ok) *reinterpret_cast<wf4*>(smem + adst + kBuf) = d;
    }
    lds_barrier();                                   // every channel group of the operand block is written
    // ---- pointwise 1x1: 16 output channels x 16 pixels x RC channels on the matrix cores --------------------
    f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
#pragma unroll
    for (int k8 = 0; k8 < KM; ++k8) {
      const float2 xc = *reinterpret_cast<const float2*>(smem + ard + kBuf + k8 * 32);
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[k8][0], xc.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[k8][1], xc.y, acc1, 0, 0, 0);
    }
    const wf4 p = (wf4){acc0[0], acc0[1], acc0[2], acc0[3]} + (wf4){acc1[0], acc1[1], acc1[2], acc1[3]};
    if constexpr (ROLE == 1) {
      // (the helper's half of row y - 1 was complete before the barrier above; that row is finished now, this row's
      //  half waits in registers for the next barrier)
      if constexpr (decltype(fin_c)::value) {           // (not in the segment's first row: a compile-time flag, the
        const wf4 phi = *reinterpret_cast<const wf4*>(smem + psrc + (decltype(next_all_c)::value ? 1024 : 0));   // loop is peeled)
        finish(plo + phi, y - 1, std::integral_constant<bool, !decltype(next_all_c)::value>{});
      }
      plo = p;
    } else {
      finish(p, y, next_all_c);
    }
  };
  // (segments start on even rows -- the launcher -- so row y_begin - 1 is odd and the rows requested from the first
  //  half of the unrolled body are even: all inputs; from the second half odd: same-level inputs only)
  if constexpr (NI > 0)
    if (y_begin - 1 >= 0) issue(y_begin - 1, std::true_type{}, raw);
  row(y_begin - 1, std::true_type{}, std::false_type{}, std::false_type{});
  row(y_begin, std::false_type{}, std::false_type{}, std::false_type{});
  lds_barrier();                                     // the helpers' share of the first three rows is in the ring
  if constexpr (ROLE == 1) {
    row(y_begin + 1, std::true_type{}, std::true_type{}, std::false_type{});
    for (int yf = y_begin + 2; yf <= y_end; yf += 2) {
      row(yf, std::false_type{}, std::true_type{}, std::true_type{});
      if (yf + 1 <= y_end) row(yf + 1, std::true_type{}, std::true_type{}, std::true_type{});
    }
  } else {
    for (int yf = y_begin + 1; yf <= y_end; yf += 2) {
      row(yf, std::true_type{}, std::true_type{}, std::false_type{});
      if (yf + 1 <= y_end) row(yf + 1, std::false_type{}, std::true_type{}, std::false_type{});
    }
  }
  if (KSPLIT) {
    lds_barrier();                                   // the helpers' sums of the last row
    if constexpr (ROLE == 1) {
      // (the pooled output needs an even height, so the last row is odd whenever its parity matters)
      const int y = y_end - 1;
      const wf4 phi = *reinterpret_cast<const wf4*>(smem + psrc + (y & 1) * 1024);
      finish(plo + phi, y, std::false_type{});
    }
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
    for (int i = 0; i < 4; ++i) {
      const float t1 = row_sum(s1[i]), t2 = row_sum(s2[i]);
      const int ch = wave * 16 + kq * 4 + i;
      if (mrow == 0 && co_ok) stat_add(a.stats + ((size_t)n * a.cout_p + ch) * kStatW, t1, t2);
    }
  }
  };
  if (wave >= NW) body(std::integral_constant<int, 2>{});
  else if (KSPLIT && wave >= NW - 2) body(std::integral_constant<int, 1>{});
  else body(std::integral_constant<int, 0>{});
}

// Shapes of the workgroup form (the channel count is checked by the caller, bifpn_rows_eligible): as many output
// channels as input channels, no padding in either.
bool bifpn_rows_wg_shape_ok(const NodeArgs& a) {
  return (a.Cp == 160 || a.Cp == 88) && a.cout_p == a.Cp && a.cout_p16 == (a.Cp + 15) / 16 * 16;
}

template <int RC, bool KSPLIT>
static int launch_rows_wg_rc(const NodeArgs& a, hipStream_t s) {
  const int strips = (a.W + 15) / 16;
  // Rows per workgroup: a function of the node and of the predictor's time-batch CLASS only (the float partial sums
  // of the statistics are taken per strip segment: bit-equal results for any number of images per launch), as in
  // bifpn_rows.hip.  a.rows == 2 (time batches below 8, the single-frame caller among them: few images, a node is a
  // latency chain of its rows): short segments -- measured per node with one 12-camera frame set, 88 / 160 channels,
  // us: 64-row levels 16 rows 33 / 62 (8: 41 / 72, 4: 50 / 79, 32: 53 / 103); 32-row levels 4 rows 18 / 29 (8: 22 / 38,
  // 2: 26 / 42); 16 rows and below 2 rows 13 / 21 (4: 15 / 26, 8: 21 / 37).
  int seg_rows = JH_ENV_KNOB("JH_NODE_SEG") > 0 ? JH_ENV_KNOB("JH_NODE_SEG")
                 : a.rows == 2 ? (a.H >= 64 ? 16 : (a.H >= 32 ? 4 : 2))
                 : (a.n_in == 2 || a.mode[1] == FUSE_SAME ? std::max(8, a.H / 2) : (a.H >= 64 ? 16 : 8));
  // (time-batch class >= 8, levels of 32 pixels and more: ONE segment per strip -- the workgroup's prologue (its waves'
  //  weight slices, the statistics' fp64 arithmetic, two warm-up rows) is paid once per 64 rows instead of per 16 or 32:
  //  measured at 160 channels, 384 images: P3 node 1.218 -> 1.133 ms, head 1.270 -> 1.024, P4 0.331 -> 0.292 / 0.431 ->
  //  0.389; the 16-pixel levels keep 8 rows, 0.106 against 0.112 with 16)
  //  (88 channels on a ragged level -- bifpn_rows_ragged88 -- keep half-image segments: 192 images x 3 strips of a 40-pixel
  //  level are 576 workgroups on 512 slots with one segment per strip)
  if (a.rows != 2 && JH_ENV_KNOB("JH_NODE_SEG") <= 0 && a.H >= 32 && !bifpn_rows_ragged88(a)) seg_rows = a.H;
  seg_rows = (seg_rows + 1) & ~1;
  if (seg_rows > a.H) seg_rows = a.H;
  const int segs = (a.H + seg_rows - 1) / seg_rows;
  const size_t lds = RowWgGeo<RC>::lds_bytes();
  const dim3 grid(strips * segs, a.N), block((RowWgGeo<RC>::NW + RowWgGeo<RC>::NH) * 64);
#define JH_ROWS(NIN, M1, M2, ACT, POOL)                                                                            \
  do {                                                                                                             \
    auto kern = bifpn_rows_wg_kernel<RC, NIN, M1, M2, ACT, POOL, KSPLIT>;                                          \
    static bool big = false;                                                                                       \
    if (!big && lds > 64 * 1024) {                                                                                 \
      JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                                        \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                   \
      big = true;                                                                                                  \
    }                                                                                                              \
    hipLaunchKernelGGL(kern, grid, block, lds, s, a, seg_rows, strips);                                            \
  } while (0)
  JH_REQUIRE(!a.y_pool || (a.act == ACT_SILU && a.H % 2 == 0 && (a.n_in == 2 || a.mode[1] == FUSE_SAME)),
             "workgroup row-streaming node: pooled output");
  if (a.n_in == 2 && a.mode[1] == FUSE_SAME) {       // (bottom-up node of the coarsest level: P7 from P7_in and pooled P6)
    JH_REQUIRE(a.act == ACT_SILU && !a.y_pool, "workgroup row-streaming node: two same-level inputs");
    JH_ROWS(2, FUSE_SAME, 0, ACT_SILU, false);
  } else if (a.n_in == 2) {
    if (a.act == ACT_SILU && a.y_pool) JH_ROWS(2, FUSE_UP2, 0, ACT_SILU, true);
    else if (a.act == ACT_SILU) JH_ROWS(2, FUSE_UP2, 0, ACT_SILU, false);
    else if (a.act == ACT_NONE) JH_ROWS(2, FUSE_UP2, 0, ACT_NONE, false);
    else JH_REQUIRE(false, "row-streaming node: activation");
  } else if (a.mode[1] == FUSE_SAME) {
    if (a.act == ACT_SILU && a.y_pool) JH_ROWS(3, FUSE_SAME, FUSE_SAME, ACT_SILU, true);
    else if (a.act == ACT_SILU) JH_ROWS(3, FUSE_SAME, FUSE_SAME, ACT_SILU, false);
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

int launch_bifpn_rows_wg(const NodeArgs& a, hipStream_t s) {
  // (JH_NODE_WG_KSPLIT=0: the helper waves only fuse pixels 16, 17 -- the form whose bits equal the helper-less kernel's)
  const bool ks = JH_ENV_KNOB("JH_NODE_WG_KSPLIT") != 0;
  if (a.Cp == 88) return ks ? launch_rows_wg_rc<88, true>(a, s) : launch_rows_wg_rc<88, false>(a, s);
  return ks ? launch_rows_wg_rc<160, true>(a, s) : launch_rows_wg_rc<160, false>(a, s);
}

}  // namespace jh
