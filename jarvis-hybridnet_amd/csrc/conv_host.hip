// Host side of the MFMA convolution: descriptors, weight packing, dispatch.
#include <vector>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include "conv_mfma.h"

namespace jh {

ConvDesc conv_desc(int nd, int k, int stride, int pad, int cin, int cout) {
  ConvDesc d{};
  d.nd = nd; d.k = k; d.stride = stride; d.ostride = 1; d.nphase = 1;
  d.cin = cin; d.cout = cout;
  d.phase[0].pad[0] = (nd == 3) ? pad : 0;
  d.phase[0].pad[1] = pad; d.phase[0].pad[2] = pad;
  d.phase[0].ooff[0] = d.phase[0].ooff[1] = d.phase[0].ooff[2] = 0;
  return d;
}

// ConvTranspose2d(k=4, s=2, p=1): output (2y+py, 2x+px) is a 2x2-tap conv of the
// input; even outputs read inputs {y-1, y} (pad 1), odd outputs {y, y+1} (pad 0).
ConvDesc deconv2d_k4s2p1_desc(int cin, int cout) {
  ConvDesc d{};
  d.nd = 2; d.k = 2; d.stride = 1; d.ostride = 2; d.nphase = 4;
  d.cin = cin; d.cout = cout;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      ConvPhase& p = d.phase[py * 2 + px];
      p.pad[0] = 0; p.pad[1] = py ? 0 : 1; p.pad[2] = px ? 0 : 1;
      p.ooff[0] = 0; p.ooff[1] = py; p.ooff[2] = px;
    }
  return d;
}

// ConvTranspose3d(k=2, s=2, p=0): each of the 8 output parities is a 1x1x1 conv.
ConvDesc deconv3d_k2s2_desc(int cin, int cout) {
  ConvDesc d{};
  d.nd = 3; d.k = 1; d.stride = 1; d.ostride = 2; d.nphase = 8;
  d.cin = cin; d.cout = cout;
  for (int p = 0; p < 8; ++p) {
    d.phase[p].pad[0] = d.phase[p].pad[1] = d.phase[p].pad[2] = 0;
    d.phase[p].ooff[0] = (p >> 2) & 1; d.phase[p].ooff[1] = (p >> 1) & 1; d.phase[p].ooff[2] = p & 1;
  }
  return d;
}

void conv_out_shape(const ConvDesc& d, int D, int H, int W, int* Do, int* Ho, int* Wo) {
  if (d.ostride > 1) {          // transposed: every phase produces the input extent
    *Do = (d.nd == 3) ? D * d.ostride : 1; *Ho = H * d.ostride; *Wo = W * d.ostride;
    return;
  }
  const int kd = (d.nd == 3) ? d.k : 1;
  *Do = (d.nd == 3) ? (D + 2 * d.phase[0].pad[0] - kd) / d.stride + 1 : 1;
  *Ho = (H + 2 * d.phase[0].pad[1] - d.k) / d.stride + 1;
  *Wo = (W + 2 * d.phase[0].pad[2] - d.k) / d.stride + 1;
}

// Which transposed-conv kernel tap feeds tap t of phase (parity) par for the
// 2D k4 s2 p1 case: even parity taps {t0: k=3, t1: k=1}, odd {t0: k=2, t1: k=0}.
static inline int deconv4_tap(int parity, int t) { return parity ? (t == 0 ? 2 : 0) : (t == 0 ? 3 : 1); }

int pack_conv_weights(const ConvDesc& d, const float* w, const float* b, bool transposed,
                      ConvWeights* out) {
  const int cin_p = cpad(d.cin);
  const int cout_p16 = round_up(d.cout, 16);
  const int kd = (d.nd == 3) ? d.k : 1;
  const int ntap = kd * d.k * d.k;
  const int nk8 = cin_p / 8, nb = cout_p16 / 16;
  const size_t phase_stride = (size_t)ntap * nk8 * nb * 128;
  std::vector<float> packed(phase_stride * d.nphase, 0.f);
  // kernels that read both operands 16 bytes at a time want two 8-channel steps per lane word
  const bool paired = d.nd == 2 && d.ostride > 1 && deconv4_eligible(cin_p, cout_p16);
  // stride-2 3D convs: two taps per 16-byte weight word (conv_mfma.h, TAPPAIR)
  const bool tap_paired = d.nd == 3 && d.k == 3 && d.stride == 2 && d.ostride == 1 && JH_ENV_KNOB("JH_CONV_TAPPAIR") != 0;
  if (tap_paired) packed.assign((size_t)((ntap + 1) / 2) * nk8 * nb * 256, 0.f);
  // geometry of the source tensor
  int skd, sk;   // source kernel extents
  if (d.ostride > 1 && d.nd == 2) { skd = 1; sk = 4; }
  else if (d.ostride > 1 && d.nd == 3) { skd = 2; sk = 2; }
  else { skd = kd; sk = d.k; }
  const size_t sktaps = (size_t)skd * sk * sk;
  for (int ph = 0; ph < d.nphase; ++ph) {
    for (int tz = 0; tz < kd; ++tz)
      for (int ty = 0; ty < d.k; ++ty)
        for (int tx = 0; tx < d.k; ++tx) {
          int sz = tz, sy = ty, sx = tx;
          if (d.ostride > 1 && d.nd == 2) {
            sy = deconv4_tap(d.phase[ph].ooff[1], ty);
            sx = deconv4_tap(d.phase[ph].ooff[2], tx);
          } else if (d.ostride > 1 && d.nd == 3) {
            sz = d.phase[ph].ooff[0]; sy = d.phase[ph].ooff[1]; sx = d.phase[ph].ooff[2];
          }
          const size_t stap = ((size_t)sz * sk + sy) * sk + sx;
          const int tap = (tz * d.k + ty) * d.k + tx;
          for (int ci = 0; ci < d.cin; ++ci)
            for (int co = 0; co < d.cout; ++co) {
              const float v = transposed ? w[((size_t)ci * d.cout + co) * sktaps + stap]
                                         : w[((size_t)co * d.cin + ci) * sktaps + stap];
              const int kc8 = ci / 8, kq = (ci % 8) / 2, j = ci % 2;
              const int nbk = co / 16, nn = co % 16;
              const int lane = kq * 16 + nn;
              if (tap_paired)
                packed[((((size_t)(tap / 2)) * nk8 + kc8) * nb + nbk) * 256 + lane * 4 + (tap & 1) * 2 + j] = v;
              else if (paired)
                packed[ph * phase_stride + (((size_t)tap * (nk8 / 2) + kc8 / 2) * nb + nbk) * 256 + lane * 4 + (kc8 & 1) * 2 + j] = v;
              else
                packed[ph * phase_stride + (((size_t)tap * nk8 + kc8) * nb + nbk) * 128 + lane * 2 + j] = v;
            }
        }
  }
  out->cin_p = cin_p; out->cout_p16 = cout_p16; out->paired = tap_paired ? 2 : (paired ? 1 : 0);
  out->phase_stride = tap_paired ? packed.size() : phase_stride;
  JH_CHECK_HIP(hipMalloc(&out->w, packed.size() * sizeof(float)));
  JH_CHECK_HIP(hipMemcpy(out->w, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  out->bias = nullptr;
  if (b) {
    std::vector<float> bp(cout_p16, 0.f);
    std::memcpy(bp.data(), b, d.cout * sizeof(float));
    JH_CHECK_HIP(hipMalloc(&out->bias, bp.size() * sizeof(float)));
    JH_CHECK_HIP(hipMemcpy(out->bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

void free_conv_weights(ConvWeights* w) {
  if (w->w) (void)hipFree(w->w);
  if (w->bias) (void)hipFree(w->bias);
  w->w = w->bias = nullptr;
}

// Column blocks (of 16 output channels) per workgroup.  A workgroup stages its input operand once per GROUP of nr
// column blocks (global loads, InstanceNorm / activation / SE gate on load, LDS writes) and every tap re-uses the staged
// patch, so the cost of a layer is about groups x (stage + nr) in units of one column block's MFMAs, with
// stage = 1.5 / taps for operands that take SiLU (two transcendentals per element) and an SE gate on load -- the project
// convolutions of the MBConv blocks -- and 0.5 / taps otherwise: for those pointwise layers a padded last group is
// cheaper than more groups, for the k x k layers the padding is what counts.  (Rounds 1-4 minimised the padding alone: 5
// and 7 column blocks -- the 80- and 112-channel project convolutions of the medium model -- ran with nr = 1, i.e. staged
// a 480- / 672-channel operand five / seven times: 0.266 -> 0.189 ms for 672 -> 112 at 384 images.)  nr only partitions
// the output channels: outputs and statistics are the same bits for any nr.
static int pick_nr(int nb, int taps, bool heavy_staging, long units) {
  if (JH_ENV_KNOB("JH_CONV_NR_RULE") == 0) {     // the old rule: least padding, larger nr first
    int best = 1, best_waste = 1 << 30;
    for (int nr = 4; nr >= 1; --nr) {
      const int waste = (nb + nr - 1) / nr * nr - nb;
      if (waste < best_waste) { best_waste = waste; best = nr; }
    }
    return best;
  }
  const double stage = (heavy_staging ? 1.5 : 0.5) / (double)taps;
  // A launch of fewer workgroups than the chip holds (`units` = tiles x images: a single frame set has 24 tiles at the
  // 16-pixel levels) is as slow as ONE workgroup: stage + nr, so the narrow groups win until the chip is full.  One
  // frame set, us per launch: k5 s2 16 -> 96 28.8 -> 21.4, k3 40 -> 240 20.7 -> 15.7; single-frame latency small / small
  // 2.06 -> 2.00 ms, medium 3.59 -> 3.40, large 6.29 -> 6.17.  JH_CONV_NR_FILL=0: off (the throughput rule alone).
  const long slots = JH_ENV_KNOB("JH_CONV_NR_FILL") >= 0 ? std::max(1, JH_ENV_KNOB("JH_CONV_NR_FILL")) : 512;
  int best = 1;
  double best_cost = 1e30;
  for (int nr = 4; nr >= 1; --nr) {
    const long groups = (nb + nr - 1) / nr;
    const double cost = (double)std::max(units * groups, slots) * (stage + nr);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = nr; }
  }
  return best;
}

int launch_conv(const ConvDesc& d, const ConvWeights& w, const Act& x, const Act& y,
                const float* gate, double* stats, hipStream_t s, const InNorm* in, const SeGate* se) {
  JH_REQUIRE(x.Cp == w.cin_p || (x.Cp == 4 && w.cin_p == 8), "conv input channel padding mismatch");
  JH_REQUIRE(y.Cp == cpad(d.cout), "conv output channel padding mismatch");
  JH_REQUIRE(x.N == y.N, "batch mismatch");
  ConvArgs a{};
  a.x = x.p; a.y = y.p; a.w = w.w; a.bias = w.bias; a.gate = gate; a.stats = stats;
  if (in && in->stats) {
    a.in_stats = in->stats; a.in_inv = in->inv; a.in_act = in->act;
    a.nrm_floats = 2 * w.cin_p;              // multiple of 16 floats
  }
  if (gate) a.nrm_floats += w.cin_p;            // the gate vector of the image, behind mean / rstd
  if (se && se->pool) {                         // ... computed in the prologue: + channel means + hidden units
    JH_REQUIRE(!gate, "either a gate tensor or a gate recipe");
    a.se = *se;
    a.nrm_floats += w.cin_p + round_up(se->C, 4) + round_up(se->S, 4);
  }
  a.N = x.N; a.Din = x.D; a.Hin = x.H; a.Win = x.W; a.cin_p = w.cin_p; a.in_px = x.Cp;
  a.Dy = y.D; a.Hy = y.H; a.Wy = y.W; a.cout_p = y.Cp; a.cout_p16 = w.cout_p16;
  a.ostride = d.ostride; a.nphase = d.nphase; a.phase_stride = w.phase_stride; a.paired = w.paired;
  for (int p = 0; p < d.nphase; ++p) a.phase[p] = d.phase[p];
  if (d.ostride > 1) { a.Dout = x.D; a.Hout = x.H; a.Wout = x.W; }
  else conv_out_shape(d, x.D, x.H, x.W, &a.Dout, &a.Hout, &a.Wout);
  int Do, Ho, Wo;
  conv_out_shape(d, x.D, x.H, x.W, &Do, &Ho, &Wo);
  JH_REQUIRE(Do == y.D && Ho == y.H && Wo == y.W, "conv output extent mismatch");
  const int taps = d.k * d.k * (d.nd == 3 ? d.k : 1);
  long units;                                   // tiles x images of the launch (the default tile of each family)
  if (d.nd == 2) units = (long)((a.Hout + 7) / 8) * ((a.Wout <= 8 ? a.Wout + 7 : a.Wout + 15) / (a.Wout <= 8 ? 8 : 16));
  else units = (long)((a.Dout + 1) / 2) * ((a.Hout + 3) / 4) * ((a.Wout + 15) / 16);
  units *= (long)a.nphase * a.N;
  const int nr = pick_nr(w.cout_p16 / 16, taps, (in && in->stats && in->act == ACT_SILU) || gate || (se && se->pool),
                         units);
  // LDS budget of the staged channel chunk (pick_kc8): 40 KB -- three to four workgroups per CU for the k5 / k4T
  // layers -- measured against 72 KB (two): k5s2 16->96 308 -> 287 us, head ConvTranspose 954 -> 928 us
  const size_t budget = 40 * 1024;
  if (d.nd == 2 && d.ostride == 2 && d.k == 2) {      // all four parities from one staged patch
    const int rc = launch_deconv4_fused(a, s);
    if (rc >= 0) return rc;
  }
  JH_REQUIRE(!a.paired || (a.paired == 2 && d.nd == 3 && d.k == 3 && d.stride == 2),
             "paired weight layout without a kernel that reads it");
  if (conv_pw_direct_eligible(d, a)) return launch_conv_pw_direct(a, s);
  if (d.nd == 2) {
    const int small = (a.Wout <= 8) ? 1 : 0;
    // 16 x 16 tiles for high-resolution layers with few input channels
    int big = (a.Wout >= 64 && a.Hout >= 64 && w.cin_p <= 32 && d.ostride == 1 && d.stride == 1 &&
               (d.k == 1 || d.k == 3)) ? 1 : 0;   // (measured: stride-2 layers do not gain)
    if (JH_ENV_KNOB("JH_CONV2D_BIG") == 0) big = 0;
    if (big) return conv_launch_2d_big(a, d.k, d.stride, nr, budget, s);
    // Pointwise layers on images that do not tile into 8 x 16 (the reference's DEFAULT 320-pixel geometry: 20 x 20 at
    // stride 16 fills 52 % of its 3 x 2 tiles, 10 x 10 39 %): a 1 x 1 convolution has no neighbourhood, so the image
    // is handed over as one row of H * W pixels and cut into 1 x 128 tiles (20 x 20: 78 %, 40 x 40: 96 %).  A function
    // of the layer's shape only (the tiling is part of the fused statistics' fp32 arithmetic).  JH_CONV_K1_FLAT=0: off.
    if (d.k == 1 && d.stride == 1 && d.ostride == 1 && d.nphase == 1 && (a.Wout % 16 != 0 || a.Hout % 8 != 0) &&
        a.Hout * a.Wout >= 96 && a.Hin == a.Hout && a.Win == a.Wout && a.Hy == a.Hout && a.Wy == a.Wout &&
        JH_ENV_KNOB("JH_CONV_K1_FLAT") != 0) {
      ConvArgs f = a;
      f.Win = f.Wout = f.Wy = a.Hout * a.Wout;
      f.Hin = f.Hout = f.Hy = 1;
      return conv_launch_2d_k1_flat(f, nr, budget, s);
    }
    if (d.k == 1 && d.stride == 1) return conv_launch_2d_k1(a, nr, small, budget, s);
    if (d.k == 2 && d.stride == 1) return conv_launch_2d_k2(a, nr, small, budget, s);
    // (a function of the layer's shape only, like every tile choice: JH_CONV_W20=0 switches it off)
    // (time-batch class >= 8 only: one tile per image leaves a 12-image launch with a sixth of the workgroups -- measured
    //  on one frame set of the default geometry: 80 -> 480 @ 20 94 us against 63)
    if (d.k == 3 && d.stride == 1 && d.ostride == 1 && a.Wout > 16 && a.Wout <= 20 && a.Hout <= 22 &&
        !d.latency_class && JH_ENV_KNOB("JH_CONV_W20") != 0)
      return conv_launch_2d_k3_w20(a, nr, std::max(budget, (size_t)64 * 1024), s);
    // (the same for the other layers of the default geometry that leave the 8 x 16 tiles half empty: 5 x 5 on 40-pixel
    //  rows, 3 x 3 stride 2 onto 20 x 20; JH_CONV_W40=0 switches both off)
    if (!d.latency_class && d.ostride == 1 && JH_ENV_KNOB("JH_CONV_W40") != 0) {
      if (d.k == 5 && d.stride <= 2 && a.Wout > 32 && a.Wout <= 40)
        return conv_launch_2d_k5_w40(a, d.stride, nr, std::max(budget, (size_t)64 * 1024), s);
      if (d.k == 3 && d.stride == 2 && a.Wout > 16 && a.Wout <= 20 && a.Hout <= 22)
        return conv_launch_2d_k3s2_w20(a, nr, std::max(budget, (size_t)96 * 1024), s);
    }
    if (d.k == 3 && d.stride <= 2) return conv_launch_2d_k3(a, d.stride, nr, small, budget, s);
    if (d.k == 5 && d.stride <= 2) return conv_launch_2d_k5(a, d.stride, nr, small, budget, s);
  } else {
    // small tile when the volume would otherwise give fewer blocks than CUs
    // The choice depends on the IMAGE only, never on the batch size: fused statistics are sums of
    // per-workgroup fp32 partials, so the tiling is part of the arithmetic, and a frame must give
    // the same bits alone and inside any time batch (tests: test_forward_is_bitwise_reproducible).
    // (`nr` above depends on the launch size, hence on the batch: the tile FORM is decided with the column-block
    // grouping of the throughput rule alone -- units = "a full chip" -- which is a function of the layer only)
    const int nr_form = pick_nr(w.cout_p16 / 16, taps,
                                (in && in->stats && in->act == ACT_SILU) || gate || (se && se->pool), 1L << 40);
    const long tiles_big = (long)((a.Dout + 1) / 2) * ((a.Hout + 3) / 4) * ((a.Wout + 15) / 16) *
                           a.nphase * ((w.cout_p16 / 16 + nr_form - 1) / nr_form);
    int small = tiles_big < 16 ? 1 : 0;
    // 256-voxel tiles (4 row blocks per wave: half the weight traffic per MFMA, 2.5x
    // instead of 3.4x halo) for volumes of at least 32^3 outputs
    if (!small && d.k == 3 && d.stride == 1 && tiles_big >= 256) small = 2;
    if (JH_ENV_KNOB("JH_CONV3D_TILE") >= 0) { const int v = JH_ENV_KNOB("JH_CONV3D_TILE"); if (!small || v == 1) small = v; }
    if (d.k == 1 && d.stride == 1) return conv_launch_3d_k1(a, nr, small, budget, s);
    if (d.k == 2 && d.stride == 2) return conv_launch_3d_k2s2(a, nr, small, budget, s);
    if (d.k == 3 && d.stride <= 2) return conv_launch_3d_k3(a, d.stride, nr, small, budget, s);
  }
  JH_REQUIRE(false, "no convolution kernel for this (nd, k, stride)");
}

}  // namespace jh
