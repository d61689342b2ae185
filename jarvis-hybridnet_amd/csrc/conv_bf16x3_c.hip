// explicit instantiations of conv_bf16x3.h (split for parallel compilation): 2D k5 s2, host side
#include "conv_bf16x3.h"
namespace jh {
JH_XCONV_DEFINE(2, 5, 2, 1, 8, 1)
JH_XCONV_DEFINE(2, 5, 2, 1, 8, 2)

template <> int launch_xconv_ncb<3, 3, 2, 2, 4, 1>(const XArgs&, hipStream_t);
template <> int launch_xconv_ncb<2, 3, 2, 1, 16, 1>(const XArgs&, hipStream_t);
template <> int launch_xconv_ncb<2, 3, 1, 1, 16, 1>(const XArgs&, hipStream_t);
template <> int launch_xconv_ncb<2, 3, 1, 1, 16, 2>(const XArgs&, hipStream_t);

static inline unsigned short x_bf16_rne(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float x_bf16_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// which (nd, k, stride) have a split-bf16 kernel, and with which channel chunk
static int xconv_ch8(const ConvDesc& d) {
  const int cin_p = cpad(d.cin);
  if (d.ostride != 1 || d.nphase != 1) return 0;
  if (d.nd == 3 && d.k == 3 && d.stride == 2) return 1;
  if (d.nd == 2 && d.k == 3 && d.stride == 2) return 1;
  if (d.nd == 2 && d.k == 3 && d.stride == 1) return cin_p % 16 == 0 ? 2 : 1;
  if (d.nd == 2 && d.k == 5 && d.stride == 2) return cin_p % 16 == 0 ? 2 : 1;
  return 0;
}
bool conv_bf16x3_eligible(const ConvDesc& d) { return xconv_ch8(d) != 0; }

// torch layout (cout, cin, k..) -> [chunk][slice][cout block][hi, lo][lane][8] bf16
int pack_conv_bf16x3_weights(const ConvDesc& d, const float* w, const float* b, ConvWeights* out) {
  const int ch8 = xconv_ch8(d);
  JH_REQUIRE(ch8 != 0, "no split-bf16 kernel for this convolution");
  const int cin_p = cpad(d.cin), cout_p16 = round_up(d.cout, 16);
  const int chk = ch8 * 8, tps = 4 / ch8;
  const int ntap = d.k * d.k * (d.nd == 3 ? d.k : 1);
  const int nsl = (ntap + tps - 1) / tps, nchunk = (cin_p + chk - 1) / chk, ncbt = cout_p16 / 16;
  std::vector<unsigned short> packed((size_t)nchunk * nsl * ncbt * 2 * 512, 0);
  for (int chunk = 0; chunk < nchunk; ++chunk)
    for (int s = 0; s < nsl; ++s)
      for (int cb = 0; cb < ncbt; ++cb)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int g = lane >> 4, co = cb * 16 + (lane & 15);
            const int tap = s * tps + g / ch8, ci = chunk * chk + (g % ch8) * 8 + e;
            if (tap >= ntap || ci >= d.cin || co >= d.cout) continue;
            const float v = w[((size_t)co * d.cin + ci) * ntap + tap];
            const unsigned short hi = x_bf16_rne(v), lo = x_bf16_rne(v - x_bf16_f32(hi));
            const size_t base = ((((size_t)chunk * nsl + s) * ncbt + cb) * 2) * 512 + (size_t)lane * 8 + e;
            packed[base] = hi;
            packed[base + 512] = lo;
          }
  out->cin_p = cin_p; out->cout_p16 = cout_p16; out->phase_stride = packed.size() / 2;
  void* dev = nullptr;
  JH_CHECK_HIP(hipMalloc(&dev, packed.size() * sizeof(unsigned short)));
  JH_CHECK_HIP(hipMemcpy(dev, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  out->w = static_cast<float*>(dev);
  out->bias = nullptr;
  if (b) {
    std::vector<float> bp(cout_p16, 0.f);
    for (int i = 0; i < d.cout; ++i) bp[i] = b[i];
    JH_CHECK_HIP(hipMalloc(&out->bias, bp.size() * sizeof(float)));
    JH_CHECK_HIP(hipMemcpy(out->bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

int launch_conv_bf16x3(const ConvDesc& d, const ConvWeights& w, const Act& x, const Act& y, double* stats,
                       hipStream_t s, const InNorm* in) {
  const int ch8 = xconv_ch8(d);
  JH_REQUIRE(ch8 != 0 && x.Cp == w.cin_p && y.Cp <= w.cout_p16 && x.N == y.N, "bf16x3 conv shape");
  XArgs a{};
  a.x = x.p; a.y = y.p; a.w = reinterpret_cast<const uint4*>(w.w); a.bias = w.bias;
  a.in_stats = in ? in->stats : nullptr; a.in_inv = in ? in->inv : 0.f; a.in_act = in ? in->act : 0;
  a.stats = stats;
  a.N = x.N; a.Din = x.D; a.Hin = x.H; a.Win = x.W; a.cin_p = x.Cp;
  a.Dout = y.D; a.Hout = y.H; a.Wout = y.W; a.cout_p = y.Cp; a.cout_p16 = w.cout_p16;
  a.nchunk = (x.Cp + ch8 * 8 - 1) / (ch8 * 8); a.ncbt = w.cout_p16 / 16; a.pad = d.phase[0].pad[2];
  a.tiles_x = (y.W + 15) / 16;
  if (d.nd == 3) {
    a.tiles_y = (y.H + 3) / 4; a.tiles_z = (y.D + 1) / 2;
    return launch_xconv_ncb<3, 3, 2, 2, 4, 1>(a, s);
  }
  a.tiles_z = 1;
  if (d.k == 5) {
    a.tiles_y = (y.H + 7) / 8;
    return ch8 == 2 ? launch_xconv_ncb<2, 5, 2, 1, 8, 2>(a, s) : launch_xconv_ncb<2, 5, 2, 1, 8, 1>(a, s);
  }
  a.tiles_y = (y.H + 15) / 16;
  if (d.stride == 2) return launch_xconv_ncb<2, 3, 2, 1, 16, 1>(a, s);
  return ch8 == 2 ? launch_xconv_ncb<2, 3, 1, 1, 16, 2>(a, s) : launch_xconv_ncb<2, 3, 1, 1, 16, 1>(a, s);
}

}  // namespace jh
