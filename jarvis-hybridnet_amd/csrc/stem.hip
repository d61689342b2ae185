// Stem convolution of the EfficientNet trunk on the vector ALUs:
// Conv2d(3 -> CO, k3, s2, p1, bias=False)  (jarvis/efficienttrack/efficientnet.py:150-152,
// model.py:536-538; CO = 16 for the `small` model -- round_filters(32, 0.5) --, 32 for `medium` and `large`:
// round 4, so that the reference's default model size gets the fused pre-processing too).
//
// On the MFMA path the 3 input channels are padded to 8 (K = 72 for 27 real taps) and every
// workgroup spends several hundred staging / epilogue instructions on 36 MFMAs: 16 TFLOP/s, a
// third of the HBM rate (2 MB per image in + out).  With K = 27 and N = 16 the op is a small dot
// product per output pixel, so it runs as plain packed FMAs:
//
//   * 256 threads own a 16 x 16 tile of OUTPUT pixels; the 33 x 33 input halo tile (one float4
//     r, g, b, 0 per pixel) is staged once in LDS, zero padding = out-of-range buffer loads;
//   * a thread produces all 16 channels of one pixel: 9 taps x 3 channels x 16 packed-FMA lanes,
//     the weights are LDS broadcasts ([tap][cin][16] floats);
//   * MODE 1 / 2 (round 3): the input pixels are not read from a pre-processed tensor but computed
//     while the patch is staged -- the bilinear resize (CenterDetect) or the bounding-box crop
//     (KeypointDetect) of the frames plus the normalisation, with the arithmetic of the stand-alone
//     kernels (preprocess.h) -- so the 1 MB per image that `preprocess_resize` / `preprocess_crop` wrote
//     and this kernel read back never exists;
//   * raw output in the library's channel-last layout (four 16-byte stores per thread), and the
//     per-(n, channel) sum / sum of squares of the tile for the InstanceNorm that follows
//     (order-independent accumulation, jh_common.h).
#include "jh_common.h"
#include "preprocess.h"

namespace jh {

namespace {
constexpr int kStT = 16, kStP = 2 * kStT + 1;       // output tile side, input patch side
typedef float sf2 __attribute__((ext_vector_type(2)));
typedef float sf4 __attribute__((ext_vector_type(4)));
}  // namespace

struct StemSrcArgs {           // MODE != 0: the frames this launch pre-processes on the fly
  const void* frames;
  const void* const* frames_cell;
  const int* center_hm;
  int Cloc, C, cam0, FH, FW;
  float sy, sx;
  float3 mean, stdv;
};

template <int MODE, int SRC, int CO = 16>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x,
                                                        const float* __restrict__ w /* [9][3][CO] */,
                                                        float* __restrict__ y, double* __restrict__ stats,
                                                        int H, int W, StemSrcArgs sa) {
  static_assert(CO == 16 || CO == 32, "stem output channels");
  // (the statistics scratch lies over the patch, which is dead once every thread has its 16 sums: 19.5 instead of
  //  35.8 KB of LDS = twice the workgroups per CU; the kernel is bound by the latency of its patch loads)
  __shared__ __attribute__((aligned(16))) float4 patch[kStP * kStP > CO * 64 ? kStP * kStP : CO * 64];
  float* red = reinterpret_cast<float*>(patch);      // statistics: [channel][thread]
  __shared__ double red2[2 * CO * 16];               // fp64 partials (jh_common.h: stat_acc)
  const int tid = threadIdx.x;
  const int Ho = H >> 1, Wo = W >> 1;
  const int tiles_x = (Wo + kStT - 1) / kStT;
  const BlockId bid = xcd_block();
  const int oy0 = (bid.x / tiles_x) * kStT, ox0 = (bid.x % tiles_x) * kStT;
  const int n = bid.y;
  // input patch: rows 2 oy0 - 1 .. 2 oy0 + 31, same in x
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(x + (size_t)n * H * W * 4), 0, H * W * 16, 0x00020000);
  const void* frames = sa.frames;
  int ccx = 0, ccy = 0;
  if (MODE != 0) {
    if (sa.frames_cell) frames = *sa.frames_cell;
    if (MODE == 2) {
      const int t = n / sa.Cloc, cl = n - t * sa.Cloc;
      ccx = sa.center_hm[(t * sa.C + sa.cam0 + cl) * 2 + 0];
      ccy = sa.center_hm[(t * sa.C + sa.cam0 + cl) * 2 + 1];
    }
  }
  for (int i = tid; i < kStP * kStP; i += 256) {
    const int py = i / kStP, px = i - py * kStP;
    const int iy = 2 * oy0 - 1 + py, ix = 2 * ox0 - 1 + px;
    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
    if (MODE == 0) {
      const int off = ok ? (iy * W + ix) * 16 : (int)0x80000000;
      const sf4 v = __builtin_bit_cast(sf4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
      patch[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      // (the convolution's zero padding lies outside the pre-processed image: zeros, not normalised zeros)
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok)
        v = MODE == 1 ? resize_px<SRC>(frames, n, iy, ix, sa.FH, sa.FW, sa.sy, sa.sx, sa.mean, sa.stdv)
                      : crop_px<SRC>(frames, n, ccx, ccy, iy, ix, sa.FH, sa.FW, H, sa.mean, sa.stdv);
      patch[i] = v;
    }
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  sf2 acc[CO / 2];
#pragma unroll
  for (int i = 0; i < CO / 2; ++i) acc[i] = (sf2){0.f, 0.f};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const float4 in = patch[(2 * ty + ky) * kStP + 2 * tx + kx];
      const float iv[3] = {in.x, in.y, in.z};
#pragma unroll
      for (int ci = 0; ci < 3; ++ci) {
        // (uniform address: the weights come through the scalar cache into SGPRs, not as 108 LDS
        //  broadcast reads per thread -- the kernel was LDS-bandwidth-bound on them)
        const float4* wq = reinterpret_cast<const float4*>(w + ((ky * 3 + kx) * 3 + ci) * CO);
        const sf2 xv = (sf2){iv[ci], iv[ci]};
#pragma unroll
        for (int q4 = 0; q4 < CO / 4; ++q4) {
          const float4 wv = wq[q4];
          acc[q4 * 2 + 0] = __builtin_elementwise_fma(xv, (sf2){wv.x, wv.y}, acc[q4 * 2 + 0]);
          acc[q4 * 2 + 1] = __builtin_elementwise_fma(xv, (sf2){wv.z, wv.w}, acc[q4 * 2 + 1]);
        }
      }
    }
  const int oy = oy0 + ty, ox = ox0 + tx;
  const bool in_img = oy < Ho && ox < Wo;
  if (in_img) {
    float4* dst = reinterpret_cast<float4*>(y + (((size_t)n * Ho + oy) * Wo + ox) * CO);
#pragma unroll
    for (int q4 = 0; q4 < CO / 4; ++q4)
      dst[q4] = make_float4(acc[q4 * 2][0], acc[q4 * 2][1], acc[q4 * 2 + 1][0], acc[q4 * 2 + 1][1]);
  }
  if (stats) {
    // [channel][thread] (consecutive threads -> consecutive banks), then 16 threads per channel
    // add 16 values each, then one thread per channel adds the 16 partials: a fixed order
    __syncthreads();                                 // (everyone is done reading the patch)
#pragma unroll
    for (int c = 0; c < CO; ++c) red[c * 256 + tid] = in_img ? acc[c >> 1][c & 1] : 0.f;
    __syncthreads();
#pragma unroll
    for (int cg = 0; cg < CO / 16; ++cg) {             // (16 channels x 16 parts per round of the 256 threads)
      const int c = cg * 16 + (tid >> 4), part = tid & 15;
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) stat_acc(s1, s2, red[c * 256 + part * 16 + i]);
      red2[(c * 16 + part) * 2 + 0] = s1;
      red2[(c * 16 + part) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < CO) {
      double s1 = 0.0, s2 = 0.0;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s1 += red2[(tid * 16 + i) * 2 + 0];
        s2 += red2[(tid * 16 + i) * 2 + 1];
      }
      stat_add(stats + ((size_t)n * CO + tid) * kStatW, s1, s2);
    }
  }
}

// w_host: torch layout (cout, 3, 3, 3) = [cout][cin][ky][kx]; w_dev receives [tap][cin][cout]
void pack_stem_weights(const float* w_host, float* packed /* 27 * cout floats */, int cout) {
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < 3; ++ci)
      for (int t = 0; t < 9; ++t) packed[(t * 3 + ci) * cout + co] = w_host[(co * 3 + ci) * 9 + t];
}

int launch_stem_conv(const Act& x, const float* w_dev, const Act& y, double* stats, hipStream_t s) {
  JH_REQUIRE(x.Cp == 4 && (y.Cp == 16 || y.Cp == 32) && x.D == 1 && y.H * 2 == x.H && y.W * 2 == x.W && x.N == y.N,
             "stem convolution shapes");
  JH_REQUIRE((size_t)x.H * x.W * 16 < ((size_t)1 << 31), "stem input too large");
  const int tiles = ((y.H + kStT - 1) / kStT) * ((y.W + kStT - 1) / kStT);
  if (y.Cp == 32)
    hipLaunchKernelGGL((stem_conv_kernel<0, 0, 32>), dim3(tiles, x.N), dim3(256), 0, s, x.p, w_dev, y.p, stats, x.H,
                       x.W, StemSrcArgs{});
  else
    hipLaunchKernelGGL((stem_conv_kernel<0, 0, 16>), dim3(tiles, x.N), dim3(256), 0, s, x.p, w_dev, y.p, stats, x.H,
                       x.W, StemSrcArgs{});
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// The same convolution fed by the frames themselves (StemSource: resize or crop + normalise fused into the
// patch staging); x only carries the shape of the image that would have been materialised.
int launch_stem_conv_src(const StemSource& src, const Act& x, const float* w_dev, const Act& y, double* stats,
                         hipStream_t s) {
  JH_REQUIRE(src.mode == 1 || src.mode == 2, "stem source mode");
  JH_REQUIRE(x.Cp == 4 && (y.Cp == 16 || y.Cp == 32) && x.D == 1 && y.H * 2 == x.H && y.W * 2 == x.W && x.N == y.N &&
                 x.H == x.W, "stem convolution shapes");
  StemSrcArgs sa{};
  sa.frames = src.frames; sa.frames_cell = src.frames_cell; sa.center_hm = src.center_hm;
  sa.Cloc = src.Cloc; sa.C = src.C; sa.cam0 = src.cam0; sa.FH = src.H; sa.FW = src.W;
  sa.sy = (float)src.H / (float)x.H; sa.sx = (float)src.W / (float)x.W;        // (as launch_preprocess_resize)
  sa.mean = make_float3(src.mean[0], src.mean[1], src.mean[2]);
  sa.stdv = make_float3(src.stdv[0], src.stdv[1], src.stdv[2]);
  const int tiles = ((y.H + kStT - 1) / kStT) * ((y.W + kStT - 1) / kStT);
  const dim3 grid(tiles, x.N);
  // (the fp32 resize form is bound by HBM on the frames it reads -- 2 of every 4 rows of 1280 x 1024 x 3 floats --
  //  and loses locality with more workgroups in flight: 654 -> 711 us at eight per CU; 16 KB of unused dynamic LDS
  //  keep it at four.  The other forms are latency-bound and want the eight.)
  const size_t pad_f32 = JH_ENV_KNOB("JH_STEM_PAD_KB") >= 0 ? (size_t)JH_ENV_KNOB("JH_STEM_PAD_KB") * 1024 : 16384;
#define JH_STEM(M, U)                                                                                                   \
  do {                                                                                                                  \
    if (y.Cp == 32)                                                                                                     \
      hipLaunchKernelGGL((stem_conv_kernel<M, U, 32>), grid, dim3(256), 0, s, x.p, w_dev, y.p, stats, x.H, x.W, sa);    \
    else                                                                                                                \
      hipLaunchKernelGGL((stem_conv_kernel<M, U, 16>), grid, dim3(256), (M == 1 && U == 0) ? pad_f32 : 0, s, x.p,       \
                         w_dev, y.p, stats, x.H, x.W, sa);                                                              \
  } while (0)
  if (src.mode == 1) { if (src.src_u8) JH_STEM(1, 1); else JH_STEM(1, 0); }
  else { if (src.src_u8) JH_STEM(2, 1); else JH_STEM(2, 0); }
#undef JH_STEM
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace jh
