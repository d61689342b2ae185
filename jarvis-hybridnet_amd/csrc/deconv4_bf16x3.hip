// ConvTranspose2d(k = 4, s = 2, p = 1) of the keypoint head with split-bf16 operands (precision mode
// bf16x3; fp32 form: deconv4.hip; reference: nn.ConvTranspose2d, jarvis/efficienttrack/model.py:90-96).
//
// Output (2y + py, 2x + px) is a 2 x 2-tap convolution of the input: parity 0 reads inputs {y - 1, y}
// with kernel taps {3, 1}, parity 1 reads {y, y + 1} with taps {2, 0} (conv_host.hip).  A workgroup
// stages the 10 x 18 halo patch of an 8 x 16 tile of INPUT pixels once -- InstanceNorm of the producer
// applied on load, every value split into hi = bf16(v), lo = bf16(v - hi), planes of [pixel][16
// channels] per 16-channel chunk -- and each of its four waves produces ONE parity: 8 row blocks of 16
// pixels x NCB blocks of 16 output channels, K = (chunk) x (tap row a) x [tap column b, 16 channels] =
// one v_mfma_f32_16x16x32_bf16 step, three MFMAs per step (hi hi + lo hi + hi lo, fp32 accumulation).
// Row, tap-row, chunk and plane offsets of the A reads are compile-time immediates; the parity only
// moves the lane base.  Operands are requested one step ahead (see conv3d_bf16x3.hip).
#include <cstring>
#include "conv_mfma.h"

namespace jh {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct D4Args {
  const float* x;          // [N][H][W][cin_p]
  float* y;                // [N][2H][2W][cout_p] raw output
  const uint4* w;          // [parity py*2+px][chunk][a][NCB][hi, lo][64 lanes] x 8 bf16
  const double* in_stats;  // InstanceNorm (+ in_act) of the input applied on load, or nullptr
  float in_inv;
  int in_act;
  int N, H, W, cin_p, cout_p, cout_p16, nchunk;
};

namespace d4 {
constexpr int TY = 8, TX = 16, PY = TY + 2, PX = TX + 2, NPIX = PY * PX;   // 180
constexpr int PLANE = NPIX * 32;                 // bytes of one (hi or lo) plane of a 16-channel chunk
}  // namespace d4

template <int NCB>
__global__ __launch_bounds__(256) void deconv4_bf16x3_kernel(D4Args a) {
  using namespace d4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* nrm = reinterpret_cast<float*>(smem + (size_t)a.nchunk * 2 * PLANE);   // [2][cin_p]
  const BlockId bid = xcd_block();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int py = wave >> 1, px = wave & 1;
  const int tiles_x = (a.W + TX - 1) / TX;
  const int tile_x = bid.x % tiles_x, tile_y = bid.x / tiles_x;
  const int n = bid.y;
  const int oy0 = tile_y * TY, ox0 = tile_x * TX;              // in input pixels

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
    __syncthreads();
  }
  // ---- stage the whole patch (all channels): norm on load, split, [chunk][hi, lo][pixel][16] ---------
  const float* __restrict__ xin = a.x + (size_t)n * a.H * a.W * a.cin_p;
  const int q4 = a.nchunk * 4;                                  // channel quads per pixel (padded to chunks)
  for (int idx = tid; idx < NPIX * q4; idx += 256) {
    const int pix = idx / q4, cq = idx - pix * q4;
    const int pcol = pix % PX, prow = pix / PX;
    const int iy = oy0 - 1 + prow, ix = ox0 - 1 + pcol;
    const int c0 = cq * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c0 < a.cin_p) {
      v = *reinterpret_cast<const float4*>(xin + ((size_t)iy * a.W + ix) * a.cin_p + c0);
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
    const bf16x4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    const bf16x4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]),
                       (__bf16)(v.z - (float)hi[2]), (__bf16)(v.w - (float)hi[3])};
    unsigned char* dst = smem + (size_t)(cq >> 2) * 2 * PLANE + pix * 32 + (cq & 3) * 8;
    *reinterpret_cast<bf16x4*>(dst) = hi;
    *reinterpret_cast<bf16x4*>(dst + PLANE) = lo;
  }
  __syncthreads();

  // A rows of this lane: x = lane & 15, K group g: tap column b = g >> 1, channels 8 (g & 1) .. + 7;
  // patch pixel of output parity (py, px), input pixel (y, x), tap (a, b): (y + a + py, x + b + px)
  const int g = lane >> 4;
  const int abase = ((py * PX) + (lane & 15) + (g >> 1) + px) * 32 + (g & 1) * 16;
  f32x4 acc[TY][NCB];
#pragma unroll
  for (int rb = 0; rb < TY; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(a.w), 0, 4 * a.nchunk * 2 * NCB * 2 * 1024, 0x00020000);
  const int wph = wave * a.nchunk * 2 * NCB * 2 * 1024;

  bf16x8 bh[2][NCB], bl[2][NCB], ah[2], al[2];
  auto load_b = [&](int step, int buf) __attribute__((always_inline)) {     // step = chunk * 2 + a
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const int so = wph + (step * NCB + cb) * 2 * 1024;
      bh[buf][cb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so, 0));
      bl[buf][cb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so + 1024, 0));
    }
  };
  auto load_a = [&](int chunk, int ta, int rb, int buf) __attribute__((always_inline)) {
    const unsigned char* p = smem + (size_t)chunk * 2 * PLANE + abase + (rb + ta) * PX * 32;
    ah[buf] = *reinterpret_cast<const bf16x8*>(p);
    al[buf] = *reinterpret_cast<const bf16x8*>(p + PLANE);
  };
  load_b(0, 0);
  load_a(0, 0, 0, 0);
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
#pragma unroll
    for (int ta = 0; ta < 2; ++ta) {
      const int step = chunk * 2 + ta, cur = ta;               // (two steps per chunk: buffer = ta)
      if (step + 1 < a.nchunk * 2) load_b(step + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rb = 0; rb < TY; ++rb) {
        const int i = ta * TY + rb;                             // A buffer parity inside the chunk
        if (rb + 1 < TY) load_a(chunk, ta, rb + 1, (i + 1) & 1);
        else if (ta == 0) load_a(chunk, 1, 0, (i + 1) & 1);
        else if (chunk + 1 < a.nchunk) load_a(chunk + 1, 0, 0, (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bh[cur][cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i & 1], bh[cur][cb], acc[rb][cb], 0, 0, 0);
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb)
          acc[rb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i & 1], bl[cur][cb], acc[rb][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  // ---- this wave's parity: output pixel (2 y + py, 2 x + px) -----------------------------------------
  EpilogueArgs e;
  e.y = a.y + (size_t)n * (2 * a.H) * (2 * a.W) * a.cout_p;
  e.bias = nullptr;
  e.stats = nullptr;
  e.Dout = 1; e.Hout = a.H; e.Wout = a.W; e.Hy = 2 * a.H; e.Wy = 2 * a.W;
  e.cout_p = a.cout_p; e.cout_p16 = a.cout_p16; e.os = 2; e.offz = 0; e.offy = py; e.offx = px; e.osz = 1;
  conv_epilogue<TY, NCB, TY, TX, 1, false>(acc, e, nullptr, 0, 0, oy0, ox0, lane);
}

static inline unsigned short d4_bf16_rne(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float d4_bf16_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}
// kernel tap that feeds tap t of parity par (conv_host.hip: deconv4_tap)
static inline int d4_tap(int par, int t) { return par ? (t == 0 ? 2 : 0) : (t == 0 ? 3 : 1); }

// torch layout (cin, cout, 4, 4) -> [parity][chunk][a][cout block][hi, lo][lane][8] bf16
int pack_deconv4_bf16x3_weights(int cin, int cout, const float* w, ConvWeights* out) {
  const int cin_p = cpad(cin), cout_p16 = round_up(cout, 16);
  const int nchunk = (cin_p + 15) / 16, ncb = cout_p16 / 16;
  std::vector<unsigned short> packed((size_t)4 * nchunk * 2 * ncb * 2 * 512, 0);
  for (int ph = 0; ph < 4; ++ph)
    for (int chunk = 0; chunk < nchunk; ++chunk)
      for (int ta = 0; ta < 2; ++ta)
        for (int cb = 0; cb < ncb; ++cb)
          for (int lane = 0; lane < 64; ++lane)
            for (int e = 0; e < 8; ++e) {
              const int g = lane >> 4, co = cb * 16 + (lane & 15);
              const int tb = g >> 1, ci = chunk * 16 + (g & 1) * 8 + e;
              if (ci >= cin || co >= cout) continue;
              const int ky = d4_tap(ph >> 1, ta), kx = d4_tap(ph & 1, tb);
              const float v = w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
              const unsigned short hi = d4_bf16_rne(v), lo = d4_bf16_rne(v - d4_bf16_f32(hi));
              const size_t base = ((((size_t)(ph * nchunk + chunk) * 2 + ta) * ncb + cb) * 2) * 512 + (size_t)lane * 8 + e;
              packed[base] = hi;
              packed[base + 512] = lo;
            }
  out->cin_p = cin_p; out->cout_p16 = cout_p16; out->phase_stride = packed.size() / 2;
  void* dev = nullptr;
  JH_CHECK_HIP(hipMalloc(&dev, packed.size() * sizeof(unsigned short)));
  JH_CHECK_HIP(hipMemcpy(dev, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  out->w = static_cast<float*>(dev);
  out->bias = nullptr;
  return 0;
}

bool deconv4_bf16x3_eligible(int cout) { const int ncb = round_up(cout, 16) / 16; return ncb == 1 || ncb == 2; }

int launch_deconv4_bf16x3(const ConvWeights& w, const Act& x, const Act& y, hipStream_t s, const InNorm* in) {
  using namespace d4;
  JH_REQUIRE(x.Cp == w.cin_p && y.H == 2 * x.H && y.W == 2 * x.W && x.N == y.N && x.D == 1, "bf16x3 deconv shape");
  D4Args a{};
  a.x = x.p; a.y = y.p; a.w = reinterpret_cast<const uint4*>(w.w);
  a.in_stats = in ? in->stats : nullptr; a.in_inv = in ? in->inv : 0.f; a.in_act = in ? in->act : 0;
  a.N = x.N; a.H = x.H; a.W = x.W; a.cin_p = x.Cp; a.cout_p = y.Cp; a.cout_p16 = w.cout_p16;
  a.nchunk = (x.Cp + 15) / 16;
  const size_t lds = (size_t)a.nchunk * 2 * PLANE + (size_t)2 * a.cin_p * sizeof(float);
  JH_REQUIRE(lds <= 150 * 1024, "bf16x3 deconv: too many input channels for one patch");
  dim3 grid(((x.W + TX - 1) / TX) * ((x.H + TY - 1) / TY), x.N);
  const int ncb = w.cout_p16 / 16;
  auto k1 = deconv4_bf16x3_kernel<1>;
  auto k2 = deconv4_bf16x3_kernel<2>;
  static bool big = false;
  if (!big) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  if (ncb == 1) hipLaunchKernelGGL(k1, grid, dim3(256), lds, s, a);
  else if (ncb == 2) hipLaunchKernelGGL(k2, grid, dim3(256), lds, s, a);
  else JH_REQUIRE(false, "bf16x3 deconv: at most 32 output channels");
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace jh
