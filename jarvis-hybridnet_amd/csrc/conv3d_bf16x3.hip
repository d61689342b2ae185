// 3x3x3 stride-1 convolution of V2V with split-bf16 operands (the opt-in reduced-precision mode
// `bf16x3`; the fp32 Winograd kernels stay the default and the parity mode).
//
// Replaces, in that mode, the Res3DBlock convolutions of jarvis/hybridnet/v2vnet.py:30-37
// (the reference's own fast path runs them in half precision: jarvis/prediction/jarvis3D.py:93,
// 107,122, enabled_precisions={torch.half}).
//
// Every fp32 operand v is split into hi = bf16(v) and lo = bf16(v - hi) (round to nearest even)
// and a product a * b is taken as a_hi b_hi + a_lo b_hi + a_hi b_lo on the bf16 matrix cores
// with fp32 accumulation: what is dropped is a_lo b_lo and the second-order remainders, about
// 2^-16 relative per product, against 2^-24 of the fp32 MFMA and 2^-9 of plain bf16.  Three
// v_mfma_f32_16x16x32_bf16 (16 cycles, 8192 MACs each) do the work of eight
// v_mfma_f32_16x16x4_f32 (32 cycles, 1024 MACs each): 48 against 256 matrix-core cycles, so the
// direct 27-tap form here needs 0.42 x the matrix time of the fp32 Winograd kernel (12 / 27 of the
// direct multiplies) -- and no transforms.
//
// Mapping: a workgroup (4 waves) owns a 4 x 4 x 16 (z, y, x) tile of output voxels, wave w the
// z slice w: four 16-voxel row blocks (the y rows) x NCB blocks of 16 output channels.  K runs
// over (chunk of 16 input channels) x (14 slices of two taps x 16 channels = 32 = one MFMA K);
// tap 27 is a zero pad.  Per chunk the 6 x 6 x 18 halo patch is staged in LDS as two planes
// (hi, lo) of [pixel][16 channels] bf16 -- 32 bytes per pixel, so the 16 rows of an A operand
// (16 consecutive x) fall on distinct 16-byte slots -- with the InstanceNorm (+ReLU) of the
// producer applied on load, exactly as the fp32 kernels do.  Weights are split and packed on the
// host into per-lane B-operand order (one coalesced 1 KB read per slice, column block and half).
// Epilogue = the fp32 kernels' (bias, fused statistics, 16-byte channel-last stores).
#include <cstring>
#include "conv_mfma.h"

namespace jh {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

struct B3Args {
  const float* x;          // [N][D][H][W][cin_p]
  float* y;                // [N][D][H][W][cout_p] raw output
  const uint4* w;          // [chunk][14 slices][NCBT][hi, lo][64 lanes] x 8 bf16
  const float* bias;       // [cout_p16] or nullptr
  const double* in_stats;  // InstanceNorm (+ in_act) of the input applied on load, or nullptr
  float in_inv;
  int in_act;
  double* stats;           // [N][cout_p][kStatW] or nullptr
  int N, D, H, W, cin_p, cout_p, cout_p16, nchunk, ncbt;
  int tiles_x, tiles_y, tiles_z;
};

namespace b3 {
constexpr int TZ = 4, TY = 4, TX = 16, PZ = TZ + 2, PY = TY + 2, PX = TX + 2;
constexpr int NPIX = PZ * PY * PX;               // 648
constexpr int NSLICE = 14;                       // 27 taps, two per slice (tap 27 = zero weights)
constexpr int PLANE = NPIX * 32;                 // bytes of the hi (or lo) plane of a 16-channel chunk
constexpr int tap_off(int t) { return ((t / 9) * PY + (t / 3) % 3) * PX + t % 3; }
// pixel distance from tap 2s to tap 2s+1: +1 (kind 0), next row (1), next plane (2)
constexpr int pair_kind(int s) { return (2 * s) % 3 < 2 ? 0 : ((2 * s) / 3 % 3 < 2 ? 1 : 2); }
constexpr int kind_delta(int k) { return k == 0 ? 1 : (k == 1 ? PX - 2 : PY * PX - 2 * PX - 2); }
}  // namespace b3

template <int NCB>
__global__ __launch_bounds__(256) void conv3d_bf16x3_kernel(B3Args a) {
  using namespace b3;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* nrm = reinterpret_cast<float*>(smem + 2 * PLANE);          // [2][cin_p]: -mean * rstd, rstd
  float* red = nrm + 2 * a.cin_p;                                   // epilogue reduction space
  const BlockId bid = xcd_block();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = bid.y, nb0 = bid.z * NCB;
  const int tile = (int)bid.x;
  const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, tz = tile / (a.tiles_x * a.tiles_y);
  const int oz0 = tz * TZ, oy0 = ty * TY, ox0 = tx * TX;

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
  }
  const float* __restrict__ xin = a.x + (size_t)n * a.D * a.H * a.W * a.cin_p;

  // A-operand addresses of this lane: row block rb = y row, row = x = lane & 15, K group g = lane >> 4:
  // tap (g >> 1) of the slice's pair, channels 8 (g & 1) .. + 7 of the chunk
  // (row block and tap offsets are compile-time immediates of the LDS reads)
  const int g = lane >> 4;
  int abase[4];        // [kind 0..2], [3]: both K-group pairs on the SAME tap (the last slice's phantom tap)
#pragma unroll
  for (int k = 0; k < 4; ++k)
    abase[k] = (((wave * PY) * PX + (lane & 15)) + ((g >> 1) && k < 3 ? kind_delta(k) : 0)) * 32 + (g & 1) * 16;

  f32x4 acc[TY][NCB];
#pragma unroll
  for (int rb = 0; rb < TY; ++rb)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) acc[rb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<uint4*>(a.w), 0, a.nchunk * NSLICE * a.ncbt * 2 * 1024, 0x00020000);

  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();                      // the previous chunk's operand reads (and the nrm table) are done
    // ---- stage the halo patch of this chunk: norm (+ReLU) on load, split, two planes -------------
    for (int idx = tid; idx < NPIX * 4; idx += 256) {
      const int pix = idx >> 2, q = idx & 3;
      const int px = pix % PX, py = (pix / PX) % PY, pz = pix / (PX * PY);
      const int iz = oz0 - 1 + pz, iy = oy0 - 1 + py, ix = ox0 - 1 + px;
      const int c0 = chunk * 16 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c0 < a.cin_p) {
        v = *reinterpret_cast<const float4*>(xin + ((size_t)(iz * a.H + iy) * a.W + ix) * a.cin_p + c0);
        if (a.in_stats) {
          const float4 m = *reinterpret_cast<const float4*>(nrm + c0);
          const float4 r = *reinterpret_cast<const float4*>(nrm + a.cin_p + c0);
          v.x = __fmaf_rn(v.x, r.x, m.x); v.y = __fmaf_rn(v.y, r.y, m.y);
          v.z = __fmaf_rn(v.z, r.z, m.z); v.w = __fmaf_rn(v.w, r.w, m.w);
          if (a.in_act == ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          }
        }
      }
      const bf16x4 hi = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
      const bf16x4 lo = {(__bf16)(v.x - (float)hi[0]), (__bf16)(v.y - (float)hi[1]),
                         (__bf16)(v.z - (float)hi[2]), (__bf16)(v.w - (float)hi[3])};
      *reinterpret_cast<bf16x4*>(smem + pix * 32 + q * 8) = hi;
      *reinterpret_cast<bf16x4*>(smem + PLANE + pix * 32 + q * 8) = lo;
    }
    __syncthreads();
    // ---- 14 slices x (TY x NCB) x 3 MFMAs; operands one step ahead: the weights of slice s+1 are
    // requested before the MFMAs of slice s, the A rows of row block rb+1 before those of rb ------------
    const int wchunk = chunk * NSLICE * a.ncbt * 2 * 1024;
    bf16x8 bh[2][NCB], bl[2][NCB];
    auto load_b = [&](int s, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int so = wchunk + ((s * a.ncbt + nb0 + cb) * 2) * 1024;
        bh[buf][cb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so, 0));
        bl[buf][cb] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, so + 1024, 0));
      }
    };
    // A rows of step i = slice * TY + row block (compile-time addresses: tap pair, row block, plane)
    bf16x8 ah[2], al[2];
    auto load_a = [&](int i, int buf) __attribute__((always_inline)) {
      const int s = i / TY, rb = i % TY;
      // (last slice: its second tap does not exist -- zero weights; its K groups re-read tap 26, any
      // FINITE values: one pixel further would leave the patch for uninitialised LDS, and 0 x NaN = NaN)
      const int ab = abase[(2 * s + 1 < 27) ? pair_kind(s) : 3];
      const int off = tap_off(2 * s) * 32 + rb * PX * 32;
      ah[buf] = *reinterpret_cast<const bf16x8*>(smem + ab + off);
      al[buf] = *reinterpret_cast<const bf16x8*>(smem + PLANE + ab + off);
    };
    load_b(0, 0);
    load_a(0, 0);
#pragma unroll
    for (int s = 0; s < NSLICE; ++s) {
      const int cur = s & 1;
      if (s + 1 < NSLICE) load_b(s + 1, cur ^ 1);
      __builtin_amdgcn_sched_barrier(0);               // the weight prefetch stays AHEAD of this slice's MFMAs
#pragma unroll
      for (int rb = 0; rb < TY; ++rb) {
        const int i = s * TY + rb;
        if (i + 1 < NSLICE * TY) load_a(i + 1, (i + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);             // ... and the next A rows ahead of this block's
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
  __syncthreads();                         // (red overlaps nothing, but the nrm reads of slow waves are done)
  EpilogueArgs e;
  e.y = a.y + (size_t)n * a.D * a.H * a.W * a.cout_p;
  e.bias = a.bias;
  e.stats = a.stats ? a.stats + (size_t)n * a.cout_p * kStatW : nullptr;
  e.Dout = a.D; e.Hout = a.H; e.Wout = a.W; e.Hy = a.H; e.Wy = a.W;
  e.cout_p = a.cout_p; e.cout_p16 = a.cout_p16; e.os = 1; e.offz = e.offy = e.offx = 0; e.osz = 1;
  conv_epilogue<TY, NCB, TY, TX, 4, false>(acc, e, red, nb0, oz0, oy0, ox0, tid);
}

static inline unsigned short bf16_rne(float f) {
  unsigned u;
  std::memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static inline float bf16_to_f32(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// torch layout (cout, cin, 3, 3, 3) -> [chunk][slice][cout block][hi, lo][lane][8] bf16:
// lane = (K group g) * 16 + (cout % 16); element e of K group g = tap 2 slice + (g >> 1),
// input channel 16 chunk + 8 (g & 1) + e
int pack_bf16x3_weights(int cin, int cout, const float* w, const float* b, ConvWeights* out) {
  using namespace b3;
  const int cin_p = cpad(cin), cout_p16 = round_up(cout, 16);
  const int nchunk = (cin_p + 15) / 16, ncbt = cout_p16 / 16;
  std::vector<unsigned short> packed((size_t)nchunk * NSLICE * ncbt * 2 * 64 * 8, 0);
  for (int chunk = 0; chunk < nchunk; ++chunk)
    for (int s = 0; s < NSLICE; ++s)
      for (int cb = 0; cb < ncbt; ++cb)
        for (int lane = 0; lane < 64; ++lane)
          for (int e = 0; e < 8; ++e) {
            const int g = lane >> 4, co = cb * 16 + (lane & 15);
            const int tap = 2 * s + (g >> 1), ci = chunk * 16 + (g & 1) * 8 + e;
            if (tap >= 27 || ci >= cin || co >= cout) continue;
            const float v = w[((size_t)co * cin + ci) * 27 + tap];
            const unsigned short hi = bf16_rne(v), lo = bf16_rne(v - bf16_to_f32(hi));
            const size_t base = ((((size_t)chunk * NSLICE + s) * ncbt + cb) * 2) * 512 + (size_t)lane * 8 + e;
            packed[base] = hi;
            packed[base + 512] = lo;
          }
  out->cin_p = cin_p; out->cout_p16 = cout_p16; out->phase_stride = packed.size() / 2;   // (in floats)
  void* dev = nullptr;
  JH_CHECK_HIP(hipMalloc(&dev, packed.size() * sizeof(unsigned short)));
  JH_CHECK_HIP(hipMemcpy(dev, packed.data(), packed.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
  out->w = static_cast<float*>(dev);
  out->bias = nullptr;
  if (b) {
    std::vector<float> bp(cout_p16, 0.f);
    for (int i = 0; i < cout; ++i) bp[i] = b[i];
    JH_CHECK_HIP(hipMalloc(&out->bias, bp.size() * sizeof(float)));
    JH_CHECK_HIP(hipMemcpy(out->bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

template <int NCB>
static int launch_b3(const B3Args& a, int groups, hipStream_t s) {
  using namespace b3;
  const size_t lds = (size_t)2 * PLANE + (size_t)2 * a.cin_p * sizeof(float) + (size_t)4 * NCB * 16 * 2 * sizeof(double);
  dim3 grid(a.tiles_x * a.tiles_y * a.tiles_z, a.N, groups);
  hipLaunchKernelGGL(conv3d_bf16x3_kernel<NCB>, grid, dim3(256), lds, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_conv3d_bf16x3(const ConvWeights& w, const Act& x, const Act& y, double* stats, hipStream_t s,
                         const InNorm* in) {
  using namespace b3;
  JH_REQUIRE(x.Cp == w.cin_p && y.Cp <= w.cout_p16 && x.D == y.D && x.H == y.H && x.W == y.W && x.N == y.N,
             "bf16x3 conv shape");
  B3Args a{};
  a.x = x.p; a.y = y.p; a.w = reinterpret_cast<const uint4*>(w.w); a.bias = w.bias;
  a.in_stats = in ? in->stats : nullptr; a.in_inv = in ? in->inv : 0.f; a.in_act = in ? in->act : 0;
  a.stats = stats;
  a.N = x.N; a.D = x.D; a.H = x.H; a.W = x.W; a.cin_p = x.Cp; a.cout_p = y.Cp; a.cout_p16 = w.cout_p16;
  a.nchunk = (x.Cp + 15) / 16; a.ncbt = w.cout_p16 / 16;
  a.tiles_x = (x.W + TX - 1) / TX; a.tiles_y = (x.H + TY - 1) / TY; a.tiles_z = (x.D + TZ - 1) / TZ;
  // column blocks per workgroup x groups of workgroups (blockIdx.z).  Any channel count is served: a group that
  // reaches past the last column block reads other (or out-of-range = zero) weights into accumulators whose
  // channels lie beyond cout_p, which the epilogue neither stores nor counts.  3 or 4 blocks per workgroup,
  // whichever pads less (4 on a tie): 5 -> 3 x 2, 6 -> 3 x 2, 7 -> 4 x 2, 8 -> 4 x 2, 9 -> 3 x 3, 12 -> 4 x 3, ...
  // (6 blocks in one workgroup would take 264 registers: one wave per SIMD)
  if (a.ncbt <= 4) {
    switch (a.ncbt) {
      case 1: return launch_b3<1>(a, 1, s);
      case 2: return launch_b3<2>(a, 1, s);
      case 3: return launch_b3<3>(a, 1, s);
      default: return launch_b3<4>(a, 1, s);
    }
  }
  const int g4 = (a.ncbt + 3) / 4, g3 = (a.ncbt + 2) / 3;
  if (4 * g4 <= 3 * g3) return launch_b3<4>(a, g4, s);
  return launch_b3<3>(a, g3, s);
  return 0;
}

}  // namespace jh
