// Conv3d(k = 3, stride 1, pad 1) as Winograd F(2x2, 3x3) over (y, x) and a direct 3-tap sum
// over z, on the fp32 matrix cores: 16 multiplies per 2x2 output tile, z tap and channel pair
// instead of 36, i.e. 2.25x fewer MFMAs than the implicit GEMM of conv_mfma.h for the layers
// that dominate the path (V2V Res3DBlocks, jarvis/hybridnet/v2vnet.py:27-43).
//
//   Y = A^T [ sum_dz sum_ci (G g_dz G^T) .* (B^T d_{z+dz} B) ] A          (per 4x4 input patch d)
//
// Workgroup (4 waves): 4 z-slices x 8 x 8 outputs = 4 row blocks of 16 tiles (2x2 outputs
// each).  Wave w owns the 4 frequencies (fy = w, fx = 0..3) for all 4 z-slices and all NR
// column blocks: 48 accumulator tiles of 16 tiles x 16 channels.  Per pass of 8 input channels:
//   1. raw patch (6 x 10 x 10 pixels x 8 ch, InstanceNorm(+act) applied on load) -> LDS R
//      (the global loads of the NEXT pass were issued before the MFMAs of this one)
//   2. input transform B^T d B of every (z-slice, tile): R -> V[pz][f][tile][8 ch]
//   3. per frequency: 6 A rows (the 6 z-slices, each used by up to 3 (z, dz) pairs) from LDS,
//      3 x NR B columns (dz) from L2, 72 MFMAs -- the same operand mix as one tap of the
//      direct kernel.
// After the last pass every wave folds its fx into the two x-outputs (A^T along x), the
// partial sums cross waves through LDS (A^T along y mixes fy, i.e. waves), wave w finishes
// z-slice w and hands its four (oy, ox) output phases to the shared epilogue (bias, 16-byte
// stores, fused InstanceNorm statistics).
#include <array>
#include <cstdlib>
#include <map>
#include <mutex>
#include <vector>
#include "conv_mfma.h"
#include "conv3d_wino.h"

namespace jh {

constexpr int kWSV = 12;                                    // V row stride (floats): b64 reads of
                                                            // 16 tiles hit 64 distinct banks

// kWTZ = z-slices per workgroup: 4 (48 accumulator tiles per wave, one workgroup per CU) or
// 2 (24 tiles, <= 256 registers and 62 KB of LDS: two workgroups per CU, so one's patch
// staging, input transform and epilogue run under the other's MFMAs).
// SH: the launch tiles its z-slices with the block shapes of conv3d_wino.h (an extent with a remainder of 1..4 voxels);
// !SH is the plain grid of 8 x 8-voxel blocks, and every shape expression below folds to that constant.
template <int NR, int kWTZ, bool SH = false>
__global__ __launch_bounds__(256, (kWTZ == 2 ? 2 : 1)) void conv3d_wino_kernel(const WinoArgs a) {
  constexpr int kWPZ = kWTZ + 2;
  constexpr int kWNP = kWPZ * (SH ? 108 : kWPY * kWPX);     // 600 / 400 patch pixels (648 / 432 with 18 x 6 patches)
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int nrm_floats = a.in_stats ? 2 * a.cin_p : 0;
  float* nrm = lds_all;
  float* R = lds_all + nrm_floats;                          // [600][8]
  float* V = R + kWNP * 8;                                  // [6][16][16][kWSV]
  float* X = R;                                             // cross-wave exchange (aliases R, V)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 15, kq = lane >> 4;

  const BlockId bid = xcd_block();
  int x0, y0, z0, lc = 2, hlim = a.H, wlim = a.W;
  if (SH) {
    const WinoTile wt = wino_decode(a.tiling, 1, a.H, a.W, kWTZ, (int)bid.x);
    x0 = wt.x0; y0 = wt.y0; z0 = wt.z0; lc = wt.lc; hlim = wt.hlim; wlim = wt.wlim;
  } else {
    const int bx_n = (a.W + kWTX - 1) / kWTX, by_n = (a.H + kWTY - 1) / kWTY;
    int t = bid.x;
    x0 = (t % bx_n) * kWTX; t /= bx_n;
    y0 = (t % by_n) * kWTY; t /= by_n;
    z0 = t * kWTZ;
  }
  // patch of this block: (2 rows + 2) x (2 cols + 2) pixels per z-slice; tile -> (row, column) of the block
  const int PW = SH ? (2 << lc) + 2 : kWPX, PH = SH ? (32 >> lc) + 2 : kWPY;
  const int cmask = (1 << lc) - 1;
  const int nb0 = bid.y * NR;
  const int n = bid.z;
  const int nk8 = a.cin_p >> 3, nb = a.cout_p16 >> 4;

  if (a.in_stats) {
    for (int c = tid; c < a.cin_p; c += 256) {
      const double* st = a.in_stats + ((size_t)n * a.cin_p + c) * kStatW;
      const double mu = exact_read(st) * (double)a.in_inv;
      double var = exact_read(st + kLimbs) * (double)a.in_inv - mu * mu;
      if (var < 0.0) var = 0.0;
      // (mean is stored as -mean * rstd: the commit is one FMA, exactly as in conv3d_wino_pw.hip --
      // the two kernels must agree bit for bit, a launch picks one or the other by its tile count)
      const float rsf = (float)(1.0 / sqrt(var + 1e-5));
      nrm[c] = -(float)mu * rsf;
      nrm[a.cin_p + c] = rsf;
    }
  }
  const float* __restrict__ xin = a.x + (size_t)n * a.D * a.H * a.W * a.cin_p;

  f32x4 acc[4][kWTZ][NR];
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int mr = 0; mr < kWTZ; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[fi][mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- raw patch loads: 600 pixels x 2 channel quads = 1200 items, 5 per thread
  constexpr int ITER = (kWNP * 2 + 255) / 256;
  float4 pf[ITER];
  // per-thread patch items: offsets and validity do not depend on the channel pass, so they are
  // computed once (the commit / issue phases were dominated by this index arithmetic)
  int poff[ITER];
  unsigned okmask = 0;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int idx = tid + it * 256;
    const int q = idx & 1, pix = idx >> 1;
    const int px = pix % PW, py = (pix / PW) % PH, pz = pix / (PW * PH);
    const int iz = z0 - 1 + pz, iy = y0 - 1 + py, ix = x0 - 1 + px;
    const bool ok = idx < kWPZ * PW * PH * 2 && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    poff[it] = ok ? ((iz * a.H + iy) * a.W + ix) * a.cin_p + q * 4 : 0;     // < 2^31 elements per image
    okmask |= ok ? (1u << it) : 0u;
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it)
    pf[it] = (okmask >> it & 1) ? *reinterpret_cast<const float4*>(xin + poff[it]) : make_float4(0.f, 0.f, 0.f, 0.f);

  const float2* __restrict__ U2 = reinterpret_cast<const float2*>(a.u);
  const float2* V2 = reinterpret_cast<const float2*>(V);
  int boff[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) boff[nr] = min(nb0 + nr, nb - 1) * 64 + lane;

  for (int c0 = 0; c0 < a.cin_p; c0 += 8) {
    // the weights of this pass's first frequency do not depend on LDS: request them now, they
    // land under the patch commit and the input transform
    float2 b0v[3][NR];
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr)
        b0v[dz][nr] = U2[(size_t)((((wave * 4) * 3 + dz) * nk8 + (c0 >> 3)) * nb) * 64 + boff[nr]];
    __syncthreads();                       // previous pass: all reads of V (and R) are done
    // 1. commit the raw patch of this pass
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 256;
      if (idx < kWNP * 2) {
        float4 v = pf[it];
        if ((okmask >> it & 1) && a.in_stats) {
          const int c = c0 + (idx & 1) * 4;
          const float4 nm = *reinterpret_cast<const float4*>(nrm + c);
          const float4 rs = *reinterpret_cast<const float4*>(nrm + a.cin_p + c);
          v.x = fmaf(v.x, rs.x, nm.x); v.y = fmaf(v.y, rs.y, nm.y);
          v.z = fmaf(v.z, rs.z, nm.z); v.w = fmaf(v.w, rs.w, nm.w);
          if (a.in_act == ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          } else if (a.in_act == ACT_SILU) {
            v.x = silu_fast(v.x); v.y = silu_fast(v.y);
            v.z = silu_fast(v.z); v.w = silu_fast(v.w);
          }
        }
        *reinterpret_cast<float4*>(R + idx * 4) = v;          // [pix][q] = idx order
      }
    }
    __syncthreads();
    // next pass's loads fly under this pass's transform and MFMAs
    if (c0 + 8 < a.cin_p) {
#pragma unroll
      for (int it = 0; it < ITER; ++it)
        pf[it] = (okmask >> it & 1) ? *reinterpret_cast<const float4*>(xin + poff[it] + c0 + 8)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // 2. input transform: thread -> (z-slice pz, tile, channel quad q); 6 x 16 x 2 = 192 items
    if (tid < kWPZ * 32) {
      const int q = tid & 1, tile = (tid >> 1) & 15, pz = tid >> 5;
      const int ty = tile >> lc, tx = tile & cmask;
      const float* rb = R + ((pz * PH + 2 * ty) * PW + 2 * tx) * 8 + q * 4;
      float4 tr[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 d0 = *reinterpret_cast<const float4*>(rb + (r * PW + 0) * 8);
        const float4 d1 = *reinterpret_cast<const float4*>(rb + (r * PW + 1) * 8);
        const float4 d2 = *reinterpret_cast<const float4*>(rb + (r * PW + 2) * 8);
        const float4 d3 = *reinterpret_cast<const float4*>(rb + (r * PW + 3) * 8);
        tr[r][0] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        tr[r][1] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        tr[r][2] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        tr[r][3] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
      }
      float* vb = V + ((pz * 16) * 16 + tile) * kWSV + q * 4;
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) {
        const float4 t0 = tr[0][fx], t1 = tr[1][fx], t2 = tr[2][fx], t3 = tr[3][fx];
        *reinterpret_cast<float4*>(vb + (0 * 4 + fx) * 16 * kWSV) =
            make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w);
        *reinterpret_cast<float4*>(vb + (1 * 4 + fx) * 16 * kWSV) =
            make_float4(t1.x + t2.x, t1.y + t2.y, t1.z + t2.z, t1.w + t2.w);
        *reinterpret_cast<float4*>(vb + (2 * 4 + fx) * 16 * kWSV) =
            make_float4(t2.x - t1.x, t2.y - t1.y, t2.z - t1.z, t2.w - t1.w);
        *reinterpret_cast<float4*>(vb + (3 * 4 + fx) * 16 * kWSV) =
            make_float4(t1.x - t3.x, t1.y - t3.y, t1.z - t3.z, t1.w - t3.w);
      }
    }
    __syncthreads();
    // 3. matrix cores: wave w, frequencies f = 4 w + fi
    const int kk = c0 >> 3;
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
      const int f = wave * 4 + fi;
      float2 av[kWPZ], bv[3][NR];
#pragma unroll
      for (int pz = 0; pz < kWPZ; ++pz) av[pz] = V2[((pz * 16 + f) * 16 + mrow) * (kWSV / 2) + kq];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
          bv[dz][nr] = fi == 0 ? b0v[dz][nr] : U2[(size_t)(((f * 3 + dz) * nk8 + kk) * nb) * 64 + boff[nr]];
      // (z tap outermost, like the persistent kernel: the same accumulation order per output)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
#pragma unroll
        for (int mr = 0; mr < kWTZ; ++mr)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].x, bv[dz][nr].x, acc[fi][mr][nr], 0, 0, 0);
#pragma unroll
        for (int mr = 0; mr < kWTZ; ++mr)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].y, bv[dz][nr].y, acc[fi][mr][nr], 0, 0, 0);
      }
    }
  }

  // ---- output transform.  Along x inside the wave: r0 = M0 + M1 + M2, r1 = M1 - M2 - M3.
  __syncthreads();                         // R and V are dead: the exchange buffer aliases them
  // X[(w * 2 + j)][mr][nr][lane] as float4 (the 4 tile rows a lane holds of one accumulator)
  float4* X4 = reinterpret_cast<float4*>(X);
#pragma unroll
  for (int mr = 0; mr < kWTZ; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const f32x4 m0 = acc[0][mr][nr], m1 = acc[1][mr][nr], m2 = acc[2][mr][nr], m3 = acc[3][mr][nr];
      X4[(((wave * 2 + 0) * kWTZ + mr) * NR + nr) * 64 + lane] =
          make_float4(m0[0] + m1[0] + m2[0], m0[1] + m1[1] + m2[1], m0[2] + m1[2] + m2[2], m0[3] + m1[3] + m2[3]);
      X4[(((wave * 2 + 1) * kWTZ + mr) * NR + nr) * 64 + lane] =
          make_float4(m1[0] - m2[0] - m3[0], m1[1] - m2[1] - m3[1], m1[2] - m2[2] - m3[2], m1[3] - m2[3] - m3[3]);
    }
  __syncthreads();
  // Along y across waves (fy = wave): out[0] = P0 + P1 + P2, out[1] = P1 - P2 - P3; wave w
  // finishes z-slice w: bias, statistics, 4x4 quad transpose, 16-byte stores.  A lane holds
  // tiles (ty = kq, tx = 0..3) of channel mrow; after the transpose, tile tx = lane & 3 of
  // channels (mrow & ~3) .. + 3.
  float* yb = a.y + (size_t)n * a.D * a.H * a.W * a.cout_p;
  const int jq = lane & 3;
  // kWTZ = 4: wave w finishes z-slice w (both x phases); kWTZ = 2: z-slice w & 1, x phase w >> 1
  constexpr int OXN = (kWTZ == 4) ? 2 : 1;
  const int mr_own = (kWTZ == 4) ? wave : (wave & 1);
  const int ox_first = (kWTZ == 4) ? 0 : (wave >> 1);
  const int oz = z0 + mr_own;
  float s1[NR], s2[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int ch = (nb0 + nr) * 16 + mrow;
    const bool ch_ok = ch < a.cout_p;
    const float bvl = (a.bias && ch < a.cout_p16) ? a.bias[ch] : 0.f;
    s1[nr] = 0.f; s2[nr] = 0.f;
#pragma unroll
    for (int oxi = 0; oxi < OXN; ++oxi) {
      const int ox = ox_first + oxi;
      float4 p[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) p[w] = X4[(((w * 2 + ox) * kWTZ + mr_own) * NR + nr) * 64 + lane];
#pragma unroll
      for (int oy = 0; oy < 2; ++oy) {
        float v[4];
        if (oy == 0) {
          v[0] = p[0].x + p[1].x + p[2].x; v[1] = p[0].y + p[1].y + p[2].y;
          v[2] = p[0].z + p[1].z + p[2].z; v[3] = p[0].w + p[1].w + p[2].w;
        } else {
          v[0] = p[1].x - p[2].x - p[3].x; v[1] = p[1].y - p[2].y - p[3].y;
          v[2] = p[1].z - p[2].z - p[3].z; v[3] = p[1].w - p[2].w - p[3].w;
        }
        // a lane holds tiles 4 kq + r (r = 0..3) of the block: tile -> (row tile >> lc, column tile & cmask)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] += bvl;
          const int tl = 4 * kq + r;
          if (ch_ok && oz < a.D && y0 + 2 * (tl >> lc) + oy < hlim && x0 + 2 * (tl & cmask) + ox < wlim) {
            s1[nr] += v[r];
            s2[nr] += v[r] * v[r];
          }
        }
        {
          float x, y;
          x = (jq & 1) ? v[0] : v[1]; y = quad_xor1(x); if (jq & 1) v[0] = y; else v[1] = y;
          x = (jq & 1) ? v[2] : v[3]; y = quad_xor1(x); if (jq & 1) v[2] = y; else v[3] = y;
          x = (jq & 2) ? v[0] : v[2]; y = quad_xor2(x); if (jq & 2) v[0] = y; else v[2] = y;
          x = (jq & 2) ? v[1] : v[3]; y = quad_xor2(x); if (jq & 2) v[1] = y; else v[3] = y;
        }
        const int tj = 4 * kq + jq;                         // (after the transpose: tile 4 kq + jq)
        const int yy = y0 + 2 * (tj >> lc) + oy, xx = x0 + 2 * (tj & cmask) + ox;
        const int c0 = (nb0 + nr) * 16 + (mrow & ~3);
        if (c0 < a.cout_p && oz < a.D && yy < hlim && xx < wlim)
          *reinterpret_cast<float4*>(yb + ((size_t)(oz * a.H + yy) * a.W + xx) * a.cout_p + c0) =
              make_float4(v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (a.stats) {
    __syncthreads();                       // the reduction scratch aliases X
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      float t1 = s1[nr], t2 = s2[nr];
      t1 = sum_xor16(t1); t2 = sum_xor16(t2);
      t1 = sum_xor32(t1); t2 = sum_xor32(t2);
      if (kq == 0) {
        X[(wave * NR * 16 + nr * 16 + mrow) * 2 + 0] = t1;
        X[(wave * NR * 16 + nr * 16 + mrow) * 2 + 1] = t2;
      }
    }
    __syncthreads();
    if (tid < NR * 16) {
      const int ch = nb0 * 16 + tid;
      if (ch < a.cout_p) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          t1 += X[(w * NR * 16 + tid) * 2 + 0];
          t2 += X[(w * NR * 16 + tid) * 2 + 1];
        }
        stat_add(a.stats + ((size_t)n * a.cout_p + ch) * kStatW, t1, t2);
      }
    }
  }
}

template <int NR, int TZ, bool SH = false>
static int launch_wino_nr(const WinoArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv3d_wino_kernel<NR, TZ, SH>;
  static bool big = false;
  if (!big) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// weights: torch layout (cout, cin, 3, 3, 3) -> U[f][dz][cin_p/8][cout_p16/16][64][2],
// U_f = (G g G^T)[fy][fx] over (ky, kx), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
int pack_wino_weights(int cin, int cout, const float* w, const float* b, ConvWeights* out) {
  const int cin_p = cpad(cin), cout_p16 = round_up(cout, 16);
  const int nk8 = cin_p / 8, nb = cout_p16 / 16;
  static const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  std::vector<float> packed((size_t)16 * 3 * nk8 * nb * 128, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int dz = 0; dz < 3; ++dz) {
        const float* g = w + (((size_t)co * cin + ci) * 3 + dz) * 9;      // [ky][kx]
        double tmp[4][3];
        for (int fy = 0; fy < 4; ++fy)
          for (int kx = 0; kx < 3; ++kx) {
            double s = 0.0;
            for (int ky = 0; ky < 3; ++ky) s += (double)G[fy][ky] * g[ky * 3 + kx];
            tmp[fy][kx] = s;
          }
        const int k8 = ci / 8, kq = (ci % 8) / 2, j = ci % 2;
        const int nbk = co / 16, nn = co % 16;
        const int lane = kq * 16 + nn;
        for (int fy = 0; fy < 4; ++fy)
          for (int fx = 0; fx < 4; ++fx) {
            double s = 0.0;
            for (int kx = 0; kx < 3; ++kx) s += tmp[fy][kx] * (double)G[fx][kx];
            const int f = fy * 4 + fx;
            packed[((((size_t)f * 3 + dz) * nk8 + k8) * nb + nbk) * 128 + lane * 2 + j] = (float)s;
          }
      }
  out->cin_p = cin_p; out->cout_p16 = cout_p16; out->phase_stride = packed.size();
  JH_CHECK_HIP(hipMalloc(&out->w, packed.size() * sizeof(float)));
  JH_CHECK_HIP(hipMemcpy(out->w, packed.data(), packed.size() * sizeof(float), hipMemcpyHostToDevice));
  out->bias = nullptr;
  if (b) {
    std::vector<float> bp(cout_p16, 0.f);
    for (int i = 0; i < cout; ++i) bp[i] = b[i];
    JH_CHECK_HIP(hipMalloc(&out->bias, bp.size() * sizeof(float)));
    JH_CHECK_HIP(hipMemcpy(out->bias, bp.data(), bp.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  return 0;
}

// variant: 4 = persistent wave-specialised form (conv3d_wino_pw.hip; the default -- it falls
// back to variant 0 for launches with fewer than two tiles per CU or fewer than three channel
// passes), 0 = one role per workgroup, 4 z-slices (JH_WINO_PW=0), 1 = the same with 2 z-slices
// (JH_WINO_TZ=2)
int wino_variant_from_env() {
  if (const char* e = getenv("JH_WINO_TZ")) { if (atoi(e) == 2) return 1; }
  if (const char* e = getenv("JH_WINO_PW")) { if (atoi(e) == 0) return 0; }
  return 4;
}

// Host side of the persistent kernel's tables (conv3d_wino.h): the loader's per-shape patch tables (the tiles themselves
// are decoded arithmetically by the kernel).  A function of the volume and the channel padding only -- not of the
// batch.  Empty for volumes without remainder strips.
std::vector<int> wino_tables(int D, int H, int W, int cin_p) {
  const WinoTiling g = wino_tiling(D, H, W, 4);
  std::vector<int> t;
  if (!g.shaped) return t;
  t.assign((size_t)3 * kWinoShapeWords, 0);
  const int rz = D % 4 ? D % 4 : 4;
  for (int lc = 1; lc <= 3; ++lc) {
    const int PW = (2 << lc) + 2, PH = (32 >> lc) + 2, TH = 32 >> lc, TW = 2 << lc;
    const int hy = H - (lc == 3 ? g.H44 : 0), wx = W - (lc == 1 ? g.W44 : 0);
    const int ry = hy % TH ? hy % TH : TH, rx = wx % TW ? wx % TW : TW;
    int* tab = t.data() + (size_t)(lc - 1) * kWinoShapeWords;
    for (int lane = 0; lane < 64; ++lane) {
      unsigned m[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int it = 0; it < kWinoShapeIter; ++it) {
        const int idx = lane + it * 64, pix = idx >> 1, q = lane & 1;
        const int px = pix % PW, py = (pix / PW) % PH, pz = pix / (PW * PH);
        tab[it * 64 + lane] = (((pz * H + py) * W + px) * cin_p + q * 4) * 4;
        m[0] |= (unsigned)(pz == 0) << it; m[1] |= (unsigned)(pz > rz) << it;
        m[2] |= (unsigned)(py == 0) << it; m[3] |= (unsigned)(py > ry) << it;
        m[4] |= (unsigned)(px == 0) << it; m[5] |= (unsigned)(px > rx) << it;
        m[6] |= (unsigned)(idx >= 6 * PW * PH * 2) << it;
      }
      for (int k = 0; k < 7; ++k) tab[(kWinoShapeIter + k) * 64 + lane] = (int)m[k];
    }
  }
  return t;
}

int launch_conv3d_wino(const ConvWeights& w, const Act& x, const Act& y, double* stats, hipStream_t s,
                       const InNorm* in, int variant, const int* tables) {
  JH_REQUIRE(x.Cp == w.cin_p && x.N == y.N && x.D == y.D && x.H == y.H && x.W == y.W, "wino shapes");
  WinoArgs a{};
  a.x = x.p; a.y = y.p; a.u = w.w; a.bias = w.bias; a.stats = stats;
  if (in && in->stats) { a.in_stats = in->stats; a.in_inv = in->inv; a.in_act = in->act; }
  a.N = x.N; a.D = x.D; a.H = x.H; a.W = x.W; a.cin_p = x.Cp; a.cout_p = y.Cp; a.cout_p16 = w.cout_p16;
  const int nb = w.cout_p16 / 16;
  const int nr_full = (nb % 3 == 0) ? 3 : ((nb % 2 == 0) ? 2 : (nb == 1 ? 1 : 3));
  // (2 z-slices per workgroup measure the same as 4: the second resident workgroup only pays for
  // the doubled per-workgroup prologue / epilogue; kept as an experiment knob)
  const int tz = variant == 1 ? 2 : 4;
  // block shapes of a z-slice (conv3d_wino.h): a function of the volume only -- the persistent and the one-role kernel
  // tile a launch the same way, so their per-block fp32 partial sums of the statistics agree bit for bit
  a.tiling = wino_tiling(x.D, x.H, x.W, tz);
  const bool sh = a.tiling.shaped != 0;
  if (sh && variant == 4 && !tables) {
    // callers outside a network plan (the op-level test entry only; plans pass their own tables, so this allocation
    // never runs under stream capture): one 5 KB table per (device, volume, channel padding), built on first use
    static std::mutex mu;
    static std::map<std::array<int, 5>, int*> cache;
    std::lock_guard<std::mutex> lock(mu);
    int device = 0;
    JH_CHECK_HIP(hipGetDevice(&device));
    int*& dev = cache[{device, x.D, x.H, x.W, x.Cp}];
    if (!dev) {
      const std::vector<int> host = wino_tables(x.D, x.H, x.W, x.Cp);
      JH_CHECK_HIP(hipMalloc(&dev, host.size() * sizeof(int)));
      JH_CHECK_HIP(hipMemcpy(dev, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    tables = dev;
  }
  a.shape_tab = tables;
  if (variant == 4) {
    const int rc = launch_conv3d_wino_pw(a, nr_full, s);
    if (rc >= 0) return rc;                 // -1: too few tiles / one channel pass -> one-role kernel
  }
  const int blocks = sh ? a.tiling.slabs * wino_blocks_per_slab(a.tiling)
                        : ((x.D + tz - 1) / tz) * ((x.H + kWTY - 1) / kWTY) * ((x.W + kWTX - 1) / kWTX);
  // Single-frame-set launches have fewer tiles than the chip has CUs (32 / 128 at 16^3 / 32^3): fewer column
  // blocks per workgroup then, i.e. more workgroups that each repeat the input transform but run a half / a
  // third of the MFMAs.  (nr only partitions the output channels: the fp32 partial sums of the statistics --
  // per channel over a tile's voxels -- and every output value are the same bits for any nr.)
  int nr_l = nr_full;
  if (JH_ENV_KNOB("JH_WINO_NR_SPLIT") != 0)
    while (nr_l > 1 && (long)blocks * ((nb + nr_l - 1) / nr_l) * x.N <= 128) --nr_l;
  const int nr = nr_l;
  dim3 grid(blocks, (nb + nr - 1) / nr, x.N);
  const size_t xbytes = (size_t)4 * 2 * tz * nr * 4 * 64 * sizeof(float);
  size_t lds = (size_t)((tz + 2) * (sh ? 108 : kWPY * kWPX) * 8 + (tz + 2) * 16 * 16 * kWSV) * sizeof(float);
  if (lds < xbytes) lds = xbytes;
  lds += (a.in_stats ? (size_t)2 * a.cin_p : 0) * sizeof(float);
  JH_REQUIRE(lds <= 160 * 1024, "wino LDS");
#define JH_WINO_CASE(NRV)                                                                               \
  if (nr == NRV && sh) return tz == 4 ? launch_wino_nr<NRV, 4, true>(a, grid, lds, s) : launch_wino_nr<NRV, 2, true>(a, grid, lds, s); \
  if (nr == NRV) return tz == 4 ? launch_wino_nr<NRV, 4>(a, grid, lds, s) : launch_wino_nr<NRV, 2>(a, grid, lds, s);
  JH_WINO_CASE(3) JH_WINO_CASE(2) JH_WINO_CASE(1)
#undef JH_WINO_CASE
  JH_REQUIRE(false, "wino NR");
}

}  // namespace jh
