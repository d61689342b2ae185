// "Ping-pong" form of the Winograd 3D convolution of conv3d_wino.hip (same mathematics, same
// weights, same output): 8 waves per workgroup, two per SIMD, in two sets that alternate roles
// inside every channel pass.
//
//   pass p, half A:  waves 0-3: MFMAs of pass p (their 2 of 4 frequencies per fy)
//                    waves 4-7: input transform of z-slices 0-2 of pass p+1
//   pass p, half B:  waves 4-7: MFMAs of pass p
//                    waves 0-3: input transform of z-slices 3-5 of pass p+1
//   all:             patch commit of pass p+2, loads of pass p+3; two barriers per pass
//
// Every SIMD hosts one wave of each set, so its matrix core always has an MFMA stream to run
// while the other wave does the LDS / vector-ALU work that the one-role kernel serialises
// (there the matrix cores idle 46 % of the time).  The price is LDS: R and V are
// double-buffered (V unpadded, 8 floats per row: its 2-way bank conflict on the 24 A-operand
// reads of a pass does not matter), 137 KB per workgroup.
//
// Wave w: fy = w & 3, set = w >> 2, frequencies f = 4 fy + 2 set + {0, 1}; 2 x 4 x NR
// accumulator tiles (96 registers for NR = 3).
#include <cstdlib>
#include "conv_mfma.h"
#include "conv3d_wino.h"

namespace jh {

constexpr int kPTZ = 4, kPPZ = kPTZ + 2;
constexpr int kPNP = kPPZ * kWPY * kWPX;                    // 600 patch pixels
constexpr int kPSV = 8;                                     // V row stride (floats), unpadded

template <int NR>
__global__ __launch_bounds__(512, 2) void conv3d_wino_pp_kernel(const WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int nrm_floats = a.in_stats ? 2 * a.cin_p : 0;
  float* nrm = lds_all;
  float* R = lds_all + nrm_floats;                          // [2][600][8]
  float* V = R + 2 * kPNP * 8;                              // [2][6][16][16][8]
  float* X = R;                                             // epilogue exchange (aliases R, V)
  constexpr int VSZ = kPPZ * 16 * 16 * kPSV;                // floats per V buffer
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int set = wave >> 2, fy = wave & 3;
  const int tl = tid & 255;                                 // thread id inside its set
  const int mrow = lane & 15, kq = lane >> 4;

  const BlockId bid = xcd_block();
  const int bx_n = (a.W + kWTX - 1) / kWTX, by_n = (a.H + kWTY - 1) / kWTY;
  int t = bid.x;
  const int x0 = (t % bx_n) * kWTX; t /= bx_n;
  const int y0 = (t % by_n) * kWTY; t /= by_n;
  const int z0 = t * kPTZ;
  const int nb0 = bid.y * NR;
  const int n = bid.z;
  const int nk8 = a.cin_p >> 3, nb = a.cout_p16 >> 4;
  const int npass = nk8;

  if (a.in_stats) {
    for (int c = tid; c < a.cin_p; c += 512) {
      const double* st = a.in_stats + ((size_t)n * a.cin_p + c) * 2;
      const double mu = st[0] * (double)a.in_inv;
      double var = st[1] * (double)a.in_inv - mu * mu;
      if (var < 0.0) var = 0.0;
      nrm[c] = (float)mu;
      nrm[a.cin_p + c] = (float)(1.0 / sqrt(var + 1e-5));
    }
  }
  const float* __restrict__ xin = a.x + (size_t)n * a.D * a.H * a.W * a.cin_p;

  f32x4 acc[2][kPTZ][NR];
#pragma unroll
  for (int fi = 0; fi < 2; ++fi)
#pragma unroll
    for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[fi][mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- raw patch: 600 pixels x 2 channel quads = 1200 items over 512 threads
  constexpr int ITER = (kPNP * 2 + 511) / 512;
  float4 pf[ITER];
  auto patch_ok = [&](int idx, int c0, const float** src) -> bool {
    const int q = idx & 1, pix = idx >> 1;
    const int px = pix % kWPX, py = (pix / kWPX) % kWPY, pz = pix / (kWPX * kWPY);
    const int iz = z0 - 1 + pz, iy = y0 - 1 + py, ix = x0 - 1 + px;
    *src = xin + ((size_t)(iz * a.H + iy) * a.W + ix) * a.cin_p + c0 + q * 4;
    return idx < kPNP * 2 && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
  };
  auto issue = [&](int c0) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const float* src;
      const bool ok = patch_ok(tid + it * 512, c0, &src);
      pf[it] = ok ? *reinterpret_cast<const float4*>(src) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit = [&](int c0, float* Rd) {
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = tid + it * 512;
      if (idx < kPNP * 2) {
        const float* src;
        const bool ok = patch_ok(idx, c0, &src);
        float4 v = pf[it];
        if (ok && a.in_stats) {
          const int c = c0 + (idx & 1) * 4;
          const float4 mu = *reinterpret_cast<const float4*>(nrm + c);
          const float4 rs = *reinterpret_cast<const float4*>(nrm + a.cin_p + c);
          v.x = (v.x - mu.x) * rs.x; v.y = (v.y - mu.y) * rs.y;
          v.z = (v.z - mu.z) * rs.z; v.w = (v.w - mu.w) * rs.w;
          if (a.in_act == ACT_RELU) {
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          } else if (a.in_act == ACT_SILU) {
            v.x = __fdividef(v.x, 1.f + __expf(-v.x)); v.y = __fdividef(v.y, 1.f + __expf(-v.y));
            v.z = __fdividef(v.z, 1.f + __expf(-v.z)); v.w = __fdividef(v.w, 1.f + __expf(-v.w));
          }
        }
        *reinterpret_cast<float4*>(Rd + idx * 4) = v;
      }
    }
  };
  // input transform of z-slices [pz0, pz0 + 3) by the 256 threads of one set (96 items)
  auto transform3 = [&](const float* Rs, float* Vd, int pz0) {
    if (tl < 96) {
      const int q = tl & 1, tile = (tl >> 1) & 15, pz = pz0 + (tl >> 5);
      const int ty = tile >> 2, tx = tile & 3;
      const float* rb = Rs + ((pz * kWPY + 2 * ty) * kWPX + 2 * tx) * 8 + q * 4;
      float4 tr[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float4 d0 = *reinterpret_cast<const float4*>(rb + (r * kWPX + 0) * 8);
        const float4 d1 = *reinterpret_cast<const float4*>(rb + (r * kWPX + 1) * 8);
        const float4 d2 = *reinterpret_cast<const float4*>(rb + (r * kWPX + 2) * 8);
        const float4 d3 = *reinterpret_cast<const float4*>(rb + (r * kWPX + 3) * 8);
        tr[r][0] = make_float4(d0.x - d2.x, d0.y - d2.y, d0.z - d2.z, d0.w - d2.w);
        tr[r][1] = make_float4(d1.x + d2.x, d1.y + d2.y, d1.z + d2.z, d1.w + d2.w);
        tr[r][2] = make_float4(d2.x - d1.x, d2.y - d1.y, d2.z - d1.z, d2.w - d1.w);
        tr[r][3] = make_float4(d1.x - d3.x, d1.y - d3.y, d1.z - d3.z, d1.w - d3.w);
      }
      float* vb = Vd + ((pz * 16) * 16 + tile) * kPSV + q * 4;
#pragma unroll
      for (int fx = 0; fx < 4; ++fx) {
        const float4 t0 = tr[0][fx], t1 = tr[1][fx], t2 = tr[2][fx], t3 = tr[3][fx];
        *reinterpret_cast<float4*>(vb + (0 * 4 + fx) * 16 * kPSV) =
            make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w);
        *reinterpret_cast<float4*>(vb + (1 * 4 + fx) * 16 * kPSV) =
            make_float4(t1.x + t2.x, t1.y + t2.y, t1.z + t2.z, t1.w + t2.w);
        *reinterpret_cast<float4*>(vb + (2 * 4 + fx) * 16 * kPSV) =
            make_float4(t2.x - t1.x, t2.y - t1.y, t2.z - t1.z, t2.w - t1.w);
        *reinterpret_cast<float4*>(vb + (3 * 4 + fx) * 16 * kPSV) =
            make_float4(t1.x - t3.x, t1.y - t3.y, t1.z - t3.z, t1.w - t3.w);
      }
    }
  };

  const float2* __restrict__ U2 = reinterpret_cast<const float2*>(a.u);
  int boff[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) boff[nr] = min(nb0 + nr, nb - 1) * 64 + lane;

  auto mfma_pass = [&](const float* Vs, int kk) {
    const float2* V2 = reinterpret_cast<const float2*>(Vs);
#pragma unroll
    for (int fi = 0; fi < 2; ++fi) {
      const int f = fy * 4 + set * 2 + fi;
      float2 av[kPPZ], bv[3][NR];
#pragma unroll
      for (int pz = 0; pz < kPPZ; ++pz) av[pz] = V2[((pz * 16 + f) * 16 + mrow) * (kPSV / 2) + kq];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
          bv[dz][nr] = U2[(size_t)(((f * 3 + dz) * nk8 + kk) * nb) * 64 + boff[nr]];
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].x, bv[dz][nr].x, acc[fi][mr][nr], 0, 0, 0);
#pragma unroll
      for (int dz = 0; dz < 3; ++dz)
#pragma unroll
        for (int mr = 0; mr < kPTZ; ++mr)
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].y, bv[dz][nr].y, acc[fi][mr][nr], 0, 0, 0);
    }
  };

  // ---- prologue: pass 0 fully staged and transformed, pass 1 committed, pass 2 in flight
  issue(0);
  __syncthreads();                         // mean / rstd visible
  commit(0, R);
  if (npass > 1) issue(8);
  __syncthreads();
  transform3(R, V, set * 3);               // both sets: the two halves of pass 0
  if (npass > 1) { commit(8, R + kPNP * 8); if (npass > 2) issue(16); }
  __syncthreads();

  for (int p = 0; p < npass; ++p) {
    const int cur = p & 1;
    const float* Vc = V + cur * VSZ;
    float* Vn = V + (cur ^ 1) * VSZ;
    const float* Rn = R + (cur ^ 1) * kPNP * 8;            // pass p+1, committed during pass p-1
    float* Rf = R + cur * kPNP * 8;                        // free: receives pass p+2
    const bool more1 = p + 1 < npass, more2 = p + 2 < npass, more3 = p + 3 < npass;
    // half A: set 0 on the matrix cores, set 1 transforms z-slices 0-2 of pass p+1
    if (set == 0) {
      mfma_pass(Vc, p);
    } else {
      if (more1) transform3(Rn, Vn, 0);
    }
    __syncthreads();      // (measured: without this barrier, i.e. with both sets' MFMA streams
                          // overlapping freely, the kernel is 24 % slower)
    // half B: set 1 on the matrix cores, set 0 transforms z-slices 3-5
    if (set == 1) {
      mfma_pass(Vc, p);
    } else {
      if (more1) transform3(Rn, Vn, 3);
    }
    // every thread commits its own share of pass p+2 (R[cur] was consumed a pass ago) and
    // requests pass p+3
    if (more2) { commit((p + 2) * 8, Rf); if (more3) issue((p + 3) * 8); }
    __syncthreads();
  }

  // ---- output transform, one column block (nr) per round through LDS.
  // x-fold inside the wave: set 0 holds fx 0,1: r0 = M0 + M1, r1 = M1; set 1 holds fx 2,3:
  // r0 = M2, r1 = -M2 - M3.  X[(w * 2 + j)][mr][lane] as float4.
  float4* X4 = reinterpret_cast<float4*>(X);
  float* yb = a.y + (size_t)n * a.D * a.H * a.W * a.cout_p;
  const int jq = lane & 3;
  const int mr_own = wave >> 1, ox = wave & 1;             // wave w finishes z-slice w / 2, x phase w % 2
  const int oz = z0 + mr_own;
  float s1[NR], s2[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
#pragma unroll
    for (int mr = 0; mr < kPTZ; ++mr) {
      const f32x4 m0 = acc[0][mr][nr], m1 = acc[1][mr][nr];
      float4 r0, r1;
      if (set == 0) {
        r0 = make_float4(m0[0] + m1[0], m0[1] + m1[1], m0[2] + m1[2], m0[3] + m1[3]);
        r1 = make_float4(m1[0], m1[1], m1[2], m1[3]);
      } else {
        r0 = make_float4(m0[0], m0[1], m0[2], m0[3]);
        r1 = make_float4(-m0[0] - m1[0], -m0[1] - m1[1], -m0[2] - m1[2], -m0[3] - m1[3]);
      }
      X4[((wave * 2 + 0) * kPTZ + mr) * 64 + lane] = r0;
      X4[((wave * 2 + 1) * kPTZ + mr) * 64 + lane] = r1;
    }
    __syncthreads();
    // y-fold across fy (and the two sets): out[0] = P0 + P1 + P2, out[1] = P1 - P2 - P3
    float4 P[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 u = X4[((g * 2 + ox) * kPTZ + mr_own) * 64 + lane];            // set 0, fy = g
      const float4 v = X4[(((g + 4) * 2 + ox) * kPTZ + mr_own) * 64 + lane];      // set 1, fy = g
      P[g] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
    const int ch = (nb0 + nr) * 16 + mrow;
    const bool ch_ok = ch < a.cout_p;
    const float bvl = (a.bias && ch < a.cout_p16) ? a.bias[ch] : 0.f;
    s1[nr] = 0.f; s2[nr] = 0.f;
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      float v[4];
      if (oy == 0) {
        v[0] = P[0].x + P[1].x + P[2].x; v[1] = P[0].y + P[1].y + P[2].y;
        v[2] = P[0].z + P[1].z + P[2].z; v[3] = P[0].w + P[1].w + P[2].w;
      } else {
        v[0] = P[1].x - P[2].x - P[3].x; v[1] = P[1].y - P[2].y - P[3].y;
        v[2] = P[1].z - P[2].z - P[3].z; v[3] = P[1].w - P[2].w - P[3].w;
      }
      const int yy = y0 + 2 * kq + oy;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[r] += bvl;
        if (ch_ok && oz < a.D && yy < a.H && x0 + 2 * r + ox < a.W) {
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
      const int xx = x0 + 2 * jq + ox;
      const int c0 = (nb0 + nr) * 16 + (mrow & ~3);
      if (c0 < a.cout_p && oz < a.D && yy < a.H && xx < a.W)
        *reinterpret_cast<float4*>(yb + ((size_t)(oz * a.H + yy) * a.W + xx) * a.cout_p + c0) =
            make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();                       // X is rewritten by the next round / the reduction
  }
  if (a.stats) {
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      float t1 = s1[nr], t2 = s2[nr];
      t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
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
        for (int w = 0; w < 8; ++w) {
          t1 += X[(w * NR * 16 + tid) * 2 + 0];
          t2 += X[(w * NR * 16 + tid) * 2 + 1];
        }
        double* st = a.stats + ((size_t)n * a.cout_p + ch) * 2;
        unsafeAtomicAdd(st + 0, (double)t1);
        unsafeAtomicAdd(st + 1, (double)t2);
      }
    }
  }
}

template <int NR>
static int launch_pp_nr(const WinoArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv3d_wino_pp_kernel<NR>;
  static bool big = false;
  if (!big) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    big = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

int launch_conv3d_wino_pp(const WinoArgs& a, int nr, hipStream_t s) {
  const int nb = a.cout_p16 / 16;
  const int blocks = ((a.D + kPTZ - 1) / kPTZ) * ((a.H + kWTY - 1) / kWTY) * ((a.W + kWTX - 1) / kWTX);
  dim3 grid(blocks, (nb + nr - 1) / nr, a.N);
  size_t lds = (size_t)(2 * kPNP * 8 + 2 * kPPZ * 16 * 16 * kPSV) * sizeof(float);
  const size_t xbytes = (size_t)8 * 2 * kPTZ * 64 * 4 * sizeof(float);
  if (lds < xbytes) lds = xbytes;
  lds += (a.in_stats ? (size_t)2 * a.cin_p : 0) * sizeof(float);
  JH_REQUIRE(lds <= 160 * 1024, "wino (ping-pong) LDS");
  if (nr == 3) return launch_pp_nr<3>(a, grid, lds, s);
  if (nr == 2) return launch_pp_nr<2>(a, grid, lds, s);
  return launch_pp_nr<1>(a, grid, lds, s);
}

}  // namespace jh
