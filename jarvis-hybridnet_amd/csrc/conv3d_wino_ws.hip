// Wave-specialised form of the Winograd 3D convolution of conv3d_wino.hip (same mathematics,
// same packed weights, same output): 8 waves per workgroup, two per SIMD, with FIXED roles.
//
//   waves 0-3 ("matrix waves", one per SIMD): nothing but the MFMA stream of conv3d_wino.hip
//       (wave w = frequencies fy = w, fx = 0..3; 4 z-slices x NR column blocks x 4 frequencies
//       = 48 accumulator tiles for NR = 3) and, at the end, the output transform + epilogue.
//   waves 4-7 ("staging waves", one per SIMD): everything else -- the global loads of the raw
//       6 x 10 x 10 x 8-channel patch, InstanceNorm(+act) on load, the commit to LDS and the
//       Winograd input transform B^T d B into the operand buffer V, always one channel pass
//       AHEAD of the matrix waves (V is double-buffered).
//
// In the one-role kernel the matrix cores idle while the same four waves commit and transform
// (54 % busy); in the two-set "ping-pong" form both sets alternate roles, so after every barrier
// the set that takes over the matrix cores first has to fetch its operands (57 %).  Here the
// matrix waves never leave the MFMA stream: their weight operands are prefetched one frequency
// ahead from L2 regardless of barriers, and the LDS operands of the next pass are complete
// before the pass ends because the staging waves need about a third of a pass for their work.
//
// Per channel pass p (two workgroup barriers, both roles):
//   matrix waves : MFMAs of frequencies fx = 0,1 from V[p & 1] | B | fx = 2,3 | B
//   staging waves: commit patch p+1 -> R, request patch p+2    | B | transform p+1 -> V[(p+1) & 1] | B
//
// LDS: mean/rstd, R [600][8] (19 KB), V [2][6][16][16][8] unpadded with an XOR swizzle of the
// channel quads (98 KB; the 16 tile rows x 4 k-quads of an A operand hit 64 distinct banks),
// statistics scratch: 120 KB, one workgroup per CU.  The epilogue's cross-wave exchange
// aliases V.  Registers: 512 threads x <= 256.
#include <cstdlib>
#include "conv_mfma.h"
#include "conv3d_wino.h"

namespace jh {

namespace {
// Workgroup barrier that waits for this wave's LDS traffic only: global loads already in flight
// (next pass's patch, next step's weights) stay in flight across it.  __syncthreads() would add
// s_waitcnt vmcnt(0) and expose their latency at every barrier.
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
constexpr int kSTZ = 4, kSPZ = kSTZ + 2;
constexpr int kSNP = kSPZ * kWPY * kWPX;                    // 600 patch pixels
constexpr int kSVSZ = kSPZ * 16 * 16 * 8;                   // floats per V buffer
}  // namespace

template <int NR, int ABL>
__global__ __launch_bounds__(512) void conv3d_wino_ws_kernel(const WinoArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds_all[];
  const int nrm_floats = a.in_stats ? 2 * a.cin_p : 0;
  float* nrm = lds_all;
  float* R = lds_all + nrm_floats;                          // [600][8]
  float* V = R + kSNP * 8;                                  // [2][6][16][16][8]
  float* S = V + 2 * kSVSZ;                                 // [4][NR*16][2] statistics scratch
  float* X = V;                                             // epilogue exchange (aliases V)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mrow = lane & 15, kq = lane >> 4;

  const BlockId bid = xcd_block();
  const int bx_n = (a.W + kWTX - 1) / kWTX, by_n = (a.H + kWTY - 1) / kWTY;
  int t = bid.x;
  const int x0 = (t % bx_n) * kWTX; t /= bx_n;
  const int y0 = (t % by_n) * kWTY; t /= by_n;
  const int z0 = t * kSTZ;
  const int nb0 = bid.y * NR;
  const int n = bid.z;
  const int nk8 = a.cin_p >> 3, nb = a.cout_p16 >> 4;
  const int npass = ABL == 3 ? 0 : nk8;

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

  if (wave >= 4) {
    // ===================================================================== staging waves
    const int ht = tid - 256;
    const float* __restrict__ xin = a.x + (size_t)n * a.D * a.H * a.W * a.cin_p;
    constexpr int ITER = (kSNP * 2 + 255) / 256;            // 1200 float4 items over 256 threads
    float4 pf[ITER];
    int poff[ITER];
    unsigned okmask = 0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int idx = ht + it * 256;
      const int q = idx & 1, pix = idx >> 1;
      const int px = pix % kWPX, py = (pix / kWPX) % kWPY, pz = pix / (kWPX * kWPY);
      const int iz = z0 - 1 + pz, iy = y0 - 1 + py, ix = x0 - 1 + px;
      const bool ok = idx < kSNP * 2 && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      poff[it] = ok ? ((iz * a.H + iy) * a.W + ix) * a.cin_p + q * 4 : 0;
      okmask |= ok ? (1u << it) : 0u;
    }
    auto issue = [&](int c0) {
#pragma unroll
      for (int it = 0; it < ITER; ++it)
        pf[it] = (okmask >> it & 1) ? *reinterpret_cast<const float4*>(xin + poff[it] + c0)
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto commit = [&](int c0) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) {
        const int idx = ht + it * 256;
        if (idx < kSNP * 2) {
          float4 v = pf[it];
          if ((okmask >> it & 1) && a.in_stats) {
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
          *reinterpret_cast<float4*>(R + idx * 4) = v;        // [pix][q] = idx order
        }
      }
    };
    // input transform B^T d B of every (z-slice, tile, channel quad): 6 x 16 x 2 = 192 items
    auto transform = [&](float* Vd) {
      if (ht < kSPZ * 32) {
        const int q = ht & 1, tile = (ht >> 1) & 15, pz = ht >> 5;
        const int ty = tile >> 2, tx = tile & 3;
        const float* rb = R + ((pz * kWPY + 2 * ty) * kWPX + 2 * tx) * 8 + q * 4;
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
        // swizzle: channel quad q of tile rows 8..15 goes to slot q ^ 1
        float* vb = Vd + ((pz * 16) * 16 + tile) * 8 + ((q ^ (tile >> 3)) & 1) * 4;
#pragma unroll
        for (int fx = 0; fx < 4; ++fx) {
          const float4 t0 = tr[0][fx], t1 = tr[1][fx], t2 = tr[2][fx], t3 = tr[3][fx];
          *reinterpret_cast<float4*>(vb + (0 * 4 + fx) * 16 * 8) =
              make_float4(t0.x - t2.x, t0.y - t2.y, t0.z - t2.z, t0.w - t2.w);
          *reinterpret_cast<float4*>(vb + (1 * 4 + fx) * 16 * 8) =
              make_float4(t1.x + t2.x, t1.y + t2.y, t1.z + t2.z, t1.w + t2.w);
          *reinterpret_cast<float4*>(vb + (2 * 4 + fx) * 16 * 8) =
              make_float4(t2.x - t1.x, t2.y - t1.y, t2.z - t1.z, t2.w - t1.w);
          *reinterpret_cast<float4*>(vb + (3 * 4 + fx) * 16 * 8) =
              make_float4(t1.x - t3.x, t1.y - t3.y, t1.z - t3.z, t1.w - t3.w);
        }
      }
    };

    issue(0);
    __syncthreads();                                        // B0: mean / rstd visible
    commit(0);
    if (npass > 1) issue(8);
    __syncthreads();                                        // B1: R(0) complete
    transform(V);
    __syncthreads();                                        // B2: V[0] complete
    for (int p = 0; p < npass; ++p) {
      const bool more = p + 1 < npass && ABL != 1;
      if (more) {
        commit((p + 1) * 8);                                // R was read by transform(p) before B(2p+2)
        if (p + 2 < npass) issue((p + 2) * 8);
      }
      wg_barrier();
      if (more) transform(V + ((p + 1) & 1) * kSVSZ);       // last read in pass p-1
      wg_barrier();
    }
    __syncthreads();                                        // (matrix waves drain their requests)
    __syncthreads();                                        // E1 (exchange written)
    if (a.stats) __syncthreads();                           // E2 (statistics scratch written)
    return;
  }

  // ======================================================================= matrix waves
  f32x4 acc[4][kSTZ][NR];
#pragma unroll
  for (int fi = 0; fi < 4; ++fi)
#pragma unroll
    for (int mr = 0; mr < kSTZ; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[fi][mr][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Weights through buffer loads: one VGPR byte offset per column block and a scalar offset per
  // (frequency, dz, pass) -- no 64-bit per-load addresses in VGPRs.
  const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.u), 0, 16 * 3 * nk8 * nb * 512, 0x00020000);
  int bvo[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) bvo[nr] = (min(nb0 + nr, nb - 1) * 64 + lane) * 8;
  typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
  auto load_b = [&](float2 (&bd)[NR], int f, int dz, int kk) {
    const int so = (((f * 3 + dz) * nk8 + kk) * nb) * 512;
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(urs, bvo[nr], so, 0);
      bd[nr] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
    }
  };
  // A operand: row = tile (mrow), k-quad kq; swizzled like the transform writes it
  const int aoff = mrow * 4 + (kq ^ ((mrow >> 3) << 1));

  // Software pipeline of the operand registers, in program order (pinned with sched_barrier):
  // the weights b[dz] of the NEXT frequency step are requested as soon as the 24 MFMAs of this
  // step's dz are issued (>= 48 MFMAs = 1536 cycles before their first use, no second register
  // set), the LDS operands of the next step go to a second set (12 registers).
  float2 b[3][NR], av[kSPZ], avn[kSPZ];
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) load_b(b[dz], wave * 4, dz, 0);
  __syncthreads();                                          // B0
  __syncthreads();                                          // B1
  __syncthreads();                                          // B2: V[0] complete
  {
    const float2* V2 = reinterpret_cast<const float2*>(V);
#pragma unroll
    for (int pz = 0; pz < kSPZ; ++pz) av[pz] = V2[((pz * 16 + wave * 4) * 16) * 4 + aoff];
  }

  for (int p = 0; p < npass; ++p) {
    const float2* V2 = reinterpret_cast<const float2*>(V + (p & 1) * kSVSZ);
    const int pn = min(p + 1, npass - 1);                   // (the last pass re-requests itself)
#pragma unroll
    for (int fi = 0; fi < 4; ++fi) {
      const int f = wave * 4 + fi;
      const int nf = fi < 3 ? f + 1 : wave * 4, nk = fi < 3 ? p : pn;
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) {
        if (ABL != 2) {
#pragma unroll
          for (int mr = 0; mr < kSTZ; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
              acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].x, b[dz][nr].x, acc[fi][mr][nr], 0, 0, 0);
#pragma unroll
          for (int mr = 0; mr < kSTZ; ++mr)
#pragma unroll
            for (int nr = 0; nr < NR; ++nr)
              acc[fi][mr][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mr + dz].y, b[dz][nr].y, acc[fi][mr][nr], 0, 0, 0);
        } else {
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) acc[fi][dz][nr][0] += av[dz].x * b[dz][nr].x + av[dz + 3].y * b[dz][nr].y;
        }
        __builtin_amdgcn_sched_barrier(0);
        load_b(b[dz], nf, dz, nk);
        if (fi < 3) {
          avn[2 * dz] = V2[(((2 * dz) * 16 + nf) * 16) * 4 + aoff];
          avn[2 * dz + 1] = V2[(((2 * dz + 1) * 16 + nf) * 16) * 4 + aoff];
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (fi < 3) {
#pragma unroll
        for (int pz = 0; pz < kSPZ; ++pz) av[pz] = avn[pz];
      }
      if (fi == 1 || fi == 3) wg_barrier();                 // mid-pass / end-of-pass barrier
      if (fi == 3) {                                        // V[(p+1) & 1] is complete now
        const float2* Vn = reinterpret_cast<const float2*>(V + ((p + 1) & 1) * kSVSZ);
#pragma unroll
        for (int pz = 0; pz < kSPZ; ++pz) av[pz] = Vn[((pz * 16 + wave * 4) * 16) * 4 + aoff];
      }
    }
  }
  __syncthreads();                                          // (drains the redundant last requests)

  // ---- output transform.  Along x inside the wave: r0 = M0 + M1 + M2, r1 = M1 - M2 - M3.
  // (the last barrier of the loop: every wave is done with V, the staging waves never write
  // it again, so the exchange buffer may alias it)
  float4* X4 = reinterpret_cast<float4*>(X);
#pragma unroll
  for (int mr = 0; mr < kSTZ; ++mr)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      const f32x4 m0 = acc[0][mr][nr], m1 = acc[1][mr][nr], m2 = acc[2][mr][nr], m3 = acc[3][mr][nr];
      X4[(((wave * 2 + 0) * kSTZ + mr) * NR + nr) * 64 + lane] =
          make_float4(m0[0] + m1[0] + m2[0], m0[1] + m1[1] + m2[1], m0[2] + m1[2] + m2[2], m0[3] + m1[3] + m2[3]);
      X4[(((wave * 2 + 1) * kSTZ + mr) * NR + nr) * 64 + lane] =
          make_float4(m1[0] - m2[0] - m3[0], m1[1] - m2[1] - m3[1], m1[2] - m2[2] - m3[2], m1[3] - m2[3] - m3[3]);
    }
  __syncthreads();                                          // E1
  // Along y across waves (fy = wave): out[0] = P0 + P1 + P2, out[1] = P1 - P2 - P3; wave w
  // finishes z-slice w: bias, statistics, 4x4 quad transpose, 16-byte stores.
  float* yb = a.y + (size_t)n * a.D * a.H * a.W * a.cout_p;
  const int jq = lane & 3;
  const int oz = z0 + wave;
  float s1[NR], s2[NR];
#pragma unroll
  for (int nr = 0; nr < NR; ++nr) {
    const int ch = (nb0 + nr) * 16 + mrow;
    const bool ch_ok = ch < a.cout_p;
    const float bvl = (a.bias && ch < a.cout_p16) ? a.bias[ch] : 0.f;
    s1[nr] = 0.f; s2[nr] = 0.f;
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) {
      float4 p[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) p[w] = X4[(((w * 2 + ox) * kSTZ + wave) * NR + nr) * 64 + lane];
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
    }
  }
  if (a.stats) {
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) {
      float t1 = s1[nr], t2 = s2[nr];
      t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
      t1 += __shfl_xor(t1, 32); t2 += __shfl_xor(t2, 32);
      if (kq == 0) {
        S[(wave * NR * 16 + nr * 16 + mrow) * 2 + 0] = t1;
        S[(wave * NR * 16 + nr * 16 + mrow) * 2 + 1] = t2;
      }
    }
    __syncthreads();                                        // E2
    if (tid < NR * 16) {
      const int ch = nb0 * 16 + tid;
      if (ch < a.cout_p) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          t1 += S[(w * NR * 16 + tid) * 2 + 0];
          t2 += S[(w * NR * 16 + tid) * 2 + 1];
        }
        double* st = a.stats + ((size_t)n * a.cout_p + ch) * 2;
        unsafeAtomicAdd(st + 0, (double)t1);
        unsafeAtomicAdd(st + 1, (double)t2);
      }
    }
  }
}

template <int NR, int ABL = 0>
static int launch_ws_nr(const WinoArgs& a, dim3 grid, size_t lds, hipStream_t s) {
  auto kern = conv3d_wino_ws_kernel<NR, ABL>;
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

int launch_conv3d_wino_ws(const WinoArgs& a, int nr, hipStream_t s) {
  const int nb = a.cout_p16 / 16;
  const int blocks = ((a.D + kSTZ - 1) / kSTZ) * ((a.H + kWTY - 1) / kWTY) * ((a.W + kWTX - 1) / kWTX);
  dim3 grid(blocks, (nb + nr - 1) / nr, a.N);
  const_cast<WinoArgs&>(a).abl = JH_ENV_KNOB("JH_WS_ABL") > 0 ? JH_ENV_KNOB("JH_WS_ABL") : 0;
  const size_t xfloats = (size_t)4 * 2 * kSTZ * nr * 64 * 4;                       // exchange, aliases V
  JH_REQUIRE(xfloats <= (size_t)2 * kSVSZ, "wino (wave-specialised) exchange buffer");
  size_t lds = (size_t)(kSNP * 8 + 2 * kSVSZ + 4 * nr * 16 * 2) * sizeof(float);
  lds += (a.in_stats ? (size_t)2 * a.cin_p : 0) * sizeof(float);
  JH_REQUIRE(lds <= 160 * 1024, "wino (wave-specialised) LDS");
  if (nr == 3 && a.abl == 1) return launch_ws_nr<3, 1>(a, grid, lds, s);
  if (nr == 3 && a.abl == 2) return launch_ws_nr<3, 2>(a, grid, lds, s);
  if (nr == 3 && a.abl == 3) return launch_ws_nr<3, 3>(a, grid, lds, s);
  if (nr == 3) return launch_ws_nr<3>(a, grid, lds, s);
  if (nr == 2) return launch_ws_nr<2>(a, grid, lds, s);
  return launch_ws_nr<1>(a, grid, lds, s);
}

}  // namespace jh
