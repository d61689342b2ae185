// ReprojectionLayer for gfx950: per-view heatmaps -> calibrated voxel grid.
//
// Replaces jarvis/hybridnet/repro_layer.py:40-119 (reprojectPoints,
// _get_heatmap_value, forward).  The reference materialises a (J, C*G^3)
// gather (289 MB at 12 cameras / 64^3); here the volume is produced directly:
//
//   1. repro_coarse_kernel  projects the (G/2)^3 coarse grid into every
//      camera (pinhole + 2-term radial distortion, clamp to the crop) and
//      stores (u, v) per camera and coarse voxel          (3 MB, L2 resident)
//   2. repro_gather_kernel  per fine voxel and camera: trilinear x2 upsampling
//      of (u, v), integer heatmap index, gather of the contiguous J-vector
//      from the channel-last heatmap, mean over cameras.
//
// INTEGER-PATH PARITY.  The heatmap index trunc(v/2)*hs + trunc(u/2) must be
// bit-identical to the reference, so the float arithmetic below reproduces
// the reference's op sequence exactly, one IEEE-rounded operation per torch
// op (explicit *_rn intrinsics, file built with -ffp-contract=off):
//   * the (x,y,z,1) @ cameraMatrix product is the k-ordered chain
//     fma(1,m3, fma(z,m2, fma(y,m1, x*m0)))  (what torch's CPU GEMM does for K=4)
//   * F.interpolate(mode='trilinear', align_corners=False) is three nested
//     lerps, W innermost, each evaluated as fma(p0, w0, p1*w1)
// both established empirically against torch 2.10 CPU in the build container
// (tests/golden/make_golden.py) and pinned by tests/golden/reprojection.npz.
//
// Memory behaviour: the gather reads Jp*4 contiguous bytes per (voxel, camera);
// a wave owns 64 consecutive voxels, computes one index per lane and camera,
// then re-distributes the work with a wave shuffle so that lane q handles
// (voxel q / Q, channel quad q % Q): loads are 16-byte, output stores are
// fully coalesced 16-byte writes of 64*Jp*4 contiguous bytes per wave.
#include <algorithm>
#include "jh_common.h"

namespace jh {

struct ReproCalib {
  const float* cam;    // [C][4][3]
  const float* intr;   // [C][3][3]  (principal point in row 2)
  const float* dist;   // [C][5]
};

__global__ __launch_bounds__(256) void repro_coarse_kernel(
    ReproCalib cal, const int* __restrict__ center3d, const int* __restrict__ center_hm,
    float2* __restrict__ coarse, int C, int Gh, float spacing, int hs) {
  const int t = blockIdx.z, c = blockIdx.y;
  const int vox = blockIdx.x * blockDim.x + threadIdx.x;
  const int nvox = Gh * Gh * Gh;
  if (vox >= nvox) return;
  const int k = vox % Gh, j = (vox / Gh) % Gh, i = vox / (Gh * Gh);
  const int half = Gh / 2;
  // grid * GRID_SPACING * 2 + center   (repro_layer.py:26-36,113)
  const float gx = __fadd_rn(__fmul_rn(__fmul_rn((float)(i - half), spacing), 2.f), (float)center3d[t * 3 + 0]);
  const float gy = __fadd_rn(__fmul_rn(__fmul_rn((float)(j - half), spacing), 2.f), (float)center3d[t * 3 + 1]);
  const float gz = __fadd_rn(__fmul_rn(__fmul_rn((float)(k - half), spacing), 2.f), (float)center3d[t * 3 + 2]);
  const float* M = cal.cam + c * 12;
  float p[3];
#pragma unroll
  for (int col = 0; col < 3; ++col) {
    float a = __fmul_rn(gx, M[0 * 3 + col]);
    a = __fmaf_rn(gy, M[1 * 3 + col], a);
    a = __fmaf_rn(gz, M[2 * 3 + col], a);
    a = __fmaf_rn(1.f, M[3 * 3 + col], a);
    p[col] = a;
  }
  const float* K = cal.intr + c * 9;
  const float cx = K[6], cy = K[7], fx = K[0], fy = K[4];
  const float k1 = cal.dist[c * 5 + 0], k2 = cal.dist[c * 5 + 1];
  float u = __fsub_rn(__fdiv_rn(p[0], p[2]), cx);
  float v = __fsub_rn(__fdiv_rn(p[1], p[2]), cy);
  const float a1 = __fdiv_rn(u, fx), a2 = __fdiv_rn(v, fy);
  const float r2 = __fadd_rn(__fmul_rn(a1, a1), __fmul_rn(a2, a2));
  const float dd = __fadd_rn(1.f, __fmul_rn(__fadd_rn(k1, __fmul_rn(k2, r2)), r2));
  u = __fadd_rn(__fmul_rn(u, dd), cx);
  v = __fadd_rn(__fmul_rn(v, dd), cy);
  const int chx = center_hm[(t * C + c) * 2 + 0], chy = center_hm[(t * C + c) * 2 + 1];
  // clamp(u, chm-(hs-1), chm+hs-2) - chm + hs - 1   (repro_layer.py:65-68)
  u = fminf(fmaxf(u, (float)(chx - (hs - 1))), (float)(chx + hs - 2));
  v = fminf(fmaxf(v, (float)(chy - (hs - 1))), (float)(chy + hs - 2));
  u = __fsub_rn(__fadd_rn(__fsub_rn(u, (float)chx), (float)hs), 1.f);
  v = __fsub_rn(__fadd_rn(__fsub_rn(v, (float)chy), (float)hs), 1.f);
  coarse[((size_t)(t * C + c)) * nvox + vox] = make_float2(u, v);
}

// source index and lambdas of the x2 linear upsampling, align_corners=False
__device__ __forceinline__ void up2_axis(int d, int Gh, int* i0, int* i1, float* w0, float* w1) {
  float real = __fsub_rn(__fmul_rn(0.5f, __fadd_rn((float)d, 0.5f)), 0.5f);
  real = fmaxf(real, 0.f);
  int a = (int)floorf(real);
  if (a > Gh - 1) a = Gh - 1;
  float l1 = __fsub_rn(real, (float)a);
  l1 = fminf(fmaxf(l1, 0.f), 1.f);
  *i0 = a;
  *i1 = a + ((a < Gh - 1) ? 1 : 0);
  *w1 = l1;
  *w0 = __fsub_rn(1.f, l1);
}

__device__ __forceinline__ float lerp_ref(float p0, float p1, float w0, float w1) {
  return __fmaf_rn(p0, w0, __fmul_rn(p1, w1));
}

// Latency structure (the kernel is latency-, not bandwidth-bound): a workgroup owns 256
// consecutive fine voxels.  (1) ONE round trip stages the coarse (u, v) table rows those
// voxels interpolate from, for all cameras, in LDS (the 8 coarse taps per voxel and camera
// were 57 % of the vector-memory instructions); (2) per camera every lane computes its
// gather offset from LDS only, then the wave issues its Q independent 16-byte heatmap
// loads back to back.  Measured: batching several cameras' loads per lane (kCamBatch > 1)
// is SLOWER (0.32 ms at 4 vs 0.23 ms at 1 per 8 frames) -- occupancy, not per-wave
// memory parallelism, hides the latency here.
constexpr int kCamBatch = 1;

template <int Q>
__global__ __launch_bounds__(256) void repro_gather_kernel(
    const float2* __restrict__ coarse, const float* __restrict__ heat, float* __restrict__ vol,
    int* __restrict__ idx_out, int C, int G, int hs, int Jp, int heat_pad, int div255, int ci_n,
    int cj_n, FastDiv fgh, FastDiv frows, FastDiv fcjn, FastDiv fbpp, HeatLayout lay) {
  extern __shared__ __attribute__((aligned(16))) float2 ctab[];   // [C][ci_n][cj_n][Gh]
  const BlockId bid = xcd_block();             // consecutive planes share heatmap regions
  const int t = bid.y;
  const int Gh = G >> 1;
  const int nvox = G * G * G, nvox_c = Gh * Gh * Gh;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  // blocks never straddle an i-plane: plane p owns blocks [p*bpp, (p+1)*bpp)
  const int bpp = (G * G + 255) / 256;
  const int plane = (int)fd_div(bid.x, fbpp);
  const int vox0 = plane * G * G + ((int)bid.x - plane * bpp) * 256;
  const int vox_end = (plane + 1) * G * G;
  const int wave_vox0 = vox0 + (tid >> 6) * 64;
  const int vox = wave_vox0 + lane;
  const bool vox_ok = vox < vox_end;
  const int vv = vox_ok ? vox : vox_end - 1;
  const int k = vv % G, j = (vv / G) % G, i = vv / (G * G);
  int i0, i1, j0, j1, k0, k1;
  float wi0, wi1, wj0, wj1, wk0, wk1;
  up2_axis(i, Gh, &i0, &i1, &wi0, &wi1);
  up2_axis(j, Gh, &j0, &j1, &wj0, &wj1);
  up2_axis(k, Gh, &k0, &k1, &wk0, &wk1);

  // coarse rows needed by this block: lowest source row of its first voxel; the block
  // spans at most ci_n x cj_n coarse (i, j) rows (host-computed bounds)
  int ci_lo, cj_lo;
  {
    const int fi = vox0 / (G * G), fj = (vox0 / G) % G;
    int a0, a1; float w0, w1;
    up2_axis(fi, Gh, &a0, &a1, &w0, &w1);
    ci_lo = a0;
    up2_axis(fj, Gh, &a0, &a1, &w0, &w1);
    cj_lo = (cj_n >= Gh) ? 0 : a0;
  }
  const int rows = ci_n * cj_n;
  // (index arithmetic through host-prepared multiply-shifts: the three run-time divisions per element
  // of this loop were a quarter of the kernel's vector instructions)
  for (int idx = tid; idx < C * rows * Gh; idx += 256) {
    const int q1 = (int)fd_div((unsigned)idx, fgh), ck = idx - q1 * Gh;
    const int c = (int)fd_div((unsigned)q1, frows), r = q1 - c * rows;
    const int ri = (int)fd_div((unsigned)r, fcjn), rj = r - ri * cj_n;
    const int ci = min(ci_lo + ri, Gh - 1), cj = min(cj_lo + rj, Gh - 1);
    ctab[idx] = coarse[(size_t)(t * C + c) * nvox_c + (ci * Gh + cj) * Gh + ck];
  }
  const int r00 = ((i0 - ci_lo) * cj_n + (j0 - cj_lo)) * Gh, r01 = ((i0 - ci_lo) * cj_n + (j1 - cj_lo)) * Gh;
  const int r10 = ((i1 - ci_lo) * cj_n + (j0 - cj_lo)) * Gh, r11 = ((i1 - ci_lo) * cj_n + (j1 - cj_lo)) * Gh;
  // heat_pad = 0: the heatmap is stored without the reference's 1-pixel zero
  // border (the border is virtual); heat_pad = 1: it is stored padded.
  const int Hh = hs - 2 + 2 * heat_pad;
  float4 acc[Q];
#pragma unroll
  for (int q = 0; q < Q; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  typedef float gf4 __attribute__((ext_vector_type(4)));
  // one buffer resource per CAMERA plane (uniform: scalar registers): camera c of frame t lives at
  // block c / cpb, frame t, local camera c % cpb of the (blocks, frames, cameras) heatmap layout
  const int plane_bytes = Hh * Hh * Jp * 4;                 // (< 2^31: checked by the launcher)
  const float* heat_t = heat + (size_t)t * lay.frame_stride;
  __syncthreads();

  int cblk = 0, cloc = 0;                                   // c = cblk * cams_per_block + cloc
  for (int cb = 0; cb < C; cb += kCamBatch) {
    static_assert(kCamBatch == 1, "one buffer resource per camera");
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(heat_t + (size_t)cblk * lay.block_stride + (size_t)cloc * (plane_bytes >> 2)), 0,
        plane_bytes, 0x00020000);
    if (++cloc == lay.cams_per_block) { cloc = 0; ++cblk; }
    int src[kCamBatch];
#pragma unroll
    for (int cc = 0; cc < kCamBatch; ++cc) {
      const int c = min(cb + cc, C - 1);
      const float2* cz = ctab + (size_t)c * rows * Gh;
      const float2 p000 = cz[r00 + k0], p001 = cz[r00 + k1];
      const float2 p010 = cz[r01 + k0], p011 = cz[r01 + k1];
      const float2 p100 = cz[r10 + k0], p101 = cz[r10 + k1];
      const float2 p110 = cz[r11 + k0], p111 = cz[r11 + k1];
      // W (k) innermost, then H (j), then D (i)
      const float u00 = lerp_ref(p000.x, p001.x, wk0, wk1), u01 = lerp_ref(p010.x, p011.x, wk0, wk1);
      const float u10 = lerp_ref(p100.x, p101.x, wk0, wk1), u11 = lerp_ref(p110.x, p111.x, wk0, wk1);
      const float v00 = lerp_ref(p000.y, p001.y, wk0, wk1), v01 = lerp_ref(p010.y, p011.y, wk0, wk1);
      const float v10 = lerp_ref(p100.y, p101.y, wk0, wk1), v11 = lerp_ref(p110.y, p111.y, wk0, wk1);
      const float u0 = lerp_ref(u00, u01, wj0, wj1), u1 = lerp_ref(u10, u11, wj0, wj1);
      const float v0 = lerp_ref(v00, v01, wj0, wj1), v1 = lerp_ref(v10, v11, wj0, wj1);
      const float u = lerp_ref(u0, u1, wi0, wi1);
      const float v = lerp_ref(v0, v1, wi0, wi1);
      const int iu = (int)__fdiv_rn(u, 2.f), iv = (int)__fdiv_rn(v, 2.f);
      if (idx_out && vox_ok && cb + cc < C) idx_out[((size_t)(t * C + c)) * nvox + vox] = iv * hs + iu;
      // padded -> stored heatmap coordinates (the zero border may be virtual)
      const int hx = iu - 1 + heat_pad, hy = iv - 1 + heat_pad;
      // byte offset inside the camera's heatmap; taps on the (virtual) zero border get bit 31: the
      // buffer load below is then out of range and returns 0 -- no branch, no 64-bit address
      src[cc] = (int)0x80000000;
      if (cb + cc < C && hx >= 0 && hy >= 0 && hx < Hh && hy < Hh)
        src[cc] = ((hy * Hh + hx) * Jp) * 4;
    }
    float4 h[Q][kCamBatch];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const int item = q * 64 + lane;          // (voxel in wave, channel quad)
      const int vsrc = item / Q, quad = item % Q;
#pragma unroll
      for (int cc = 0; cc < kCamBatch; ++cc) {
        const int off = __shfl(src[cc], vsrc);
        const gf4 v = __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(hrs, off + quad * 16, 0, 0));
        h[q][cc] = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
#pragma unroll
      for (int cc = 0; cc < kCamBatch; ++cc) {     // camera order = the reference's sum order
        acc[q].x += h[q][cc].x; acc[q].y += h[q][cc].y; acc[q].z += h[q][cc].z; acc[q].w += h[q][cc].w;
      }
  }
  // mean over cameras, then the /255 of hybridnet/model.py:72: two IEEE divisions per value in the
  // reference.  x / c is evaluated as q = x rc, r = fma(-q, c, x), q' = fma(r, rc, q) with rc = RN(1 / c):
  // by Markstein's theorem q' = RN(x / c) whenever nothing underflows -- checked exhaustively over all
  // 2^32 inputs for c = 3 ... 16, 18, 20, 24, 32 and 255: identical bits for every |x| >= 1e-37 (below
  // that, where the quotient is denormal, it can differ in the last denormal bit).  3 instead of 11
  // instructions per division; the epilogue was 21 % of the kernel's vector instructions.
  const float fc = (float)C;
  const float rfc = __fdiv_rn(1.f, fc), r255 = __fdiv_rn(1.f, 255.f);
  auto divc = [](float x, float c, float rc) __attribute__((always_inline)) {
    const float q = __fmul_rn(x, rc);
    return __fmaf_rn(__fmaf_rn(-q, c, x), rc, q);
  };
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int item = q * 64 + lane;
    const int vsrc = item / Q, quad = item % Q;
    if (wave_vox0 + vsrc < vox_end) {
      float4 r;
      r.x = divc(acc[q].x, fc, rfc); r.y = divc(acc[q].y, fc, rfc);
      r.z = divc(acc[q].z, fc, rfc); r.w = divc(acc[q].w, fc, rfc);
      if (div255) {
        r.x = divc(r.x, 255.f, r255); r.y = divc(r.y, 255.f, r255);
        r.z = divc(r.z, 255.f, r255); r.w = divc(r.w, 255.f, r255);
      }
      *reinterpret_cast<float4*>(vol + ((size_t)t * nvox + wave_vox0 + vsrc) * Jp + quad * 4) = r;
    }
  }
}

int launch_reproject(const float* cam, const float* intr, const float* dist, const int* center3d,
                     const int* center_hm, const float* heat, float2* coarse, float* vol,
                     int* idx_out, int T, int C, int G, float spacing, int hs, int Jp,
                     int heat_pad, int div255, hipStream_t s, const HeatLayout* layout) {
  const int Gh = G / 2;
  const int Hst = hs - 2 + 2 * heat_pad;
  HeatLayout lay;                        // default: dense (T, C, Hh, Hh, Jp)
  lay.cams_per_block = C;
  lay.frame_stride = (size_t)C * Hst * Hst * Jp;
  lay.block_stride = 0;
  if (layout) lay = *layout;
  JH_REQUIRE(lay.cams_per_block >= 1 && C % lay.cams_per_block == 0, "cameras per heatmap block");
  JH_REQUIRE(G % 2 == 0 && Jp % 8 == 0 && Jp <= 64, "reprojection shape");
  JH_REQUIRE((size_t)hs * hs * Jp * 4 < ((size_t)1 << 31), "one camera's heatmap exceeds 2 GB");
  ReproCalib cal{cam, intr, dist};
  const int nvc = Gh * Gh * Gh;
  hipLaunchKernelGGL(repro_coarse_kernel, dim3((nvc + 255) / 256, C, T), dim3(256), 0, s, cal,
                     center3d, center_hm, coarse, C, Gh, spacing, hs);
  JH_CHECK_HIP(hipGetLastError());
  // one block = up to 256 consecutive voxels of ONE i-plane
  const int bpp = (G * G + 255) / 256;
  dim3 grid(G * bpp, T);
  // coarse rows such a block interpolates from: 2 in i, and in j the rows spanned / 2 + 2
  const int ci_n = 2;
  const int cj_n = (256 >= G * G) ? Gh : std::min(Gh, ((256 + G - 1) / G + 1) / 2 + 2);
  const size_t lds = (size_t)C * ci_n * cj_n * Gh * sizeof(float2);
  JH_REQUIRE(lds <= 64 * 1024, "coarse table tile does not fit LDS");
#define JH_RG(QV)                                                                              \
  case QV:                                                                                     \
    hipLaunchKernelGGL(repro_gather_kernel<QV>, grid, dim3(256), lds, s, coarse, heat, vol,    \
                       idx_out, C, G, hs, Jp, heat_pad, div255, ci_n, cj_n, make_fastdiv(Gh),   \
                       make_fastdiv(ci_n * cj_n), make_fastdiv(cj_n), make_fastdiv(bpp), lay); \
    break;
  switch (Jp / 4) {
    JH_RG(2) JH_RG(4) JH_RG(6) JH_RG(8) JH_RG(10) JH_RG(12) JH_RG(14) JH_RG(16)
    default: JH_REQUIRE(false, "unsupported joint count");
  }
#undef JH_RG
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

}  // namespace jh
