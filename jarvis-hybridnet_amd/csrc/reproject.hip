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
#include <atomic>
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
// ... of a (u, v) pair at once: v_pk_mul_f32 + v_pk_fma_f32, the same two IEEE operations per component
typedef float rp2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ rp2 lerp_ref2(rp2 p0, rp2 p1, float w0, float w1) {
  return __builtin_elementwise_fma(p0, (rp2){w0, w0}, p1 * (rp2){w1, w1});
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

// ---------------------------------------------------------------------------------------------
// Cube form of the gather (round 3).  The gather above reads 16 bytes per lane straight from the
// heatmaps: 302 MB of logical L1 traffic per frame in (voxel, camera)-scattered 96-byte segments,
// with the per-camera chain LDS -> index -> gather exposed on every camera (0.93 ms per 32 frames,
// 18 % of HBM).  Here a workgroup owns a CI x 16 x 16 cube of fine voxels and walks the cameras;
// per camera
//   * the bounding box of the cube's projection comes from the 8 corner nodes of the coarse (u, v)
//     field that encloses the cube (a pinhole projection is monotone along lines, so the extremes
//     over the cube sit at corners; one pixel of margin for the rounding of the interpolation),
//   * the pixels inside it -- whole rows of the channel-last heatmap, i.e. fully coalesced 16-byte
//     loads -- are staged ONCE in LDS and every tap is Q ds_read_b128 of the lane's OWN voxel (no
//     cross-lane index shuffle): the 2.5 .. 5 uses a staged pixel sees inside one cube come out of
//     LDS instead of L1.  A staged pixel takes Q + 1 sixteen-byte slots and a row an odd number of
//     pixels, so the start banks of 16 neighbouring pixels -- along x or along y -- are all
//     different and the 16 lanes of one LDS pass do not collide,
//   * the patch of camera c+1 is in flight -- global_load_lds straight into the second patch
//     buffer, no staging registers -- and the indices of camera c+1 are being interpolated while
//     camera c is gathered: one workgroup barrier per camera, no exposed round trip.
// The integer index path is the one above, instruction for instruction (up2_axis, lerp_ref, the
// truncating /2).  A tap that falls outside the staged patch (never seen; a non-monotone lens model
// would be needed) or a patch larger than the LDS budget is read from global memory by that lane:
// the result does not depend on the box, only the speed does.
constexpr int kCubeJ = 16, kCubeK = 16, kTabJ = 10, kTabK = 10;
constexpr int kGeoBytes = 64 * 16;                   // boxes of up to 64 cameras
constexpr int kCubePatchOff(int ci) { return 2 * (ci / 2 + 2) * kTabJ * kTabK * 8 + 256 + kGeoBytes; }

struct CubeArgs {
  const float2* coarse;
  const float* heat;
  float* vol;
  int* idx_out;
  int C, G, hs, heat_pad, div255, patch_bytes;   // patch_bytes: size of each of the two LDS patch buffers
  int patch_limit;                               // largest box that is staged (<= patch_bytes; test knob)
  HeatLayout lay;
  int abl;       // JH_REPRO_ABL bit mask (timing experiments only): 1 no patch loads, 2 no LDS gather,
                 // 4 no tap interpolation after camera 0, 8 no stores, 16 no table prefetch
};

// raw LDS reads of Q consecutive 16-byte words (see repro_cube_kernel: invisible to the compiler's wait-count pass)
typedef float cube_f4 __attribute__((ext_vector_type(4)));
template <int Q>
__device__ __forceinline__ void lds_read_quads(cube_f4 (&h)[Q], unsigned addr) {
  if constexpr (Q == 2)
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(h[0]), "=&v"(h[1]) : "v"(addr));
  else if constexpr (Q == 4)
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\t"
                 "ds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]) : "v"(addr));
  else if constexpr (Q == 6)
    asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %6 offset:16\n\tds_read_b128 %2, %6 offset:32\n\t"
                 "ds_read_b128 %3, %6 offset:48\n\tds_read_b128 %4, %6 offset:64\n\tds_read_b128 %5, %6 offset:80\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]) : "v"(addr));
  else {
    static_assert(Q == 8, "channel quads of the cube gather");
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\tds_read_b128 %2, %8 offset:32\n\t"
                 "ds_read_b128 %3, %8 offset:48\n\tds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                 "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(h[0]), "=&v"(h[1]), "=&v"(h[2]), "=&v"(h[3]), "=&v"(h[4]), "=&v"(h[5]), "=&v"(h[6]), "=&v"(h[7])
                 : "v"(addr));
  }
}

template <int Q, int CI, int NT>
__global__ __launch_bounds__(NT) void repro_cube_kernel(CubeArgs a) {
  // voxels per thread: a wave's group g = w * VPT + v is the i-plane g % CI and the j-rows 4 (g / CI) .. + 3 of the
  // cube, 16 k each -- j-row groups vary slowest over the waves, so in a cube that is ragged along j (72 = 4.5 x 16)
  // the waves without a voxel inside the grid are spread evenly over the SIMDs and skip the tap arithmetic
  constexpr int VPT = CI * 256 / NT;
  constexpr int NI = CI / 2 + 2, NTAB = NI * kTabJ * kTabK;
  constexpr int TPT = (NTAB + NT - 1) / NT;         // table entries staged per thread
  constexpr int JPB = Q * 16;                       // bytes per heatmap pixel in memory
  constexpr int SPX = Q + 1;                        // 16-byte slots per staged pixel in LDS
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* ctab = reinterpret_cast<float2*>(smem);   // [2][NTAB] cube-local coarse (u, v) of two cameras
  constexpr int kZeroOff = 2 * NTAB * 8;            // one all-zero pixel (the virtual border)
  constexpr int kGeoOff = kZeroOff + 256;           // [C] int4 (x0, y0, pw, ph): the patch boxes of all cameras
  constexpr int kPatchOff = kCubePatchOff(CI);
  typedef cube_f4 gf4;
  // The LDS reads of the camera loop -- tap interpolation tables and the gather itself -- are raw ds_read instructions
  // (inline asm, waited for explicitly): the compiler cannot tell that the patch DMA in flight (global_load_lds into the
  // OTHER patch buffer) does not alias them and puts `s_waitcnt vmcnt(0)` in front of the first LDS read it emits after
  // a DMA -- a wave then waits for the patch of camera c+1 it has just requested BEFORE it gathers camera c, and the round
  // trip is exposed in every camera step instead of running under the step's arithmetic.
  typedef __attribute__((address_space(3))) unsigned char lds_u8;
  const unsigned lds0 = (unsigned)(size_t)(lds_u8*)smem;

  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.G, Gh = G >> 1, C = a.C, hs = a.hs;
  // (G not a multiple of 16 -- the reference's shipped 72^3 grid: the last cube along j and k is ragged; its voxels past
  //  the grid are computed from clamped table entries like any others and dropped at the stores)
  const int nck = (G + kCubeK - 1) / kCubeK, ncj = (G + kCubeJ - 1) / kCubeJ;
  // (tried, round 5: a PERSISTENT form -- one workgroup per CU walking the (frame, cube) list, so that the stores of cube k
  //  drain under the prologue of cube k+1 instead of holding the CU's LDS until the workgroup has ended: 3 % faster than
  //  one workgroup per cube inside the same binary, but the loop costs 10 registers -- 128 with 11 spills -- and the
  //  binary 8 %: 0.741 against 0.685 ms)
  const BlockId bid = xcd_block();
  const int t = bid.y;
  const int cube = (int)bid.x;
  const int ck = cube % nck, cj = (cube / nck) % ncj, ci = cube / (nck * ncj);
  const int I0 = ci * CI, J0 = cj * kCubeJ, K0 = ck * kCubeK;
  const int bi = I0 / 2 - 1, bj = J0 / 2 - 1, bk = K0 / 2 - 1;       // coarse index of local entry 0
  const int Hh = hs - 2 + 2 * a.heat_pad;
  const int plane_bytes = Hh * Hh * JPB;
  const size_t nvox = (size_t)G * G * G, nvox_c = (size_t)Gh * Gh * Gh;
  const float* heat_t = a.heat + (size_t)t * a.lay.frame_stride;

  if (tid < JPB / 4) reinterpret_cast<float*>(smem + kZeroOff)[tid] = 0.f;

  // ---- per-lane constants: the voxels this lane owns ---------------------------------------------
  int cidx[VPT];                     // entry of p000 in the local table
  float wi0[VPT], wi1[VPT], wj0[VPT], wj1[VPT], wk0, wk1;
  {
    int k0, k1;
    up2_axis(K0 + (lane & 15), Gh, &k0, &k1, &wk0, &wk1);
#pragma unroll
    for (int g = 0; g < VPT; ++g) {
      const int gi = w * VPT + g, il = gi % CI, jl = ((gi / CI) << 2) + (lane >> 4);
      int i0, i1, j0, j1;
      up2_axis(I0 + il, Gh, &i0, &i1, &wi0[g], &wi1[g]);
      up2_axis(J0 + jl, Gh, &j0, &j1, &wj0[g], &wj1[g]);
      cidx[g] = ((i0 - bi) * kTabJ + (j0 - bj)) * kTabK + (k0 - bk);
    }
  }
  // local table entries of this thread (staging role)
  int ctab_src[TPT];
#pragma unroll
  for (int e = 0; e < TPT; ++e) {
    const int n = min(e * NT + tid, NTAB - 1);
    const int li = n / (kTabJ * kTabK), lj = (n / kTabK) % kTabJ, lk = n % kTabK;
    ctab_src[e] = (min(max(bi + li, 0), Gh - 1) * Gh + min(max(bj + lj, 0), Gh - 1)) * Gh +
                  min(max(bk + lk, 0), Gh - 1);
  }
  // min / max over the 8 lanes of a half row: xor 1, xor 2 (quad permutes), mirror of the half row
  auto min8 = [](float x) __attribute__((always_inline)) {
    x = fminf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)));
    x = fminf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)));
    return fminf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true)));
  };
  auto max8 = [](float x) __attribute__((always_inline)) {
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0xB1, 0xF, 0xF, true)));
    x = fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x4E, 0xF, 0xF, true)));
    return fmaxf(x, __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), 0x141, 0xF, 0xF, true)));
  };
  // (the tables of cameras 0 and 1 are requested BEFORE the boxes are computed -- their round trip then runs under
  //  the boxes' own loads and arithmetic instead of after them: two dependent round trips in the prologue, not three)
  float2 tpre[2][TPT];
#pragma unroll
  for (int e = 0; e < TPT; ++e)
    if (e * NT + tid < NTAB) {
      tpre[0][e] = a.coarse[(size_t)(t * C) * nvox_c + ctab_src[e]];
      if (C > 1) tpre[1][e] = a.coarse[(size_t)(t * C + 1) * nvox_c + ctab_src[e]];
    }
  // ---- boxes of ALL cameras, once per cube (prologue; wave w takes cameras w, w + waves, ...): lane
  // n & 7 interpolates the cube's n-th corner VOXEL from the coarse field in global memory (the field is
  // piecewise trilinear and the projection monotone along lines: the extremes of (u, v) over the cube
  // are at its corner voxels); padded-heatmap pixel = trunc(u / 2), stored pixel = that - 1 +
  // heat_pad, one pixel of margin for the rounding of the interpolation
  {
    int i0, i1, j0, j1, k0, k1;
    float ki0, ki1, kj0, kj1, kk0, kk1;
    up2_axis(I0 + ((lane & 4) ? CI - 1 : 0), Gh, &i0, &i1, &ki0, &ki1);
    up2_axis(J0 + ((lane & 2) ? kCubeJ - 1 : 0), Gh, &j0, &j1, &kj0, &kj1);
    up2_axis(K0 + ((lane & 1) ? kCubeK - 1 : 0), Gh, &k0, &k1, &kk0, &kk1);
    for (int c = w; c < C; c += NT / 64) {
      const float2* cz = a.coarse + (size_t)(t * C + c) * nvox_c;
      const float2 p000 = cz[(i0 * Gh + j0) * Gh + k0], p001 = cz[(i0 * Gh + j0) * Gh + k1];
      const float2 p010 = cz[(i0 * Gh + j1) * Gh + k0], p011 = cz[(i0 * Gh + j1) * Gh + k1];
      const float2 p100 = cz[(i1 * Gh + j0) * Gh + k0], p101 = cz[(i1 * Gh + j0) * Gh + k1];
      const float2 p110 = cz[(i1 * Gh + j1) * Gh + k0], p111 = cz[(i1 * Gh + j1) * Gh + k1];
      const float u00 = lerp_ref(p000.x, p001.x, kk0, kk1), u01 = lerp_ref(p010.x, p011.x, kk0, kk1);
      const float u10 = lerp_ref(p100.x, p101.x, kk0, kk1), u11 = lerp_ref(p110.x, p111.x, kk0, kk1);
      const float v00 = lerp_ref(p000.y, p001.y, kk0, kk1), v01 = lerp_ref(p010.y, p011.y, kk0, kk1);
      const float v10 = lerp_ref(p100.y, p101.y, kk0, kk1), v11 = lerp_ref(p110.y, p111.y, kk0, kk1);
      const float uu = lerp_ref(lerp_ref(u00, u01, kj0, kj1), lerp_ref(u10, u11, kj0, kj1), ki0, ki1);
      const float vv = lerp_ref(lerp_ref(v00, v01, kj0, kj1), lerp_ref(v10, v11, kj0, kj1), ki0, ki1);
      const float ulo = min8(uu), uhi = max8(uu), vlo = min8(vv), vhi = max8(vv);
      const int sh = -1 + a.heat_pad;
      const int x0 = max((int)(ulo * 0.5f) + sh - 1, 0), y0 = max((int)(vlo * 0.5f) + sh - 1, 0);
      const int x1 = min((int)(uhi * 0.5f) + sh + 1, Hh - 1), y1 = min((int)(vhi * 0.5f) + sh + 1, Hh - 1);
      if (lane == 0)
        *reinterpret_cast<int4*>(smem + kGeoOff + c * 16) = make_int4(x0, y0, max(x1 - x0 + 1, 1), max(y1 - y0 + 1, 1));
    }
  }

  // patch geometry of one camera (uniform values): box origin, width, height, odd LDS row pitch
  struct Geo { int x0, y0, pw, ph, pwp, slots, big; };
  auto geometry = [&](int4 b) __attribute__((always_inline)) {
    Geo g;
    g.x0 = __builtin_amdgcn_readfirstlane(b.x); g.y0 = __builtin_amdgcn_readfirstlane(b.y);
    g.pw = __builtin_amdgcn_readfirstlane(b.z); g.ph = __builtin_amdgcn_readfirstlane(b.w);
    g.pwp = g.pw | 1;
    g.slots = g.ph * g.pwp * SPX;                        // 16-byte LDS slots
    g.big = g.slots * 16 > a.patch_limit;
    return g;
  };
  auto box = [&](int c) __attribute__((always_inline)) {
    return *reinterpret_cast<const int4*>(smem + kGeoOff + min(c, C - 1) * 16);
  };
  // heatmap plane of a camera: camera c lives at block c / cams_per_block, local camera c % cams_per_block of the
  // (blocks, frames, cameras) layout.  Walked incrementally -- a run-time integer division per camera is ~28
  // vector instructions even for a uniform value, and the kernel is bound by vector-instruction issue.
  struct CamPos { int blk, loc; };
  auto cam_next = [&](CamPos p) __attribute__((always_inline)) {
    if (++p.loc == a.lay.cams_per_block) { p.loc = 0; ++p.blk; }
    return p;
  };
  auto cam_base = [&](CamPos p) __attribute__((always_inline)) {
    return heat_t + (size_t)p.blk * a.lay.block_stride + (size_t)p.loc * (plane_bytes >> 2);
  };
  // where the taps of camera c (geometry g) of this lane's voxels are: an LDS byte offset (>= 0) or
  // bit 31 + the byte offset inside the camera's heatmap (that lane then reads global memory)
  // Tap of voxel v in camera c, in two halves so that the camera loop can put ONE LDS round trip under both its table
  // reads and the gather of the previous camera: tap_request issues the four ds_read2_b64 of the table entries (cidx,
  // cidx + 1 | + kTabK | + kTabJ kTabK | + both; offsets in 8-byte units) WITHOUT waiting, tap_finish interpolates
  // (packed fp32: (u, v) pairs, the same IEEE operation per component as the scalar form; W (k) innermost, then H (j),
  // then D (i)) and turns the index into an LDS / global offset.
  struct TapRaw { gf4 t0, t1, t2, t3; };
  auto tap_request = [&](int c, int v, TapRaw& r) __attribute__((always_inline)) {
    const unsigned ta = lds0 + (unsigned)(((c & 1) * NTAB + cidx[v]) * 8);
    asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:1\n\t"
                 "ds_read2_b64 %1, %4 offset0:%5 offset1:%6\n\t"
                 "ds_read2_b64 %2, %4 offset0:%7 offset1:%8\n\t"
                 "ds_read2_b64 %3, %4 offset0:%9 offset1:%10"
                 : "=&v"(r.t0), "=&v"(r.t1), "=&v"(r.t2), "=&v"(r.t3)
                 : "v"(ta), "n"(kTabK), "n"(kTabK + 1), "n"(kTabJ * kTabK), "n"(kTabJ * kTabK + 1),
                   "n"(kTabJ * kTabK + kTabK), "n"(kTabJ * kTabK + kTabK + 1));
  };
  auto tap_wait = [&](TapRaw& r) __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r.t0), "+v"(r.t1), "+v"(r.t2), "+v"(r.t3));
  };
  auto tap_finish = [&](int c, const Geo& g, int v, const TapRaw& r) __attribute__((always_inline)) -> int {
    const rp2 p000 = (rp2){r.t0[0], r.t0[1]}, p001 = (rp2){r.t0[2], r.t0[3]};
    const rp2 p010 = (rp2){r.t1[0], r.t1[1]}, p011 = (rp2){r.t1[2], r.t1[3]};
    const rp2 p100 = (rp2){r.t2[0], r.t2[1]}, p101 = (rp2){r.t2[2], r.t2[3]};
    const rp2 p110 = (rp2){r.t3[0], r.t3[1]}, p111 = (rp2){r.t3[2], r.t3[3]};
    const rp2 c00 = lerp_ref2(p000, p001, wk0, wk1), c01 = lerp_ref2(p010, p011, wk0, wk1);
    const rp2 c10 = lerp_ref2(p100, p101, wk0, wk1), c11 = lerp_ref2(p110, p111, wk0, wk1);
    const rp2 d0 = lerp_ref2(c00, c01, wj0[v], wj1[v]), d1 = lerp_ref2(c10, c11, wj0[v], wj1[v]);
    const rp2 uv = lerp_ref2(d0, d1, wi0[v], wi1[v]);
    const float uu = uv[0], vv = uv[1];
    const int iu = (int)__fdiv_rn(uu, 2.f), iv = (int)__fdiv_rn(vv, 2.f);
    if (a.idx_out) {
      int ln = lane;
      asm volatile("" : "+v"(ln));
      const int gi = w * VPT + v, il = gi % CI, jl = ((gi / CI) << 2) + (ln >> 4);
      if (J0 + jl < G && K0 + (ln & 15) < G)
        a.idx_out[((size_t)(t * C + c)) * nvox + ((size_t)(I0 + il) * G + (J0 + jl)) * G + K0 + (ln & 15)] =
            iv * hs + iu;
    }
    const int hx = iu - 1 + a.heat_pad, hy = iv - 1 + a.heat_pad;
    const int px = hx - g.x0, py = hy - g.y0;
    int o = kZeroOff;                                                     // the (virtual) zero border
    if (hx >= 0 && hy >= 0 && hx < Hh && hy < Hh) {
      if (!g.big && px >= 0 && py >= 0 && px < g.pw && py < g.ph)
        o = kPatchOff + (c & 1) * a.patch_bytes + (py * g.pwp + px) * (SPX * 16);
      else
        o = (int)0x80000000 | ((hy * Hh + hx) * JPB);
    }
    return o;
  };
  auto tap_offsets = [&](int c, const Geo& g, int* off) __attribute__((always_inline)) {
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      TapRaw r;
      tap_request(c, v, r);
      tap_wait(r);
      off[v] = tap_finish(c, g, v, r);
    }
  };
  // patch of camera c -> LDS buffer c & 1.  LDS slot s (16 bytes) of a patch row = quad s % SPX of staged pixel
  // s / SPX; rows have a pitch of pwp pixels.  One WAVE stages one row at a time (rows w, w + waves, ...):
  // global_load_lds writes at a wave-uniform LDS base + 16 * lane, so within a round of 64 slots lane <-> slot,
  // and the pixels of a heatmap row are contiguous in memory -- a lane's source offset inside the row,
  // px * JPB + quad * 16, is a CONSTANT of the lane and the round (kept in registers for the first four rounds =
  // rows of up to 256 slots), pad slots carry an offset that fails the one range check `offset < pw * JPB`.
  // (Round 3 linearised the whole patch over the workgroup: a float multiply, two integer divisions and four
  // compares per slot -- a third of the kernel's vector instructions, and the kernel is bound by their issue.)
  constexpr int kPre = 4;
  int goff[kPre];
#pragma unroll
  for (int r = 0; r < kPre; ++r) {
    const int slot = r * 64 + lane, px = slot / SPX, quad = slot - px * SPX;
    goff[r] = quad < Q ? px * JPB + quad * 16 : 0x7fffffff;
  }
  auto load_patch = [&](int c, CamPos cp, const Geo& g) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) void lds_void;
    typedef const __attribute__((address_space(1))) void glb_void;
    const int rowslots = g.pwp * SPX, wlim = g.pw * JPB;                     // uniform
    const char* src = reinterpret_cast<const char*>(cam_base(cp)) + ((size_t)g.y0 * Hh + g.x0) * JPB;
    unsigned char* dst = smem + kPatchOff + (c & 1) * a.patch_bytes;
#pragma nounroll
    for (int row = w; row < g.ph; row += NT / 64) {                          // uniform
      const char* srow = src + (size_t)row * Hh * JPB;
      unsigned char* drow = dst + (size_t)row * rowslots * 16;
#pragma unroll
      for (int r = 0; r < kPre; ++r)
        if (r * 64 < rowslots && goff[r] < wlim)
          __builtin_amdgcn_global_load_lds((glb_void*)(srow + goff[r]), (lds_void*)(drow + r * 1024), 16, 0, 0);
#pragma nounroll
      for (int r = kPre; r * 64 < rowslots; ++r) {                           // (rows wider than 256 slots: rare)
        const int slot = r * 64 + lane, px = slot / SPX, quad = slot - px * SPX;
        if (quad < Q && px < g.pw)
          __builtin_amdgcn_global_load_lds((glb_void*)(srow + px * JPB + quad * 16), (lds_void*)(drow + r * 1024), 16, 0, 0);
      }
    }
  };

  // ---- prologue: tables of cameras 0 and 1, taps and patch of camera 0 ----------------------------
#pragma unroll
  for (int e = 0; e < TPT; ++e)
    if (e * NT + tid < NTAB) {
      ctab[e * NT + tid] = tpre[0][e];
      if (C > 1) ctab[NTAB + e * NT + tid] = tpre[1][e];
    }
  __syncthreads();
  Geo gc = geometry(box(0));
  CamPos cam_c{0, 0};                                  // camera c of the loop below
  if (!gc.big) load_patch(0, cam_c, gc);
  int4 box_n = box(1);                                // (a box is fetched one camera ahead of its use)
  int off_c[VPT];
  // (uniform per wave: does any of this wave's j-row groups lie inside the grid?)
  const bool wave_in = J0 + 4 * ((w * VPT) / CI) < G;
  if (wave_in) tap_offsets(0, gc, off_c);
  __syncthreads();

  rp2 acc[VPT][Q][2];                                 // (channel pairs: v_pk_add_f32, half the additions' instructions)
#pragma unroll
  for (int v = 0; v < VPT; ++v)
#pragma unroll
    for (int q = 0; q < Q; ++q) { acc[v][q][0] = (rp2){0.f, 0.f}; acc[v][q][1] = (rp2){0.f, 0.f}; }

  float2 tnext[TPT];                                  // table of camera c+2 at the top of iteration c
#pragma unroll
  for (int e = 0; e < TPT; ++e) {
    tnext[e] = make_float2(0.f, 0.f);
    if (2 < C && e * NT + tid < NTAB) tnext[e] = a.coarse[(size_t)(t * C + 2) * nvox_c + ctab_src[e]];
  }
  for (int c = 0; c < ((a.abl & 32) ? 1 : C); ++c) {
    // (the table entry requested in the previous iteration is taken over BEFORE this iteration's DMA is issued: vector
    //  memory operations return in order, and the number of DMA instructions is not a compile-time constant -- a wait for
    //  the entry placed after them would be a wait for all of them)
    float2 tv[TPT];
#pragma unroll
    for (int e = 0; e < TPT; ++e) {
      tv[e] = tnext[e];
      asm volatile("" : "+v"(tv[e].x), "+v"(tv[e].y));
    }
    // camera c+1: box, patch in flight into the other buffer; camera c+2: table entry in flight
    Geo gn = gc;
    const bool more = c + 1 < C;
    const CamPos cam_n = cam_next(cam_c);
    if (more) {
      if (!(a.abl & 64)) gn = geometry(box_n);
      box_n = box(c + 2);
      if (!gn.big && !(a.abl & 1)) load_patch(c + 1, cam_n, gn);
    }
    // table of camera c+3 requested now, committed at the END OF THE NEXT iteration (as camera (c+1)+2): a table entry
    // has two iterations to arrive.  (Requested and committed inside one iteration, its round trip was the floor under
    // every camera step: with patch loads, LDS gather and tap arithmetic ablated the kernel still took 0.57 ms, 0.35
    // without these loads.)
#pragma unroll
    for (int e = 0; e < TPT; ++e)
      if (c + 3 < C && e * NT + tid < NTAB && !(a.abl & 16)) tnext[e] = a.coarse[(size_t)(t * C + c + 3) * nvox_c + ctab_src[e]];
    // gather camera c: every lane reads the Q quads of its own voxels' pixels
    const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cam_base(cam_c)), 0,
                                                                           plane_bytes, 0x00020000);
    if (!(a.abl & 2) && wave_in)
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      const int o = off_c[v] >= 0 ? off_c[v] : kZeroOff;
      gf4 h[Q];
      lds_read_quads<Q>(h, lds0 + (unsigned)o);
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        acc[v][q][0] += __builtin_shufflevector(h[q], h[q], 0, 1);
        acc[v][q][1] += __builtin_shufflevector(h[q], h[q], 2, 3);
      }
    }
    // taps that are not in LDS (a box over the LDS budget; never seen otherwise): from global memory.
    // Those lanes added the zero pixel above, so the camera order of the sum is unchanged.
    if (wave_in)
#pragma unroll
    for (int v = 0; v < VPT; ++v) {
      if (__builtin_amdgcn_ballot_w64(off_c[v] < 0) != 0) {                  // uniform per wave
        if (off_c[v] < 0) {
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            const gf4 h = __builtin_bit_cast(gf4, __builtin_amdgcn_raw_buffer_load_b128(
                                                      rs_c, (off_c[v] & 0x7fffffff) + q * 16, 0, 0));
            acc[v][q][0] += __builtin_shufflevector(h, h, 0, 1);
            acc[v][q][1] += __builtin_shufflevector(h, h, 2, 3);
          }
        }
      }
    }
    // (tried, round 5: (1) half of the waves interpolating camera c+1's taps BEFORE they gather camera c, so that the
    //  workgroup's waves do not hit the LDS pipe and then the vector ALU in lock step -- as a wave-uniform branch over two
    //  copies of this body it takes the kernel from 116 to 128 registers with 50 spills; (2) per voxel, the table reads
    //  of camera c+1 and the pixel reads of camera c under ONE wait: 128 registers with 10 spills at 24 channels, 0.81
    //  against 0.68 ms; without spills, at 32 channels, 1.22 against 1.23 ms)
    if (more && !(a.abl & 4) && wave_in) tap_offsets(c + 1, gn, off_c);          // (reads table (c+1) & 1; off_c of camera c is spent)
    // table of camera c+2 over the table of camera c (read for the last time one iteration ago)
#pragma unroll
    for (int e = 0; e < TPT; ++e)
      if (c + 2 < C && e * NT + tid < NTAB) ctab[(c & 1) * NTAB + e * NT + tid] = tv[e];
    // one barrier per camera: patch c+1 has landed (vmcnt) and is visible, table c+2 is visible, and
    // nobody still reads patch c, whose buffer the next iteration's prefetch overwrites
    // patch c+1 (this wave's rows of it) has landed: the DMA is waited for HERE, after the step's arithmetic
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (!(a.abl & 128)) __syncthreads();
    gc = gn;
    cam_c = cam_n;
  }

  // ---- mean over cameras, / 255 (see repro_gather_kernel); a lane stores its voxels' Jp channels ----
  const float fc = (float)C;
  const float rfc = __fdiv_rn(1.f, fc), r255 = __fdiv_rn(1.f, 255.f);
  // (packed: q = x rc, r = fma(-q, c, x), q' = fma(r, rc, q) on channel pairs)
  auto divc2 = [](rp2 x, float cc, float rc) __attribute__((always_inline)) {
    const rp2 c2 = (rp2){cc, cc}, r2 = (rp2){rc, rc};
    const rp2 q = x * r2;
    return __builtin_elementwise_fma(__builtin_elementwise_fma(-q, c2, x), r2, q);
  };
  // The results leave through LDS (the patch buffers are free now): a lane parks the Jp channels of
  // its voxels, then the wave streams its VPT * 4 runs of 16 voxels -- 16 * Jp * 4 contiguous bytes
  // each in the volume -- as fully coalesced 16-byte stores (a lane writing its own voxel's 96 bytes
  // directly is a 16-byte store at a 96-byte stride: 6 half-empty transactions per voxel).
  int lane_late = lane;                               // (opaque: keeps the epilogue's addresses out of the loop's live set)
  asm volatile("" : "+v"(lane_late));
  unsigned char* park = smem + kPatchOff + (size_t)w * (64 * JPB);           // this wave's region
  constexpr int kRunChunks = 16 * Q;                  // 16-byte chunks per run of 16 voxels
#pragma unroll
  for (int v = 0; v < VPT; ++v) {
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      rp2 lo = divc2(acc[v][q][0], fc, rfc), hi = divc2(acc[v][q][1], fc, rfc);
      if (a.div255) {
        lo = divc2(lo, 255.f, r255);
        hi = divc2(hi, 255.f, r255);
      }
      // voxel `lane` of group v = run lane / 16 (a j-row), position lane % 16 inside the run
      *reinterpret_cast<float4*>(park + (lane_late * Q + q) * 16) = make_float4(lo[0], lo[1], hi[0], hi[1]);
    }
    // (the same wave wrote and reads: the LDS operations of a wave complete in order, no barrier)
    const int gi = w * VPT + v, il = gi % CI;
#pragma unroll
    for (int n = 0; n < Q; ++n) {
      const int ch = n * 64 + lane_late;              // chunk of this group's 64 * Q
      const int run = ch / kRunChunks, within = ch - run * kRunChunks;
      const int jl = ((gi / CI) << 2) + run;
      const size_t vox0 = ((size_t)(I0 + il) * G + (J0 + jl)) * G + K0;
      const float4 r = *reinterpret_cast<const float4*>(park + ch * 16);
      // (ragged last cubes: rows j >= G and, inside a run, voxels k >= G are not part of the grid)
      if (!(a.abl & 8) && J0 + jl < G && K0 + within / Q < G)
        *reinterpret_cast<float4*>(a.vol + ((size_t)t * nvox + vox0) * (JPB / 4) + within * 4) = r;
    }
  }
}

template <int Q, int CI, int NT>
static int launch_cube(const CubeArgs& a, int T, hipStream_t s) {
  auto kern = repro_cube_kernel<Q, CI, NT>;
  // the attribute belongs to the CURRENT device: one flag per device (a process may drive several GPUs, and
  // two host threads may build predictors at the same time -- setting it twice is harmless)
  static std::atomic<bool> big[64];
  int devid = 0;
  JH_CHECK_HIP(hipGetDevice(&devid));
  if (devid < 0 || devid >= 64 || !big[devid].load(std::memory_order_acquire)) {
    JH_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (devid >= 0 && devid < 64) big[devid].store(true, std::memory_order_release);
  }
  const int cubes = (a.G / CI) * ((a.G + kCubeJ - 1) / kCubeJ) * ((a.G + kCubeK - 1) / kCubeK);
  hipLaunchKernelGGL(kern, dim3(cubes, T), dim3(NT), (size_t)kCubePatchOff(CI) + 2 * a.patch_bytes, s, a);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
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
  // cube form: G a multiple of 8 (the cubes' thickness; j and k may leave a ragged last cube: 72 = 4.5 x 16), at most
  // 32 channels (JH_REPRO_CUBE=0: the voxel-row form below; JH_REPRO_CUBE=16: only grids that are multiples of 16)
  if (G % 8 == 0 && G >= 16 && Jp <= 32 && JH_ENV_KNOB("JH_REPRO_CUBE") != 0 &&
      (G % 16 == 0 || JH_ENV_KNOB("JH_REPRO_CUBE") != 16)) {
    CubeArgs ca{coarse, heat, vol, idx_out, C, G, hs, heat_pad, div255, 0, 0, lay, 0};
    ca.abl = std::max(0, JH_ENV_KNOB("JH_REPRO_ABL"));
    const int Q = Jp / 4;
    ca.patch_bytes = ((160 * 1024 - kCubePatchOff(8)) / 2) & ~1023;     // two patch buffers (CI <= 8)
    ca.patch_limit = ca.patch_bytes;
    // (test knob: a smaller limit sends boxes to the per-lane global-memory path, which must give the same bits)
    if (getenv("JH_REPRO_PATCH_KB")) ca.patch_limit = std::min(ca.patch_bytes, std::max(1, atoi(getenv("JH_REPRO_PATCH_KB"))) * 1024);
    switch (Q) {
      case 2: return launch_cube<2, 8, 1024>(ca, T, s);
      case 4: return launch_cube<4, 8, 512>(ca, T, s);
      case 6:
        if (JH_ENV_KNOB("JH_REPRO_CI4") > 0) {
          // experiment (round 6): 4-voxel-thick cubes on 512 threads with HALF the patch buffers -- two workgroups per
          // CU that cover each other's per-camera barriers; boxes over the limit take the per-lane global path
          ca.patch_bytes = ((80 * 1024 - kCubePatchOff(4)) / 2) & ~1023;
          ca.patch_limit = std::min(ca.patch_limit, ca.patch_bytes);
          return launch_cube<6, 4, 512>(ca, T, s);
        }
        return JH_ENV_KNOB("JH_REPRO_NT") == 512 ? launch_cube<6, 8, 512>(ca, T, s)
                                                 : launch_cube<6, 8, 1024>(ca, T, s);
      // 32 channels: 4-voxel-thick cubes, one voxel per lane (configs[4]: 1.37 against 1.48 ms per 8 frames
      // for the 8-thick cube on 512 threads, 1.52 for the voxel-row kernel)
      case 8:
        if (JH_ENV_KNOB("JH_REPRO_CI4") > 0) {       // (the same experiment at 32 channels: configs[4])
          ca.patch_bytes = ((80 * 1024 - kCubePatchOff(4)) / 2) & ~1023;
          ca.patch_limit = std::min(ca.patch_limit, ca.patch_bytes);
          return launch_cube<8, 4, 512>(ca, T, s);
        }
        return JH_ENV_KNOB("JH_REPRO_Q8") == 0 ? launch_cube<8, 8, 512>(ca, T, s)
                                               : launch_cube<8, 4, 1024>(ca, T, s);
      default: break;
    }
  }
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
