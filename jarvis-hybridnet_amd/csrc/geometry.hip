// Small kernels around the networks: frame pre-processing, centre argmax,
// confidence-weighted triangulation, projection, and the soft-argmax tail.
//
//  preprocess_resize   torchvision tensor resize (bilinear, align_corners =
//                      False, no antialias) + (x-mean)/std
//                      jarvis/prediction/jarvis3D.py:143-145
//  preprocess_crop     bounding-box crop + normalisation   jarvis3D.py:168-178
//  center_argmax       per-camera argmax of the centre heatmap, jarvis3D.py:147-155
//  triangulate         ReprojectionTool.reconstructPoint + reprojectPoint +
//                      integer clamp, jarvis/utils/reprojection.py:49-90,
//                      jarvis3D.py:157-166
//  softargmax          softplus + spatial soft-argmax + confidences,
//                      jarvis/hybridnet/model.py:73-88
#include "jh_common.h"
#include "preprocess.h"

namespace jh {

// ------------------------------------------------------------------ preprocess
// out: [N][S][S][4] channel-last (one float4 per pixel: r, g, b, 0).
template <int SRC>
__global__ __launch_bounds__(256) void preprocess_resize_kernel(
    const void* __restrict__ frames, float* __restrict__ out, int N, int H, int W, int S,
    float sy, float sx, float3 mean, float3 stdv, const void* const* __restrict__ frames_cell) {
  // graph replays: the frame pointer of THIS call is read from a device cell (a captured launch
  // would otherwise keep the pointer of the call it was captured on)
  if (frames_cell) frames = *frames_cell;
  const size_t total = (size_t)N * S * S;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % S), oy = (int)((i / S) % S), n = (int)(i / ((size_t)S * S));
    *reinterpret_cast<float4*>(out + i * 4) = resize_px<SRC>(frames, n, oy, ox, H, W, sy, sx, mean, stdv);
  }
}

int launch_preprocess_resize(const void* frames, int src_u8, float* out, int N, int H, int W, int S,
                             const float* mean, const float* stdv, hipStream_t s,
                             const void* const* frames_cell) {
  const size_t total = (size_t)N * S * S;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  const float3 m = make_float3(mean[0], mean[1], mean[2]), sd = make_float3(stdv[0], stdv[1], stdv[2]);
  if (src_u8)
    hipLaunchKernelGGL(preprocess_resize_kernel<1>, dim3(blocks), dim3(256), 0, s, frames, out, N, H,
                       W, S, (float)H / (float)S, (float)W / (float)S, m, sd, frames_cell);
  else
    hipLaunchKernelGGL(preprocess_resize_kernel<0>, dim3(blocks), dim3(256), 0, s, frames, out, N, H,
                       W, S, (float)H / (float)S, (float)W / (float)S, m, sd, frames_cell);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// frames: [T][Cloc] images; center_hm: [T][C][2] (all cameras); out [T*Cloc][B][B][4]
template <int SRC>
__global__ __launch_bounds__(256) void preprocess_crop_kernel(
    const void* __restrict__ frames, const int* __restrict__ center_hm, float* __restrict__ out,
    int T, int Cloc, int C, int cam0, int H, int W, int B, float3 mean, float3 stdv,
    const void* const* __restrict__ frames_cell) {
  if (frames_cell) frames = *frames_cell;
  const size_t total = (size_t)T * Cloc * B * B;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ox = (int)(i % B), oy = (int)((i / B) % B);
    const int n = (int)(i / ((size_t)B * B));
    const int t = n / Cloc, cl = n % Cloc;
    const int cx = center_hm[(t * C + cam0 + cl) * 2 + 0], cy = center_hm[(t * C + cam0 + cl) * 2 + 1];
    *reinterpret_cast<float4*>(out + i * 4) = crop_px<SRC>(frames, n, cx, cy, oy, ox, H, W, B, mean, stdv);
  }
}

int launch_preprocess_crop(const void* frames, int src_u8, const int* center_hm, float* out, int T,
                           int Cloc, int C, int cam0, int H, int W, int B, const float* mean,
                           const float* stdv, hipStream_t s, const void* const* frames_cell) {
  const size_t total = (size_t)T * Cloc * B * B;
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  const float3 m = make_float3(mean[0], mean[1], mean[2]), sd = make_float3(stdv[0], stdv[1], stdv[2]);
  if (src_u8)
    hipLaunchKernelGGL(preprocess_crop_kernel<1>, dim3(blocks), dim3(256), 0, s, frames, center_hm,
                       out, T, Cloc, C, cam0, H, W, B, m, sd, frames_cell);
  else
    hipLaunchKernelGGL(preprocess_crop_kernel<0>, dim3(blocks), dim3(256), 0, s, frames, center_hm,
                       out, T, Cloc, C, cam0, H, W, B, m, sd, frames_cell);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// --------------------------------------------------------------- centre argmax
// heat: [N][Hh][Wh][Cp], channel 0.  det[n] = (x, y, maxval); first maximum wins
// (torch.argmax on CPU returns the lowest index among equal maxima).
__global__ __launch_bounds__(1024) void center_argmax_kernel(const float* __restrict__ heat,
                                                            float* __restrict__ det, int Hh,
                                                            int Wh, int Cp) {
  __shared__ float sv[1024];
  __shared__ int si[1024];
  const int n = blockIdx.x;
  const int P = Hh * Wh;
  const float* h = heat + (size_t)n * P * Cp;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  // (eight independent loads in flight per thread: the scan is a chain of load latencies otherwise -- 26 us for
  //  a 256 x 256 map; the compares keep the ascending-index order, so the first maximum still wins)
  constexpr int U = 8;
  int p = threadIdx.x;
  for (; p + (U - 1) * (int)blockDim.x < P; p += U * blockDim.x) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = h[(size_t)(p + u * blockDim.x) * Cp];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * blockDim.x;
      if (v[u] > best || (v[u] == best && pp < bi)) { best = v[u]; bi = pp; }
    }
  }
  for (; p < P; p += blockDim.x) {
    const float v = h[(size_t)p * Cp];
    if (v > best || (v == best && p < bi)) { best = v; bi = p; }
  }
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 512; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = sv[threadIdx.x + s];
      const int ii = si[threadIdx.x + s];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ii < si[threadIdx.x])) {
        sv[threadIdx.x] = v; si[threadIdx.x] = ii;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int m = si[0];
    // preds = (m % shape[2], m // shape[3])   jarvis3D.py:151
    det[n * 3 + 0] = (float)(m % Hh);
    det[n * 3 + 1] = (float)(m / Wh);
    det[n * 3 + 2] = sv[0];
  }
}

int launch_center_argmax(const float* heat, float* det, int N, int Hh, int Wh, int Cp,
                         hipStream_t s) {
  hipLaunchKernelGGL(center_argmax_kernel, dim3(N), dim3(1024), 0, s, heat, det, Hh, Wh, Cp);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------- 2D predictor glue kernels
// JarvisPredictor2D (jarvis/prediction/jarvis2D.py:121-129): crop centre of each image
// from its centre detection: trunc((x, y) * scale * 2), clamped to
// [hw, size - hw - 1]; valid = maxval > 40.
__global__ void center2d_kernel(const float* __restrict__ det, int* __restrict__ center_hm,
                                int* __restrict__ valid, int T, float sx, float sy, int hw, int W,
                                int H) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float* d = det + (size_t)t * 3;
  int cx = (int)__fmul_rn(__fmul_rn(d[0], sx), 2.f);
  int cy = (int)__fmul_rn(__fmul_rn(d[1], sy), 2.f);
  cx = min(max(cx, hw), W - hw - 1);
  cy = min(max(cy, hw), H - hw - 1);
  center_hm[t * 2 + 0] = cx;
  center_hm[t * 2 + 1] = cy;
  valid[t] = d[2] > 40.f ? 1 : 0;
}

int launch_center2d(const float* det, int* center_hm, int* valid, int T, float sx, float sy, int hw,
                    int W, int H, hipStream_t s) {
  hipLaunchKernelGGL(center2d_kernel, dim3((T + 63) / 64), dim3(64), 0, s, det, center_hm, valid, T,
                     sx, sy, hw, W, H);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// Per-joint argmax of the keypoint heatmaps [T][Hh][Wh][Jp] (jarvis2D.py:139-149):
// points2D = (m % Hh, m // Wh) * 2 + centerHM - hw, confidence = min(max, 255) / 255.
// One block per (joint, image); the lowest index wins among equal maxima.
__global__ __launch_bounds__(256) void joint_argmax_kernel(
    const float* __restrict__ heat, const int* __restrict__ center_hm, int* __restrict__ points,
    float* __restrict__ conf, int J, int Jp, int Hh, int Wh, int hw) {
  __shared__ float sv[256];
  __shared__ int si[256];
  const int j = blockIdx.x, t = blockIdx.y;
  const int P = Hh * Wh;
  const float* h = heat + (size_t)t * P * Jp + j;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int p = threadIdx.x; p < P; p += blockDim.x) {
    const float v = h[(size_t)p * Jp];
    if (v > best || (v == best && p < bi)) { best = v; bi = p; }
  }
  sv[threadIdx.x] = best; si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      const float v = sv[threadIdx.x + s];
      const int ii = si[threadIdx.x + s];
      if (v > sv[threadIdx.x] || (v == sv[threadIdx.x] && ii < si[threadIdx.x])) {
        sv[threadIdx.x] = v; si[threadIdx.x] = ii;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const int m = si[0];
    points[((size_t)t * J + j) * 2 + 0] = (m % Hh) * 2 + center_hm[t * 2 + 0] - hw;
    points[((size_t)t * J + j) * 2 + 1] = (m / Wh) * 2 + center_hm[t * 2 + 1] - hw;
    conf[(size_t)t * J + j] = __fdiv_rn(fminf(sv[0], 255.f), 255.f);
  }
}

int launch_joint_argmax(const float* heat, const int* center_hm, int* points, float* conf, int T,
                        int J, int Jp, int Hh, int Wh, int hw, hipStream_t s) {
  hipLaunchKernelGGL(joint_argmax_kernel, dim3(J, T), dim3(256), 0, s, heat, center_hm, points, conf,
                     J, Jp, Hh, Wh, hw);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------- triangulation
// Smallest-eigenvalue eigenvector of a symmetric 4x4 matrix by cyclic Jacobi in
// fp64.  A^T A of the weighted DLT system shares its right singular vectors with
// A (reprojection.py:85-89 takes V[:, -1] of the SVD); fp64 keeps the squared
// condition number harmless.
__device__ void smallest_eigvec4(double a[4][4], double out[4]) {
  double v[4][4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) off += a[p][q] * a[p][q];
    double diag = 0.0;
    for (int p = 0; p < 4; ++p) diag += a[p][p] * a[p][p];
    if (off <= 1e-30 * diag) break;
    for (int p = 0; p < 4; ++p)
      for (int q = p + 1; q < 4; ++q) {
        if (a[p][q] == 0.0) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), sn = tt * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - sn * akq;
          a[k][q] = sn * akp + c * akq;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - sn * aqk;
          a[q][k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - sn * vkq;
          v[k][q] = sn * vkp + c * vkq;
        }
      }
  }
  int m = 0;
  for (int i = 1; i < 4; ++i)
    if (a[i][i] < a[m][m]) m = i;
  for (int k = 0; k < 4; ++k) out[k] = v[k][m];
}

__device__ __forceinline__ void project_one(const float* M, const float* K, const float* D,
                                            float x, float y, float z, float* u, float* v) {
  float p[3];
#pragma unroll
  for (int col = 0; col < 3; ++col) {
    float a = __fmul_rn(x, M[0 * 3 + col]);
    a = __fmaf_rn(y, M[1 * 3 + col], a);
    a = __fmaf_rn(z, M[2 * 3 + col], a);
    a = __fmaf_rn(1.f, M[3 * 3 + col], a);
    p[col] = a;
  }
  const float cx = K[6], cy = K[7], fx = K[0], fy = K[4];
  float uu = __fsub_rn(__fdiv_rn(p[0], p[2]), cx);
  float vv = __fsub_rn(__fdiv_rn(p[1], p[2]), cy);
  const float a1 = __fdiv_rn(uu, fx), a2 = __fdiv_rn(vv, fy);
  const float r2 = __fadd_rn(__fmul_rn(a1, a1), __fmul_rn(a2, a2));
  const float dd = __fadd_rn(1.f, __fmul_rn(__fadd_rn(D[0], __fmul_rn(D[1], r2)), r2));
  *u = __fadd_rn(__fmul_rn(uu, dd), cx);
  *v = __fadd_rn(__fmul_rn(vv, dd), cy);
}

// One 64-thread block per frame t.  det: [T][C][3] (x, y, raw maxval).
// Outputs: center3d_f [T][3] float, center3d_i [T][3] int (truncated),
// center_hm [T][C][2] int (truncated + clamped crop centres), valid [T].
__global__ __launch_bounds__(64) void triangulate_kernel(
    const float* __restrict__ det, const float* __restrict__ cam, const float* __restrict__ intr,
    const float* __restrict__ dist, float* __restrict__ center3d_f, int* __restrict__ center3d_i,
    int* __restrict__ center_hm, int* __restrict__ valid, int C, float sx2, float sy2, float wdiv,
    int hw, int W, int H) {
  __shared__ double ata[16];
  __shared__ double contrib[64][17];            // per-camera terms of A^T A (summed in camera order)
  __shared__ float ctr[3];
  __shared__ int cnt;
  const int t = blockIdx.x, c = threadIdx.x;
  if (c == 0) cnt = 0;
  __syncthreads();
  if (c < C) {
    const float* d = det + ((size_t)t * C + c) * 3;
    const float* K = intr + c * 9;
    const float* M = cam + c * 12;
    const float cx = K[6], cy = K[7], fx = K[0], fy = K[4];
    const float k1 = dist[c * 5 + 0], k2 = dist[c * 5 + 1];
    if (d[2] > 50.f) atomicAdd(&cnt, 1);
    const float wgt = __fdiv_rn(d[2], wdiv);
    // undistort the detection (single-step inverse), reprojection.py:71-78
    float u = __fsub_rn(__fmul_rn(d[0], sx2), cx);
    float v = __fsub_rn(__fmul_rn(d[1], sy2), cy);
    const float a1 = __fdiv_rn(u, fx), a2 = __fdiv_rn(v, fy);
    const float r2 = __fadd_rn(__fmul_rn(a1, a1), __fmul_rn(a2, a2));
    const float dd = __fadd_rn(1.f, __fmul_rn(__fadd_rn(k1, __fmul_rn(k2, r2)), r2));
    u = __fadd_rn(__fdiv_rn(u, dd), cx);
    v = __fadd_rn(__fdiv_rn(v, dd), cy);
    // rows u*P2 - P0 and v*P2 - P1 with P = cameraMatrix^T, weighted by maxval
    float r0[4], r1[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      r0[k] = __fmul_rn(__fsub_rn(__fmul_rn(u, M[k * 3 + 2]), M[k * 3 + 0]), wgt);
      r1[k] = __fmul_rn(__fsub_rn(__fmul_rn(v, M[k * 3 + 2]), M[k * 3 + 1]), wgt);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        contrib[c][i * 4 + j] = (double)r0[i] * (double)r0[j] + (double)r1[i] * (double)r1[j];
  }
  __syncthreads();
  if (c < 16) {                                 // fixed order: the result does not depend on timing
    double s = 0.0;
    for (int k = 0; k < C; ++k) s += contrib[k][c];
    ata[c] = s;
  }
  __syncthreads();
  if (c == 0) {
    double a[4][4], x[4];
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) a[i][j] = 0.5 * (ata[i * 4 + j] + ata[j * 4 + i]);
    smallest_eigvec4(a, x);
    for (int k = 0; k < 3; ++k) {
      const float f = (float)(x[k] / x[3]);
      ctr[k] = f;
      center3d_f[t * 3 + k] = f;
      center3d_i[t * 3 + k] = (int)f;          // .int() truncates, jarvis3D.py:183
    }
    valid[t] = cnt >= 2 ? 1 : 0;
  }
  __syncthreads();
  if (c < C) {
    float u, v;
    project_one(cam + c * 12, intr + c * 9, dist + c * 5, ctr[0], ctr[1], ctr[2], &u, &v);
    int iu = (int)u, iv = (int)v;
    iu = min(max(iu, hw), W - hw);             // jarvis3D.py:163-166
    iv = min(max(iv, hw), H - hw);
    center_hm[((size_t)t * C + c) * 2 + 0] = iu;
    center_hm[((size_t)t * C + c) * 2 + 1] = iv;
  }
}

int launch_triangulate(const float* det, const float* cam, const float* intr, const float* dist,
                       float* center3d_f, int* center3d_i, int* center_hm, int* valid, int T, int C,
                       float sx2, float sy2, float wdiv, int hw, int W, int H, hipStream_t s) {
  JH_REQUIRE(C <= 64, "at most 64 cameras");
  hipLaunchKernelGGL(triangulate_kernel, dim3(T), dim3(64), 0, s, det, cam, intr, dist, center3d_f,
                     center3d_i, center_hm, valid, C, sx2, sy2, wdiv, hw, W, H);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// points [P][3] -> uv [C][P][2]  (ReprojectionTool.reprojectPoint for tests/API)
__global__ void project_points_kernel(const float* __restrict__ pts, const float* cam,
                                      const float* intr, const float* dist, float* uv, int P,
                                      int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P * C) return;
  const int c = i / P, p = i % P;
  float u, v;
  project_one(cam + c * 12, intr + c * 9, dist + c * 5, pts[p * 3], pts[p * 3 + 1], pts[p * 3 + 2], &u, &v);
  uv[(size_t)i * 2] = u;
  uv[(size_t)i * 2 + 1] = v;
}

int launch_project_points(const float* pts, const float* cam, const float* intr, const float* dist,
                          float* uv, int P, int C, hipStream_t s) {
  hipLaunchKernelGGL(project_points_kernel, dim3((P * C + 63) / 64), dim3(64), 0, s, pts, cam, intr,
                     dist, uv, P, C);
  JH_CHECK_HIP(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------ soft-argmax
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

// x: [T][Gh^3][Jp].  partial: [T][Jp][4][kLimbs] doubles (sum h, sum h*i, sum h*j, sum h*k; order-
// independent accumulation, jh_common.h),
// pmax: [T][Jp] floats as ordered ints (h > 0 so the int order equals the float order).
__global__ __launch_bounds__(256) void softargmax_partial_kernel(
    const float* __restrict__ x, double* __restrict__ partial, int* __restrict__ pmax, int Gh,
    int Jp, int ppb) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int q = Jp >> 2, rows = 256 / q, tid = threadIdx.x;
  const bool active = tid < rows * q;
  const int c4 = tid % q, row = tid / q;
  const int t = blockIdx.y;
  const int P = Gh * Gh * Gh;
  const int p0 = blockIdx.x * ppb, p1 = min(P, p0 + ppb);
  float4 s0 = make_float4(0, 0, 0, 0), si = s0, sj = s0, sk = s0, mx = s0;
  if (active) {
    for (int p = p0 + row; p < p1; p += rows) {
      const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)t * P + p) * Jp + c4 * 4);
      const float4 h = make_float4(softplus_f(v.x), softplus_f(v.y), softplus_f(v.z), softplus_f(v.w));
      const float fk = (float)(p % Gh), fj = (float)((p / Gh) % Gh), fi = (float)(p / (Gh * Gh));
      s0.x += h.x; s0.y += h.y; s0.z += h.z; s0.w += h.w;
      si.x += h.x * fi; si.y += h.y * fi; si.z += h.z * fi; si.w += h.w * fi;
      sj.x += h.x * fj; sj.y += h.y * fj; sj.z += h.z * fj; sj.w += h.w * fj;
      sk.x += h.x * fk; sk.y += h.y * fk; sk.z += h.z * fk; sk.w += h.w * fk;
      mx.x = fmaxf(mx.x, h.x); mx.y = fmaxf(mx.y, h.y); mx.z = fmaxf(mx.z, h.z); mx.w = fmaxf(mx.w, h.w);
    }
  }
  // block reduce: sm [rows][q][5][4]
  if (active) {
    float4* p = reinterpret_cast<float4*>(sm) + ((size_t)row * q + c4) * 5;
    p[0] = s0; p[1] = si; p[2] = sj; p[3] = sk; p[4] = mx;
  }
  __syncthreads();
  for (int i = tid; i < q * 4 * 5; i += 256) {
    const int comp = i & 3, v = (i >> 2) % 5, cq = (i >> 2) / 5;
    const int ch = cq * 4 + comp;
    if (v < 4) {
      float acc = 0.f;
      for (int r = 0; r < rows; ++r) acc += sm[(((size_t)r * q + cq) * 5 + v) * 4 + comp];
      exact_add(partial + (((size_t)t * Jp + ch) * 4 + v) * kLimbs, (double)acc);
    } else {
      float m = 0.f;
      for (int r = 0; r < rows; ++r) m = fmaxf(m, sm[(((size_t)r * q + cq) * 5 + v) * 4 + comp]);
      atomicMax(pmax + (size_t)t * Jp + ch, __float_as_int(m));
    }
  }
}

__global__ void softargmax_final_kernel(const double* __restrict__ partial,
                                        const int* __restrict__ pmax,
                                        const int* __restrict__ center3d, float* __restrict__ points,
                                        float* __restrict__ conf, int J, int Jp, float spacing,
                                        float roi) {
  const int t = blockIdx.x, j = threadIdx.x;
  if (j >= J) return;
  const double* pl = partial + ((size_t)t * Jp + j) * 4 * kLimbs;
  const double p[4] = {exact_read(pl), exact_read(pl + kLimbs), exact_read(pl + 2 * kLimbs),
                       exact_read(pl + 3 * kLimbs)};
  const float norm = (float)p[0];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float idx = __fdiv_rn((float)p[1 + a], norm);
    // idx * GRID_SPACING * 2 - ROI_CUBE_SIZE / 2 + center3D   (model.py:86-87)
    const float mm = __fadd_rn(__fsub_rn(__fmul_rn(__fmul_rn(idx, spacing), 2.f), __fdiv_rn(roi, 2.f)),
                               (float)center3d[t * 3 + a]);
    points[((size_t)t * J + j) * 3 + a] = mm;
  }
  const float m = __int_as_float(pmax[(size_t)t * Jp + j]);
  conf[(size_t)t * J + j] = __fdiv_rn(fminf(m, 255.f), 255.f);
}

// heatmap_final = softplus(softplus(x)) in the reference's NCDHW layout
__global__ __launch_bounds__(256) void heatmap_final_kernel(const float* __restrict__ x,
                                                            float* __restrict__ out, int J, int Jp,
                                                            size_t P, int T) {
  const size_t total = (size_t)T * J * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % P;
    const int j = (int)((i / P) % J);
    const size_t t = i / (P * J);
    out[i] = softplus_f(softplus_f(x[(t * P + p) * Jp + j]));
  }
}

int launch_softargmax(const float* x, const int* center3d, double* partial, int* pmax,
                      float* points, float* conf, float* heatmap_final, int T, int J, int Jp,
                      int Gh, float spacing, float roi, hipStream_t s) {
  const int P = Gh * Gh * Gh;
  const int q = Jp / 4;
  JH_REQUIRE(q >= 1 && q <= 64 && J <= 256, "soft-argmax joint count");
  const int rows = 256 / q;
  const size_t pbytes = (size_t)T * Jp * 4 * kLimbs * sizeof(double), mbytes = (size_t)T * Jp * sizeof(int);
  if (reinterpret_cast<char*>(pmax) == reinterpret_cast<char*>(partial) + pbytes) {
    if (launch_zero(partial, pbytes + mbytes, s)) return 1;      // (one launch: the predictor allocates them as one)
  } else {
    if (launch_zero(partial, pbytes, s)) return 1;
    if (launch_zero(pmax, mbytes, s)) return 1;
  }
  const int ppb = rows * 8;
  dim3 grid((P + ppb - 1) / ppb, T);
  hipLaunchKernelGGL(softargmax_partial_kernel, grid, dim3(256), (size_t)rows * q * 20 * sizeof(float),
                     s, x, partial, pmax, Gh, Jp, ppb);
  JH_CHECK_HIP(hipGetLastError());
  hipLaunchKernelGGL(softargmax_final_kernel, dim3(T), dim3(256), 0, s, partial, pmax, center3d,
                     points, conf, J, Jp, spacing, roi);
  JH_CHECK_HIP(hipGetLastError());
  if (heatmap_final) {
    const size_t total = (size_t)T * J * P;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(heatmap_final_kernel, dim3(blocks), dim3(256), 0, s, x, heatmap_final, J, Jp,
                       (size_t)P, T);
    JH_CHECK_HIP(hipGetLastError());
  }
  return 0;
}

}  // namespace jh
