// Do fp32 MFMAs and vector-ALU instructions of DIFFERENT waves on one SIMD overlap?
//
// A workgroup of 512 threads = 8 waves = 2 per SIMD (one workgroup per CU, one launch over all
// CUs).  Waves 0-3 ("matrix") issue a stream of independent MFMAs, waves 4-7 ("vector") a stream
// of independent vector-ALU instructions of a chosen kind.  Three launches per kind: matrix role
// alone, vector role alone, both.  If the two pipes were independent, "both" would take
// max(alone, alone); if they share issue / the FMA datapath it takes the sum.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_coissue.hip -o /tmp/coissue && /tmp/coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MK: 0 = v_mfma_f32_16x16x4_f32, 1 = v_mfma_f32_16x16x32_bf16
// VK: 0 = v_fma_f32, 1 = v_pk_fma_f32, 2 = v_add_u32 (integer), 3 = ds_read_b128 + 4 v_add, 4 = v_exp_f32,
//     5 = ds_read_b128 only, 6 = global_load_dwordx4 only (L1/L2 hits), 7 = s_mul_i32 (scalar ALU)
// FILL (matrix waves, after every MFMA): 0 nothing, 1 = 3 x s_nop 7, 2 = s_nop 15, 3 = s_sleep 1 every 8th,
//   4 = vector waves at s_setprio 3, 5 = s_nop 0; same-wave instruction costs: 6 = 4 s_mul_i32, 7 = 1 v_fma_f32,
//   8 = 1 ds_read_b64, 9 = 2 v_fma_f32, 10 = 1 global_load_dwordx2, 11 = 1 v_pk_fma_f32, 12 = 1 v_exp_f32
template <int MK, int VK, int NACC = 8, int FILL = 0>
__global__ __launch_bounds__(512) void k(float* out, int iters, int roles, float a0, float b0) {
  __shared__ float4 lds[1024];
  const int wave = threadIdx.x >> 6;
  const bool matrix = wave < 4;
  if (threadIdx.x < 1024) lds[threadIdx.x] = make_float4(a0, b0, a0, b0);
  lds[threadIdx.x + 512] = make_float4(b0, a0, b0, a0);
  __syncthreads();
  float s = 0.f;
  if (matrix) {
    if (!(roles & 1)) return;
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    bf16x8 ah, bh;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 20, 0x00020000);
    int sx[4] = {1, 2, 3, 4};
    float fv[8], fw[8]; f32x2 fp[8];
    const float fm = b0 * 0.999f, fc = a0 * 0.001f;
    for (int i = 0; i < 8; ++i) { fv[i] = a + i; fw[i] = b + i; fp[i] = (f32x2){a + i, b - i}; }
    for (int i = 0; i < 8; ++i) { ah[i] = (__bf16)(a + i); bh[i] = (__bf16)(b - i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int i0 = 0; i0 < 8; ++i0) {
          const int i = i0 % NACC;               // NACC accumulators in rotation: dependency distance NACC
          if (MK == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
          else acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc[i], 0, 0, 0);
          if (FILL == 1) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
          if (FILL == 2) asm volatile("s_nop 15");
          if (FILL == 3 && i0 == 7) asm volatile("s_sleep 1");
          if (FILL == 5) asm volatile("s_nop 0");
          if (FILL == 6) asm volatile("s_mul_i32 %0, %0, 3\n\ts_mul_i32 %1, %1, 3\n\ts_mul_i32 %2, %2, 3\n\ts_mul_i32 %3, %3, 3"
                                      : "+s"(sx[0]), "+s"(sx[1]), "+s"(sx[2]), "+s"(sx[3]));
          if (FILL == 7 || FILL == 9) fv[i] = __builtin_fmaf(fv[i], fm, fc);
          if (FILL == 9) fw[i] = __builtin_fmaf(fw[i], fm, fc);
          if (FILL == 8) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x & 511) * 8))); if (it == iters + 5) fv[0] += t[0]; }
          if (FILL == 10) { f32x2 t; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(out + (threadIdx.x & 511) * 2)); if (it == iters + 5) fv[0] += t[0]; }
          if (FILL == 13 && (i0 & 1)) { f32x4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x & 511) * 16))); if (it == iters + 5) fv[0] += t[0]; }
          if (FILL == 14 && (i0 & 1)) { f32x4 t; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(out + (threadIdx.x & 511) * 4)); if (it == iters + 5) fv[0] += t[0]; }
          if (FILL == 15 && (i0 & 1)) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)((threadIdx.x & 511) * 8))); if (it == iters + 5) fv[0] += t[0]; }
          if (FILL == 16 && (i0 & 1)) { f32x2 t; asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen" : "=v"(t) : "v"((unsigned)((threadIdx.x & 511) * 8)), "s"(rsrc)); if (it == iters + 5) fv[0] += t[0]; }
          if (FILL == 17 && i0 == 7) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)(((threadIdx.x + 64 * j) & 511) * 8))); if (it == iters + 5) fv[j] += t[0]; }
          }
          if (FILL == 18 && i0 == 7 && (r & 1)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)(((threadIdx.x + 64 * j) & 511) * 8))); if (it == iters + 5) fv[j] += t[0]; }
          }
          if (FILL == 19 && i0 == 7) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { f32x2 t; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(out + ((threadIdx.x + 64 * j) & 511) * 2)); if (it == iters + 5) fv[j] += t[0]; }
          }
          if (FILL == 20 && i0 == 7) {
#pragma unroll
            for (int j = 0; j < 2; ++j) { f32x2 t; asm volatile("ds_read_b64 %0, %1" : "=v"(t) : "v"((unsigned)(((threadIdx.x + 64 * j) & 511) * 8))); if (it == iters + 5) fv[j] += t[0]; }
#pragma unroll
            for (int j = 0; j < 2; ++j) { f32x2 t; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(t) : "v"(out + ((threadIdx.x + 64 * j) & 511) * 2)); if (it == iters + 5) fv[j] += t[0]; }
          }
          if (FILL == 11) fp[i] = __builtin_elementwise_fma(fp[i], (f32x2){fm, fm}, (f32x2){fc, fc});
          if (FILL == 12) fv[i] = __builtin_amdgcn_exp2f(fv[i]);
        }
    }
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + fv[i] + fw[i] + fp[i][0] + fp[i][1];
    s += sx[0] + sx[1] + sx[2] + sx[3];
    if (FILL >= 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
  } else {
    if (!(roles & 2)) return;
    if (FILL == 4) __builtin_amdgcn_s_setprio(3);
    float v[16];
    f32x2 p[8];
    unsigned u[16];
    for (int i = 0; i < 16; ++i) { v[i] = a0 + i + threadIdx.x; u[i] = i + threadIdx.x; }
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){a0 + i, b0 - i};
    const float m = b0 * 0.999f, c = a0 * 0.001f;
    const f32x2 pm = (f32x2){m, m}, pc = (f32x2){c, c};
    float4 q[4];
    for (int i = 0; i < 4; ++i) q[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (VK == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], m, c);
        } else if (VK == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
#pragma unroll
          for (int i = 0; i < 8; ++i) p[i] = __builtin_elementwise_fma(p[i], pm, pc);
        } else if (VK == 2) {
#pragma unroll
          for (int i = 0; i < 16; ++i) u[i] = u[i] * 3u + (unsigned)it;
        } else if (VK == 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 t = lds[(threadIdx.x + i * 64 + it) & 1023];
            q[i].x += t.x; q[i].y += t.y; q[i].z += t.z; q[i].w += t.w;
          }
        } else if (VK == 4) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = __builtin_amdgcn_exp2f(v[i]);
        } else if (VK == 5) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            f32x4 t;
            asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"((unsigned)(((threadIdx.x + i * 64) & 1023) * 16)));
            if (it == iters + 5) q[i & 3].x += t[0];          // (never: keeps the result formally used)
          }
          asm volatile("s_waitcnt lgkmcnt(0)");
        } else if (VK == 6) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            f32x4 t;
            const float* gp = out + ((threadIdx.x + i * 512) & 4095) * 4;
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gp));
            if (it == iters + 5) q[i & 3].x += t[0];
          }
          asm volatile("s_waitcnt vmcnt(0)");
        } else {
          int sv = __builtin_amdgcn_readfirstlane(u[0]);
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sv));
          u[0] += sv;
        }
      }
    }
    for (int i = 0; i < 16; ++i) s += v[i] + (float)u[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    for (int i = 0; i < 4; ++i) s += q[i].x + q[i].y + q[i].z + q[i].w;
  }
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MK, int VK, int NACC = 8, int FILL = 0>
static void run(const char* name, float* out) {
  const int iters = 4000;
  float t[4] = {0, 0, 0, 0};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int roles = 1; roles <= 3; ++roles) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<MK, VK, NACC, FILL>), dim3(256), dim3(512), 0, 0, out, iters, roles, 1.f, 1.0001f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&t[roles], e0, e1);
    }
  }
  printf("%-44s matrix alone %.3f ms, vector alone %.3f ms, both %.3f ms  (max %.3f, sum %.3f)\n", name, t[1], t[2],
         t[3], t[1] > t[2] ? t[1] : t[2], t[1] + t[2]);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  run<0, 0>("mfma f32 16x16x4  + v_fma_f32", out);
  run<0, 1>("mfma f32 16x16x4  + v_pk_fma_f32", out);
  run<0, 2>("mfma f32 16x16x4  + v_mul/add_u32", out);
  run<0, 3>("mfma f32 16x16x4  + ds_read_b128 (+4 v_add)", out);
  run<0, 4>("mfma f32 16x16x4  + v_exp_f32", out);
  run<0, 5>("mfma f32 16x16x4  + ds_read_b128 only", out);
  run<0, 6>("mfma f32 16x16x4  + global_load_dwordx4 only", out);
  run<0, 7>("mfma f32 16x16x4  + s_mul_i32", out);
  run<0, 0, 1>("mfma f32, 1 accumulator   + v_fma_f32", out);
  run<0, 0, 2>("mfma f32, 2 accumulators  + v_fma_f32", out);
  run<0, 0, 4>("mfma f32, 4 accumulators  + v_fma_f32", out);
  run<0, 5, 2>("mfma f32, 2 accumulators  + ds_read_b128 only", out);
  run<0, 6, 2>("mfma f32, 2 accumulators  + global_load only", out);
  run<0, 0, 8, 1>("mfma f32 + 3 x s_nop 7  | v_fma_f32", out);
  run<0, 5, 8, 1>("mfma f32 + 3 x s_nop 7  | ds_read only", out);
  run<0, 6, 8, 1>("mfma f32 + 3 x s_nop 7  | global_load only", out);
  run<0, 0, 8, 2>("mfma f32 + s_nop 15     | v_fma_f32", out);
  run<0, 5, 8, 2>("mfma f32 + s_nop 15     | ds_read only", out);
  run<0, 0, 8, 5>("mfma f32 + s_nop 0      | v_fma_f32", out);
  run<0, 5, 8, 5>("mfma f32 + s_nop 0      | ds_read only", out);
  run<0, 0, 8, 3>("mfma f32 + s_sleep 1 /8 | v_fma_f32", out);
  run<0, 5, 8, 3>("mfma f32 + s_sleep 1 /8 | ds_read only", out);
  run<0, 0, 8, 4>("mfma f32 | v_fma_f32 at prio 3", out);
  run<0, 5, 8, 4>("mfma f32 | ds_read only at prio 3", out);
  printf("--- same-wave costs: read the 'matrix alone' column against 1.73 ms (128 k MFMAs x 32 cycles) ---\n");
  run<0, 0, 8, 6>("mfma f32 + 4 s_mul_i32 each (same wave)", out);
  run<0, 0, 8, 7>("mfma f32 + 1 v_fma_f32 each (same wave)", out);
  run<0, 0, 8, 9>("mfma f32 + 2 v_fma_f32 each (same wave)", out);
  run<0, 0, 8, 11>("mfma f32 + 1 v_pk_fma_f32 each (same wave)", out);
  run<0, 0, 8, 12>("mfma f32 + 1 v_exp_f32 each (same wave)", out);
  run<0, 0, 8, 8>("mfma f32 + 1 ds_read_b64 each (same wave)", out);
  run<0, 0, 8, 10>("mfma f32 + 1 global_load_dwordx2 each (same wave)", out);
  run<0, 0, 8, 15>("mfma f32 + 1 ds_read_b64 per 2 (same wave)", out);
  run<0, 0, 8, 13>("mfma f32 + 1 ds_read_b128 per 2 (same wave)", out);
  run<0, 0, 8, 14>("mfma f32 + 1 global_load_dwordx4 per 2 (same wave)", out);
  run<0, 0, 8, 16>("mfma f32 + 1 buffer_load_dwordx2 per 2 (same wave)", out);
  run<0, 0, 8, 17>("mfma f32 + 4 ds_read_b64 in a row per 8 (same wave)", out);
  run<0, 0, 8, 18>("mfma f32 + 8 ds_read_b64 in a row per 16 (same wave)", out);
  run<0, 0, 8, 19>("mfma f32 + 4 global_load in a row per 8 (same wave)", out);
  run<0, 0, 8, 20>("mfma f32 + 2 ds_read + 2 global_load in a row per 8", out);
  run<1, 0>("mfma bf16 16x16x32 + v_fma_f32", out);
  run<1, 1>("mfma bf16 16x16x32 + v_pk_fma_f32", out);
  run<1, 3>("mfma bf16 16x16x32 + ds_read_b128 (+4 v_add)", out);
  return 0;
}
