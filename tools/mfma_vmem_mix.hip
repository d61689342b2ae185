// Cost model probe: how much does a global load (L1/L2-resident, consumed one iteration
// later) cost a wave that is otherwise issuing back-to-back fp32 MFMAs?  Mirrors the tap loop
// of conv_mfma.h: 72 MFMAs per iteration, NL loads of W bytes per lane.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_vmem_mix.hip -o scratch/mfma_vmem_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NL, int W4, int ND>   // NL global loads (W4 ? 16 : 8 bytes per lane) and ND ds_read_b64 per iteration
__global__ __launch_bounds__(256) void k(const float* __restrict__ w, float* out, int iters) {
  __shared__ float2 sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) sm[i] = make_float2(1.f, 2.f);
  __syncthreads();
  float2 acur[ND > 0 ? ND : 1], anxt[ND > 0 ? ND : 1];
  for (int j = 0; j < (ND > 0 ? ND : 1); ++j) acur[j] = make_float2(1.f, 1.f);
  f32x4 acc[12];
  for (int i = 0; i < 12; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  float4 cur[NL > 0 ? NL : 1], nxt[NL > 0 ? NL : 1];
  for (int j = 0; j < (NL > 0 ? NL : 1); ++j) cur[j] = make_float4(1.f, 2.f, 3.f, 4.f);
  const float* base = w + lane * (W4 ? 4 : 2);
  for (int it = 0; it < iters; ++it) {
    const float* p = base + (size_t)(it & 63) * NL * 256;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      if (W4) nxt[j] = *reinterpret_cast<const float4*>(p + j * 256);
      else { const float2 t = *reinterpret_cast<const float2*>(p + j * 128); nxt[j] = make_float4(t.x, t.y, t.x, t.y); }
    }
#pragma unroll
    for (int j = 0; j < ND; ++j) anxt[j] = sm[((it * 7 + j * 64) & 4032) + lane];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 c = cur[NL > 0 ? (r * 12 + i) % NL : 0];
        const float2 av = acur[ND > 0 ? (r * 12 + i) % ND : 0];
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(c.x + av.x, c.y, acc[i], 0, 0, 0);
      }
#pragma unroll
    for (int j = 0; j < NL; ++j) cur[j] = nxt[j];
#pragma unroll
    for (int j = 0; j < ND; ++j) acur[j] = anxt[j];
  }
  float s = 0.f;
  for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NL, int W4, int ND>
void run(const float* w, float* out, int wg_per_cu) {
  const int iters = 4000, blocks = 256 * wg_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<NL, W4, ND>), dim3(blocks), dim3(256), 0, 0, w, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double flops = (double)blocks * 4 * iters * 72 * 2048.0;
  printf("wg/cu=%d loads/iter=%2d x%d ds_reads=%2d  %.1f TFLOP/s\n", wg_per_cu, NL, W4 ? 4 : 2, ND, flops / best / 1e9);
}

int main() {
  float *w, *out;
  (void)hipMalloc(&w, 64 * 18 * 256 * sizeof(float) + 4096);
  (void)hipMemset(w, 0, 64 * 18 * 256 * sizeof(float) + 4096);
  (void)hipMalloc(&out, 256 * 256 * 8 * sizeof(float));
  for (int o = 1; o <= 2; ++o) {
    run<0, 0, 0>(w, out, o); run<3, 0, 0>(w, out, o); run<9, 0, 0>(w, out, o); run<18, 0, 0>(w, out, o);
    run<5, 1, 0>(w, out, o); run<0, 0, 12>(w, out, o); run<9, 0, 12>(w, out, o); run<5, 1, 12>(w, out, o);
  }
  return 0;
}
