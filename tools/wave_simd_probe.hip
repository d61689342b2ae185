// Which SIMD does wave w of a workgroup run on?  (HW_REG_HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SE [15:13].)
// The helper waves of csrc/bifpn_rows_wg.hip are placed by wave index; this prints the placement for workgroups of
// 6..16 waves, alone on the chip and with every CU busy.
//   hipcc --offload-arch=gfx950 -O2 tools/wave_simd_probe.hip -o /tmp/wave_simd_probe && /tmp/wave_simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
  const unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
  for (volatile int i = 0; i < spin; ++i) {}
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id;
}

int main() {
  unsigned* d;
  const int blocks = 1024;
  hipMalloc(&d, blocks * 16 * sizeof(unsigned));
  std::vector<unsigned> h(blocks * 16);
  for (int waves : {6, 8, 10, 12, 16}) {
    for (int nb : {1, blocks}) {
      hipMemset(d, 0xff, blocks * 16 * sizeof(unsigned));
      hipLaunchKernelGGL(probe, dim3(nb), dim3(waves * 64), 0, 0, d, nb == 1 ? 0 : 2000);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), d, blocks * 16 * sizeof(unsigned), hipMemcpyDeviceToHost);
      // histogram of the per-workgroup SIMD pattern
      int same = 0;
      std::vector<int> first(waves);
      for (int w = 0; w < waves; ++w) first[w] = (h[w] >> 4) & 3;
      for (int b = 0; b < nb; ++b) {
        bool eq = true;
        for (int w = 0; w < waves; ++w) eq = eq && (int)((h[b * 16 + w] >> 4) & 3) == (first[0] + w) % 4;
        same += eq;
      }
      printf("%2d waves, %4d workgroups: SIMD of wave 0.. =", waves, nb);
      for (int w = 0; w < waves; ++w) printf(" %d", first[w]);
      printf("   (wave w on SIMD (s0 + w) %% 4 in %d of %d workgroups)\n", same, nb);
      if (nb > 1) {
        printf("     a later workgroup:");
        for (int w = 0; w < waves; ++w) printf(" %d", (h[777 * 16 + w] >> 4) & 3);
        printf("\n");
      }
    }
  }
  return 0;
}
