// wave_shr:1 / wave_shl:1 DPP on gfx950: which lane does lane i read?   hipcc --offload-arch=gfx950 -O3 -o /tmp/dppw tools/ubench/dpp_wave_shift.hip && /tmp/dppw
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  const int v = 100 + lane;
  out[lane] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
}
int main() {
  int* d; hipMalloc(&d, 128 * sizeof(int));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("wave_shr:1  lane0 %d lane1 %d lane15 %d lane16 %d lane17 %d lane32 %d lane63 %d\n", h[0], h[1], h[15], h[16], h[17], h[32], h[63]);
  printf("wave_shl:1  lane0 %d lane1 %d lane15 %d lane16 %d lane31 %d lane62 %d lane63 %d\n", h[64], h[65], h[79], h[80], h[95], h[126], h[127]);
  return 0;
}
