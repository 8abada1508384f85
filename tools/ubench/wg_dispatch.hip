// Workgroup dispatch rate and relaunch gap on gfx950, as a function of the workgroup's shape.
// hipcc --offload-arch=gfx950 -O3 wg_dispatch.hip -o wg_dispatch && ./wg_dispatch
//
// s_memtime on gfx950 counts at a fixed 100 MHz only nominally: the tick rate is calibrated at run time.
// (a) empty workgroups: grid of G workgroups of NT threads with L bytes of LDS and R VGPRs that return at once
//     -> time per launch / G = what the dispatcher needs per workgroup when nothing else limits it;
// (b) workgroups that stay for `spin` microseconds (s_memtime loop, no memory traffic): G = slots x generations
//     -> time per launch against generations x spin = ramp + relaunch gaps + tail of the dispatcher alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NT, int WPE, bool BIG = false>
__global__ __launch_bounds__(NT, WPE) void k_wg(int* out, long long spin_ticks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  if (BIG) asm volatile("v_mov_b32 v124, 0" ::: "v124");  // a 128-VGPR wave, like the f64 tile kernel's
  if (spin_ticks > 0) {
    const long long t0 = __builtin_amdgcn_s_memtime();
    while ((long long)__builtin_amdgcn_s_memtime() - t0 < spin_ticks) __builtin_amdgcn_s_sleep(8);
  }
  if (out != nullptr && threadIdx.x == 0) {
    lds[0] = 1;
    out[blockIdx.x] = lds[0];
  }
}

template <int NT, int WPE, bool BIG = false>
static double run(int grid, size_t lds, long long spin_ticks, int* d_out, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wg<NT, WPE, BIG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_wg<NT, WPE, BIG>), dim3(grid), dim3(NT), lds, 0, d_out, spin_ticks);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_wg<NT, WPE, BIG>), dim3(grid), dim3(NT), lds, 0, d_out, spin_ticks);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;  // us per launch
}

int main() {
  int* d_out;
  hipMalloc(&d_out, sizeof(int) * 65536);
  // clock ramp
  run<512, 4>(8192, 0, 0, d_out, 2000);
  printf("(a) empty workgroups: us per launch (minus the 2304-workgroup figure's intercept) -> workgroups per us\n");
  printf("%-34s %10s %10s %12s\n", "shape", "G=2304", "G=9216", "WG/us slope");
  struct Row { const char* name; double a, b; };
  auto pr = [&](const char* name, double a, double b) {
    printf("%-34s %10.2f %10.2f %12.1f\n", name, a, b, (9216 - 2304) / (b - a));
  };
  pr("256 thr,  0 KB LDS", run<256, 4>(2304, 0, 0, d_out, 300), run<256, 4>(9216, 0, 0, d_out, 300));
  pr("256 thr, 36 KB LDS", run<256, 4>(2304, 36 * 1024, 0, d_out, 300), run<256, 4>(9216, 36 * 1024, 0, d_out, 300));
  pr("512 thr,  0 KB LDS", run<512, 4>(2304, 0, 0, d_out, 300), run<512, 4>(9216, 0, 0, d_out, 300));
  pr("512 thr, 36 KB LDS", run<512, 4>(2304, 36 * 1024, 0, d_out, 300), run<512, 4>(9216, 36 * 1024, 0, d_out, 300));
  pr("512 thr, 72 KB LDS (the tile)", run<512, 4>(2304, 72 * 1024, 0, d_out, 300), run<512, 4>(9216, 72 * 1024, 0, d_out, 300));
  pr("512 thr, 72 KB LDS, 128 VGPRs", run<512, 4, true>(2304, 72 * 1024, 0, d_out, 300), run<512, 4, true>(9216, 72 * 1024, 0, d_out, 300));
  pr("1024 thr, 72 KB LDS", run<1024, 4>(2304, 72 * 1024, 0, d_out, 300), run<1024, 4>(9216, 72 * 1024, 0, d_out, 300));
  pr("1024 thr, 144 KB LDS", run<1024, 4>(2304, 144 * 1024, 0, d_out, 300), run<1024, 4>(9216, 144 * 1024, 0, d_out, 300));
  pr("64 thr,  0 KB LDS", run<64, 4>(2304, 0, 0, d_out, 300), run<64, 4>(9216, 0, 0, d_out, 300));

  // s_memtime ticks per microsecond: one generation of long-lived workgroups, two lengths
  const double c1 = run<512, 4, true>(512, 72 * 1024, 20000, d_out, 50), c2 = run<512, 4, true>(512, 72 * 1024, 60000, d_out, 50);
  const double tpu = 40000.0 / (c2 - c1);
  printf("\ns_memtime: %.1f ticks per us (a 20 000-tick generation takes %.2f us, a 60 000-tick one %.2f us)\n", tpu, c1, c2);
  printf("\n(b) workgroups that stay `spin` us: 512 thr, 72 KB LDS, 128 VGPRs = 2 per CU = 512 slots\n");
  printf("%-28s %8s %10s %12s %10s\n", "grid", "spin us", "ideal us", "measured us", "overhead");
  for (int gens : {1, 2, 4, 8}) {
    for (double spin : {4.0, 8.0}) {
      const int grid = 512 * gens;
      const double t = run<512, 4, true>(grid, 72 * 1024, (long long)(spin * tpu), d_out, 100);
      printf("%4d = 512 x %-15d %8.1f %10.1f %12.2f %10.2f\n", grid, gens, spin, gens * spin, t, t - gens * spin);
    }
  }
  printf("\n(c) the same with 256-thread workgroups, 36 KB LDS = 4 per CU = 1024 slots (same waves per CU)\n");
  for (int gens : {1, 4, 8}) {
    const double spin = 4.0;
    const int grid = 1024 * gens;
    const double t = run<256, 4, true>(grid, 36 * 1024, (long long)(spin * tpu), d_out, 100);
    printf("%4d = 1024 x %-14d %8.1f %10.1f %12.2f %10.2f\n", grid, gens, spin, gens * spin, t, t - gens * spin);
  }
  printf("\n(d) 1024-thread workgroups, 144 KB LDS = 1 per CU = 256 slots\n");
  for (int gens : {1, 4}) {
    const double spin = 16.0;
    const int grid = 256 * gens;
    const double t = run<1024, 4, true>(grid, 144 * 1024, (long long)(spin * tpu), d_out, 100);
    printf("%4d = 256 x %-15d %8.1f %10.1f %12.2f %10.2f\n", grid, gens, spin, gens * spin, t, t - gens * spin);
  }
  return 0;
}
