// Issue-rate microbenchmark for gfx950: cycles per wave64 VALU instruction per SIMD, by instruction kind.
// hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP 64
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float a, double da, int ia) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0 + 4, i5 = i0 + 5, i6 = i0 + 6, i7 = i0 + 7;
  __shared__ float lds[4096];
  lds[threadIdx.x] = x0;
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < REP / 8; ++r) {
      if (KIND == 0) {  // v_fma_f32
        x0 = __builtin_fmaf(x0, a, x1); x1 = __builtin_fmaf(x1, a, x2); x2 = __builtin_fmaf(x2, a, x3); x3 = __builtin_fmaf(x3, a, x4);
        x4 = __builtin_fmaf(x4, a, x5); x5 = __builtin_fmaf(x5, a, x6); x6 = __builtin_fmaf(x6, a, x7); x7 = __builtin_fmaf(x7, a, x0);
      } else if (KIND == 1) {  // v_fma_f64
        d0 = __builtin_fma(d0, da, d1); d1 = __builtin_fma(d1, da, d2); d2 = __builtin_fma(d2, da, d3); d3 = __builtin_fma(d3, da, d4);
        d4 = __builtin_fma(d4, da, d5); d5 = __builtin_fma(d5, da, d6); d6 = __builtin_fma(d6, da, d7); d7 = __builtin_fma(d7, da, d0);
      } else if (KIND == 2) {  // v_add_f64
        d0 = d0 + d1; d1 = d1 + d2; d2 = d2 + d3; d3 = d3 + d4; d4 = d4 + d5; d5 = d5 + d6; d6 = d6 + d7; d7 = d7 + d0;
      } else if (KIND == 3) {  // int add / xor mix
        i0 = (i0 + i1) ^ ia; i1 = (i1 + i2) ^ ia; i2 = (i2 + i3) ^ ia; i3 = (i3 + i4) ^ ia;
        i4 = (i4 + i5) ^ ia; i5 = (i5 + i6) ^ ia; i6 = (i6 + i7) ^ ia; i7 = (i7 + i0) ^ ia;
      } else if (KIND == 4) {  // ds_read_b32 (conflict-free) + add
        x0 += lds[(i0 + r * 64) & 4095]; x1 += lds[(i0 + r * 64 + 512) & 4095]; x2 += lds[(i0 + r * 64 + 1024) & 4095]; x3 += lds[(i0 + r * 64 + 1536) & 4095];
        x4 += lds[(i0 + r * 64 + 2048) & 4095]; x5 += lds[(i0 + r * 64 + 2560) & 4095]; x6 += lds[(i0 + r * 64 + 3072) & 4095]; x7 += lds[(i0 + r * 64 + 3584) & 4095];
      } else if (KIND == 5) {  // v_mul_f64
        d0 = d0 * da; d1 = d1 * da; d2 = d2 * da; d3 = d3 * da; d4 = d4 * da; d5 = d5 * da; d6 = d6 * da; d7 = d7 * da;
      } else if (KIND == 6) {  // v_cndmask on f64 (2 x b32) via compare
        d0 = d0 > da ? d1 : d0; d1 = d1 > da ? d2 : d1; d2 = d2 > da ? d3 : d2; d3 = d3 > da ? d4 : d3;
        d4 = d4 > da ? d5 : d4; d5 = d5 > da ? d6 : d5; d6 = d6 > da ? d7 : d6; d7 = d7 > da ? d0 : d7;
      } else if (KIND == 8) {  // v_ldexp_f64
        d0 = __builtin_ldexp(d0, ia); d1 = __builtin_ldexp(d1, ia); d2 = __builtin_ldexp(d2, ia); d3 = __builtin_ldexp(d3, ia);
        d4 = __builtin_ldexp(d4, ia); d5 = __builtin_ldexp(d5, ia); d6 = __builtin_ldexp(d6, ia); d7 = __builtin_ldexp(d7, ia);
      } else if (KIND == 9) {  // v_min_f64
        d0 = __builtin_fmin(d0, d1); d1 = __builtin_fmin(d1, d2); d2 = __builtin_fmin(d2, d3); d3 = __builtin_fmin(d3, d4);
        d4 = __builtin_fmin(d4, d5); d5 = __builtin_fmin(d5, d6); d6 = __builtin_fmin(d6, d7); d7 = __builtin_fmin(d7, d0);
      } else if (KIND == 10) {  // v_cmp_gt_f64 -> ballot accumulate (cmp only + salu)
        unsigned long long m = 0;
        m += __builtin_amdgcn_ballot_w64(d0 > da); m += __builtin_amdgcn_ballot_w64(d1 > da); m += __builtin_amdgcn_ballot_w64(d2 > da); m += __builtin_amdgcn_ballot_w64(d3 > da);
        m += __builtin_amdgcn_ballot_w64(d4 > da); m += __builtin_amdgcn_ballot_w64(d5 > da); m += __builtin_amdgcn_ballot_w64(d6 > da); m += __builtin_amdgcn_ballot_w64(d7 > da);
        da += (double)(m & 1);
      } else if (KIND == 11) {  // v_cndmask_b32 with a fixed mask
        const bool c = (threadIdx.x & 1);
        i0 = c ? i1 : i0; i1 = c ? i2 : i1; i2 = c ? i3 : i2; i3 = c ? i4 : i3; i4 = c ? i5 : i4; i5 = c ? i6 : i5; i6 = c ? i7 : i6; i7 = c ? i0 : i7;
        asm volatile("" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7));
      } else if (KIND == 12) {  // v_bfi_b32
        i0 = (i0 & ia) | (i1 & ~ia); i1 = (i1 & ia) | (i2 & ~ia); i2 = (i2 & ia) | (i3 & ~ia); i3 = (i3 & ia) | (i4 & ~ia);
        i4 = (i4 & ia) | (i5 & ~ia); i5 = (i5 & ia) | (i6 & ~ia); i6 = (i6 & ia) | (i7 & ~ia); i7 = (i7 & ia) | (i0 & ~ia);
      } else if (KIND == 13) {  // f32: ldexp + med3 (sgn_scaled<float>)
        x0 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x0, 200), -a, a); x1 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x1, 200), -a, a);
        x2 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x2, 200), -a, a); x3 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x3, 200), -a, a);
        x4 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x4, 200), -a, a); x5 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x5, 200), -a, a);
        x6 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x6, 200), -a, a); x7 = __builtin_amdgcn_fmed3f(__builtin_ldexpf(x7, 200), -a, a);
      } else if (KIND == 7) {  // v_pk_fma_f32
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, aa = {a, a};
        p0 = __builtin_elementwise_fma(p0, aa, p1); p1 = __builtin_elementwise_fma(p1, aa, p2);
        p2 = __builtin_elementwise_fma(p2, aa, p3); p3 = __builtin_elementwise_fma(p3, aa, p0);
        p0 = __builtin_elementwise_fma(p0, aa, p1); p1 = __builtin_elementwise_fma(p1, aa, p2);
        p2 = __builtin_elementwise_fma(p2, aa, p3); p3 = __builtin_elementwise_fma(p3, aa, p0);
        x0 = p0.x; x1 = p0.y; x2 = p1.x; x3 = p1.y; x4 = p2.x; x5 = p2.y; x6 = p3.x; x7 = p3.y;
      }
    }
  }
  float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) + (float)(i0 + i1 + i2 + i3 + i4 + i5 + i6 + i7);
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, int waves_per_simd) {
  float* out; hipMalloc(&out, 4);
  const int iters = 2000;
  const int blocks = 256 * waves_per_simd / 2;  // 512 threads = 8 waves = 2 per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<KIND><<<blocks, 512>>>(out, 10, 1.0001f, 1.0001, 3);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<KIND><<<blocks, 512>>>(out, iters, 1.0001f, 1.0001, 3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)iters * REP * waves_per_simd;
  printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4 GHz = %.2f cyc)\n", name, waves_per_simd, ms,
         ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
  hipFree(out);
}

int main() {
  for (int w : {4, 8}) {
    run<0>("v_fma_f32", w); run<7>("v_pk_fma_f32", w); run<1>("v_fma_f64", w); run<2>("v_add_f64", w); run<5>("v_mul_f64", w);
    run<3>("int add+xor", w); run<6>("cmp+cndmask64", w); run<4>("ds_read+add", w);
    run<8>("v_ldexp_f64", w); run<9>("v_min_f64", w); run<10>("v_cmp_gt_f64", w); run<11>("v_cndmask_b32", w); run<12>("v_bfi_b32", w);
    run<13>("ldexp+med3 f32", w);
  }
  return 0;
}
