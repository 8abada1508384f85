// Exact per-instruction issue cost on gfx950 (inline asm so the compiler cannot fold the stream).
// hipcc --offload-arch=gfx950 -O3 valu_asm.hip -o valu_asm && ./valu_asm
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY8(INS) R8(INS)
#define REP64(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS) BODY8(INS)

#define DEFK(NAME, ASMSTR, CONSTR_OUT, CONSTR_IN)                                           \
  __global__ __launch_bounds__(512) void NAME(float* out, int iters, int ia) {                \
    int v[8];                                                                                 \
    double d[8];                                                                              \
    for (int i = 0; i < 8; ++i) { v[i] = threadIdx.x + i; d[i] = threadIdx.x + i; }           \
    unsigned long long m = (threadIdx.x & 1) ? 0x5555555555555555ull : 0xaaaaaaaaaaaaaaaaull; \
    m = __builtin_amdgcn_readfirstlane((int)m) | ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(m >> 32)) << 32); \
    for (int it = 0; it < iters; ++it) {                                                      \
      REP64(ASMSTR)                                                                           \
    }                                                                                         \
    float s = 0;                                                                              \
    for (int i = 0; i < 8; ++i) s += v[i] + (float)d[i];                                      \
    if (s == 12345.678f) out[0] = s;                                                          \
  }

#define I_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "s"(m));
#define I_CNDVCC(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_BFI(i) asm volatile("v_bfi_b32 %0, %2, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(ia));
#define I_MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_MED3(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(ia));
#define I_CMPF64(i) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(d[i]), "v"(d[(i + 1) & 7]) : "vcc");
#define I_CMPF32(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(v[i]), "v"(v[(i + 1) & 7]) : "vcc");
#define I_CMPU32(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" : : "v"(v[i]), "v"(v[(i + 1) & 7]) : "vcc");
#define I_MINF64(i) asm volatile("v_min_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_LDEXP64(i) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(d[i]) : "v"(ia));
#define I_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(v[(i + 1) & 7]), "v"(ia));
#define I_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_FMAABS64(i) asm volatile("v_fma_f64 %0, |%0|, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_READLANE(i) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[i]) : "s20");
#define I_CNDVCC64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_CMPCND(i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[i]) : "v"(v[(i + 1) & 7]) : "vcc");
#define I_CMPCNDS(i) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(v[i]) : "v"(v[(i + 1) & 7]) : "s20", "s21");
#define I_ADDCO(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]) : "vcc");
#define I_MULF(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_MINF(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_LSHL(i) asm volatile("v_lshlrev_b32 %0, 1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_SUB(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define I_LDEXP32(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(v[i]) : "v"(ia));
#define I_CVT(i) asm volatile("v_cvt_f32_i32 %0, %1" : "+v"(v[i]) : "v"(v[(i + 1) & 7]));
#define I_FMAS(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "s"(ia));
#define I_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));

DEFK(k_cnd, I_CND, , )
DEFK(k_cndvcc, I_CNDVCC, , )
DEFK(k_add, I_ADD, , )
DEFK(k_bfi, I_BFI, , )
DEFK(k_maxf, I_MAXF, , )
DEFK(k_med3, I_MED3, , )
DEFK(k_cmpf64, I_CMPF64, , )
DEFK(k_cmpf32, I_CMPF32, , )
DEFK(k_cmpu32, I_CMPU32, , )
DEFK(k_minf64, I_MINF64, , )
DEFK(k_ldexp64, I_LDEXP64, , )
DEFK(k_fma64, I_FMA64, , )
DEFK(k_fma32, I_FMA32, , )
DEFK(k_mov, I_MOV, , )
DEFK(k_mov64, I_MOV64, , )
DEFK(k_and, I_AND, , )
DEFK(k_andor, I_ANDOR, , )
DEFK(k_pkfma, I_PKFMA, , )
DEFK(k_pkadd, I_PKADD, , )
DEFK(k_fmaabs64, I_FMAABS64, , )
DEFK(k_readlane, I_READLANE, , )
DEFK(k_cndvcc64, I_CNDVCC64, , )
DEFK(k_cmpcnd, I_CMPCND, , )
DEFK(k_cmpcnds, I_CMPCNDS, , )
DEFK(k_addco, I_ADDCO, , )
DEFK(k_mulf, I_MULF, , )
DEFK(k_addf, I_ADDF, , )
DEFK(k_fmac, I_FMAC, , )
DEFK(k_minf, I_MINF, , )
DEFK(k_lshl, I_LSHL, , )
DEFK(k_lshladd, I_LSHLADD, , )
DEFK(k_add3, I_ADD3, , )
DEFK(k_mad24, I_MAD24, , )
DEFK(k_mullo, I_MULLO, , )
DEFK(k_or, I_OR, , )
DEFK(k_xor, I_XOR, , )
DEFK(k_sub, I_SUB, , )
DEFK(k_add64, I_ADD64, , )
DEFK(k_mul64, I_MUL64, , )
DEFK(k_ldexp32, I_LDEXP32, , )
DEFK(k_cvt, I_CVT, , )
DEFK(k_fmas, I_FMAS, , )
DEFK(k_pkmul, I_PKMUL, , )

typedef void (*kfn)(float*, int, int);
static void run(const char* name, kfn f, int wps) {
  float* out; (void)hipMalloc(&out, 4);
  const int iters = 2000, blocks = 256 * wps / 2;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  f<<<blocks, 512>>>(out, 10, 3);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  f<<<blocks, 512>>>(out, iters, 3);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)iters * 64 * wps;
  printf("%-16s waves/SIMD=%d %8.3f ms  %6.2f ns/instr/SIMD\n", name, wps, ms, ms * 1e6 / n); fflush(stdout);
  (void)hipFree(out);
}

int main() {
  for (int w : {8}) {
#define RUN(k) run(#k, k, w);
    RUN(k_fma32) RUN(k_fma64) RUN(k_fmaabs64) RUN(k_pkfma) RUN(k_pkadd) RUN(k_add) RUN(k_and) RUN(k_andor) RUN(k_bfi) RUN(k_mov) RUN(k_mov64)
    RUN(k_maxf) RUN(k_med3) RUN(k_minf64) RUN(k_ldexp64) RUN(k_cmpf64) RUN(k_cmpf32) RUN(k_cmpu32) RUN(k_cnd) RUN(k_cndvcc)
    RUN(k_readlane) RUN(k_cndvcc64) RUN(k_cmpcnd) RUN(k_cmpcnds) RUN(k_addco) RUN(k_mulf) RUN(k_addf) RUN(k_fmac) RUN(k_minf) RUN(k_lshl) RUN(k_lshladd) RUN(k_add3) RUN(k_mad24) RUN(k_mullo) RUN(k_or) RUN(k_xor) RUN(k_sub) RUN(k_add64) RUN(k_mul64) RUN(k_ldexp32) RUN(k_cvt) RUN(k_fmas) RUN(k_pkmul)
  }
  return 0;
}
