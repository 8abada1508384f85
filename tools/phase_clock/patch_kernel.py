"""Instruments a COPY of csrc/kernels_ztile.hip (written to /tmp/spx/kz_time.hip) with per-wave s_memtime stamps at the
phase boundaries of k_eval_z and a setter srmap_dbg_set_timing(buffer).  Timing-only: see build.sh / run.py."""
import sys
root = sys.argv[1] if len(sys.argv) > 1 else "/root/repo"
src=open(root + '/super-resolution_amd/csrc/kernels_ztile.hip').read()
s=src
anchor = 'template <typename T, int S, int B, int REGK, int R, bool WD, bool SP>\n__global__'
assert anchor in s
s=s.replace(anchor,'__device__ unsigned long long* g_ztdbg = nullptr;\n#define ZT_STAMP(i) do { zts[i] = __builtin_readcyclecounter(); } while (0)\n\n' + anchor,1)
def rep(old,new):
    global s
    assert old in s, old
    s=s.replace(old,new,1)
rep('''  const int R0 = tby * C::TH, CJ0 = tbx * C::CW, C0 = CJ0 * S;''','''  const int R0 = tby * C::TH, CJ0 = tbx * C::CW, C0 = CJ0 * S;
  unsigned long long* dbg = g_ztdbg ? g_ztdbg + ((size_t)(by * gridDim.x + blockIdx.x) * 8 + wv) * 16 : nullptr;
  unsigned long long zts[10];
  const unsigned long long rt0 = wall_clock64();  // s_memrealtime: 100 MHz, common to the whole chip
  const unsigned hwid = __builtin_amdgcn_s_getreg(63492), xccid = __builtin_amdgcn_s_getreg(63508);
  ZT_STAMP(0);''')
rep('''  T ypre[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ypre[v] = T(0);''','''  ZT_STAMP(1);
  T ypre[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) ypre[v] = T(0);''')
rep('''  if (!SP && has_z_halo) z_row_prefetch<T, S, B, C>(A, hrowz, R0, CJ0, lane, edge, ybase, ypre2);''','''  ZT_STAMP(2);
  if (!SP && has_z_halo) z_row_prefetch<T, S, B, C>(A, hrowz, R0, CJ0, lane, edge, ybase, ypre2);
  ZT_STAMP(3);''')
rep('''  // ---------------- x tile -> LDS, polyphase ----------------''','''  ZT_STAMP(4);
  // ---------------- x tile -> LDS, polyphase ----------------''')
rep('''  // in-image mask of this thread's pixels (partial tiles at the right / bottom edge)''','''  ZT_STAMP(5);
  // in-image mask of this thread's pixels (partial tiles at the right / bottom edge)''')
rep('''  // ---------------- phase 1: regulariser ----------------''','''  ZT_STAMP(6);
  // ---------------- phase 1: regulariser ----------------''')
rep('''  __syncthreads();

  // ---------------- phase 2 ----------------''','''  ZT_STAMP(7);
  __syncthreads();
  ZT_STAMP(8);

  // ---------------- phase 2 ----------------''')
rep('''  // ---------------- cost partial of this workgroup ----------------''','''  ZT_STAMP(9);
  if (dbg && lane == 0) { for (int q = 0; q < 10; ++q) dbg[q] = zts[q]; dbg[10] = rt0; dbg[11] = wall_clock64(); dbg[12] = ((unsigned long long)xccid << 32) | hwid; }
  // ---------------- cost partial of this workgroup ----------------''')
rep('''    if (bidx * C::NT < Bd.n_ring) {''','''    const unsigned long long brt0 = wall_clock64();
    if (bidx * C::NT < Bd.n_ring) {''')
rep('''    else if (threadIdx.x == 0) {
      const int nbb = A.nby * gridDim.x;''','''    if (g_ztdbg && threadIdx.x == 0) { unsigned long long* d = g_ztdbg + ((size_t)(2048 + bidx) * 8) * 16; d[10] = brt0; d[11] = wall_clock64(); d[12] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); d[13] = (bidx * C::NT < Bd.n_ring); }
    if (!(bidx * C::NT < Bd.n_ring) && threadIdx.x == 0) {
      const int nbb = A.nby * gridDim.x;''')
rep('// ---------------------------------------------------------------------------------------------------------\n// host side: plan','extern "C" int srmap_dbg_set_timing(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ztdbg), &buf, sizeof(buf)); }\n// ---------------------------------------------------------------------------------------------------------\n// host side: plan')
open('/tmp/spx/kz_time.hip','w').write(s)
