// channel_map.hip -- dense per-pixel spectral maps:
// out = M * (in - offset_in) + offset_out over planar images, the projection /
// back-projection of the reference's SpectralPCA (spectral_pca.cpp:94-161, which
// calls cv::PCA::project / backProject once per pixel).  One DGEMM per chunk of
// pixels (rocBLAS: a plain library GEMM, MFMA f64 underneath), the offsets folded
// into a bias that pre-fills the output (beta = 1).  Host buffers cross PCIe
// through the context's pinned staging chunks.
#include <rocblas/rocblas.h>

#include <vector>

#include "srmap_internal.hpp"

namespace srmap {

__global__ void k_fill_rows(double* __restrict__ out, const double* __restrict__ bias, size_t n, int rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (i < n && r < rows) out[(size_t)r * n + i] = bias[r];
}

}  // namespace srmap

using namespace srmap;

extern "C" int srmap_channel_map(srmap_ctx* ctx, int rows_out, int rows_in, size_t n, const double* M,
                                 const double* offset_in, const double* offset_out, const double* in_host,
                                 double* out_host) {
  if (!ctx || !M || !in_host || !out_host || rows_out <= 0 || rows_in <= 0 || n == 0) return SRMAP_EINVAL;
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  // bias[r] = offset_out[r] - sum_c M[r][c] * offset_in[c]
  std::vector<double> bias((size_t)rows_out, 0.0);
  for (int r = 0; r < rows_out; ++r) {
    double b = offset_out ? offset_out[r] : 0.0;
    if (offset_in)
      for (int c = 0; c < rows_in; ++c) b -= M[(size_t)r * rows_in + c] * offset_in[c];
    bias[r] = b;
  }
  static rocblas_handle handle = nullptr;  // one per process (one context per process/GPU)
  if (!handle) {
    if (rocblas_create_handle(&handle) != rocblas_status_success)
      return set_error(ctx, SRMAP_EHIP, "rocblas_create_handle failed");
  }
  if (rocblas_set_stream(handle, st) != rocblas_status_success) return set_error(ctx, SRMAP_EHIP, "rocblas_set_stream failed");
  // pixels are processed in chunks so that in + out stay within ~2 GiB of HBM whatever the cube size
  const size_t per_pixel = (size_t)(rows_in + rows_out) * sizeof(double);
  size_t chunk = ((size_t)2 << 30) / per_pixel;
  if (chunk > n) chunk = n;
  if (chunk == 0) chunk = 1;
  double *d_in = nullptr, *d_out = nullptr, *d_M = nullptr, *d_bias = nullptr;
  int rc = SRMAP_OK;
  auto fail = [&](int code, const char* what) { rc = set_error(ctx, code, "%s", what); };
  if (hipMalloc((void**)&d_in, chunk * rows_in * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_out, chunk * rows_out * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_M, (size_t)rows_out * rows_in * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_bias, (size_t)rows_out * sizeof(double)) != hipSuccess)
    fail(SRMAP_ENOMEM, "hipMalloc failed");
  if (rc == SRMAP_OK &&
      (hipMemcpyAsync(d_M, M, (size_t)rows_out * rows_in * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
       hipMemcpyAsync(d_bias, bias.data(), (size_t)rows_out * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
       hipStreamSynchronize(st) != hipSuccess))
    fail(SRMAP_EHIP, "upload of the map failed");
  srmap_problem tmp;  // staging helpers take a problem for its context / dtype
  tmp.ctx = ctx;
  tmp.dtype = SRMAP_F64;
  for (size_t p0 = 0; p0 < n && rc == SRMAP_OK; p0 += chunk) {
    const size_t m = n - p0 < chunk ? n - p0 : chunk;
    for (int c = 0; c < rows_in && rc == SRMAP_OK; ++c)  // planar rows of this chunk
      rc = convert_upload(&tmp, in_host + (size_t)c * n + p0, d_in + (size_t)c * m, m, st);
    if (rc) break;
    hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)((m + 255) / 256), rows_out), dim3(256), 0, st, d_out, d_bias, m, rows_out);
    // row-major planar [rows][m] == column-major (m x rows), ld = m:  OUT(m x ro) = IN(m x ri) * M^T(ri x ro) + OUT
    const double one = 1.0;
    const rocblas_status bs = rocblas_dgemm(handle, rocblas_operation_none, rocblas_operation_none, (rocblas_int)m,
                                            rows_out, rows_in, &one, d_in, (rocblas_int)m, d_M, rows_in, &one, d_out,
                                            (rocblas_int)m);
    if (bs != rocblas_status_success) { fail(SRMAP_EHIP, "rocblas_dgemm failed"); break; }
    for (int r = 0; r < rows_out && rc == SRMAP_OK; ++r)
      rc = convert_download(&tmp, d_out + (size_t)r * m, out_host + (size_t)r * n + p0, m, st);
  }
  if (d_in) (void)hipFree(d_in);
  if (d_out) (void)hipFree(d_out);
  if (d_M) (void)hipFree(d_M);
  if (d_bias) (void)hipFree(d_bias);
  return rc;
}
