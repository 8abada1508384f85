// channel_map.hip -- dense per-pixel spectral maps:
// out = M * (in - offset_in) + offset_out over planar images, the projection /
// back-projection of the reference's SpectralPCA (spectral_pca.cpp:94-161, which
// calls cv::PCA::project / backProject once per pixel).  One DGEMM per chunk of
// pixels (rocBLAS: a plain library GEMM, MFMA f64 underneath), the offsets folded
// into a bias that pre-fills the output (beta = 1).  Host buffers cross PCIe
// through the context's pinned staging chunks; srmap_channel_map_device works on
// device-resident cubes.  The PCA training itself (spectral_pca.cpp:30-88 over
// cv::PCA: mean, covariance / n, eigenvectors by descending eigenvalue) also runs
// on the device: row means and the centred sample matrix by two small kernels,
// the covariance as one DGEMM on it, the eigen-decomposition by rocSOLVER's
// dsyevd -- library calls, the one place on the path where the matrix cores apply.
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>

#include <algorithm>
#include <cmath>
#include <numeric>

#include <vector>

#include "srmap_internal.hpp"

namespace srmap {

__global__ void k_fill_rows(double* __restrict__ out, const double* __restrict__ bias, size_t n, int rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (i < n && r < rows) out[(size_t)r * n + i] = bias[r];
}

// mean[r] = (1 / count) * sum_j in[r * n + first + j * stride]   (one block per row, fixed order)
__global__ __launch_bounds__(256) void k_row_means(const double* __restrict__ in, size_t n, size_t first, size_t stride,
                                                  size_t count, double* __restrict__ mean) {
  __shared__ double red[4];
  const int r = blockIdx.x;
  double s = 0.0;
  for (size_t j = threadIdx.x; j < count; j += 256) s += in[(size_t)r * n + first + j * stride];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  if (lane == 0) red[wid] = s;
  __syncthreads();
  if (threadIdx.x == 0) mean[r] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)count;
}

// t[r * count + j] = in[r * n + first + j * stride] - mean[r]
__global__ void k_center_samples(const double* __restrict__ in, size_t n, size_t first, size_t stride, size_t count,
                                 const double* __restrict__ mean, double* __restrict__ t) {
  const size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = blockIdx.y;
  if (j < count) t[(size_t)r * count + j] = in[(size_t)r * n + first + j * stride] - mean[r];
}

static int blas_handle(srmap_ctx* ctx, hipStream_t st, rocblas_handle* out) {
  if (!ctx->blas) {
    rocblas_handle h = nullptr;
    if (rocblas_create_handle(&h) != rocblas_status_success) return set_error(ctx, SRMAP_EHIP, "rocblas_create_handle failed");
    ctx->blas = h;
  }
  if (rocblas_set_stream((rocblas_handle)ctx->blas, st) != rocblas_status_success)
    return set_error(ctx, SRMAP_EHIP, "rocblas_set_stream failed");
  *out = (rocblas_handle)ctx->blas;
  return SRMAP_OK;
}

void blas_release(srmap_ctx* ctx) {
  if (ctx->blas) { (void)rocblas_destroy_handle((rocblas_handle)ctx->blas); ctx->blas = nullptr; }
}

// out_dev = M (in_dev - offset_in) + offset_out on device-resident planar f64 cubes, enqueued on st
static int channel_map_core(srmap_ctx* ctx, int rows_out, int rows_in, size_t n, const double* d_M, const double* d_bias,
                            const double* d_in, double* d_out, hipStream_t st) {
  rocblas_handle handle;
  int rc = blas_handle(ctx, st, &handle);
  if (rc) return rc;
  hipLaunchKernelGGL(k_fill_rows, dim3((unsigned)((n + 255) / 256), rows_out), dim3(256), 0, st, d_out, d_bias, n, rows_out);
  // row-major planar [rows][n] == column-major (n x rows), ld = n:  OUT(n x ro) = IN(n x ri) * M^T(ri x ro) + OUT
  const double one = 1.0;
  if (rocblas_dgemm(handle, rocblas_operation_none, rocblas_operation_none, (rocblas_int)n, rows_out, rows_in, &one, d_in,
                    (rocblas_int)n, d_M, rows_in, &one, d_out, (rocblas_int)n) != rocblas_status_success)
    return set_error(ctx, SRMAP_EHIP, "rocblas_dgemm failed");
  return SRMAP_OK;
}

}  // namespace srmap

using namespace srmap;

extern "C" int srmap_channel_map_device(srmap_ctx* ctx, int rows_out, int rows_in, size_t n, const double* M,
                                        const double* offset_in, const double* offset_out, const double* in_dev,
                                        double* out_dev, void* hip_stream) {
  if (!ctx || !M || !in_dev || !out_dev || rows_out <= 0 || rows_in <= 0 || n == 0) return SRMAP_EINVAL;
  if (n > (size_t)0x7fffffff) return set_error(ctx, SRMAP_EUNSUPPORTED, "more than 2^31 pixels per call");
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  std::vector<double> bias((size_t)rows_out, 0.0);
  for (int r = 0; r < rows_out; ++r) {
    double b = offset_out ? offset_out[r] : 0.0;
    if (offset_in)
      for (int c = 0; c < rows_in; ++c) b -= M[(size_t)r * rows_in + c] * offset_in[c];
    bias[r] = b;
  }
  double *d_M = nullptr, *d_bias = nullptr;
  SRMAP_HIP(ctx, hipMalloc((void**)&d_M, (size_t)rows_out * rows_in * sizeof(double)));
  if (hipMalloc((void**)&d_bias, (size_t)rows_out * sizeof(double)) != hipSuccess) { (void)hipFree(d_M); return set_error(ctx, SRMAP_ENOMEM, "hipMalloc failed"); }
  int rc = SRMAP_OK;
  if (hipMemcpyAsync(d_M, M, (size_t)rows_out * rows_in * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipMemcpyAsync(d_bias, bias.data(), (size_t)rows_out * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess)
    rc = set_error(ctx, SRMAP_EHIP, "upload of the map failed");
  if (rc == SRMAP_OK) rc = channel_map_core(ctx, rows_out, rows_in, n, d_M, d_bias, in_dev, out_dev, st);
  (void)hipStreamSynchronize(st);  // the map / bias buffers and the host bias vector are released below
  (void)hipFree(d_M);
  (void)hipFree(d_bias);
  return rc;
}

// PCA of the samples in_dev[r][first + j * stride], j < count, of a device-resident planar f64 cube [rows][n]:
// mean (host, rows), eigenvalues of the covariance / count in descending order (host, rows), basis (host,
// rows x rows, row k = k-th eigenvector, sign: its largest-magnitude component is positive).
extern "C" int srmap_channel_pca_device(srmap_ctx* ctx, int rows, size_t n, const double* in_dev, size_t first,
                                        size_t stride, size_t count, double* mean_out, double* eigenvalues_out,
                                        double* basis_out, void* hip_stream) {
  if (!ctx || !in_dev || !mean_out || !eigenvalues_out || !basis_out || rows <= 0 || count == 0 || stride == 0)
    return SRMAP_EINVAL;
  if (first + (count - 1) * stride >= n) return set_error(ctx, SRMAP_EINVAL, "PCA samples outside the cube");
  if (count > (size_t)0x7fffffff) return set_error(ctx, SRMAP_EUNSUPPORTED, "more than 2^31 samples");
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : ctx->stream;
  rocblas_handle handle;
  int rc = blas_handle(ctx, st, &handle);
  if (rc) return rc;
  double *d_mean = nullptr, *d_t = nullptr, *d_cov = nullptr, *d_w = nullptr, *d_e = nullptr;
  rocblas_int* d_info = nullptr;
  auto cleanup = [&]() {
    void* b[] = {d_mean, d_t, d_cov, d_w, d_e, d_info};
    for (void* q : b) if (q) (void)hipFree(q);
  };
  if (hipMalloc((void**)&d_mean, rows * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_t, (size_t)rows * count * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_cov, (size_t)rows * rows * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_w, rows * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_e, rows * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&d_info, sizeof(rocblas_int)) != hipSuccess) {
    cleanup();
    return set_error(ctx, SRMAP_ENOMEM, "hipMalloc failed");
  }
  hipLaunchKernelGGL(k_row_means, dim3(rows), dim3(256), 0, st, in_dev, n, first, stride, count, d_mean);
  hipLaunchKernelGGL(k_center_samples, dim3((unsigned)((count + 255) / 256), rows), dim3(256), 0, st, in_dev, n, first, stride,
                     count, (const double*)d_mean, d_t);
  // planar [rows][count] row-major == column-major (count x rows): cov = T^T T / count
  const double alpha = 1.0 / (double)count, beta = 0.0;
  rocblas_status bs = rocblas_dgemm(handle, rocblas_operation_transpose, rocblas_operation_none, rows, rows, (rocblas_int)count,
                                    &alpha, d_t, (rocblas_int)count, d_t, (rocblas_int)count, &beta, d_cov, rows);
  if (bs == rocblas_status_success)
    bs = rocsolver_dsyevd(handle, rocblas_evect_original, rocblas_fill_lower, rows, d_cov, rows, d_w, d_e, d_info);
  if (bs != rocblas_status_success) { cleanup(); return set_error(ctx, SRMAP_EHIP, "covariance / eigen-decomposition failed"); }
  std::vector<double> w(rows), V((size_t)rows * rows);
  rocblas_int info = 0;
  hipError_t e = hipMemcpyAsync(w.data(), d_w, rows * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(V.data(), d_cov, (size_t)rows * rows * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(mean_out, d_mean, rows * sizeof(double), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipMemcpyAsync(&info, d_info, sizeof(info), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  cleanup();
  SRMAP_HIP(ctx, e);
  if (info != 0) return set_error(ctx, SRMAP_EHIP, "dsyevd did not converge (info %d)", (int)info);
  // dsyevd: ascending eigenvalues, eigenvector k = column k (column-major).  cv::PCA: descending, rows.
  for (int k = 0; k < rows; ++k) {
    const int src = rows - 1 - k;
    eigenvalues_out[k] = w[src];
    const double* col = &V[(size_t)src * rows];
    int big = 0;
    for (int c = 1; c < rows; ++c)
      if (std::fabs(col[c]) > std::fabs(col[big])) big = c;
    const double sign = col[big] < 0 ? -1.0 : 1.0;
    for (int c = 0; c < rows; ++c) basis_out[(size_t)k * rows + c] = sign * col[c];
  }
  return SRMAP_OK;
}

// The same from a host sample matrix [rows][count] (planar): what SpectralPCA's sampling rule produces.
extern "C" int srmap_channel_pca(srmap_ctx* ctx, int rows, size_t count, const double* samples_host, double* mean_out,
                                 double* eigenvalues_out, double* basis_out) {
  if (!ctx || !samples_host || rows <= 0 || count == 0) return SRMAP_EINVAL;
  SRMAP_HIP(ctx, hipSetDevice(ctx->device));
  double* d = nullptr;
  SRMAP_HIP(ctx, hipMalloc((void**)&d, (size_t)rows * count * sizeof(double)));
  hipError_t e = hipMemcpy(d, samples_host, (size_t)rows * count * sizeof(double), hipMemcpyHostToDevice);
  int rc = SRMAP_OK;
  if (e != hipSuccess) rc = set_error(ctx, SRMAP_EHIP, "upload of the PCA samples failed");
  if (rc == SRMAP_OK) rc = srmap_channel_pca_device(ctx, rows, count, d, 0, 1, count, mean_out, eigenvalues_out, basis_out, nullptr);
  (void)hipFree(d);
  return rc;
}

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
  rocblas_handle handle;
  {
    const int hrc = blas_handle(ctx, st, &handle);
    if (hrc) return hrc;
  }
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
