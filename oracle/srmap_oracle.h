/*
 * srmap_oracle.h -- CPU restatement of the reference MAP super-resolution
 * gradient path (rteammco/super-resolution).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product library (libsrmap.so, super-resolution_amd/csrc) never links,
 * includes or calls anything in oracle/.
 *
 * Everything is IEEE double, single threaded, planar [C][H][W] images with
 * index c*W*H + row*W + col (reference src/util/util.cpp:81-89), and keeps
 * the reference's pass structure (whole-image passes per frame, residual
 * evaluated at HR resolution on nearest-neighbour upsampled images).
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - decimation / zero insertion / NN + additive resize, Gaussian blur, TV
 *     values, 3-D TV values, BTV values, TV 2-D gradient (finite differences),
 *     PSNR, integer-shift motion matrices, the SmallDataTest solve: pinned by
 *     the reference's own test literals (tests/golden/reference_literals.json).
 *   - nonlinear CG (mincg + More'-Thuente search): pinned against the
 *     reference's vendored ALGLIB 3.10.0 compiled from /root/reference
 *     (oracle/_ref) and by committed trajectories (tests/golden/cg_*.json).
 *   - BTV gradient, 3-D TV gradient, sub-pixel (fractional) warpAffine: PARITY
 *     UNPINNED -- the reference has no test for them and its sources for them
 *     cannot be built here without stand-in headers (OpenCV/glog absent), so
 *     they are restated line-by-line from the cited reference lines only.
 */
#ifndef SRMAP_ORACLE_H_
#define SRMAP_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

/* ---- third-party (OpenCV 3.2+, not in /root/reference) restatements ---- */

/* cv::warpAffine(src, dst, [1 0 dx; 0 1 dy], size) with the defaults the
 * reference uses (INTER_LINEAR, BORDER_CONSTANT 0); call site
 * src/image_model/motion_module.cpp:18-25.  src != dst. */
void sro_warp_shift(const double* src, double* dst, int W, int H,
                    double dx, double dy);
/* The fixed-point source coordinate tables warpAffine builds: X[x] (W ints),
 * Y[y] (H ints) in 1/32 px units (integer part X>>5, fraction X&31). */
void sro_warp_tables(int W, int H, double dx, double dy, int* X, int* Y);

/* cv::getGaussianKernel(ksize, sigma, CV_64F) for sigma > 0; call site
 * src/image_model/blur_module.cpp:20-22.  out2d (ksize*ksize) = k * k^T. */
void sro_gaussian_kernel(int ksize, double sigma, double* out1d, double* out2d);

/* cv::filter2D(src, dst, -1, kernel, anchor centre, 0, BORDER_CONSTANT);
 * call site src/util/matrix_util.cpp:12-29.  Correlation, direct sum. */
void sro_filter2d(const double* src, double* dst, int W, int H,
                  const double* kernel, int kw, int kh);

/* cv::resize(INTER_NEAREST) index map dst->src for one axis; call site
 * src/image/image_data.cpp:338-350. */
void sro_nearest_map(int src_len, int dst_len, int* map);
void sro_resize_nearest(const double* src, int sw, int sh,
                        double* dst, int dw, int dh);

/* ---- reference's own code ---- */

/* ResizeAdditiveInterpolation, src/image/image_data.cpp:80-134. */
void sro_resize_additive(const double* src, int sw, int sh,
                         double* dst, int dw, int dh);

/* (int)(len * (1.0/scale)) -- DownsamplingModule::ApplyToImage size rule,
 * src/image_model/downsampling_module.cpp:19-27 + image_data.cpp:353-364. */
int sro_downsampled_len(int len, int scale);

typedef struct {
  int scale;            /* DownsamplingModule scale, >= 1 */
  int num_frames;       /* K; 0 = no MotionModule in the chain */
  const double* shifts; /* K pairs (dx, dy) */
  int blur_ksize;       /* "blur_radius" = kernel size; 0 = no BlurModule */
  double blur_sigma;
} sro_model;

/* ImageModel::ApplyToImage(ImageData*, k): M_k, B, D in that order
 * (src/image_model/image_model.cpp:86-91).  hr is [C][H][W]; lr is
 * [C][h][w] with w = sro_downsampled_len(W, scale). */
void sro_model_apply(const sro_model* m, int k, const double* hr,
                     int W, int H, int C, double* lr);
/* ImageModel::ApplyTransposeToImage: D^T, B^T, M_k^T
 * (src/image_model/image_model.cpp:93-101).  lr [C][h][w] -> hr [C][h*s][w*s]. */
void sro_model_apply_transpose(const sro_model* m, int k, const double* lr,
                               int w, int h, int C, double* hr);

/* Regularizers.  kind: 0 = TV 2-D, 1 = TV 3-D, 2 = BTV. */
enum { SRO_REG_TV = 0, SRO_REG_TV3D = 1, SRO_REG_BTV = 2 };
typedef struct {
  int kind;
  int btv_range;     /* scale_range_ */
  double btv_decay;  /* spatial_decay_ */
} sro_regularizer;

/* Regularizer::ApplyToImage (tv_regularizer.cpp:110-132,
 * btv_regularizer.cpp:67-90). */
void sro_reg_values(const sro_regularizer* r, const double* x,
                    int W, int H, int C, double* values);
/* Regularizer::ApplyToImageWithDifferentiation (tv_regularizer.cpp:135-227,
 * btv_regularizer.cpp:93-170).  gradient is overwritten. */
void sro_reg_values_and_gradient(const sro_regularizer* r, const double* x,
                                 const double* gradient_constants,
                                 int W, int H, int C,
                                 double* values, double* gradient);

/* A MAP problem as MapSolver holds it (map_solver.cpp:52-86): observations
 * are stored NN-upsampled to HR size. */
typedef struct sro_problem sro_problem;
sro_problem* sro_problem_create(const sro_model* m, const double* lr_frames,
                                int w, int h, int C);
void sro_problem_destroy(sro_problem* p);
/* regularizers_ entry: (regularizer, lambda); weights may be NULL (= ones).
 * Returns the regularizer index.  Weights are copied. */
int sro_problem_add_regularizer(sro_problem* p, const sro_regularizer* r,
                                double lambda);
void sro_problem_set_irls_weights(sro_problem* p, int reg,
                                  const double* weights);
int sro_problem_hr_width(const sro_problem* p);
int sro_problem_hr_height(const sro_problem* p);

/* ObjectiveDataTerm::Compute (objective_data_term.cpp:15-116): gradient may
 * be NULL, otherwise it is accumulated into. */
double sro_data_term(const sro_problem* p, const double* x, double* gradient);
/* ObjectiveIRLSRegularizationTerm::Compute
 * (objective_irls_regularization_term.cpp:10-58). */
double sro_irls_reg_term(const sro_problem* p, int reg, const double* x,
                         double* gradient);
/* ObjectiveFunction::ComputeAllTerms (objective_function.cpp:5-20): zeroes
 * gradient, then data term + every regularizer term. */
double sro_objective(const sro_problem* p, const double* x, double* gradient);

/* ---- nonlinear CG: ALGLIB 3.10.0 mincg (libs/alglib/src/optimization.cpp
 * :17137-17880) + mcsrch (alglibinternal.cpp:12313-12632), restated ---- */
typedef double (*sro_fg_fn)(void* ctx, const double* x, double* g);
typedef void (*sro_rep_fn)(void* ctx, const double* x, double f);
typedef struct {
  int termination_type;
  int iterations;
  int nfev;
  double f;          /* state.f when the optimizer returns */
} sro_cg_report;
void sro_mincg(int n, double* x, double epsg, double epsf, double epsx,
               int maxits, sro_fg_fn fg, sro_rep_fn rep, void* ctx,
               sro_cg_report* report);

/* IRLSMapSolver::Solve (irls_map_solver.cpp:45-157, 192-265). */
typedef struct {
  int max_num_solver_iterations;        /* 50 */
  double gradient_norm_threshold;       /* 1e-6 */
  double cost_decrease_threshold;       /* 1e-6 */
  double parameter_variation_threshold; /* 1e-6 */
  int split_channels;                   /* 0 */
  int max_num_irls_iterations;          /* 20 */
  double irls_cost_difference_threshold;/* 1e-5 */
} sro_irls_options;
void sro_irls_options_default(sro_irls_options* o);
typedef struct {
  int irls_rounds;
  int cg_iterations;
  int nfev;
  double final_cost;
} sro_solve_report;
/* cg == NULL uses sro_mincg; tests may plug the real ALGLIB (oracle/_ref)
 * through a function with sro_mincg's signature. */
typedef void (*sro_cg_fn)(int n, double* x, double epsg, double epsf,
                          double epsx, int maxits, sro_fg_fn fg,
                          sro_rep_fn rep, void* ctx, sro_cg_report* report);
void sro_irls_solve(sro_problem* p, const sro_irls_options* o, const double* x0,
                    double* x_out, sro_cg_fn cg, sro_solve_report* report);

/* PeakSignalToNoiseRatioEvaluator::Evaluate
 * (src/evaluation/peak_signal_to_noise_ratio.cpp:11-54). */
double sro_psnr(const double* ground_truth, const double* image, long count);

#ifdef __cplusplus
}
#endif
#endif  /* SRMAP_ORACLE_H_ */
