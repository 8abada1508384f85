/*
 * srmap.h -- C ABI of libsrmap.so: the MI355X-native (HIP, gfx950) MAP
 * super-resolution gradient path.
 *
 * This is the drop-in boundary for ONE path of rteammco/super-resolution: the
 * per-iteration MAP cost + gradient (ObjectiveFunction::ComputeAllTerms) and
 * the operators and solver loop around it.  The reference has no FFI of its
 * own -- the path sits behind plain C++ virtual interfaces -- so every entry
 * point below names the reference interface it replaces (file:line relative to
 * the reference repository), and super-resolution_amd/host/ holds C++ classes
 * with the reference's names that forward here.
 *
 * Conventions
 *   - every function returns an srmap_status (0 = ok); nothing aborts or throws
 *     across the ABI; srmap_last_error() gives the message.  The reference's
 *     CHECK-class violations (glog abort) map to SRMAP_EINVAL.
 *   - images are planar [C][H][W], index c*W*H + row*W + col (util.cpp:81-89);
 *     LR stacks are [K][C][h][w].  Host buffers are IEEE double, owned by the
 *     caller.  Device buffers passed to *_device entry points hold the
 *     problem's dtype (double or float) and are owned by the caller.
 *   - one context = one GPU (one process per GPU); a problem is used from one
 *     host thread at a time.
 *   - there is no CPU fallback: without a usable HIP device every entry point
 *     that computes fails with SRMAP_EHIP.
 *
 * Streams (the ordering contract of every entry point that takes a DEVICE pointer)
 *   - a context owns one NON-BLOCKING HIP stream: it is not ordered against the
 *     legacy default stream or any other stream.  Every *_device entry point
 *     takes `hip_stream` (a hipStream_t; NULL = the context's stream) and
 *     enqueues ALL its work there, so a device buffer the caller produced on
 *     stream S is ordered by passing S -- or by completing S first.  Nothing the
 *     library enqueues runs on the legacy stream.
 *   - entry points that take HOST pointers (srmap_eval, srmap_solve*,
 *     srmap_cg_trace, srmap_apply*, srmap_reg_values*, srmap_set_observations,
 *     srmap_set_irls_weights, srmap_channel_map, srmap_channel_pca,
 *     srmap_register_translational, srmap_upload / srmap_download) run on the
 *     context's stream and are complete when they return.
 *   - the problem's device state (observations, IRLS weights) is ordered by the
 *     library itself: a write through srmap_update_irls_weights_device on one
 *     stream is waited for (an event) by evaluations on another, and a writer
 *     first drains the stream of the LAST evaluation when it is a different one.
 *     Only that one: a problem may have evaluations in flight on ONE stream at a
 *     time (it owns one set of cost partials and scratch buffers anyway) -- finish
 *     the evaluations on stream A (or order B after A) before evaluating the same
 *     problem on stream B.  srmap_set_observations_device is complete when it returns.
 *   - srmap_eval_device / srmap_eval_sharded_device with cost == NULL return
 *     right after enqueueing; x_dev / g_dev must stay alive and untouched by
 *     other streams until the stream reaches that point.
 *
 * Input domain
 *   - pixel values, observations and weights are finite IEEE numbers.  The tile
 *     kernels stage x multiplied by 2^512 (f64) / 2^64 (f32) (an exact scaling
 *     that turns the regulariser's sign() into a clamp):
 *     results equal the reference's for |x| < 2^508 (f64) / 2^60 (f32) and for
 *     pixel differences that are 0 or >= 2^-512 (f64) / 2^-64 (f32) in
 *     magnitude.  (Beyond 2^511 the reference's own cost, a sum of squared
 *     residuals, overflows.)  Inputs outside that range: force
 *     SRMAP_IMPL_DIRECT, which has no such scaling.
 */
#ifndef SRMAP_H_
#define SRMAP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srmap_ctx srmap_ctx;
typedef struct srmap_problem srmap_problem;

typedef enum {
  SRMAP_OK = 0,
  SRMAP_EINVAL = 1,       /* a reference CHECK would have fired */
  SRMAP_ENOMEM = 2,
  SRMAP_EHIP = 3,         /* HIP runtime / device error */
  SRMAP_EUNSUPPORTED = 4  /* valid for the reference, not representable here */
} srmap_status;

typedef enum { SRMAP_F64 = 0, SRMAP_F32 = 1 } srmap_dtype;

/* Regularizer kinds: TotalVariationRegularizer (tv_regularizer.h),
 * the same with SetUse3dTotalVariation(true), and
 * BilateralTotalVariationRegularizer (btv_regularizer.h). */
typedef enum { SRMAP_REG_TV = 0, SRMAP_REG_TV3D = 1, SRMAP_REG_BTV = 2 } srmap_reg_kind;

/* Which ObjectiveTerms an evaluation includes (objective_function.h:18-26). */
enum {
  SRMAP_TERM_DATA = 1u,  /* ObjectiveDataTerm */
  SRMAP_TERM_REG = 2u,   /* every ObjectiveIRLSRegularizationTerm */
  SRMAP_TERM_ALL = 3u
};

/* Kernel family selection (for tests and A/B measurements).  AUTO picks the
 * workgroup-tile kernels whenever the problem geometry admits them (scale
 * 2..4, blur size 1 or 3, integer or sub-pixel shifts), else the direct
 * kernels.  TILED forces that family and fails with SRMAP_EUNSUPPORTED when it
 * does not cover the problem.  (Value 3 was round 5's marching evaluation
 * kernel: bit-equal to the tiles, not faster on gfx950 -- profiles/r05_march.txt
 * -- and removed from the library in round 6; srmap_problem_set_impl answers
 * SRMAP_EINVAL for it.) */
typedef enum { SRMAP_IMPL_AUTO = 0, SRMAP_IMPL_DIRECT = 1, SRMAP_IMPL_TILED = 2 } srmap_impl;

/* ---------------------------------------------------------------- context */
/* Binds HIP device `device_id`.  Replaces nothing in the reference (it has no
 * device notion); it is what a maintainer creates once per process. */
int srmap_ctx_create(int device_id, srmap_ctx** out);
void srmap_ctx_destroy(srmap_ctx* ctx);
const char* srmap_last_error(const srmap_ctx* ctx);
/* Library/version string, e.g. "srmap 0.1 (gfx950)". */
const char* srmap_version(void);

/* ---------------------------------------------------------------- problem */
/* ImageModelParameters + MapSolver geometry: image_model.h:26-44,
 * ImageModel::CreateImageModel image_model.cpp:17-61, MapSolver::MapSolver
 * map_solver.cpp:52-86. */
typedef struct {
  int hr_width, hr_height;  /* HR image size (= LR size * scale when solving) */
  int channels;             /* C */
  int frames;               /* K observations / motion shifts */
  int scale;                /* DownsamplingModule scale >= 1 */
  const double* shifts_xy;  /* K x (dx, dy) MotionShift; NULL = no MotionModule */
  int blur_ksize;           /* BlurModule "blur_radius" = kernel size (odd);
                               0 (or sigma <= 0) = no BlurModule */
  double blur_sigma;
  int dtype;                /* srmap_dtype: arithmetic/storage type on device */
} srmap_problem_desc;

int srmap_problem_create(srmap_ctx* ctx, const srmap_problem_desc* desc,
                         srmap_problem** out);
void srmap_problem_destroy(srmap_problem* p);
/* Under frame sharding (srmap_eval_sharded_device / srmap_solve_sharded with SRMAP_SHARD_FRAMES) this call and the
 * regulariser calls (add / clear) are COLLECTIVE: every rank makes them in the same order between the same
 * evaluations -- the ranks agree by an all-reduce on how the regulariser is split whenever one of them changes. */
int srmap_problem_set_impl(srmap_problem* p, int impl /* srmap_impl */);
/* The family the next evaluation will run (SRMAP_IMPL_DIRECT / TILED): how a caller learns that AUTO fell
 * back to the direct kernels (geometry outside the tile kernels' coverage, or a sub-pixel shift on a 1/32-px
 * rounding tie, whose per-row table only the direct kernels read). */
int srmap_problem_active_impl(const srmap_problem* p, int* impl);

/* Row-band sharding (no reference counterpart: the reference is single-process).
 * A rank that owns HR rows [r0, r1) of a larger image creates its problem on the
 * band extended by halo rows and restricts the COST to the rows it owns:
 * regulariser pixels of HR rows [hr_row0, hr_row1) and data residuals of LR rows
 * [hr_row0/scale, hr_row1/scale) (rows relative to this problem; multiples of the
 * scale).  The gradient is always produced for every row; the caller keeps the
 * owned ones.  Default: the whole image. */
int srmap_problem_set_cost_rows(srmap_problem* p, int hr_row0, int hr_row1);

/* LR size the model produces: (int)(len * (1.0/scale)),
 * DownsamplingModule::ApplyToImage downsampling_module.cpp:19-27. */
int srmap_problem_lr_size(const srmap_problem* p, int* lr_width, int* lr_height);

/* MapSolver's low_res_images (map_solver.cpp:52-86): [K][C][h][w] doubles at LR
 * resolution.  (The reference stores them NN-upsampled to HR; this library
 * keeps LR and accounts for the s*s replication arithmetically.)  Requires
 * hr size == lr size * scale. */
int srmap_set_observations(srmap_problem* p, const double* lr_host);
/* Same from a device buffer holding the problem dtype, copied on hip_stream
 * (NULL = the context's stream); complete on return. */
int srmap_set_observations_device(srmap_problem* p, const void* lr_dev, void* hip_stream);

/* MapSolver::AddRegularizer(regularizer, regularization_parameter)
 * map_solver.cpp:88-94; constructors tv_regularizer.h / btv_regularizer.cpp
 * :50-65 (range >= 1, 0 < decay <= 1).  Gradients are bug-compatible with the
 * reference.  *reg_index receives the handle. */
int srmap_add_regularizer(srmap_problem* p, int kind, double lambda,
                          int btv_range, double btv_decay, int* reg_index);
int srmap_clear_regularizers(srmap_problem* p);
/* The irls_weights_ vector an ObjectiveIRLSRegularizationTerm holds
 * (objective_irls_regularization_term.h:40); NULL = all ones. */
int srmap_set_irls_weights(srmap_problem* p, int reg, const double* w_host);
/* w = 1 / max(1e-5, regularizer(x)) on device, irls_map_solver.cpp:128-143.
 * Enqueued on hip_stream (NULL = the context's stream) and NOT waited for:
 * later evaluations on any stream are ordered after it by the library. */
int srmap_update_irls_weights_device(srmap_problem* p, int reg, const void* x_dev, void* hip_stream);

/* ------------------------------------------------------- operators (host) */
/* ImageModel::ApplyToImage(ImageData*, index) image_model.cpp:86-91:
 * hr [C][H][W] -> lr [C][h][w]. */
int srmap_apply(srmap_problem* p, int frame, const double* hr, double* lr);
/* ImageModel::ApplyTransposeToImage image_model.cpp:93-101:
 * lr [C][h][w] -> hr [C][h*s][w*s]. */
int srmap_apply_transpose(srmap_problem* p, int frame, const double* lr, double* hr);
/* Regularizer::ApplyToImage regularizer.h:13-30 (values only). */
int srmap_reg_values(srmap_problem* p, int reg, const double* x, double* values);
/* Regularizer::ApplyToImageWithDifferentiation regularizer.h:32-45: values and
 * the gradient for the given per-pixel gradient_constants. */
int srmap_reg_values_and_gradient(srmap_problem* p, int reg, const double* x,
                                  const double* gradient_constants,
                                  double* values, double* gradient);

/* ---------------------------------------------------------- objective */
/* ObjectiveFunction::ComputeAllTerms(x, gradient) objective_function.cpp:5-20
 * restricted to `terms`: zeroes the gradient, then adds the selected terms.
 * grad may be NULL (cost only, objective_function.h:23). */
int srmap_eval(srmap_problem* p, unsigned terms, const double* x, double* cost,
               double* grad);
/* Device-resident form: x_dev / g_dev hold the problem dtype ([C][H][W]).
 * Work is enqueued on `hip_stream` (NULL = the context's stream, which is a
 * NON-BLOCKING stream: it does not order itself against the legacy default
 * stream, so buffers produced by other streams must be complete -- or pass the
 * producing stream here).  When
 * cost != NULL the call synchronises the stream and returns the cost; when
 * cost == NULL it returns right after enqueueing (the cost stays on device
 * until srmap_last_cost()). */
int srmap_eval_device(srmap_problem* p, unsigned terms, const void* x_dev,
                      void* g_dev, double* cost, void* hip_stream);
int srmap_last_cost(srmap_problem* p, double* cost);

/* Device memory helpers for C/C++ hosts that do not bring their own allocator
 * (element = the problem dtype). */
int srmap_device_alloc(srmap_ctx* ctx, size_t bytes, void** dev);
int srmap_device_free(srmap_ctx* ctx, void* dev);
int srmap_upload(srmap_problem* p, const double* host, void* dev, size_t count);
int srmap_download(srmap_problem* p, const void* dev, double* host, size_t count);
int srmap_synchronize(srmap_ctx* ctx);

/* ---------------------------------------------- spectral (channel) maps */
/* out[r][p] = sum_c M[r][c] * (in[c][p] - offset_in[c]) + offset_out[r] on planar
 * images in[rows_in][n], out[rows_out][n] (host, double): the per-pixel
 * projection of SpectralPCA::GetPCAImage / ReconstructImage
 * (spectral_pca.cpp:94-161: cv::PCA::project / backProject pixel by pixel),
 * done as one dense contraction on the GPU (rocBLAS DGEMM: this is the one place
 * on the path where the matrix cores apply).  M is rows_out x rows_in, row
 * major; the offsets may be NULL (zero). */
int srmap_channel_map(srmap_ctx* ctx, int rows_out, int rows_in, size_t n,
                      const double* M, const double* offset_in,
                      const double* offset_out, const double* in_host,
                      double* out_host);

/* The same on device-resident planar f64 cubes (in_dev [rows_in][n], out_dev [rows_out][n]); M and the offsets
 * are host arrays.  Enqueued on hip_stream (NULL = the context's stream); returns when the result is complete. */
int srmap_channel_map_device(srmap_ctx* ctx, int rows_out, int rows_in, size_t n,
                             const double* M, const double* offset_in,
                             const double* offset_out, const double* in_dev,
                             double* out_dev, void* hip_stream);
/* SpectralPCA training (spectral_pca.cpp:30-88 over cv::PCA): mean, covariance / count and its
 * eigen-decomposition of `count` spectral samples, on the GPU (row means, centred samples, the covariance as one
 * DGEMM, rocSOLVER dsyevd).  samples_host is planar [rows][count].  Outputs (host): mean[rows],
 * eigenvalues[rows] in descending order, basis[rows][rows] with row k = k-th eigenvector, its
 * largest-magnitude component positive. */
int srmap_channel_pca(srmap_ctx* ctx, int rows, size_t count, const double* samples_host,
                      double* mean_out, double* eigenvalues_out, double* basis_out);
/* The same on a device-resident planar f64 cube in_dev [rows][n]: the samples are the pixels
 * first + j * stride, j < count.  Enqueued on hip_stream (NULL = the context's stream); returns when the
 * (host) results are complete. */
int srmap_channel_pca_device(srmap_ctx* ctx, int rows, size_t n, const double* in_dev,
                             size_t first, size_t stride, size_t count, double* mean_out,
                             double* eigenvalues_out, double* basis_out, void* hip_stream);

/* ------------------------------------------------------- registration */
/* registration::TranslationalRegistration (src/motion/registration.h:19-22, registration.cpp:161-201): the
 * shift (dx, dy) of every image relative to the first one -- content at p in image 0 sits at p + (dx, dy) in
 * image i, the convention of MotionModule / MotionShift -- estimated on the GPU (box pyramid, exhaustive
 * integer search coarse to fine, Gauss-Newton sub-pixel refinement; see csrc/registration.hip for why this is
 * not the reference's OpenCV feature pipeline).  images_host: num_images planes [height][width] (the reference
 * registers on channel 0, registration.cpp:41-46).  shifts_xy_out: 2 * num_images doubles, image 0 -> (0, 0).
 * num_images == 0 returns SRMAP_OK and writes nothing (registration.cpp:165-168).  SRMAP_EINVAL when no shift
 * can be determined (the reference CHECK-fails, registration.cpp:193-194). */
int srmap_register_translational(srmap_ctx* ctx, int num_images, int width, int height,
                                 const double* images_host, double* shifts_xy_out);
/* What this estimator is NOT: the reference finds features (BRISK), fits a RANSAC homography / rigid transform and
 * keeps its translation, so it tolerates some rotation, scale and outliers.  This one assumes a PURE TRANSLATION
 * of at most a quarter of the frame (16 pixels of the coarsest pyramid level), has no outlier rejection, and its
 * sub-pixel step stays within +-1 px of the integer search.  On periodic texture it can lock onto a wrong period;
 * rotation / scale between frames bias the result.  The _ex form reports how trustworthy each shift is --
 * quality_out (optional, 2 doubles per image): [2i] separation = 1 - best / runner-up mean squared difference of
 * the coarsest search (runner-up at least 2 coarse pixels away; near 1 = one clear minimum, near 0 = ambiguous),
 * [2i + 1] the root mean squared residual at the returned shift.  Callers with real (non-synthetic) stacks
 * should check both, or supply shifts from their own registration (MotionShiftSequence accepts any). */
int srmap_register_translational_ex(srmap_ctx* ctx, int num_images, int width, int height,
                                    const double* images_host, double* shifts_xy_out, double* quality_out);

/* ------------------------------------------------------------- solver */
/* IRLSMapSolverOptions (irls_map_solver.h:14-36) + MapSolverOptions
 * (map_solver.h:28-79); srmap_irls_options_default() fills the reference
 * defaults.  Only CG with analytic differentiation is provided
 * (alglib_objective.cpp:47-75); L-BFGS / numeric differentiation are test-only
 * alternatives in the reference and are out of scope. */
typedef struct {
  int struct_size;                       /* sizeof(srmap_irls_options) of the header the caller was built with: filled by
                                            srmap_irls_options_default(); srmap_solve answers SRMAP_EINVAL when it is
                                            not this library's (a caller built against another version of this header
                                            would otherwise have its fields read at the wrong offsets) */
  int max_num_solver_iterations;         /* 50 */
  double gradient_norm_threshold;        /* 1e-6 */
  double cost_decrease_threshold;        /* 1e-6 */
  double parameter_variation_threshold;  /* 1e-6 */
  int split_channels;                    /* 0 */
  int max_num_irls_iterations;           /* 20 */
  double irls_cost_difference_threshold; /* 1e-5 */
  int host_paced_passes;                 /* 0.  No reference counterpart: 1 = every CG pass waits for the host's
                                            answer before the next is queued (the order up to round 3) instead of
                                            chaining the passes whose inputs are already on the device, and every
                                            trial point of the line search is formed by its own n-vector pass
                                            instead of inside the evaluation.  Same arithmetic, same result bit
                                            for bit: a debugging / measurement switch. */
} srmap_irls_options;
void srmap_irls_options_default(srmap_irls_options* o);

typedef struct {
  int irls_rounds;
  int cg_iterations;
  int evaluations;       /* cost+gradient evaluations ("MAP gradient iterations") */
  int last_termination;  /* ALGLIB-style code of the last CG run */
  double final_cost;
  double loop_seconds;   /* wall time of the IRLS / CG loop itself (device-resident part: no allocation,
                            no upload / download of x) */
  double wait_seconds;   /* part of loop_seconds the host spent waiting for device scalars */
  int waits;             /* number of such waits (2 + nfev per CG iteration) */
} srmap_solve_report;

/* IRLSMapSolver::Solve(initial_estimate) irls_map_solver.cpp:192-265:
 * x0 / x_out are [C][H][W] host doubles.  The iterate, gradient and CG vectors
 * stay on the GPU; only scalars cross PCIe per evaluation.  Passes whose inputs are
 * already on the device are queued without waiting for the host (un-sharded solves;
 * options->host_paced_passes = 1 restores the host-paced order).  The library reads no
 * environment variable. */
int srmap_solve(srmap_problem* p, const srmap_irls_options* options,
                const double* x0, double* x_out, srmap_solve_report* report);

/* Self-check of the solver's derived sums (no reference counterpart).  mincg forms the DY / HS betas with the
 * denominator y.dk summed over the vectors (optimization.cpp:17700-17760); this solver derives it from sums it already
 * holds, y.dk = (g.d) / (s1 s2) - g_prev.dk.  With srmap_irls_options::host_paced_passes = 1 the beta pass ALSO sums y.dk
 * directly; *beta_denominator_rel_dev receives the largest relative deviation |derived - direct| / |direct| seen by the
 * host-paced solves of this problem so far (0 if none ran).  Expected: reduction-order level (<= 1e-12 f64, <= 1e-6 f32). */
int srmap_problem_selfcheck(const srmap_problem* p, double* beta_denominator_rel_dev);

/* ------------------------------------------------- multi-GPU (one rank per GPU) */
/* The reference is single-process.  Its objective shards three ways (SURVEY.md
 * section 8e); a communicator carries the exchanges the sharded evaluation and the
 * sharded solve need, INSIDE the library, on the evaluation's HIP stream:
 *   - RCCL over xGMI (the production backend, ncclAllReduce / ncclSend / ncclRecv),
 *   - or caller-supplied host callbacks (MPI / gloo harnesses and the one-GPU
 *     tests: the library stages device buffers through pinned host memory). */
typedef struct srmap_comm srmap_comm;
#define SRMAP_UNIQUE_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it and hands the 128 bytes to every rank. */
int srmap_comm_get_unique_id(srmap_ctx* ctx, char* id128);
/* ncclCommInitRank on this context's device. */
int srmap_comm_create_rccl(srmap_ctx* ctx, const char* id128, int rank, int world,
                           srmap_comm** out);
/* op: 0 = sum, 1 = max.  dtype: srmap_dtype of the elements.  In place, all ranks. */
typedef int (*srmap_host_allreduce_fn)(void* buf, size_t count, int dtype, int op, void* user);
/* Send `send_bytes` to rank dst (skip if dst < 0) and receive `recv_bytes` from rank
 * src (skip if src < 0); must not deadlock when every rank calls it at once. */
typedef int (*srmap_host_sendrecv_fn)(const void* send, size_t send_bytes, int dst,
                                      void* recv, size_t recv_bytes, int src, void* user);
int srmap_comm_create_host(srmap_ctx* ctx, int rank, int world,
                           srmap_host_allreduce_fn allreduce,
                           srmap_host_sendrecv_fn sendrecv, void* user, srmap_comm** out);
void srmap_comm_destroy(srmap_comm* comm);
/* What the communicator itself reports: rank, size (ncclCommCount for RCCL), backend (1 = RCCL, 0 = host
 * callbacks).  Any out pointer may be NULL.  No reference counterpart (the reference is single-process). */
int srmap_comm_info(srmap_comm* comm, int* rank, int* world, int* backend);
/* Row shards: post the halo exchange of x on the communicator's side stream UNDER the tile rows that read no halo row
 * (on != 0), or exchange first and evaluate afterwards (on == 0).  Default: on for the host-callback backend (whose
 * callbacks block anyway), off for RCCL -- the overlapped form uses one ncclComm_t from two streams and is opt-in until
 * it has been validated on the deployment's RCCL.  No reference counterpart. */
int srmap_comm_set_overlap(srmap_comm* comm, int on);
/* Which collective library the communicator runs on, as text: "rccl <ncclGetVersion> <path of the loaded librccl>" or
 * "host callbacks" (a process may carry several librccl copies: PyTorch ships its own).  No reference counterpart. */
int srmap_comm_describe(srmap_comm* comm, char* buf, size_t cap);
/* ncclCommSplit: ranks passing the same color form a new communicator ordered by key; the caller states its rank
 * and size in it (RCCL backend; host-callback harnesses build the sub-communicator themselves).  Collective. */
int srmap_comm_split(srmap_comm* comm, int color, int key, int new_rank, int new_world, srmap_comm** out);
/* In-place all-reduce of a device buffer (op 0 = sum, 1 = max) on `hip_stream`: what the sharded evaluation
 * and solver use internally, exposed for harnesses. */
int srmap_comm_allreduce(srmap_comm* comm, void* dev_buf, size_t count, int dtype, int op,
                         void* hip_stream);

typedef enum {
  SRMAP_SHARD_NONE = 0,
  SRMAP_SHARD_FRAMES = 1,   /* rank owns a frame subset + a replica of x: all-reduce of the
                               gradient (C*N elements) and of the cost (one group) after every
                               evaluation (objective_data_term.cpp:98-116 summed over ranks).
                               The regulariser is split over the ranks by HR row band (whole
                               tile rows) when the tile kernels produce it in their one pass;
                               otherwise (3-D TV, a second regulariser, direct kernels) rank
                               reg_rank evaluates it */
  SRMAP_SHARD_ROWS = 2,     /* rank owns a band of HR rows; its problem is the band + halo
                               rows; halo rows of x are exchanged with the two neighbours
                               before every evaluation, scalars all-reduced */
  SRMAP_SHARD_CHANNELS = 3, /* rank owns a channel block (+ one halo channel plane per
                               neighbour when a 3-D TV regulariser couples them,
                               tv_regularizer.cpp:205-222); scalars all-reduced */
  SRMAP_SHARD_GRID = 4      /* frames x channels (BASELINE configs[4]): world rank =
                               channel_block * frame_groups + frame_group.  The rank owns a
                               channel block (as CHANNELS) and a frame subset (as FRAMES) of
                               it: the gradient of the block is all-reduced over the
                               frame_groups ranks that share the block (frame_comm), halo
                               planes travel between ranks rank +- frame_groups, the
                               regulariser terms are evaluated by frame group 0, scalars
                               are all-reduced over the world communicator */
} srmap_shard_mode;

typedef struct {
  int mode;                        /* srmap_shard_mode */
  int own_row0, own_row1;          /* ROWS: HR rows of THIS problem the rank owns (the rest is halo) */
  int send_up_rows, send_down_rows;/* ROWS: owned boundary rows the upper / lower neighbour's halo holds */
  int own_ch0, own_ch1;            /* CHANNELS: channels of THIS problem the rank owns (others: halo planes) */
  int reg_rank;                    /* FRAMES: the rank whose evaluation carries the regulariser terms when they
                                      cannot be split by row band (see SRMAP_SHARD_FRAMES) */
  int frame_groups;                /* GRID: ranks per channel block (0 / 1 elsewhere) */
  srmap_comm* frame_comm;          /* GRID: communicator of the frame_groups ranks sharing this rank's channel
                                      block (srmap_comm_split, or a host communicator of that group) */
} srmap_shard_desc;

/* One ObjectiveFunction::ComputeAllTerms of the JOINT objective on device buffers of
 * this rank's shard: halo exchange (ROWS / CHANNELS), the local evaluation, the
 * gradient all-reduce (FRAMES) and -- when cost != NULL -- the all-reduced cost.
 * comm == NULL or mode NONE: plain srmap_eval_device. */
int srmap_eval_sharded_device(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard,
                              unsigned terms, void* x_dev, void* g_dev, double* cost,
                              void* hip_stream);
/* IRLSMapSolver::Solve of the joint problem with the unknowns sharded as described:
 * every rank calls it with its shard of x0 and receives its shard of the result (halo
 * rows / planes of x_out are valid copies of the neighbours' values).  The CG scalars
 * (dot products, max-norm) are reduced over owned elements only and all-reduced, so
 * every rank follows the trajectory of the single-GPU solve up to reduction order. */
int srmap_solve_sharded(srmap_problem* p, srmap_comm* comm, const srmap_shard_desc* shard,
                        const srmap_irls_options* options, const double* x0, double* x_out,
                        srmap_solve_report* report);

/* One run of the nonlinear CG (alglib_objective.cpp:47-75 / mincg, no IRLS re-weighting) on
 * the problem's current objective, with the cost of EVERY evaluation recorded in order
 * (f_trace[0 .. min(*trace_len, trace_cap))): the trajectory the parity tests compare with
 * ALGLIB's.  epsg / epsf / epsx / maxits are mincgsetcond's arguments. */
int srmap_cg_trace(srmap_problem* p, double epsg, double epsf, double epsx, int maxits,
                   const double* x0, double* x_out, int* iterations, int* nfev,
                   int* termination, double* f_trace, int trace_cap, int* trace_len);

#ifdef __cplusplus
}
#endif
#endif  /* SRMAP_H_ */
