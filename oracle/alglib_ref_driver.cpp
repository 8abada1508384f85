// alglib_ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C entry point over the reference's vendored ALGLIB 3.10.0 nonlinear CG,
// compiled together with the ALGLIB sources WHERE THEY LIE under
// /root/reference/libs/alglib/src into oracle/_ref/libalglib_ref.so (see
// oracle/Makefile).  No reference source is copied into this repository.
//
// It drives mincg exactly the way the reference does in
// src/optimization/alglib_objective.cpp:47-75 (RunCGSolverAnalyticalDiff):
// mincgcreate / mincgsetcond / mincgsetxrep(true) / mincgoptimize /
// mincgresults, returning state.f.  The signature equals sro_mincg's
// (oracle/srmap_oracle.h) so tests can swap one for the other.
#include "optimization.h"

extern "C" {
typedef double (*sro_fg_fn)(void* ctx, const double* x, double* g);
typedef void (*sro_rep_fn)(void* ctx, const double* x, double f);
typedef struct {
  int termination_type;
  int iterations;
  int nfev;
  double f;
} sro_cg_report;
}

namespace {
struct Thunk {
  sro_fg_fn fg;
  sro_rep_fn rep;
  void* ctx;
};
void GradCallback(const alglib::real_1d_array& x, double& f,
                  alglib::real_1d_array& g, void* ptr) {
  Thunk* t = reinterpret_cast<Thunk*>(ptr);
  f = t->fg(t->ctx, x.getcontent(), g.getcontent());
}
void RepCallback(const alglib::real_1d_array& x, double f, void* ptr) {
  Thunk* t = reinterpret_cast<Thunk*>(ptr);
  if (t->rep) t->rep(t->ctx, x.getcontent(), f);
}
}  // namespace

extern "C" void ref_mincg(int n, double* x, double epsg, double epsf,
                          double epsx, int maxits, sro_fg_fn fg,
                          sro_rep_fn rep, void* ctx, sro_cg_report* report) {
  alglib::real_1d_array data;
  data.setcontent(n, x);
  alglib::mincgstate state;
  alglib::mincgreport rep_out;
  alglib::mincgcreate(data, state);
  alglib::mincgsetcond(state, epsg, epsf, epsx, maxits);
  alglib::mincgsetxrep(state, true);
  Thunk thunk = {fg, rep, ctx};
  alglib::mincgoptimize(state, GradCallback, RepCallback, &thunk);
  alglib::mincgresults(state, data, rep_out);
  for (int i = 0; i < n; ++i) x[i] = data[i];
  if (report) {
    report->termination_type = static_cast<int>(rep_out.terminationtype);
    report->iterations = static_cast<int>(rep_out.iterationscount);
    report->nfev = static_cast<int>(rep_out.nfev);
    report->f = state.f;
  }
}
