// mlf_walk_api.hip -- C ABI of the population step-sampler path (include/mlfriends_hip.h, section
// "population step sampler"): argument checks, device buffers, staging, kernel sequencing.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/mlfriends_hip.h"
#include "mlf_ctx.hpp"
#include "mlf_misc.hpp"
#include "mlf_sample.hpp"
#include "mlf_walk.hpp"

using namespace mlf;

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e_ = (x);                                                       \
    if (e_ != hipSuccess) return ctx_fail_hip(e_, #x, "mlf_walk_api.hip", __LINE__); \
  } while (0)

struct mlf_walkers {
  int P = 0, nsteps = 0, d = 0, nparams = 0;
  DevBuf allu, allL, generation, currentt, currentv, left, right, sl, sr, currentp;
  DevBuf unew, movable, acceptable, success, pnew, Lnew, dist2;
  DevBuf gmax, flags, snap, idx, rows, vals, vidx, vrows, unif, blk, compact, pc, Lc, rec, aux;
  DevBuf axes, live, std, lay_ctr, lay_mat, lay_wrap, liveL, ring, partials;
  bool have_liveL = false;
  int nlive = 0;
  bool have_axes = false, have_live = false, have_std = false;
  int layer_kind = -1;
  bool layer_wrap = false;
  double r2 = 1.0;
  unsigned nblk = 0;
  bool proposed = false, compacted = false;
  std::vector<uint8_t> host_snap;
  // whole-step hipGraph: one launch replays the ~12 kernels + the record copy; the values that change per
  // call travel through a pinned StepParams block
  hipGraphExec_t gexec = nullptr;
  StepParams *h_sp = nullptr;      // pinned
  double *h_rec = nullptr;         // pinned
  DevBuf d_sp;
  std::vector<unsigned long long> gkey;
  // several rounds per call (mlf_walkers_rounds_dev)
  DevBuf r_ctl, r_flags, r_dist2, r_out, r_sp, r_last, r_parts, live_stage;
  hipGraphExec_t rgexec = nullptr; // the launch sequence of mlf_walkers_rounds_dev (parameter copy, four kernels, record copy) as ONE graph launch
  std::vector<unsigned long long> rgkey;
  double *h_live = nullptr;        // pinned staging of mlf_walkers_update_live
  size_t h_live_bytes = 0;
  StepParams *h_rsp = nullptr;     // pinned
  double *h_rout = nullptr;        // pinned: record + per-round statistics
  size_t h_rout_doubles = 0;
};

namespace {

struct Scratch {
  DevBuf a, b, c, d, e, f, g, h, i, j, k, l, m;
};
Scratch g_s;

int upload(DevBuf &b, const void *host, size_t bytes, hipStream_t s) {
  CK(b.reserve(bytes ? bytes : 1));
  if (bytes) CK(hipMemcpyAsync(b.p, host, bytes, hipMemcpyHostToDevice, s));
  return 0;
}

int download(void *host, const DevBuf &b, size_t bytes, hipStream_t s) {
  if (bytes) CK(hipMemcpyAsync(host, b.p, bytes, hipMemcpyDeviceToHost, s));
  return 0;
}

int check_nd(size_t n, size_t d) {
  if (d == 0) return ctx_fail_arg(MLF_E_BADARG, "dimensionality must be positive");
  if (n > 0x7fffffffull / (d ? d : 1)) return ctx_fail_arg(MLF_E_BADARG, "population too large");
  return 0;
}

WalkState state_of(const mlf_walkers *w) {
  WalkState s{};
  s.P = w->P;
  s.G = w->nsteps + 1;
  s.d = w->d;
  s.nparams = w->nparams;
  s.allu = w->allu.as<double>();
  s.allL = w->allL.as<double>();
  s.generation = w->generation.as<long long>();
  s.currentt = w->currentt.as<double>();
  s.currentv = w->currentv.as<double>();
  s.left = w->left.as<double>();
  s.right = w->right.as<double>();
  s.sl = w->sl.as<uint8_t>();
  s.sr = w->sr.as<uint8_t>();
  s.currentp = w->currentp.as<double>();
  s.unew = w->unew.as<double>();
  s.movable = w->movable.as<uint8_t>();
  s.acceptable = w->acceptable.as<uint8_t>();
  s.success = w->success.as<uint8_t>();
  s.pnew = w->pnew.as<double>();
  s.Lnew = w->Lnew.as<double>();
  s.dist2 = w->dist2.as<double>();
  return s;
}

WalkLayer layer_of(const mlf_walkers *w) {
  WalkLayer l{};
  l.kind = w->layer_kind;
  l.ctr = w->lay_ctr.as<double>();
  l.mat = w->lay_mat.as<double>();
  l.wrap = w->layer_wrap ? w->lay_wrap.as<double>() : nullptr;
  l.r2 = w->r2;
  return l;
}

int ensure_params(mlf_walkers *w, size_t nparams) {
  if (nparams == 0) return ctx_fail_arg(MLF_E_BADARG, "nparams must be positive");
  if (w->nparams == (int)nparams) return 0;
  if (w->nparams != 0) return ctx_fail_arg(MLF_E_STATE, "number of transformed parameters changed between calls");
  w->nparams = (int)nparams;
  CK(w->currentp.reserve((size_t)w->P * nparams * sizeof(double)));
  CK(w->pnew.reserve((size_t)w->P * nparams * sizeof(double)));
  CK(hipMemsetAsync(w->currentp.p, 0xff, (size_t)w->P * nparams * sizeof(double), ctx_stream()));   // NaN
  return 0;
}

int finish_common(mlf_walkers *w, double Lmin, int64_t ringindex, double *rec) {
  hipStream_t s = ctx_stream();
  if (ringindex < 0 || ringindex >= w->P) return ctx_fail_arg(MLF_E_BADARG, "ringindex out of range");
  const size_t nrec = 9 + (size_t)w->d + (size_t)w->nparams;
  CK(w->rec.reserve(nrec * sizeof(double)));
  const WalkState st = state_of(w);
  launch_walk_update(st, Lmin, layer_of(w), s);
  launch_walk_harvest(st, ringindex, nullptr, w->r2, w->rec.as<double>(), w->partials.as<double>(), s);
  CK(hipGetLastError());
  if (int rc = download(rec, w->rec, nrec * sizeof(double), s)) return rc;
  CK(hipStreamSynchronize(s));
  w->proposed = false;
  return 0;
}

}  // namespace

extern "C" {

int mlf_walkers_create(mlf_walkers **out, size_t popsize, size_t nsteps, size_t d) {
  if (!out) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  *out = nullptr;
  if (popsize == 0 || nsteps == 0 || d == 0 || popsize > (1u << 24) || nsteps > 65535)
    return ctx_fail_arg(MLF_E_BADARG, "mlf_walkers_create: popsize, nsteps, d must be positive");
  if (d > 128) return ctx_fail_arg(MLF_E_DIM, "the resident walkers (one wave per walker, lane = coordinate pair) cover up to 128 dimensions");
  if (int rc = ctx_ensure()) return rc;
  mlf_walkers *w = new mlf_walkers();
  w->P = (int)popsize;
  w->nsteps = (int)nsteps;
  w->d = (int)d;
  const size_t P = popsize, G = nsteps + 1;
  struct {
    DevBuf *b;
    size_t bytes;
  } plan[] = {{&w->allu, P * G * d * 8}, {&w->allL, P * G * 8}, {&w->generation, P * 8}, {&w->currentt, P * 8},
              {&w->currentv, P * d * 8}, {&w->left, P * 8},     {&w->right, P * 8},      {&w->sl, P},
              {&w->sr, P},               {&w->unew, P * d * 8}, {&w->movable, P},        {&w->acceptable, P},
              {&w->success, P},          {&w->Lnew, P * 8},     {&w->dist2, P * 8},      {&w->gmax, 8},
              {&w->flags, P},            {&w->unif, P * 8},     {&w->blk, ((P + 255) / 256 + 1) * 4},
              {&w->compact, P * d * 8}, {&w->partials, ((P + 1023) / 1024) * 6 * 8}};
  for (auto &e : plan) {
    hipError_t err = e.b->reserve(e.bytes);
    if (err != hipSuccess) {
      mlf_walkers_destroy(w);
      return ctx_fail_hip(err, "device allocation for the walker population", "mlf_walk_api.hip", __LINE__);
    }
  }
  launch_walk_reset(state_of(w), ctx_stream());
  hipError_t err = hipStreamSynchronize(ctx_stream());
  if (err != hipSuccess) {
    mlf_walkers_destroy(w);
    return ctx_fail_hip(err, "walker reset", "mlf_walk_api.hip", __LINE__);
  }
  *out = w;
  return 0;
}

int mlf_walkers_destroy(mlf_walkers *w) {
  if (!w) return 0;
  DevBuf *all[] = {&w->allu, &w->allL, &w->generation, &w->currentt, &w->currentv, &w->left, &w->right, &w->sl,
                   &w->sr, &w->currentp, &w->unew, &w->movable, &w->acceptable, &w->success, &w->pnew, &w->Lnew,
                   &w->dist2, &w->gmax, &w->flags, &w->snap, &w->idx, &w->rows, &w->vals, &w->vidx, &w->vrows, &w->unif, &w->blk, &w->compact,
                   &w->pc, &w->Lc, &w->rec, &w->aux, &w->axes, &w->live, &w->std, &w->lay_ctr, &w->lay_mat,
                   &w->lay_wrap, &w->liveL, &w->ring, &w->partials};
  for (DevBuf *b : all) b->release();
  w->d_sp.release();
  for (DevBuf *b : {&w->r_ctl, &w->r_flags, &w->r_dist2, &w->r_out, &w->r_sp, &w->r_last, &w->r_parts, &w->live_stage}) b->release();
  if (w->h_live) (void)hipHostFree(w->h_live);
  if (w->rgexec) (void)hipGraphExecDestroy(w->rgexec);
  if (w->h_rsp) (void)hipHostFree(w->h_rsp);
  if (w->h_rout) (void)hipHostFree(w->h_rout);
  if (w->gexec) (void)hipGraphExecDestroy(w->gexec);
  if (w->h_sp) (void)hipHostFree(w->h_sp);
  if (w->h_rec) (void)hipHostFree(w->h_rec);
  delete w;
  return 0;
}

int mlf_walkers_reset(mlf_walkers *w) {
  if (!w) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  launch_walk_reset(state_of(w), ctx_stream());
  if (w->nparams) CK(hipMemsetAsync(w->currentp.p, 0xff, (size_t)w->P * w->nparams * sizeof(double), ctx_stream()));
  if (w->ring.p) CK(hipMemsetAsync(w->ring.p, 0, 8, ctx_stream()));
  CK(hipGetLastError());
  w->proposed = false;
  return 0;
}

int mlf_walkers_begin(mlf_walkers *w, double Lmin, int64_t *generation, uint8_t *flags) {
  if (!w || !generation || !flags) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  hipStream_t s = ctx_stream();
  const size_t P = (size_t)w->P;
  // snapshot = generation (8 P bytes) followed by the flags (P bytes): one device-to-host copy
  CK(w->snap.reserve(9 * P));
  uint8_t *d_flags = w->snap.as<uint8_t>() + 8 * P;
  launch_walk_step_back(state_of(w), Lmin, w->gmax.as<long long>(), d_flags, s);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(w->snap.p, w->generation.p, 8 * P, hipMemcpyDeviceToDevice, s));
  w->host_snap.resize(9 * P);
  CK(hipMemcpyAsync(w->host_snap.data(), w->snap.p, 9 * P, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  memcpy(generation, w->host_snap.data(), 8 * P);
  memcpy(flags, w->host_snap.data() + 8 * P, P);
  return 0;
}

int mlf_walkers_start(mlf_walkers *w, const int64_t *idx, size_t n, const double *u_rows, const double *L) {
  if (!w || (n && (!idx || !u_rows || !L))) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (n == 0) return 0;
  for (size_t j = 0; j < n; ++j)
    if (idx[j] < 0 || idx[j] >= w->P) return ctx_fail_arg(MLF_E_BADARG, "walker index out of range");
  hipStream_t s = ctx_stream();
  if (int rc = upload(w->idx, idx, n * 8, s)) return rc;
  if (int rc = upload(w->rows, u_rows, n * (size_t)w->d * 8, s)) return rc;
  if (int rc = upload(w->vals, L, n * 8, s)) return rc;
  launch_walk_start(state_of(w), w->idx.as<long long>(), (int)n, w->rows.as<double>(), w->vals.as<double>(), s);
  CK(hipGetLastError());
  return 0;
}

int mlf_walkers_points(mlf_walkers *w, const int64_t *idx, size_t n, double *out_rows) {
  if (!w || (n && (!idx || !out_rows))) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (n == 0) return 0;
  for (size_t j = 0; j < n; ++j)
    if (idx[j] < 0 || idx[j] >= w->P) return ctx_fail_arg(MLF_E_BADARG, "walker index out of range");
  hipStream_t s = ctx_stream();
  if (int rc = upload(w->idx, idx, n * 8, s)) return rc;
  CK(w->rows.reserve(n * (size_t)w->d * 8));
  launch_walk_points(state_of(w), w->idx.as<long long>(), (int)n, w->rows.as<double>(), s);
  CK(hipGetLastError());
  if (int rc = download(out_rows, w->rows, n * (size_t)w->d * 8, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_walkers_brackets(mlf_walkers *w, const int64_t *idx, size_t n, double scale, const double *v_rows) {
  if (!w || (n && (!idx || !v_rows))) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (n == 0) return 0;
  for (size_t j = 0; j < n; ++j)
    if (idx[j] < 0 || idx[j] >= w->P) return ctx_fail_arg(MLF_E_BADARG, "walker index out of range");
  hipStream_t s = ctx_stream();
  if (int rc = upload(w->vidx, idx, n * 8, s)) return rc;
  if (int rc = upload(w->vrows, v_rows, n * (size_t)w->d * 8, s)) return rc;
  launch_walk_brackets(state_of(w), w->vidx.as<long long>(), (int)n, scale, w->vrows.as<double>(), s);
  CK(hipGetLastError());
  return 0;
}

int mlf_walkers_set_direction_data(mlf_walkers *w, const double *axes, const double *live, size_t nlive,
                                   const double *std) {
  if (!w) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  hipStream_t s = ctx_stream();
  const size_t d = (size_t)w->d;
  if (axes) {
    if (int rc = upload(w->axes, axes, d * d * 8, s)) return rc;
    w->have_axes = true;
  }
  if (live) {
    if (nlive < 2) return ctx_fail_arg(MLF_E_BADARG, "differential directions need at least two live points");
    if (int rc = upload(w->live, live, nlive * d * 8, s)) return rc;
    w->nlive = (int)nlive;
    w->have_live = true;
  }
  if (std) {
    if (int rc = upload(w->std, std, d * 8, s)) return rc;
    w->have_std = true;
  }
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_walkers_brackets_philox(mlf_walkers *w, double scale, int kind, double dirscale, uint64_t seed,
                                uint64_t offset, uint64_t *next_offset) {
  if (!w || !next_offset) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (kind < 0 || kind > DIR_MIXTURE) return ctx_fail_arg(MLF_E_BADARG, "unknown direction kind");
  const bool need_axes = kind == DIR_REGION_ORIENTED || kind == DIR_REGION_RANDOM || kind == DIR_MIXTURE;
  const bool need_live = kind == DIR_DIFFERENTIAL || kind == DIR_MIXTURE;
  if ((need_axes && !w->have_axes) || (need_live && !w->have_live) ||
      (kind == DIR_CUBE_ORIENTED_SCALED && !w->have_std))
    return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_direction_data has not provided what this direction kind needs");
  WalkDirData dd{};
  dd.axes = w->axes.as<double>();
  dd.live = w->live.as<double>();
  dd.nlive = w->nlive;
  dd.std = w->std.as<double>();
  launch_walk_brackets_philox(state_of(w), scale, kind, dirscale, dd, seed, offset, ctx_stream());
  CK(hipGetLastError());
  *next_offset = offset + (uint64_t)w->P * (uint64_t)((w->d + 1) / 2 + 2);
  return 0;
}

int mlf_walkers_set_layer(mlf_walkers *w, int kind, const double *ctr, const double *mat, const double *wrap,
                          double maxradiussq) {
  if (!w) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (kind < 0) {
    w->layer_kind = -1;
    return 0;
  }
  if (kind > 1 || !ctr || !mat) return ctx_fail_arg(MLF_E_BADARG, "layer kind must be 0 (affine) or 1 (scaling)");
  hipStream_t s = ctx_stream();
  const size_t d = (size_t)w->d;
  if (int rc = upload(w->lay_ctr, ctr, d * 8, s)) return rc;
  if (int rc = upload(w->lay_mat, mat, (kind == 0 ? d * d : d) * 8, s)) return rc;
  w->layer_wrap = wrap != nullptr;
  if (wrap)
    if (int rc = upload(w->lay_wrap, wrap, d * 8, s)) return rc;
  CK(hipStreamSynchronize(s));
  w->layer_kind = kind;
  w->r2 = maxradiussq;
  return 0;
}

int mlf_walkers_propose(mlf_walkers *w, const double *unif, uint64_t seed, uint64_t offset, double *unew_out,
                        size_t *nacc) {
  if (!w) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if ((unew_out == nullptr) != (nacc == nullptr)) return ctx_fail_arg(MLF_E_BADARG, "unew_out and nacc go together");
  hipStream_t s = ctx_stream();
  const WalkState st = state_of(w);
  const double *d_unif = nullptr;
  if (unif) {
    if (int rc = upload(w->unif, unif, (size_t)w->P * 8, s)) return rc;
    d_unif = w->unif.as<double>();
  }
  launch_walk_propose(st, d_unif, seed, offset, s);
  CK(hipGetLastError());
  w->proposed = true;
  w->compacted = false;
  if (!nacc) return 0;
  // host likelihood: hand back the acceptable rows in walker order
  launch_compact(st.unew, st.acceptable, w->P, w->d, w->blk.as<unsigned>(), w->compact.as<double>(),
                 (unsigned)w->P, s);
  CK(hipGetLastError());
  w->nblk = (unsigned)((w->P + 255) / 256);
  unsigned count = 0;
  CK(hipMemcpyAsync(&count, w->blk.as<unsigned>() + w->nblk, sizeof count, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  if (count) {
    if (int rc = download(unew_out, w->compact, (size_t)count * w->d * 8, s)) return rc;
    CK(hipStreamSynchronize(s));
  }
  *nacc = count;
  w->compacted = true;
  return 0;
}

int mlf_walkers_finish(mlf_walkers *w, double Lmin, const double *pnew, const double *Lnew, size_t nacc,
                       size_t nparams, int64_t ringindex, double *rec) {
  if (!w || !rec || (nacc && (!pnew || !Lnew))) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (!w->proposed || !w->compacted)
    return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_finish without a preceding mlf_walkers_propose(unew_out != NULL)");
  if (int rc = ensure_params(w, nparams)) return rc;
  hipStream_t s = ctx_stream();
  if (nacc) {
    if (int rc = upload(w->pc, pnew, nacc * nparams * 8, s)) return rc;
    if (int rc = upload(w->Lc, Lnew, nacc * 8, s)) return rc;
    launch_walk_expand(state_of(w), w->blk.as<unsigned>(), w->pc.as<double>(), w->Lc.as<double>(), s);
    CK(hipGetLastError());
  }
  return finish_common(w, Lmin, ringindex, rec);
}

int mlf_walkers_finish_dev(mlf_walkers *w, double Lmin, int tkind, double ta, double tb, int lkind,
                           const double *aux, double sigma, int64_t ringindex, double *rec) {
  if (!w || !rec) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (!w->proposed) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_finish_dev without a preceding mlf_walkers_propose");
  if (tkind < 0 || tkind > 2 || lkind < 0 || lkind > 3) return ctx_fail_arg(MLF_E_BADARG, "unknown transform / likelihood kind");
  if (lkind == 0 && !aux) return ctx_fail_arg(MLF_E_BADARG, "the Gaussian likelihood needs its centres");
  if (int rc = ensure_params(w, (size_t)w->d)) return rc;
  hipStream_t s = ctx_stream();
  const WalkState st = state_of(w);
  if (aux)
    if (int rc = upload(w->aux, aux, (size_t)w->d * 8, s)) return rc;
  launch_walk_transform(st, tkind, ta, tb, s);
  launch_loglike(lkind, st.pnew, w->d, w->P, w->aux.as<double>(), sigma, st.Lnew, s);
  CK(hipGetLastError());
  return finish_common(w, Lmin, ringindex, rec);
}

int mlf_walkers_set_live(mlf_walkers *w, const double *us, const double *Ls, size_t nlive) {
  if (!w || !us || !Ls) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (nlive < 2) return ctx_fail_arg(MLF_E_BADARG, "at least two live points are needed");
  hipStream_t s = ctx_stream();
  if (int rc = upload(w->live, us, nlive * (size_t)w->d * 8, s)) return rc;
  if (int rc = upload(w->liveL, Ls, nlive * 8, s)) return rc;
  w->nlive = (int)nlive;
  w->have_live = true;
  w->have_liveL = true;
  return 0;
}

int mlf_walkers_update_live(mlf_walkers *w, const int64_t *rows, size_t count, const double *us_rows, const double *Ls_rows) {
  if (!w || (count && (!rows || !us_rows || !Ls_rows))) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (!w->have_liveL) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_live not called");
  if (count == 0) return 0;
  const size_t d = (size_t)w->d;
  for (size_t j = 0; j < count; ++j)
    if (rows[j] < 0 || rows[j] >= w->nlive) return ctx_fail_arg(MLF_E_BADARG, "live point index out of range");
  hipStream_t s = ctx_stream();
  // pinned staging, reused from call to call: every path that reads the device copy ends with a synchronisation of the
  // library's stream, so the copy queued by the previous update has long been made
  const size_t need = count * (d + 2) * sizeof(double);
  if (w->h_live_bytes < need) {
    if (w->h_live) CK(hipHostFree(w->h_live));
    w->h_live = nullptr;
    w->h_live_bytes = 0;
    const size_t cap = need < 4096 ? 4096 : need;
    CK(hipHostMalloc(reinterpret_cast<void **>(&w->h_live), cap, hipHostMallocDefault));
    w->h_live_bytes = cap;
  }
  CK(w->live_stage.reserve(w->h_live_bytes));
  double *hs = w->h_live;
  memcpy(hs, us_rows, count * d * sizeof(double));
  memcpy(hs + count * d, Ls_rows, count * sizeof(double));
  memcpy(hs + count * (d + 1), rows, count * sizeof(int64_t));
  CK(hipMemcpyAsync(w->live_stage.p, hs, need, hipMemcpyHostToDevice, s));
  const double *ds = w->live_stage.as<double>();
  launch_walk_scatter_live(ds, ds + count * d, reinterpret_cast<const long long *>(ds + count * (d + 1)), (int)count, (int)d,
                           w->live.as<double>(), w->liveL.as<double>(), s);
  CK(hipGetLastError());
  return 0;
}

int mlf_walkers_step_dev(mlf_walkers *w, double Lmin, double scale, int dirkind, double dirscale, uint64_t seed,
                         uint64_t offset, int tkind, double ta, double tb, int lkind, const double *aux, double sigma,
                         double *rec, uint64_t *next_offset) {
  if (!w || !rec || !next_offset) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (!w->have_liveL) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_live not called");
  if (dirkind < 0 || dirkind > DIR_MIXTURE) return ctx_fail_arg(MLF_E_BADARG, "unknown direction kind");
  if (tkind < 0 || tkind > 2 || lkind < 0 || lkind > 3) return ctx_fail_arg(MLF_E_BADARG, "unknown transform / likelihood kind");
  if (lkind == 0 && !aux) return ctx_fail_arg(MLF_E_BADARG, "the Gaussian likelihood needs its centres");
  const bool need_axes = dirkind == DIR_REGION_ORIENTED || dirkind == DIR_REGION_RANDOM || dirkind == DIR_MIXTURE;
  if ((need_axes && !w->have_axes) || (dirkind == DIR_CUBE_ORIENTED_SCALED && !w->have_std))
    return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_direction_data has not provided what this direction kind needs");
  if (int rc = ensure_params(w, (size_t)w->d)) return rc;
  hipStream_t s = ctx_stream();
  if (!w->ring.p) {
    CK(w->ring.reserve(8));
    CK(hipMemsetAsync(w->ring.p, 0, 8, s));
  }
  const size_t nrec = 10 + 2 * (size_t)w->d;
  CK(w->rec.reserve(nrec * sizeof(double)));
  if (aux)
    if (int rc = upload(w->aux, aux, (size_t)w->d * 8, s)) return rc;
  const WalkState st = state_of(w);
  WalkDirData dd{};
  dd.axes = w->axes.as<double>();
  dd.live = w->live.as<double>();
  dd.nlive = w->nlive;
  dd.std = w->std.as<double>();
  // one stream of kernels, one record back: step_back, restarts, new slices, proposal, likelihood, update, harvest
  StepParams p{};
  p.Lmin = Lmin;
  p.scale = scale;
  p.dirscale = dirscale;
  p.r2 = w->r2;
  p.seed = seed;
  p.offset = offset;
  launch_walk_prologue(st, w->live.as<double>(), w->liveL.as<double>(), w->nlive, dirkind, dd, tkind, ta, tb,
                       w->flags.as<uint8_t>(), p, nullptr, s);
  launch_loglike(lkind, st.pnew, w->d, w->P, w->aux.as<double>(), sigma, st.Lnew, s);
  launch_walk_update(st, Lmin, layer_of(w), s);
  launch_walk_harvest(st, 0, w->ring.as<long long>(), w->r2, w->rec.as<double>(), w->partials.as<double>(), s, nullptr,
                      w->flags.as<uint8_t>());
  CK(hipGetLastError());
  if (int rc = download(rec, w->rec, nrec * sizeof(double), s)) return rc;
  CK(hipStreamSynchronize(s));
  const uint64_t per = (uint64_t)((w->d + 1) / 2 + 2);
  *next_offset = offset + (uint64_t)w->P * (per > 64 ? per : 64);
  return 0;
}

int mlf_walkers_step_graph(mlf_walkers *w, double Lmin, double scale, int dirkind, double dirscale, uint64_t seed,
                           uint64_t offset, int tkind, double ta, double tb, int lkind, const double *aux, double sigma,
                           double *rec, uint64_t *next_offset) {
  if (!w || !rec || !next_offset) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (!w->have_liveL) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_live not called");
  if (dirkind < 0 || dirkind > DIR_MIXTURE) return ctx_fail_arg(MLF_E_BADARG, "unknown direction kind");
  if (tkind < 0 || tkind > 2 || lkind < 0 || lkind > 3) return ctx_fail_arg(MLF_E_BADARG, "unknown transform / likelihood kind");
  if (lkind == 0 && !aux) return ctx_fail_arg(MLF_E_BADARG, "the Gaussian likelihood needs its centres");
  const bool need_axes = dirkind == DIR_REGION_ORIENTED || dirkind == DIR_REGION_RANDOM || dirkind == DIR_MIXTURE;
  if ((need_axes && !w->have_axes) || (dirkind == DIR_CUBE_ORIENTED_SCALED && !w->have_std))
    return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_direction_data has not provided what this direction kind needs");
  if (int rc = ensure_params(w, (size_t)w->d)) return rc;
  hipStream_t s = ctx_stream();
  const size_t nrec = 10 + 2 * (size_t)w->d;
  if (!w->ring.p) {
    CK(w->ring.reserve(8));
    CK(hipMemsetAsync(w->ring.p, 0, 8, s));
  }
  if (!w->h_sp) {
    CK(hipHostMalloc(reinterpret_cast<void **>(&w->h_sp), sizeof(StepParams), hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void **>(&w->h_rec), nrec * sizeof(double), hipHostMallocDefault));
    CK(w->d_sp.reserve(sizeof(StepParams)));
  }
  CK(w->rec.reserve(nrec * sizeof(double)));
  CK(w->aux.reserve((size_t)w->d * 8));
  if (aux)
    if (int rc = upload(w->aux, aux, (size_t)w->d * 8, s)) return rc;
  // everything a captured kernel argument depends on: a change means a new capture
  auto bits = [](double v) {
    unsigned long long u;
    memcpy(&u, &v, sizeof u);
    return u;
  };
  auto addr = [](const void *p) { return (unsigned long long)(uintptr_t)p; };
  std::vector<unsigned long long> key = {
      (unsigned long long)dirkind, (unsigned long long)tkind, bits(ta), bits(tb), (unsigned long long)lkind, bits(sigma),
      (unsigned long long)(w->layer_kind + 1), (unsigned long long)w->layer_wrap, (unsigned long long)w->nlive,
      addr(w->live.p), addr(w->liveL.p), addr(w->axes.p), addr(w->std.p), addr(w->lay_ctr.p), addr(w->lay_mat.p),
      addr(w->lay_wrap.p), addr(w->aux.p), addr(w->rec.p), addr(w->pnew.p), addr(w->currentp.p)};
  if (!w->gexec || key != w->gkey) {
    if (w->gexec) {
      CK(hipGraphExecDestroy(w->gexec));
      w->gexec = nullptr;
    }
    const WalkState st = state_of(w);
    WalkDirData dd{};
    dd.axes = w->axes.as<double>();
    dd.live = w->live.as<double>();
    dd.nlive = w->nlive;
    dd.std = w->std.as<double>();
    const StepParams *sp = w->d_sp.as<StepParams>();
    CK(hipStreamSynchronize(s));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    (void)hipMemcpyAsync(w->d_sp.p, w->h_sp, sizeof(StepParams), hipMemcpyHostToDevice, s);
    launch_walk_prologue(st, w->live.as<double>(), w->liveL.as<double>(), w->nlive, dirkind, dd, tkind, ta, tb,
                         w->flags.as<uint8_t>(), StepParams{}, sp, s);
    launch_loglike(lkind, st.pnew, w->d, w->P, w->aux.as<double>(), sigma, st.Lnew, s);
    launch_walk_update(st, 0.0, layer_of(w), s, sp);
    launch_walk_harvest(st, 0, w->ring.as<long long>(), 0.0, w->rec.as<double>(), w->partials.as<double>(), s, sp,
                        w->flags.as<uint8_t>());
    (void)hipMemcpyAsync(w->h_rec, w->rec.p, nrec * sizeof(double), hipMemcpyDeviceToHost, s);
    hipGraph_t graph = nullptr;
    CK(hipStreamEndCapture(s, &graph));
    hipError_t e = hipGraphInstantiate(&w->gexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return ctx_fail_hip(e, "hipGraphInstantiate", "mlf_walk_api.hip", __LINE__);
    w->gkey = key;
  }
  w->h_sp->Lmin = Lmin;
  w->h_sp->scale = scale;
  w->h_sp->dirscale = dirscale;
  w->h_sp->r2 = w->r2;
  w->h_sp->seed = seed;
  w->h_sp->offset = offset;
  CK(hipGraphLaunch(w->gexec, s));
  CK(hipStreamSynchronize(s));
  memcpy(rec, w->h_rec, nrec * sizeof(double));
  const uint64_t per = (uint64_t)((w->d + 1) / 2 + 2);
  *next_offset = offset + (uint64_t)w->P * (per > 64 ? per : 64);
  return 0;
}

int mlf_walkers_rounds_dev(mlf_walkers *w, double Lmin, double scale, int dirkind, double dirscale, uint64_t seed,
                           uint64_t offset, int tkind, double ta, double tb, int lkind, const double *aux, double sigma,
                           int max_rounds, double *rec, double *round_rows, int *rounds, uint64_t *next_offset) {
  if (!w || !rec || !round_rows || !rounds || !next_offset) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (!w->have_liveL) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_live not called");
  if (dirkind < 0 || dirkind > DIR_MIXTURE) return ctx_fail_arg(MLF_E_BADARG, "unknown direction kind");
  if (tkind < 0 || tkind > 2 || lkind < 0 || lkind > 3) return ctx_fail_arg(MLF_E_BADARG, "unknown transform / likelihood kind");
  if (lkind == 0 && !aux) return ctx_fail_arg(MLF_E_BADARG, "the Gaussian likelihood needs its centres");
  const int force_memory_form = max_rounds < 0;   // test hook: every round through global memory (the first form of this path)
  if (force_memory_form) max_rounds = -max_rounds;
  if (max_rounds < 1) return ctx_fail_arg(MLF_E_BADARG, "max_rounds must not be 0");
  const bool need_axes = dirkind == DIR_REGION_ORIENTED || dirkind == DIR_REGION_RANDOM || dirkind == DIR_MIXTURE;
  if ((need_axes && !w->have_axes) || (dirkind == DIR_CUBE_ORIENTED_SCALED && !w->have_std))
    return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_set_direction_data has not provided what this direction kind needs");
  if (int rc = ensure_params(w, (size_t)w->d)) return rc;
  // the per-round flag / distance arrays are [max_rounds][P]: keep them within 64 MiB
  const size_t per_round = (size_t)w->P * 9;
  const size_t cap = ((size_t)64 << 20) / per_round;
  if ((size_t)max_rounds > cap) max_rounds = cap < 1 ? 1 : (int)cap;
  if (max_rounds > 4096) max_rounds = 4096;
  hipStream_t s = ctx_stream();
  if (!w->ring.p) {
    CK(w->ring.reserve(8));
    CK(hipMemsetAsync(w->ring.p, 0, 8, s));
  }
  const size_t nrec = 10 + 2 * (size_t)w->d;
  const size_t nout = nrec + 5 * (size_t)max_rounds;
  if (!w->h_rsp) CK(hipHostMalloc(reinterpret_cast<void **>(&w->h_rsp), sizeof(StepParams), hipHostMallocDefault));
  if (w->h_rout_doubles < nout) {
    if (w->h_rout) CK(hipHostFree(w->h_rout));
    w->h_rout = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&w->h_rout), nout * sizeof(double), hipHostMallocDefault));
    w->h_rout_doubles = nout;
  }
  CK(w->r_sp.reserve(sizeof(StepParams)));
  CK(w->r_ctl.reserve(8 * sizeof(int)));
  CK(w->r_flags.reserve((size_t)max_rounds * w->P));
  CK(w->r_dist2.reserve((size_t)max_rounds * w->P * sizeof(double)));
  CK(w->r_out.reserve(nout * sizeof(double)));
  CK(w->r_last.reserve((size_t)w->P * sizeof(int)));
  const size_t nchunks = ((size_t)w->P + 1023) / 1024;
  if (nchunks > 1) CK(w->r_parts.reserve((size_t)max_rounds * nchunks * 5 * sizeof(double)));
  CK(w->aux.reserve((size_t)w->d * 8));
  if (aux)
    if (int rc = upload(w->aux, aux, (size_t)w->d * 8, s)) return rc;
  w->h_rsp->Lmin = Lmin;
  w->h_rsp->scale = scale;
  w->h_rsp->dirscale = dirscale;
  w->h_rsp->r2 = w->r2;
  w->h_rsp->seed = seed;
  w->h_rsp->offset = offset;
  const uint64_t per = (uint64_t)((w->d + 1) / 2 + 2);
  RoundsArgs a{};
  a.w = state_of(w);
  a.live = w->live.as<double>();
  a.Ls = w->liveL.as<double>();
  a.nlive = w->nlive;
  a.dirkind = dirkind;
  a.dd.axes = w->axes.as<double>();
  a.dd.live = w->live.as<double>();
  a.dd.nlive = w->nlive;
  a.dd.std = w->std.as<double>();
  a.tkind = tkind;
  a.ta = ta;
  a.tb = tb;
  a.lkind = lkind;
  a.aux = w->aux.as<double>();
  a.sigma = sigma;
  a.ly = layer_of(w);
  a.was_starting = w->flags.as<uint8_t>();
  a.sp = w->r_sp.as<StepParams>();
  a.ring = w->ring.as<long long>();
  a.ctl = w->r_ctl.as<int>();
  a.rflags = w->r_flags.as<uint8_t>();
  a.rdist2 = w->r_dist2.as<double>();
  a.rlast = w->r_last.as<int>();
  a.rparts = nchunks > 1 ? w->r_parts.as<double>() : nullptr;
  a.force_memory_form = force_memory_form;
  a.rec = w->r_out.as<double>();
  a.rows = w->r_out.as<double>() + nrec;
  a.max_rounds = max_rounds;
  a.per_call = (unsigned long long)w->P * (per > 64 ? per : 64);
  // Everything a captured argument depends on; the values that change from call to call (threshold, scale, radius, seed,
  // offset) travel through the pinned parameter block.  A change (new region: layer buffers, radius-independent) means a new capture
  auto bits = [](double v) {
    unsigned long long u;
    memcpy(&u, &v, sizeof u);
    return u;
  };
  auto addr = [](const void *p) { return (unsigned long long)(uintptr_t)p; };
  std::vector<unsigned long long> key = {
      (unsigned long long)dirkind, (unsigned long long)tkind, bits(ta), bits(tb), (unsigned long long)lkind, bits(sigma),
      (unsigned long long)(w->layer_kind + 1), (unsigned long long)w->layer_wrap, (unsigned long long)w->nlive,
      (unsigned long long)max_rounds, (unsigned long long)force_memory_form, (unsigned long long)nout, bits(w->r2),
      addr(w->live.p), addr(w->liveL.p), addr(w->axes.p), addr(w->std.p), addr(w->lay_ctr.p), addr(w->lay_mat.p),
      addr(w->lay_wrap.p), addr(w->aux.p), addr(w->r_out.p), addr(w->pnew.p), addr(w->currentp.p), addr(w->r_flags.p),
      addr(w->r_dist2.p), addr(w->r_last.p), addr(w->r_parts.p), addr(w->r_ctl.p), addr(w->r_sp.p), addr(w->h_rout), addr(w->h_rsp), addr(w->flags.p),
      addr(w->ring.p)};
  if (!w->rgexec || key != w->rgkey) {
    if (w->rgexec) {
      CK(hipGraphExecDestroy(w->rgexec));
      w->rgexec = nullptr;
    }
    CK(hipStreamSynchronize(s));
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    (void)hipMemcpyAsync(w->r_sp.p, w->h_rsp, sizeof(StepParams), hipMemcpyHostToDevice, s);
    launch_walk_rounds(a, s);
    (void)hipMemcpyAsync(w->h_rout, w->r_out.p, nout * sizeof(double), hipMemcpyDeviceToHost, s);
    hipGraph_t graph = nullptr;
    CK(hipStreamEndCapture(s, &graph));
    hipError_t e = hipGraphInstantiate(&w->rgexec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return ctx_fail_hip(e, "hipGraphInstantiate", "mlf_walk_api.hip", __LINE__);
    w->rgkey = key;
  }
  CK(hipGraphLaunch(w->rgexec, s));
  CK(hipStreamSynchronize(s));
  const int R = (int)w->h_rout[4];
  if (w->h_rout[5] != 0.0) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_rounds_dev: a walker gave up waiting for the ring walker's rounds");
  if (R < 1 || R > max_rounds) return ctx_fail_arg(MLF_E_STATE, "mlf_walkers_rounds_dev: the device reported an impossible round count");
  memcpy(rec, w->h_rout, nrec * sizeof(double));
  memcpy(round_rows, w->h_rout + nrec, (size_t)R * 5 * sizeof(double));
  *rounds = R;
  *next_offset = offset + (uint64_t)R * a.per_call;
  return 0;
}

int mlf_walkers_export(mlf_walkers *w, double *allu, double *allL, int64_t *generation, double *currentt,
                       double *currentv, double *left, double *right, uint8_t *sl, uint8_t *sr) {
  if (!w) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  hipStream_t s = ctx_stream();
  const size_t P = (size_t)w->P, G = (size_t)w->nsteps + 1, d = (size_t)w->d;
  if (allu) CK(hipMemcpyAsync(allu, w->allu.p, P * G * d * 8, hipMemcpyDeviceToHost, s));
  if (allL) CK(hipMemcpyAsync(allL, w->allL.p, P * G * 8, hipMemcpyDeviceToHost, s));
  if (generation) CK(hipMemcpyAsync(generation, w->generation.p, P * 8, hipMemcpyDeviceToHost, s));
  if (currentt) CK(hipMemcpyAsync(currentt, w->currentt.p, P * 8, hipMemcpyDeviceToHost, s));
  if (currentv) CK(hipMemcpyAsync(currentv, w->currentv.p, P * d * 8, hipMemcpyDeviceToHost, s));
  if (left) CK(hipMemcpyAsync(left, w->left.p, P * 8, hipMemcpyDeviceToHost, s));
  if (right) CK(hipMemcpyAsync(right, w->right.p, P * 8, hipMemcpyDeviceToHost, s));
  if (sl) CK(hipMemcpyAsync(sl, w->sl.p, P, hipMemcpyDeviceToHost, s));
  if (sr) CK(hipMemcpyAsync(sr, w->sr.p, P, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  return 0;
}

// ------------------------------------------------------------------ stateless forms ------------
int mlf_within_unit_cube(const double *u, size_t n, size_t d, uint8_t *out) {
  if (n == 0) return 0;
  if (!u || !out) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(n, d)) return rc;
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  if (int rc = upload(g_s.a, u, n * d * 8, s)) return rc;
  CK(g_s.b.reserve(n));
  launch_within_unit_cube(g_s.a.as<double>(), (int)n, (int)d, g_s.b.as<uint8_t>(), s);
  CK(hipGetLastError());
  if (int rc = download(out, g_s.b, n, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_evolve_propose(const double *currentu, const double *currentv, const double *left, const double *right,
                       const uint8_t *sl, const uint8_t *sr, const double *unif_full, double *currentt, size_t n,
                       size_t d, double *unew, uint8_t *acceptable) {
  if (n == 0) return 0;
  if (!currentu || !currentv || !left || !right || !sl || !sr || !currentt || !unew || !acceptable)
    return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(n, d)) return rc;
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  if (int rc = upload(g_s.a, currentu, n * d * 8, s)) return rc;
  if (int rc = upload(g_s.b, currentv, n * d * 8, s)) return rc;
  if (int rc = upload(g_s.c, left, n * 8, s)) return rc;
  if (int rc = upload(g_s.d, right, n * 8, s)) return rc;
  if (int rc = upload(g_s.e, sl, n, s)) return rc;
  if (int rc = upload(g_s.f, sr, n, s)) return rc;
  if (int rc = upload(g_s.g, currentt, n * 8, s)) return rc;
  if (unif_full) {
    if (int rc = upload(g_s.h, unif_full, n * 8, s)) return rc;
    launch_bisect_draw(g_s.c.as<double>(), g_s.d.as<double>(), g_s.e.as<uint8_t>(), g_s.f.as<uint8_t>(),
                       g_s.h.as<double>(), (int)n, g_s.g.as<double>(), s);
  }
  CK(g_s.i.reserve(n * d * 8));
  CK(g_s.j.reserve(n));
  launch_evolve_propose(g_s.a.as<double>(), g_s.b.as<double>(), g_s.c.as<double>(), g_s.d.as<double>(),
                        g_s.e.as<uint8_t>(), g_s.f.as<uint8_t>(), g_s.g.as<double>(), (int)n, (int)d,
                        g_s.i.as<double>(), s);
  launch_within_unit_cube(g_s.i.as<double>(), (int)n, (int)d, g_s.j.as<uint8_t>(), s);
  CK(hipGetLastError());
  if (int rc = download(currentt, g_s.g, n * 8, s)) return rc;
  if (int rc = download(unew, g_s.i, n * d * 8, s)) return rc;
  if (int rc = download(acceptable, g_s.j, n, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_evolve_update(const uint8_t *acceptable, const double *Lnew_full, double Lmin, double *currentt, double *left,
                      double *right, uint8_t *sl, uint8_t *sr, uint8_t *success, size_t n) {
  if (n == 0) return 0;
  if (!acceptable || !Lnew_full || !currentt || !left || !right || !sl || !sr || !success)
    return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(n, 1)) return rc;
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  if (int rc = upload(g_s.a, acceptable, n, s)) return rc;
  if (int rc = upload(g_s.b, Lnew_full, n * 8, s)) return rc;
  if (int rc = upload(g_s.c, currentt, n * 8, s)) return rc;
  if (int rc = upload(g_s.d, left, n * 8, s)) return rc;
  if (int rc = upload(g_s.e, right, n * 8, s)) return rc;
  if (int rc = upload(g_s.f, sl, n, s)) return rc;
  if (int rc = upload(g_s.g, sr, n, s)) return rc;
  CK(g_s.h.reserve(n));
  launch_evolve_update(g_s.a.as<uint8_t>(), g_s.b.as<double>(), Lmin, g_s.c.as<double>(), g_s.d.as<double>(),
                       g_s.e.as<double>(), g_s.f.as<uint8_t>(), g_s.g.as<uint8_t>(), g_s.h.as<uint8_t>(), (int)n, s);
  CK(hipGetLastError());
  if (int rc = download(currentt, g_s.c, n * 8, s)) return rc;
  if (int rc = download(left, g_s.d, n * 8, s)) return rc;
  if (int rc = download(right, g_s.e, n * 8, s)) return rc;
  if (int rc = download(sl, g_s.f, n, s)) return rc;
  if (int rc = download(sr, g_s.g, n, s)) return rc;
  if (int rc = download(success, g_s.h, n, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_step_back(double Lmin, double *allL, size_t n, size_t ngen, int64_t *generation, double *currentt) {
  if (n == 0) return 0;
  if (!allL || !generation || !currentt) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(n, ngen)) return rc;
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  if (int rc = upload(g_s.a, allL, n * ngen * 8, s)) return rc;
  if (int rc = upload(g_s.b, generation, n * 8, s)) return rc;
  if (int rc = upload(g_s.c, currentt, n * 8, s)) return rc;
  CK(g_s.d.reserve(8));
  launch_step_back(Lmin, g_s.a.as<double>(), (int)n, (int)ngen, g_s.b.as<long long>(), g_s.c.as<double>(),
                   g_s.d.as<long long>(), s);
  CK(hipGetLastError());
  if (int rc = download(allL, g_s.a, n * ngen * 8, s)) return rc;
  if (int rc = download(generation, g_s.b, n * 8, s)) return rc;
  if (int rc = download(currentt, g_s.c, n * 8, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_unitcube_line_intersection(const double *origin, const double *direction, size_t n, size_t d, double *tleft,
                                   double *tright) {
  if (n == 0) return 0;
  if (!origin || !direction || !tleft || !tright) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(n, d)) return rc;
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  if (int rc = upload(g_s.a, origin, n * d * 8, s)) return rc;
  if (int rc = upload(g_s.b, direction, n * d * 8, s)) return rc;
  CK(g_s.c.reserve(n * 8));
  CK(g_s.d.reserve(n * 8));
  launch_line_intersection(g_s.a.as<double>(), g_s.b.as<double>(), (int)n, (int)d, g_s.c.as<double>(),
                           g_s.d.as<double>(), s);
  CK(hipGetLastError());
  if (int rc = download(tleft, g_s.c, n * 8, s)) return rc;
  if (int rc = download(tright, g_s.d, n * 8, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_update_vectorised_slice_sampler(const double *t, double *tleft, double *tright, const double *proposed_L,
                                        const double *proposed_u, const double *proposed_p, int64_t *worker_running,
                                        int64_t *status, double threshold, double shrink_factor, double *allu,
                                        double *allL, double *allp, size_t popsize, size_t d, size_t nparams,
                                        int64_t *discarded) {
  if (!discarded) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  *discarded = 0;
  if (popsize == 0) return 0;
  if (!t || !tleft || !tright || !proposed_L || !proposed_u || !proposed_p || !worker_running || !status || !allu ||
      !allL || !allp)
    return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(popsize, d > nparams ? d : nparams)) return rc;
  for (size_t l = 0; l < popsize; ++l)
    if (worker_running[l] < 0 || (size_t)worker_running[l] >= popsize)
      return ctx_fail_arg(MLF_E_BADARG, "worker_running entry out of range");
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  const size_t P = popsize;
  if (int rc = upload(g_s.a, t, P * 8, s)) return rc;
  if (int rc = upload(g_s.b, tleft, P * 8, s)) return rc;
  if (int rc = upload(g_s.c, tright, P * 8, s)) return rc;
  if (int rc = upload(g_s.d, proposed_L, P * 8, s)) return rc;
  if (int rc = upload(g_s.e, proposed_u, P * d * 8, s)) return rc;
  if (int rc = upload(g_s.f, proposed_p, P * nparams * 8, s)) return rc;
  if (int rc = upload(g_s.g, worker_running, P * 8, s)) return rc;
  if (int rc = upload(g_s.h, status, P * 8, s)) return rc;
  if (int rc = upload(g_s.i, allu, P * d * 8, s)) return rc;
  if (int rc = upload(g_s.j, allL, P * 8, s)) return rc;
  if (int rc = upload(g_s.k, allp, P * nparams * 8, s)) return rc;
  CK(g_s.l.reserve((P + 1) * 8));
  launch_slice_update(g_s.a.as<double>(), g_s.b.as<double>(), g_s.c.as<double>(), g_s.d.as<double>(),
                      g_s.e.as<double>(), g_s.f.as<double>(), g_s.g.as<long long>(), g_s.h.as<long long>(), threshold,
                      shrink_factor, g_s.i.as<double>(), g_s.j.as<double>(), g_s.k.as<double>(), (int)P, (int)d,
                      (int)nparams, g_s.l.as<long long>(), s);
  CK(hipGetLastError());
  if (int rc = download(tleft, g_s.b, P * 8, s)) return rc;
  if (int rc = download(tright, g_s.c, P * 8, s)) return rc;
  if (int rc = download(worker_running, g_s.g, P * 8, s)) return rc;
  if (int rc = download(status, g_s.h, P * 8, s)) return rc;
  if (int rc = download(allu, g_s.i, P * d * 8, s)) return rc;
  if (int rc = download(allL, g_s.j, P * 8, s)) return rc;
  if (int rc = download(allp, g_s.k, P * nparams * 8, s)) return rc;
  if (int rc = download(discarded, g_s.l, 8, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

int mlf_row_dist2(const double *a, const double *b, size_t n, size_t d, double *out) {
  if (n == 0) return 0;
  if (!a || !b || !out) return ctx_fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = check_nd(n, d)) return rc;
  if (int rc = ctx_ensure()) return rc;
  hipStream_t s = ctx_stream();
  if (int rc = upload(g_s.a, a, n * d * 8, s)) return rc;
  if (int rc = upload(g_s.b, b, n * d * 8, s)) return rc;
  CK(g_s.c.reserve(n * 8));
  launch_row_dist2(g_s.a.as<double>(), g_s.b.as<double>(), (int)n, (int)d, g_s.c.as<double>(), s);
  CK(hipGetLastError());
  if (int rc = download(out, g_s.c, n * 8, s)) return rc;
  CK(hipStreamSynchronize(s));
  return 0;
}

}  // extern "C"
