// mlf_api.hip -- the C ABI of libmlfriends_hip.so (include/mlfriends_hip.h): argument checks,
// device buffers, host<->device staging and kernel sequencing.  No numerics live here.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <limits>
#include <vector>

#include "../../include/mlfriends_hip.h"
#include "mlf_ctx.hpp"
#include "mlf_filter.hpp"
#include "mlf_misc.hpp"
#include <atomic>
#include "mlf_small.hpp"
#include "mlf_prep3.hpp"
#include "mlf_prep4.hpp"
#include "mlf_prep64.hpp"
#include "mlf_sample.hpp"

namespace {

using namespace mlf;

thread_local std::string g_err;

int fail_hip(hipError_t e, const char *what, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at mlf_api.hip:%d: %s", (int)e, hipGetErrorString(e),
           line, what);
  g_err = buf;
  return -(int)e;
}

int fail_arg(int code, const char *msg) {
  g_err = msg;
  return code;
}

#define CK(x)                                                   \
  do {                                                          \
    hipError_t e_ = (x);                                        \
    if (e_ != hipSuccess) return fail_hip(e_, #x, __LINE__);    \
  } while (0)

// Buffers and host-side state of the MFMA pre-filter (mlf_filter.hip) for one set of live points.
constexpr unsigned kFilterSegCap = 2048;
constexpr long long kFilterMinQueriesDefault = 257;   // = everything the single-launch path does not take (with the tile
// ranges of filter_tile_split a 1024-query batch takes 33 us through the filter, 175 us through the exact scan)

// ---- tuning options -----------------------------------------------------------------------------------------------
// Every option has a PROCESS default (mlf_set_option) and may be overridden per region handle
// (mlf_region_set_option): two regions of one process can run different routings, and nothing a test or a benchmark
// flips on one handle reaches another.  Results never depend on them.
enum Opt : int {
  OPT_FILTER,              // "filter" 0/1: matrix-core pre-filter in front of the exact scan
  OPT_FIRST_RANGE_PCT,     // "filter_first_range_pct" 10 ... 90: share of the live-point tiles in the first of two ranges
  OPT_SPLIT_WAVES,         // "filter_split_waves": waves a single-sweep launch aims at when it splits the tiles (1 ... 16 ranges)
  OPT_NARROW_TAIL,         // "filter_narrow_tail" 0/1: later ranges of a phased sweep with 2 query groups per wave
  OPT_SMALL_PATH,          // "small_path" 0/1: one launch for up to 256 proposals handed over on the host
  OPT_FUSED_PREP,          // "fused_prep" 0/1: fused per-proposal stage (off: k_prep + separate quantisation)
  OPT_PHASE_MIN_QUERIES,   // "filter_phase_min_queries": smaller batches sweep all tiles in one launch
  OPT_PHASES,              // "filter_phases": 0 single sweep, 1 default phase count, n >= 2 exactly n phases
  OPT_TIME_LAUNCHES,       // "time_filter_launches" 0/1: event pairs around every matrix-kernel launch of every call
  OPT_PREP_BOUNDED,        // "prep_bounded" 0/1: bounded matrix-core per-proposal stage (mlf_prep4.hip) or the binary64 one
  OPT_MIN_QUERIES,         // "filter_min_queries": smaller batches go straight to the exact scan
  OPT_SWEEP_MIN,           // "sweep_min" 0/1: two-range batches through the min-only sweep (mlf_sweepmin.hip) or k_sweep
  OPT_MID_MAX,             // "mid_max_queries": batches up to this size take the one-launch path (mlf_mid.hip); 0 = never
  OPT_FUSED_FIRST,         // "fused_first_range" 0/1: per-proposal stage and first range of the min-only sweep in one launch (mlf_fused.hip)
  OPT_BOOT_SYM,            // "boot_symmetric" 0/1: whole-range bootstrap passes compute every pair distance once (k_boot_sym)
  OPT_ORDER,               // "filter_order" 0/1: mask-mode operand in storage order / nearest to the centre first (k_ref_rank)
  OPT_SECOND_RANGE_PCT,    // "filter_second_range_pct" 0 ... 90: min-only sweep in three ranges, the second ending at this share of the tiles (0: two ranges)
  OPT_THIRD_MIN_WORK,      // "filter_third_range_min_work": three ranges from this many (proposals x 32-row live-point tiles) on
  OPT_FUSED_WAVES,         // "fused_waves" 4 / 8: waves per workgroup of k_prep_sweep (4, default: two workgroups per CU up to d = 50; 8: one)
  OPT_FUSED_VARIANT,       // "fused_variant": bit 0 = k_prep_sweep loads its matrix fragments by LDS-DMA (default) or by a load / store loop;
                           // bit 1 = the ellipsoid form read off the whitening chain where the region allows it (default; region_prep4_setup)
  OPT_COUNT
};
const char *const kOptNames[OPT_COUNT] = {"filter", "filter_first_range_pct", "filter_split_waves", "filter_narrow_tail",
                                          "small_path", "fused_prep", "filter_phase_min_queries", "filter_phases",
                                          "time_filter_launches", "prep_bounded", "filter_min_queries", "sweep_min", "mid_max_queries", "fused_first_range",
                                          "boot_symmetric", "filter_order", "filter_second_range_pct", "filter_third_range_min_work", "fused_waves", "fused_variant"};
}  // namespace
namespace mlf {
std::atomic<unsigned> g_grant_epoch{0u};
std::atomic<unsigned long long> g_grant_calls{0ull};
}  // namespace mlf
namespace {
long long g_opt[OPT_COUNT] = {1, 30, 2048, 1, 1, 1, 32768, 1, 0, 1, kFilterMinQueriesDefault, 1, 2048, 1, 1, 1, 50, 100000000ll, 4, 3};

struct OptOverrides {
  long long v[OPT_COUNT] = {};
  bool set[OPT_COUNT] = {};
};

int opt_id(const char *name) {
  for (int i = 0; i < OPT_COUNT; ++i)
    if (!strcmp(name, kOptNames[i])) return i;
  return -1;
}

long long opt_clamp(int id, long long value) {
  switch (id) {
    case OPT_FIRST_RANGE_PCT: return value < 10 ? 10 : (value > 90 ? 90 : value);
    case OPT_SPLIT_WAVES: return value < 256 ? 256 : (value > 16384 ? 16384 : value);
    case OPT_PHASES: return value < 0 ? 0 : (value > 64 ? 64 : value);
    case OPT_SECOND_RANGE_PCT: return value < 0 ? 0 : (value > 90 ? 90 : value);
    case OPT_FUSED_WAVES: return value == 4 ? 4 : 8;
    case OPT_FUSED_VARIANT: return value & 3;
    case OPT_BOOT_SYM: return value < 0 ? 0 : (value > 2 ? 2 : value);
    case OPT_THIRD_MIN_WORK: return value < 0 ? 0 : value;
    case OPT_PHASE_MIN_QUERIES:
    case OPT_MID_MAX:
    case OPT_MIN_QUERIES: return value;
    default: return value != 0;
  }
}

struct FilterCtx {
  bool refs_ready = false;   // live points quantised
  bool refs_dirty = false;   // a live point was replaced since: requantise before the next batch that uses the operands
  bool usable = false;       // statistics are finite and the dimensionality is covered
  int ks = 0, ntiles32 = 0;
  double sigma = 1.0, amax = 0.0;
  DevBuf stats, statscratch, refF, qF, tlo, thi, route, best, counters, list, segcnt, gate2;
  // mask-mode operand: the live points nearest to the centre first (launch_ref_order): binary16 fragments, the rows the exact
  // re-check reads (same order), keys and permutation (slot -> storage row).  The first-index operand refF keeps storage order.
  DevBuf refFm, refRm, okeys, operm;
  bool ordered = false;      // refFm / refRm are current
  int order_n = -1;          // the live-set size the permutation was ranked for
  // phased sweep: two compacted query sets (ping-pong)
  DevBuf pqF[2], ptlo[2], pthi[2], pmap[2], png, pmin, pmin2;
  DevBuf mid_rec, mid_meta, mid_arrive;   // one-launch path (mlf_mid.hip): records of the tile ranges, arrival counters
  bool mid_last = false;                  // the last batch took that path (debug_stats)
  bool mid_dirty = false;                 // a launch of that path failed: its self-resetting counters are zeroed before the next batch
  DevBuf fstamps;                         // diagnostics: stage stamps of one k_prep_sweep wave
  int stamp_block = -1;
  bool png_dirty = false;                 // a phased batch did not reach its scan launch (whose tail returns the slot counters to zero)
  // bounded per-proposal stage (mlf_prep4.hip): ellipsoid band list, per-call counters
  // misc: [0] band proposals, [1] k_ell_exact workgroups done -- both return to zero by themselves (no memset per batch),
  // zeroed once when the buffer is allocated; [2], [3] "a proposal is routed to the exact scan", used alternately by
  // successive batches (the scan launch of a batch clears the word of the next one); [4] band proposals of the last
  // batch (mlf_region_debug_stats)
  DevBuf ell_list, misc;
  unsigned batch_parity = 0;
  size_t last_nsegs = 0;      // list segments of the last filtered batch
  int last_cut[2] = {0, 0};   // tile cuts of the last min-only batch (second: 0 with two ranges)
  bool last_same_form = false;   // the last fused launch read the ellipsoid form off the whitening chain (debug_stats)
  bool ell_pending = false;   // band list of the running call not decided yet (the tail of the re-check launch decides it)
  EllExactArgs ell_args{};
  // (start, stop) event pairs around every k_filter launch of the timed calls
  std::vector<hipEvent_t> kev;
  size_t kev_used = 0;
  OptOverrides ov;            // per-handle tuning (mlf_region_set_option); the stateless calls' context has none
  void release() {
    DevBuf *b[] = {&stats, &statscratch, &refF, &qF, &tlo, &thi, &route, &best, &counters, &list, &segcnt, &gate2, &refFm, &refRm, &okeys, &operm,
                   &pqF[0], &pqF[1], &ptlo[0], &ptlo[1], &pthi[0], &pthi[1], &pmap[0], &pmap[1], &png, &pmin, &pmin2, &mid_rec, &mid_meta, &mid_arrive,
                   &ell_list, &misc, &fstamps};
    for (DevBuf *x : b) x->release();
    refs_ready = usable = ordered = false;
    order_n = -1;
  }
};

inline long long opt(const FilterCtx &f, int id) { return f.ov.set[id] ? f.ov.v[id] : g_opt[id]; }

struct Ctx {
  bool ready = false;
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;   // host batches in chunks: copies here, kernels on `stream` (mlf_region_inside)
  hipEvent_t copy_event = nullptr;
  // scratch used by the stateless host-pointer entry points
  unsigned long long *pin_adj = nullptr;   // pinned host copy of the adjacency bits (mlf_cluster_labels)
  size_t pin_adj_cap = 0;
  DevBuf src, refT, refR, q, out, flags, sel, selmask, selbytes, M, small0, small1, small2, small3, mask, tq;
  FilterCtx filter;
  // single-launch path for a handful of proposals (mlf_small.hip): pinned, device-mapped staging + two scratch words
  // per proposal
  double *pin_pts = nullptr, *pin_pts_dev = nullptr;
  uint8_t *pin_mask = nullptr, *pin_mask_dev = nullptr;   // mask bytes, then (at kSmallMaxPoints) the completion word
  unsigned small_seq = 0;
  DevBuf small_words;
};

Ctx g_ctx;

int ensure_ctx() {
  if (g_ctx.ready) return 0;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_err = "libmlfriends_hip: no HIP device visible (this library has no CPU fallback)";
    return MLF_E_NODEVICE;
  }
  CK(hipSetDevice(g_ctx.device));
  CK(hipStreamCreate(&g_ctx.stream));
  g_ctx.ready = true;
  return 0;
}

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// d x d row-major -> d rows of dp doubles, zero padded; optional transpose
std::vector<double> pad_matrix(const double *m, int d, int dp, bool transpose) {
  std::vector<double> o((size_t)d * dp, 0.0);
  for (int r = 0; r < d; ++r)
    for (int c = 0; c < d; ++c) o[(size_t)r * dp + c] = transpose ? m[(size_t)c * d + r] : m[(size_t)r * d + c];
  return o;
}

std::vector<double> pad_vector(const double *v, int d, int dp, double fill = 0.0) {
  std::vector<double> o((size_t)dp, fill);
  for (int k = 0; k < d; ++k) o[k] = v[k];
  return o;
}

// `src` may be a host or a device pointer (unified addressing picks the direction): the array arguments of the
// stateless entry points can stay on the device between calls (device-resident rebuild, ultranest_amd.device_rebuild)
// Pinned staging for the many small constant uploads of one call (mlf_region_set sends ~14 matrices and fragment sets):
// while an arena is active, a small upload copies its source into the arena and leaves from there -- truly
// asynchronous, so the caller needs no stream synchronisation before its host vectors go out of scope (round 2: nine
// synchronisations and a dozen pageable copies, 0.2 of the call's 0.6 ms).  The arena is rewound by the caller once the
// stream has been synchronised.
struct HostArena {
  unsigned char *p = nullptr;       // pinned host memory ...
  unsigned char *p_dev = nullptr;   // ... as the device sees it
  size_t cap = 0, used = 0;
  ScatterArgs pending{};            // uploads staged but not yet sent (arena_flush)
  int npending = 0;
  void *take(size_t bytes) {
    const size_t at = (used + 63) / 64 * 64;
    if (!p || at + bytes > cap) return nullptr;
    used = at + bytes;
    return p + at;
  }
};
HostArena *g_arena = nullptr;
constexpr size_t kArenaBytes = 1u << 20, kArenaMaxPiece = 128u << 10;

bool arena_active() { return g_arena != nullptr; }

// Everything staged so far leaves in ONE launch (the device reads the pinned arena itself).  To be called in front of every
// kernel that reads a constant uploaded under the arena, and before the arena is dropped.
int arena_flush(hipStream_t s) {
  if (!g_arena || g_arena->npending == 0) return 0;
  launch_scatter_copy(g_arena->pending, g_arena->npending, s);
  g_arena->npending = 0;
  CK(hipGetLastError());
  return 0;
}

bool is_device_pointer(const void *p);

int upload(DevBuf &b, const void *src, size_t bytes, hipStream_t s) {
  CK(b.reserve(bytes ? bytes : 1));
  if (!bytes) return 0;
  // the arena branch reads `src` with a host memcpy: never for a device pointer (mlf_region_set accepts the live points
  // from either side; VRAM is host-readable only on large-BAR boxes)
  if (g_arena && bytes <= kArenaMaxPiece && !is_device_pointer(src)) {
    if (void *stage = g_arena->take(bytes)) {
      memcpy(stage, src, bytes);
      if (g_arena->npending == kScatterMax)
        if (int rc = arena_flush(s)) return rc;
      HostArena &a = *g_arena;
      a.pending.dst[a.npending] = b.p;
      a.pending.src[a.npending] = a.p_dev + (static_cast<unsigned char *>(stage) - a.p);
      a.pending.bytes[a.npending] = (unsigned)bytes;
      ++a.npending;
      return 0;
    }
  }
  if (int rc = arena_flush(s)) return rc;   // keep the order of the stream
  CK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyDefault, s));
  if (g_arena) CK(hipStreamSynchronize(s));   // did not fit: the caller relies on the source being consumed
  return 0;
}

bool is_device_pointer(const void *p) {
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
    (void)hipGetLastError();   // plain host memory the runtime has never seen
    return false;
  }
  return attr.type == hipMemoryTypeDevice;
}

// like upload(), but `src` may be a host OR a device pointer (unified addressing picks the direction): the bootstrap
// selection masks arrive from a device-side broadcast in the multi-GPU rebuild
int upload_any(DevBuf &b, const void *src, size_t bytes, hipStream_t s) {
  CK(b.reserve(bytes ? bytes : 1));
  if (bytes) CK(hipMemcpyAsync(b.p, src, bytes, hipMemcpyDefault, s));
  return 0;
}

int check_dims(size_t d) {
  if (d == 0) return fail_arg(MLF_E_BADARG, "dimensionality must be positive");
  if (d > MLF_MAX_DIM) return fail_arg(MLF_E_DIM, "dimensionality above MLF_MAX_DIM (1024) is not supported");
  return 0;
}

// Live points (row-major, host) -> device layouts in the context scratch.
int stage_live_points(const double *pts, size_t n, size_t d, int dp, int npad, bool want_rows) {
  Ctx &c = g_ctx;
  int rc = upload(c.src, pts, n * d * sizeof(double), c.stream);
  if (rc) return rc;
  CK(c.refT.reserve((size_t)npad * dp * sizeof(double)));
  CK(c.refR.reserve((size_t)(npad + 1) * dp * sizeof(double)));   // k_boot requests the first block of the row after its last
  (void)want_rows;
  launch_build_layouts(c.src.as<double>(), (int)n, (int)d, dp, npad, c.refT.as<double>(),
                       c.refR.as<double>(), c.stream);
  CK(hipGetLastError());
  return 0;
}

// Quantise the live points for the filter.  host_sync = true also fetches the statistics that the
// host-side eligibility test uses (done when the live set is installed, not per batch).
int filter_prepare_refs(FilterCtx &f, const double *refR, int n, int d, int dp, hipStream_t s,
                        bool host_sync) {
  f.refs_ready = false;
  const int ks = (dp + 6 + 15) / 16;   // filter dimensionality = padded DP (zero columns are harmless)
  if (ks > 9 || n < 1 || (long long)round_up(n, 32) / 32 * ks * 1024 >= (1ll << 31)) {   // k_sweep addresses the tiles with 32-bit buffer offsets
    f.usable = false;
    return 0;
  }
  f.ks = ks;
  const int npad32 = round_up(n, 32);
  f.ntiles32 = npad32 / 32;
  CK(f.stats.reserve((8 + MLF_FILTER_MAXD) * sizeof(double)));
  CK(f.refF.reserve((size_t)npad32 * ks * 16 * 2));
  (void)d;
  {
    const void *before = f.statscratch.p;
    const size_t bytes = ((size_t)64 * 128 + 2) * sizeof(double);
    CK(f.statscratch.reserve(bytes));
    if (f.statscratch.p != before) CK(hipMemsetAsync(f.statscratch.p, 0, bytes, s));   // running maxima start at zero
  }
  // the mask-mode operand: the live points nearest to the centre first.  A NEW live set (host_sync) is ranked -- the keys
  // come out of the statistics pass --; a refresh behind row replacements keeps the permutation (the replaced rows stay in
  // their slots: the order is a heuristic of the sweep, not part of any answer) and only requantises
  const bool want_order = opt(f, OPT_ORDER) && n <= 65536;
  const bool rerank = want_order && (host_sync || f.order_n != n);
  if (want_order) {
    CK(f.okeys.reserve((size_t)n * sizeof(unsigned long long)));
    CK(f.operm.reserve((size_t)n * sizeof(int)));
  }
  launch_ref_stats(refR, n, dp, dp, f.stats.as<double>(), f.statscratch.as<double>(), s, rerank ? f.okeys.as<unsigned long long>() : nullptr);
  f.ordered = false;
  if (want_order) {
    const int nrows = round_up(n, 64) + 1;   // the re-check requests whole 16-coordinate blocks: a spare row behind the last
    CK(f.refFm.reserve((size_t)npad32 * ks * 16 * 2));
    CK(f.refRm.reserve((size_t)nrows * dp * sizeof(double)));
    if (rerank) launch_ref_rank(f.okeys.as<unsigned long long>(), n, f.operm.as<int>(), s);
    launch_quant_refs(refR, n, npad32, dp, dp, ks, f.stats.as<double>(), f.refF.p, s, f.refFm.p, f.operm.as<int>(), f.refRm.as<double>(), nrows);
    CK(hipGetLastError());
    f.ordered = true;
    f.order_n = n;
  } else {
    launch_quant_refs(refR, n, npad32, dp, dp, ks, f.stats.as<double>(), f.refF.p, s);
    CK(hipGetLastError());
    if (host_sync) f.order_n = -1;
  }
  if (host_sync) {
    double h[4];
    CK(hipMemcpyAsync(h, f.stats.p, sizeof h, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    f.sigma = h[0];
    f.amax = h[2];
    f.usable = h[3] == 1.0 && h[2] > 0.0 && h[2] < 1e150;
  }
  f.refs_ready = true;
  f.refs_dirty = false;
  return 0;
}

// requantise the live points if one was replaced since the operands were built (mlf_region_update_point)
int filter_refresh_refs(FilterCtx &f, const double *refR, int n, int d, int dp, hipStream_t s) {
  if (!f.refs_ready || !f.refs_dirty) return 0;
  if (int rc = filter_prepare_refs(f, refR, n, d, dp, s, false)) return rc;
  f.refs_dirty = false;
  return 0;
}

// host-side eligibility of one batch (the kernels re-check per query and fall back on their own)
bool filter_applies(const FilterCtx &f, long long nq, double r2) {
  if (!opt(f, OPT_FILTER) || !f.refs_ready || !f.usable || nq < opt(f, OPT_MIN_QUERIES)) return false;
  // below ~2000 proposals the five launches of the filter path (~45 us) only pay when the exact scan has real work:
  // 400 proposals against 400 x 5 live coordinates take 43 us through the exact scan, 51 us through the filter
  if (opt(f, OPT_MIN_QUERIES) == kFilterMinQueriesDefault && nq < 2048 && (long long)f.ntiles32 * 32 * f.ks * 16 < 40000) return false;
  if (!(r2 > 0.0) || !(r2 < 1e150)) return false;
  const double sr2 = f.sigma * f.sigma * r2;
  return sr2 < 4096.0 && sr2 > 1e-30;
}

// self-resetting counters of the bounded stage: zeroed once, when allocated
int misc_reserve(FilterCtx &f) {
  if (f.misc.p) return 0;
  CK(f.misc.reserve(8 * sizeof(unsigned)));
  CK(hipMemset(f.misc.p, 0, 8 * sizeof(unsigned)));
  return 0;
}

// Device buffers of one filtered batch.
int filter_reserve(FilterCtx &f, long long nq, unsigned *cap_out) {
  const long long nqpad = (nq + 31) / 32 * 32;
  long long nwaves = filter_wave_count(f.ks, nqpad / 32, f.ks == 4 ? 2 : (f.ks < 4 ? 1 : 0));   // room for a narrow later range
  {   // ... and for the tile ranges of a small batch (filter_tile_split): one segment per (wave, range)
    const long long plain = (filter_wave_count(f.ks, nqpad / 32) + 3) / 4 * 4;
    const long long split = filter_tile_split(f.ks, nqpad / 32, f.ntiles32, (int)opt(f, OPT_SPLIT_WAVES));
    if (split * plain > nwaves) nwaves = split * plain;
  }
  const unsigned cap = kFilterSegCap;   // uncertain pairs per filter wave (expected: tens)
  CK(f.qF.reserve((size_t)nqpad * f.ks * 16 * 2));
  CK(f.tlo.reserve((size_t)nqpad * sizeof(float)));
  CK(f.thi.reserve((size_t)nqpad * sizeof(float)));
  CK(f.route.reserve((size_t)nq));
  CK(f.best.reserve((size_t)nq * sizeof(int)));
  CK(f.counters.reserve(4 * sizeof(unsigned)));
  CK(f.list.reserve((size_t)nwaves * cap * sizeof(unsigned long long)));
  CK(f.segcnt.reserve((size_t)nwaves * sizeof(unsigned)));
  CK(f.gate2.reserve((size_t)nq));
  if (int rc = misc_reserve(f)) return rc;
  *cap_out = cap;
  return 0;
}

// Filter pipeline on device data: answers for all nq queries in out_mask (bytes) and/or out_idx.
// Query element (j, k) is q[j*ldq + k*ldk].  quantised = true: the fused per-proposal stage has already
// produced the binary16 fragments, thresholds and routes of this batch.
// Where the exact whitened coordinates come from when the per-proposal stage did not store them (k_prep4): the
// proposals themselves and the layer; the queries that need coordinates are whitened after the sweeps.
struct ExactSrc {
  const double *pts;
  const double *lay_ctr;
  const double *T8;   // row-major layer matrix, row stride ldt
  int ldt;
  const double *T64;  // the same as 64 x 64, zero padded (d <= 64)
  const Prep4Args *prep;   // not null: the per-proposal stage has NOT run yet -- the min-only path runs it inside its first launch
  const Prep4Consts *same; // not null: constants of the "same quadratic form" variant of that launch (region_prep4_setup)
  bool pregated;           // the gate array holds a pre-gate on entry (device-side sampling: the cube test) for that launch to honour
};

// does a batch of nq queries take the min-only path?  (the phase rule of filter_run)
// Tile ranges of the min-only sweep, every range at least 4 tiles.  Two ranges: [0, c1) [c1, ntiles32) with c1 = first_pct of
// the tiles (c2 = 0).  Three -- batches of at least min_work (proposals x tiles), where the pairs a third range saves outweigh
// its launch and its compaction: the last range starts at c2 = second_pct of the tiles and [0, c2) is split at first_pct of
// c2 (30 % of 50 % = 15 %, 50 %: profiles/r05_three_range_ab.jsonl).
void filter_range_cuts(int ntiles32, long long nq, int first_pct, int second_pct, long long min_work, int *c1, int *c2) {
  const int cut = (int)((long long)ntiles32 * first_pct / 100);
  *c1 = cut < 4 ? 4 : (cut > ntiles32 - 4 ? ntiles32 - 4 : cut);
  *c2 = 0;
  if (second_pct <= 0 || nq * ntiles32 < min_work) return;
  const int cut2 = (int)((long long)ntiles32 * second_pct / 100);
  int a = (int)((long long)cut2 * first_pct / 100);
  a = a < 4 ? 4 : a;
  if (cut2 >= a + 4 && cut2 <= ntiles32 - 4) {
    *c1 = a;
    *c2 = cut2;
  }
}

bool filter_takes_min_path(const FilterCtx &f, long long nq) {
  const int want_phases = (int)opt(f, OPT_PHASES);
  const long long phase_min = opt(f, OPT_PHASE_MIN_QUERIES);
  int nphase = 1;
  if (want_phases && nq >= phase_min && (phase_min < 32768 || nq * f.ntiles32 >= 30000000ll))
    nphase = want_phases >= 2 ? (want_phases <= f.ntiles32 / 4 ? want_phases : (f.ntiles32 >= 16 ? 2 : 1)) : (f.ntiles32 >= 16 ? 2 : 1);
  return nphase == 2 && f.ks <= 4 && opt(f, OPT_SWEEP_MIN);
}

int filter_run(FilterCtx &f, const double *refT, const double *refR, int n, int npad, int d, int dp,
               const double *q, long long ldq, long long ldk, long long nq, double r2, const uint8_t *gate,
               uint8_t *out_mask, long long *out_idx, hipStream_t s, bool quantised,
               hipEvent_t ev_after_filter = nullptr, const ExactSrc *xs = nullptr) {
  const long long ngroups = (nq + 31) / 32;
  const long long nqpad = ngroups * 32;
  unsigned cap = 0;
  f.mid_last = false;
  if (int rc = filter_reserve(f, nq, &cap)) return rc;
  // mask mode sweeps the centre-first copy of the live points (any hit decides); the first-index mode keeps storage order
  const bool use_m = out_idx == nullptr && f.ordered && opt(f, OPT_ORDER);
  const void *const opF = use_m ? f.refFm.p : f.refF.p;
  const double *const opR = use_m ? f.refRm.as<double>() : refR;
  if (!quantised)
  launch_quant_queries(q, ldq, nq, nqpad, d, dp, f.ks, f.stats.as<double>(), r2, gate, f.qF.p, f.tlo.as<float>(),
                       f.thi.as<float>(), f.route.as<uint8_t>(), f.best.as<int>(), f.counters.as<unsigned>(), s);
  CK(hipGetLastError());
  FilterArgs fa{};
  fa.refF = opF;
  fa.qF = f.qF.p;
  fa.tlo = f.tlo.as<float>();
  fa.thi = f.thi.as<float>();
  fa.ntiles32 = f.ntiles32;
  fa.ngroups = ngroups;
  fa.nq = nq;
  fa.best = f.best.as<int>();
  fa.list = f.list.as<unsigned long long>();
  fa.seg_cap = cap;
  fa.seg_count = f.segcnt.as<unsigned>();
  fa.counters = f.counters.as<unsigned>();
  // Phased sweep: the live-point tiles are split into nphase ranges; after each range the queries that
  // are decided (certain hit) leave, the rest is compacted into fresh 32-query groups.  Every launch
  // is sized for the worst case and reads the actual group count from device memory: no host sync.
  int nphase = 1;
  // worth it when the sweep is long compared with one compaction (~25 us + 40 us per 10^6 queries)
  const int want_phases = (int)opt(f, OPT_PHASES);
  const long long phase_min = opt(f, OPT_PHASE_MIN_QUERIES);
  const int narrow_tail = (int)opt(f, OPT_NARROW_TAIL);
  if (want_phases && nq >= phase_min && (phase_min < 32768 || nq * f.ntiles32 >= 30000000ll))
    nphase = want_phases >= 2 ? (want_phases <= f.ntiles32 / 4 ? want_phases : (f.ntiles32 >= 16 ? 2 : 1))
                                  : (f.ntiles32 >= 16 ? 2 : 1);
  if (nphase > 1) {
    for (int i = 0; i < 2; ++i) {
      CK(f.pqF[i].reserve((size_t)nqpad * f.ks * 16 * 2));
      CK(f.ptlo[i].reserve((size_t)nqpad * sizeof(float)));
      CK(f.pthi[i].reserve((size_t)nqpad * sizeof(float)));
      CK(f.pmap[i].reserve((size_t)nqpad * sizeof(int)));
    }
    if (!f.png.p) {   // group counts + the slot counter of the fused compaction (returns to zero by itself: k_phase_finish)
      // words: [0] groups after the first range, [1] size of the uncertain set (statistics of the last batch), [2] [3] [4] slot
      // counters of the compacted sets, [5] groups after the second of three ranges
      CK(f.png.reserve(8 * sizeof(unsigned)));
      CK(hipMemset(f.png.p, 0, 8 * sizeof(unsigned)));
    } else if (f.png_dirty) {   // an earlier phased batch failed between its first compaction and its scan launch
      CK(hipMemsetAsync(f.png.p, 0, 8 * sizeof(unsigned), s));
    }
    f.png_dirty = true;   // until the scan launch of this batch is queued
  }
  const bool fused = nphase > 1;   // the compaction of the undecided queries rides in the matrix kernel's epilogue
  // mask mode behind the bounded per-proposal stage, single-sweep batches (below ~262144 proposals at N = 4000): the
  // re-check (binary64 whitening of the queries of the uncertain pairs + the reference's distance loop) runs inside the
  // sweep launch, each wave on the segment it has just written, and the ellipsoid band rides in the same launch: three
  // launches per batch instead of four (131072 proposals: 104 -> 96 us).  Phased sweeps keep the separate re-check launch:
  // there the waves' re-check tails cost more than the launch they save (10^6: 0.473 -> 0.51 ms)
  const bool own_recheck = xs != nullptr && out_idx == nullptr && nphase == 1;
  if (own_recheck) {
    fa.rw.list = f.list.as<unsigned long long>();
    fa.rw.seg_cap = cap;
    fa.rw.seg_count = f.segcnt.as<unsigned>();
    fa.rw.refR = opR;
    fa.rw.n = n;
    fa.rw.d = d;
    fa.rw.dp = dp;
    fa.rw.pts = xs->pts;
    fa.rw.nq = nq;
    fa.rw.lay_ctr = xs->lay_ctr;
    fa.rw.T64 = xs->T64;
    fa.rw.r2 = r2;
    fa.rw.best = f.best.as<int>();
  }
  fa.split = nphase == 1 ? filter_tile_split(f.ks, ngroups, f.ntiles32, (int)opt(f, OPT_SPLIT_WAVES)) : 1;
  // two ranges, compaction inside the first, finalise tail in the scan launch: nobody pads the compacted set or publishes
  // its group count (k_phase_finish: a 5 us launch) -- the second range reads the slot counter itself, the tail resets it
  const bool fold_finish = fused && nphase == 2 && xs != nullptr;

  // Two ranges in mask mode behind the bounded per-proposal stage: the min-only sweep (mlf_sweepmin.hip).  The two long
  // launches carry the running minimum only; the queries whose minimum ends in the band are swept once more by
  // k_uncertain (band pairs found, queries whitened, pairs decided in one launch), which also carries the ellipsoid band.
  const bool min_path = fold_finish && out_idx == nullptr && f.ks <= 4 && opt(f, OPT_SWEEP_MIN);
  // filter_second_range_pct > 0: three ranges (the middle one reads set 0 and writes set 1 with a second array of minima;
  // the uncertain set then reuses set 0's arrays, which nobody reads any more)
  int c = 0, c2 = 0;
  filter_range_cuts(f.ntiles32, nq, (int)opt(f, OPT_FIRST_RANGE_PCT), (int)opt(f, OPT_SECOND_RANGE_PCT), opt(f, OPT_THIRD_MIN_WORK), &c, &c2);
  const bool min3 = min_path && c2 > c;
  if (xs && xs->prep && !min_path) CK(launch_prep4(*xs->prep, s));   // (the caller's forecast and this routing agree: belt and braces)
  if (min_path) {
    const long long lw = uncertain_blocks();
    CK(f.segcnt.reserve((size_t)(lw + 8) * sizeof(unsigned)));
    CK(f.pmin.reserve((size_t)nqpad * sizeof(int)));
    if (min3) CK(f.pmin2.reserve((size_t)nqpad * sizeof(int)));
    f.last_cut[0] = c;
    f.last_cut[1] = min3 ? c2 : 0;
    auto timed = [&](auto &&launch) -> int {
      const bool time_launch = ev_after_filter || opt(f, OPT_TIME_LAUNCHES);
      if (time_launch) {
        while (f.kev.size() < f.kev_used + 2) {
          hipEvent_t e;
          CK(hipEventCreate(&e));
          f.kev.push_back(e);
        }
        CK(hipEventRecord(f.kev[f.kev_used], s));
      }
      CK(launch());
      if (time_launch) {
        CK(hipEventRecord(f.kev[f.kev_used + 1], s));
        f.kev_used += 2;
      }
      return 0;
    };
    MinArgs m{};
    m.refF = opF;
    m.ntiles32 = f.ntiles32;
    m.nq = nq;
    m.best = f.best.as<int>();
    m.ngroups = ngroups;
    m.ccap = (unsigned)nqpad;
    // first range: slot = query; the queries without a certain hit go to set 0 with their minimum
    m.tile0 = 0;
    m.tile1 = c;
    m.qF = f.qF.p;
    m.tlo = f.tlo.as<float>();
    m.thi = f.thi.as<float>();
    m.cq = f.pqF[0].p;
    m.ctlo = f.ptlo[0].as<float>();
    m.cthi = f.pthi[0].as<float>();
    m.cmap = f.pmap[0].as<int>();
    m.cmin = f.pmin.as<int>();
    m.ccount = f.png.as<unsigned>() + 2;
    m.last = 0;
    if (xs->prep) {   // per-proposal stage + first range in one launch: the operand never leaves the registers
      FusedArgs fu{};
      fu.p = *xs->prep;
      fu.refF = opF;
      fu.ntiles32 = f.ntiles32;
      fu.tile0 = 0;
      fu.tile1 = c;
      fu.cq = m.cq;
      fu.ctlo = m.ctlo;
      fu.cthi = m.cthi;
      fu.cmap = m.cmap;
      fu.cmin = m.cmin;
      fu.ccount = m.ccount;
      fu.ccap = m.ccap;
      if (f.stamp_block >= 0 && f.stamp_block < 1000000) {   // diagnostics (mlf_region_debug_fused_stamps)
        CK(f.fstamps.reserve(16 * sizeof(unsigned long long)));
        fu.stamps = f.fstamps.as<unsigned long long>();
        fu.stamp_block = (unsigned)f.stamp_block;
      }
      fu.variant = (unsigned)(opt(f, OPT_FUSED_VARIANT) & 0xff);
      if ((fu.variant & 2u) && xs->same)
        fu.p.c = *xs->same;
      else
        fu.variant &= ~2u;
      f.last_same_form = (fu.variant & 2u) != 0u;
      if (xs->pregated) fu.variant |= 4u;
      const int fused_waves = (int)opt(f, OPT_FUSED_WAVES);
      if (int rc = timed([&] { return launch_prep_sweep(fu, s, fused_waves); })) return rc;
    } else if (int rc = timed([&] { return launch_sweep_min(f.ks, filter_groups_per_wave(f.ks, 0), m, s); })) return rc;
    // next range: set 0 with its minima.  Two ranges: this is the last, the uncertain queries go to set 1.  Three: the
    // queries still without a certain hit go to set 1 with their minima, and the last range sweeps those into set 0's arrays
    const int usrc = min3 ? 0 : 1;   // the set that holds the uncertain queries in the end
    m.tile0 = c;
    m.tile1 = min3 ? c2 : f.ntiles32;
    m.qF = f.pqF[0].p;
    m.tlo = f.ptlo[0].as<float>();
    m.thi = f.pthi[0].as<float>();
    m.qmap = f.pmap[0].as<int>();
    m.qmin = f.pmin.as<int>();
    m.nslots_dev = f.png.as<unsigned>() + 2;
    m.cq = f.pqF[1].p;
    m.ctlo = f.ptlo[1].as<float>();
    m.cthi = f.pthi[1].as<float>();
    m.cmap = f.pmap[1].as<int>();
    m.cmin = min3 ? f.pmin2.as<int>() : nullptr;
    m.ccount = f.png.as<unsigned>() + 3;
    m.last = min3 ? 0 : 1;
    // diagnostics: stage stamps of one k_sweep_min workgroup -- block 1 000 000 + b: the launch that follows the first range,
    // 2 000 000 + b: the one after it
    auto sweep_stamps = [&](int which) -> int {
      m.stamps = nullptr;
      if (f.stamp_block >= which * 1000000 && f.stamp_block < (which + 1) * 1000000) {
        CK(f.fstamps.reserve(16 * sizeof(unsigned long long)));
        m.stamps = f.fstamps.as<unsigned long long>();
        m.stamp_block = (unsigned)(f.stamp_block - which * 1000000);
      }
      return 0;
    };
    if (int rc = sweep_stamps(1)) return rc;
    if (min3) {
      if (int rc = timed([&] { return launch_sweep_min(f.ks, filter_groups_per_wave(f.ks, 0), m, s); })) return rc;
      if (int rc = sweep_stamps(2)) return rc;
      m.tile0 = c2;
      m.tile1 = f.ntiles32;
      m.qF = f.pqF[1].p;
      m.tlo = f.ptlo[1].as<float>();
      m.thi = f.pthi[1].as<float>();
      m.qmap = f.pmap[1].as<int>();
      m.qmin = f.pmin2.as<int>();
      m.nslots_dev = f.png.as<unsigned>() + 3;
      m.cq = f.pqF[0].p;
      m.ctlo = f.ptlo[0].as<float>();
      m.cthi = f.pthi[0].as<float>();
      m.cmap = f.pmap[0].as<int>();
      m.cmin = nullptr;
      m.ccount = f.png.as<unsigned>() + 4;
      m.last = 1;
    }
    // four query groups per wave here too: with the min-only loop the narrow form (two groups, twice the waves) that paid for
    // k_sweep's second range loses (0.080 against 0.085-0.088 ms; profiles/r04_first_range_ab.jsonl)
    if (int rc = timed([&] { return launch_sweep_min(f.ks, filter_groups_per_wave(f.ks, 0), m, s); })) return rc;
    // the uncertain set: band pairs, exact whitening, exact distances -- one launch; the ellipsoid band rides along
    UncertainArgs ua{};
    ua.refF = opF;
    ua.ntiles32 = f.ntiles32;
    ua.qF = f.pqF[usrc].p;
    ua.thi = f.pthi[usrc].as<float>();
    ua.qmap = f.pmap[usrc].as<int>();
    ua.nslots_dev = f.png.as<unsigned>() + (min3 ? 4 : 3);
    ua.pts = xs->pts;
    ua.d = d;
    ua.dp = dp;
    ua.lay_ctr = xs->lay_ctr;
    ua.T8 = xs->T8;
    ua.ldt8 = xs->ldt;
    ua.refR = opR;
    ua.n = n;
    ua.r2 = r2;
    ua.best = f.best.as<int>();
    ua.counters = f.counters.as<unsigned>();
    ua.seg_count = f.segcnt.as<unsigned>();
    if (f.ell_pending) {   // the band proposals of k_prep4: trailing workgroups of this launch
      ua.ell = f.ell_args;
      f.ell_pending = false;
    }
    if (int rc = timed([&] { return launch_uncertain(f.ks, ua, s); })) return rc;
    if (ev_after_filter) CK(hipEventRecord(ev_after_filter, s));
    f.last_nsegs = (size_t)lw;
  } else {
  for (int ph = 0; ph < nphase; ++ph) {
    fa.tile0 = (int)((long long)f.ntiles32 * ph / nphase);
    fa.tile1 = (int)((long long)f.ntiles32 * (ph + 1) / nphase);
    if (nphase == 2) {   // two ranges: the first takes filter_first_range_pct per cent of the tiles
      const int cut = (int)((long long)f.ntiles32 * opt(f, OPT_FIRST_RANGE_PCT) / 100);
      const int c = cut < 4 ? 4 : (cut > f.ntiles32 - 4 ? f.ntiles32 - 4 : cut);
      fa.tile0 = ph == 0 ? 0 : c;
      fa.tile1 = ph == 0 ? c : f.ntiles32;
    }
    fa.append = ph > 0;
    fa.seg_first_extra = filter_wave_count(f.ks, ngroups);   // segments beyond the first launch's own: for a narrow later range
    fa.seg_extra = (narrow_tail && nphase > 1 && f.ks <= 4) ? filter_wave_count(f.ks, ngroups, narrow_tail) - fa.seg_first_extra : 0;
    if (fa.seg_extra < 0) fa.seg_extra = 0;
    fa.cq = nullptr;
    if (fused && ph > 0) {   // the previous launch compacted its undecided queries into set (ph - 1) & 1
      const int src = (ph - 1) & 1;
      fa.qF = f.pqF[src].p;
      fa.tlo = f.ptlo[src].as<float>();
      fa.thi = f.pthi[src].as<float>();
      fa.qmap = f.pmap[src].as<int>();
      fa.ngroups_dev = f.png.as<unsigned>() + src;
      if (fold_finish) {
        fa.ngroups_dev = nullptr;
        fa.nslots_dev = f.png.as<unsigned>() + 2;
      }
    }
    if (fused && ph + 1 < nphase) {   // ... and this one compacts into set ph & 1
      const int dst = ph & 1;
      fa.cq = f.pqF[dst].p;
      fa.ctlo = f.ptlo[dst].as<float>();
      fa.cthi = f.pthi[dst].as<float>();
      fa.cmap = f.pmap[dst].as<int>();
      fa.ccount = f.png.as<unsigned>() + 2;
      fa.ccap = (unsigned)nqpad;
      fa.route = f.route.as<uint8_t>();
    }
    const bool time_launch = ev_after_filter || opt(f, OPT_TIME_LAUNCHES);
    if (time_launch) {   // timed call: bracket the matrix kernel itself (the compaction kernels stay outside)
      while (f.kev.size() < f.kev_used + 2) {
        hipEvent_t e;
        CK(hipEventCreate(&e));
        f.kev.push_back(e);
      }
      CK(hipEventRecord(f.kev[f.kev_used], s));
    }
    const int narrow = (f.ks <= 4 && ph > 0) ? narrow_tail : 0;
    if (own_recheck) {   // the sweeping waves re-check their own segments; the first launch carries the ellipsoid band
      fa.append = 0;
      fa.seg_extra = 0;
      fa.rw.ell = EllExactArgs{};
      if (ph == 0 && f.ell_pending) {
        fa.rw.ell = f.ell_args;
        f.ell_pending = false;
      }
    }
    CK(launch_filter(f.ks, fa, out_idx != nullptr, s, narrow));
    if (time_launch) {
      CK(hipEventRecord(f.kev[f.kev_used + 1], s));
      f.kev_used += 2;
    }
    if (fa.cq && !fold_finish) {
      const int dst = ph & 1;
      launch_phase_finish(fa.cq, fa.ctlo, fa.cthi, fa.cmap, fa.ccount, f.png.as<unsigned>() + dst, f.ks, s);
      CK(hipGetLastError());
    }
  }
  if (ev_after_filter) CK(hipEventRecord(ev_after_filter, s));
  const int any_narrow = (f.ks <= 4 && nphase > 1) ? narrow_tail : 0;
  long long nsegs_all = filter_wave_count(f.ks, ngroups, any_narrow);
  if (nsegs_all < filter_wave_count(f.ks, ngroups)) nsegs_all = filter_wave_count(f.ks, ngroups);
  if (fa.split > 1) nsegs_all = (filter_wave_count(f.ks, ngroups) + 3) / 4 * 4 * fa.split;
  f.last_nsegs = (size_t)nsegs_all;
  }   // k_sweep / k_filter phases
  long long nsegs_all = f.last_nsegs;
  if (own_recheck || min_path) {
    // nothing left to do here: every sweeping wave has re-checked its own pairs
  } else if (xs) {   // no whitened coordinates were stored: the re-check whitens the queries of its pairs itself
    RecheckWArgs rw{};
    rw.list = f.list.as<unsigned long long>();
    rw.seg_cap = cap;
    rw.seg_count = f.segcnt.as<unsigned>();
    rw.nsegs = nsegs_all;
    rw.unit_cap = (unsigned)(nphase * (f.ks <= 4 ? 4 : (f.ks <= 8 ? 2 : 1)) * 32);   // queries per filter wave, all phases (a narrow later range has fewer)
    rw.refR = opR;
    rw.n = n;
    rw.d = d;
    rw.dp = dp;
    rw.pts = xs->pts;
    rw.nq = nq;
    rw.lay_ctr = xs->lay_ctr;
    rw.T64 = xs->T64;
    rw.r2 = r2;
    rw.best = f.best.as<int>();
    if (f.ell_pending) {   // the band proposals of k_prep4 are decided by the last waves of this launch (before the finalise)
      rw.ell = f.ell_args;
      f.ell_pending = false;
    }
    launch_recheck_whiten(rw, s);
    CK(hipGetLastError());
  } else {
    RecheckArgs ra{};
    ra.list = f.list.as<unsigned long long>();
    ra.seg_cap = cap;
    ra.seg_count = f.segcnt.as<unsigned>();
    ra.refR = opR;
    ra.n = n;
    ra.d = d;
    ra.dp = dp;
    ra.q = q;
    ra.ldq = ldq;
    ra.ldk = ldk;
    ra.nq = nq;
    ra.r2 = r2;
    ra.best = f.best.as<int>();
    launch_recheck(ra, nsegs_all, s);
    CK(hipGetLastError());
  }
  // answers of the filtered queries + the gate of the exact scan that follows: (a) queries that do not
  // fit binary16 and (b) every filtered query if the uncertain-pair list overflowed
  if (!xs)
    launch_filter_finalize(f.route.as<uint8_t>(), f.best.as<int>(), f.counters.as<unsigned>(), nq, out_mask,
                           out_idx, f.gate2.as<uint8_t>(), s);
  {
    ScanArgs a{};
    a.refT = refT;
    a.n = n;
    a.npad = npad;
    a.ntiles = npad / kWave;
    a.q = q;
    a.ldq = ldq;
    a.ldk = ldk;
    a.nq = nq;
    a.d = d;
    a.r2 = r2;
    a.mode = out_idx ? SCAN_FIRST : SCAN_MASK;
    a.gate = f.gate2.as<uint8_t>();
    a.out_idx = out_idx;
    a.out_mask = out_mask;
    a.only_gated = 1;   // leave the outputs of ungated queries alone
    if (xs) {   // the finalise work rides in the scan launch; its gate is the routing itself; the (rare) workgroups
                // that do scan whiten their own queries
      a.q = xs->pts;
      a.ldq = d;
      a.ldk = 1;
      a.raw_ctr = xs->lay_ctr;
      a.raw_T8 = xs->T8;
      a.raw_ldt = xs->ldt;
      a.route = f.route.as<uint8_t>();
      a.counters = f.counters.as<unsigned>();
      a.fin_best = f.best.as<int>();
      a.any_flag = f.misc.as<unsigned>() + 2 + f.batch_parity;
      a.fin_reset = f.misc.as<unsigned>() + 2 + (f.batch_parity ^ 1u);
      if (fold_finish) {
        a.fin_slots = f.png.as<unsigned>() + 2;
        a.fin_groups = f.png.as<unsigned>();   // where k_phase_finish would have left the group count (debug_stats)
        if (min_path) a.fin_slots2 = f.png.as<unsigned>() + (min3 ? 4 : 3);
        if (min3) a.fin_slots3 = f.png.as<unsigned>() + 3;
      }
    }
    CK(launch_scan(dp, a, s));
  }
  CK(hipGetLastError());
  // only a PHASED batch's scan tail (or its k_phase_finish) returns the slot counters to zero: a single-sweep batch that
  // follows a failed phased one must leave the flag standing for the next phased batch
  if (nphase > 1) f.png_dirty = false;
  return 0;
}

int scan_host(const double *apts, size_t na, const double *bpts, size_t nb, size_t d, double r2,
              int mode, int64_t *out) {
  if (int rc = check_dims(d)) return rc;
  if (nb == 0) return 0;
  if (!bpts || !out || (na && !apts)) return fail_arg(MLF_E_BADARG, "null pointer");
  if (na == 0) {  // the reference loops over zero live points
    for (size_t j = 0; j < nb; ++j) out[j] = mode == SCAN_FIRST ? -1 : 0;
    return 0;
  }
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  const int npad = round_up((int)na, kWave);
  if (int rc = stage_live_points(apts, na, d, dp, npad, false)) return rc;
  if (int rc = upload(c.q, bpts, nb * d * sizeof(double), c.stream)) return rc;
  CK(c.out.reserve(nb * sizeof(long long)));
  ScanArgs a{};
  a.refT = c.refT.as<double>();
  a.n = (int)na;
  a.npad = npad;
  a.ntiles = npad / kWave;
  a.q = c.q.as<double>();
  a.ldq = (long long)d;
  a.nq = (long long)nb;
  a.d = (int)d;
  a.r2 = r2;
  a.mode = mode;
  a.out_idx = c.out.as<long long>();
  bool filtered = false;
  // The stateless call pays the filter's live-point preparation every time and first-index mode keeps sweeping after
  // a hit, so the pre-filter only pays for larger batches than in the resident mask path (measured at N = 4000,
  // d = 50: exact scan 0.23-0.35 ms against 0.45-1.05 ms at 4000 queries; break-even between 16000 and 50000)
  const long long host_min = opt(c.filter, OPT_MIN_QUERIES) != kFilterMinQueriesDefault ? opt(c.filter, OPT_MIN_QUERIES) : 32768;
  if (mode == SCAN_FIRST && opt(c.filter, OPT_FILTER) && (long long)nb >= host_min && na >= 256) {
    if (int rc = filter_prepare_refs(c.filter, c.refR.as<double>(), (int)na, (int)d, dp, c.stream, true)) return rc;
    if (filter_applies(c.filter, (long long)nb, r2)) {
      if (int rc = filter_run(c.filter, c.refT.as<double>(), c.refR.as<double>(), (int)na, npad, (int)d, dp,
                              c.q.as<double>(), (long long)d, 1, (long long)nb, r2, nullptr, nullptr,
                              c.out.as<long long>(), c.stream, false))
        return rc;
      filtered = true;
    }
  }
  if (!filtered) CK(launch_scan(dp, a, c.stream));
  CK(hipMemcpyAsync(out, c.out.p, nb * sizeof(long long), hipMemcpyDefault, c.stream));   // host or device destination
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int prep_consts(DevBuf &ctr_b, DevBuf &mat_b, const double *ctr, const double *mat, int d, int dp,
                bool transpose, hipStream_t s) {
  std::vector<double> pc = pad_vector(ctr, d, dp);
  std::vector<double> pm = pad_matrix(mat, d, dp, transpose);
  if (int rc = upload(ctr_b, pc.data(), pc.size() * sizeof(double), s)) return rc;
  if (int rc = upload(mat_b, pm.data(), pm.size() * sizeof(double), s)) return rc;
  if (!arena_active()) CK(hipStreamSynchronize(s));  // host vectors go out of scope
  return 0;
}

}  // namespace

// ============================================================================================
struct mlf_region {
  bool ready = false;
  int n = 0, d = 0, dp = 0, npad = 0;
  int layer_kind = 0, use_scan = 1, live_space = 0;
  bool has_wrap = false;
  double enlarge = 0.0, r2 = 0.0;
  double live_extent_hint = -1.0;   // mlf_region_hint_live_extent, consumed by the next mlf_region_set
  HostArena arena;                  // pinned staging of the constants sent by mlf_region_set
  DevBuf refT, refR, lay_ctr, lay_mat, lay_T8, wrap, ell_ctr, ell_A, ell_Lt, ell_LtF, lay_TtF;
  bool chol_ready = false, chol_ok = false;
  double ell_eps_scale = 0.0;
  DevBuf tq, gate, pts, mask, row;
  // bounded per-proposal stage (mlf_prep4.hip): binary32 fragments, chain start values, error constants
  DevBuf p4_LtF, p4_TtF, p4_y0, lay_T64, ell_L;
  Prep4Consts p4c{};
  bool p4_ready = false;
  // "same quadratic form": A = T T^T + E with |E|_F measured (same_matrix) and c_lay == c_ell bit for bit (same_centres):
  // k_prep_sweep<.., true> reads delta^T A delta off the whitening chain; p4c_same = the constants of that form
  Prep4Consts p4c_same{};
  bool same_matrix = false, same_centres = false;
  std::vector<double> h_L, h_lay_ctr, h_ell_ctr;   // host copies: y0 = L^T (c_lay - c_ell) follows the ellipsoid centre
  FilterCtx filter;
  DevBuf gen, gen2, cube, smask, blk, sout, ax_zero, ax_mat, ax_pad;   // device-side sampling
  DevBuf s_invT, s_lo, s_hi, s_thin, s_count, rf_p, rf_L, rf_out, rf_aux, rf_keep;
  DevBuf s_invT_pad, s_tc, s_wc, s_thc, s_gate;   // t-space sampling: padded invT, survivors of the cheap tests (rows, cube rows, thinning draws)
  bool axes_ready = false, sampling_ready = false;
  std::vector<hipEvent_t> events;  // 4 per timed call
  size_t events_used = 0;
};

namespace {

float f32_up(double x) {
  float f = (float)x;
  if ((double)f < x) f = nextafterf(f, INFINITY);
  return f;
}

float f32_dn(double x) {
  float f = (float)x;
  if ((double)f > x) f = nextafterf(f, -INFINITY);
  return f;
}

// power of two s with s * amax in (2^(e-1), 2^e]
double pow2_scale(double amax, int e) {
  int ex = 0;
  std::frexp(amax, &ex);   // amax = m 2^ex, m in [0.5, 1)
  return std::ldexp(1.0, e - ex);
}

// k_prep4 subtracts the LAYER centre from every proposal; the ellipsoid form then starts its chain at
// y0 = L^T (c_lay - c_ell) (scaled like the accumulator: s_L s_x).  Recomputed whenever one of the two centres changes.
int region_prep4_centres(mlf_region *r, hipStream_t s) {
  const int d = r->d;
  const std::vector<double> &L = r->h_L;
  std::vector<double> s0((size_t)d);
  double s0n2 = 0.0, y0n2 = 0.0;
  for (int k = 0; k < d; ++k) {
    s0[k] = r->h_lay_ctr[k] - r->h_ell_ctr[k];
    s0n2 += s0[k] * s0[k];
  }
  const double acc_scale = 1.0 / (double)r->p4c.inv_sl_sx;
  std::vector<float> y0f((size_t)32 * ((r->dp + 31) / 32), 0.0f);
  for (int i = 0; i < d; ++i) {
    double y = 0.0;
    for (int k = i; k < d; ++k) y += L[(size_t)k * d + i] * s0[k];
    y0n2 += y * y;
    y0f[i] = (float)(y * acc_scale);
  }
  if (!std::isfinite(s0n2) || !std::isfinite(y0n2) || std::sqrt(y0n2) * acc_scale > 1e30) {
    r->p4_ready = false;
    return 0;
  }
  r->same_centres = s0n2 == 0.0;   // every difference an exact zero
  r->p4c.s0n = f32_up(std::sqrt(s0n2) * (1.0 + 1e-12));
  r->p4c.y0n = f32_up(std::sqrt(y0n2) * (1.0 + 1e-12));
  if (int rc = upload(r->p4_y0, y0f.data(), y0f.size() * sizeof(float), s)) return rc;
  if (!arena_active()) CK(hipStreamSynchronize(s));
  return 0;
}

// Fragments and error constants of the bounded per-proposal stage.  L: lower Cholesky factor of the ellipsoid matrix,
// fro2 = |A|_F^2; layer_T / layer_ctr may be null for regions without a neighbour scan; `live` = the cube-space live
// points (n x d) or null: their spread around the layer centre fixes the scale of the binary16 proposal operand.
int region_prep4_setup(mlf_region *r, const std::vector<double> &L, double fro2, const double *ell_center,
                       const double *layer_ctr, const double *layer_T, const double *live, size_t nlive, hipStream_t s,
                       const double *ell_invcov) {
  r->p4_ready = false;
  r->same_matrix = r->same_centres = false;
  const int d = r->d, dp = r->dp;
  if (!prep4_usable(d) || (dp & 1) || dp > 64 || !r->chol_ok || r->has_wrap) return 0;
  if (r->use_scan && (r->layer_kind != 0 || !layer_T || !layer_ctr)) return 0;
  double lf2 = 0.0, lmax = 0.0, dmin = INFINITY;
  for (int i = 0; i < d; ++i)
    for (int k = 0; k <= i; ++k) {
      const double v = L[(size_t)i * d + k];
      lf2 += v * v;
      lmax = std::fmax(lmax, std::fabs(v));
      if (k == i) dmin = std::fmin(dmin, v);
    }
  if (!std::isfinite(lf2) || !(lmax > 0.0) || !(lmax < 1e100) || !(dmin > 0.0)) return 0;
  const int nsteps = 3 * ((dp + 15) / 16);   // matrix instructions per output chain
  const int kdim = 16 * ((dp + 15) / 16);
  const double g = (4.0 * nsteps + 8.0) * std::ldexp(1.0, -24) * (1.0 + std::ldexp(1.0, -8)) + std::pow(2.0, -21.6) +
                   std::pow(2.0, -21.9);
  const double lf = std::sqrt(lf2);
  // share of the proposals near the boundary that the split-binary16 chain cannot decide ~ d g |L|_F / sigma_min(L):
  // beyond a few per cent the binary64 test behind it would dominate, the binary64 stage (k_prep3) is used instead
  if (d * g * lf / dmin > 0.02) return 0;
  // scale of the proposal operand: the live points' largest centred coordinate lands in (16, 32]; a region without
  // live points uses the ellipsoid's extent, 1 / (smallest diagonal entry of L) being a bound on its semi-axes' scale
  const double *ctr = r->use_scan ? layer_ctr : ell_center;
  double amax = 0.0;
  if (live) {   // four running maxima (a NaN never wins a comparison, as with fmax): one chain of dependent maxima cost 0.1 ms at N = 4000, d = 50
    double m[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t i = 0; i < nlive; ++i) {
      const double *row = live + i * d;
      int k = 0;
      for (; k + 4 <= d; k += 4)
        for (int q = 0; q < 4; ++q) {
          const double v = std::fabs(row[k + q] - ctr[k + q]);
          m[q] = v > m[q] ? v : m[q];
        }
      for (; k < d; ++k) {
        const double v = std::fabs(row[k] - ctr[k]);
        m[0] = v > m[0] ? v : m[0];
      }
    }
    amax = std::fmax(std::fmax(m[0], m[1]), std::fmax(m[2], m[3]));
  }
  if (!(amax > 0.0) || !std::isfinite(amax)) amax = 4.0 / dmin;
  if (!(amax > 1e-60) || !(amax < 1e60)) return 0;
  const double sx = pow2_scale(amax, 5);
  const double sl = pow2_scale(lmax, 8);            // largest |s_L L| entry in (128, 256]
  Prep4Consts &c = r->p4c;
  c = Prep4Consts{};
  c.g_chain = f32_up(g);
  c.lf = f32_up(lf * (1.0 + 1e-12));
  c.eps_scale = f32_up(std::ldexp(1.0, -34) * std::sqrt(fro2) * (1.0 + 1e-12));
  c.s_x = (float)sx;
  c.inv_sx = (float)(1.0 / sx);
  c.inv_sl_sx = (float)(1.0 / (sl * sx));
  c.l_abs = f32_up(lf * std::sqrt((double)kdim) * std::ldexp(1.0, -25) / sx * (1.0 + 1e-12));
  if (!(c.inv_sl_sx > 0.0f) || !std::isfinite(1.0f / c.inv_sl_sx) || !(c.inv_sx > 0.0f)) return 0;
  r->h_L = L;
  r->h_ell_ctr.assign(ell_center, ell_center + d);
  r->h_lay_ctr.assign(ctr, ctr + d);
  {   // the lower factor itself, row-major with stride dp (the wave-per-proposal exact test reads its columns)
    std::vector<double> lrm((size_t)dp * dp, 0.0);
    for (int j = 0; j < d; ++j)
      for (int k = 0; k <= j; ++k) lrm[(size_t)j * dp + k] = L[(size_t)j * d + k];
    if (int rc = upload(r->ell_L, lrm.data(), lrm.size() * sizeof(double), s)) return rc;
    if (!arena_active()) CK(hipStreamSynchronize(s));
  }
  std::vector<uint16_t> ltf(prep4_ltf_count(dp));
  const double el = prep4_lt_fragments(L.data(), d, dp, sl, ltf.data());
  c.el = f32_up(el / sl * (1.0 + 1e-12));
  if (int rc = upload(r->p4_LtF, ltf.data(), ltf.size() * sizeof(uint16_t), s)) return rc;
  if (r->use_scan) {
    double tf2 = 0.0, tmax = 0.0, cmin = INFINITY, cmax = 0.0;
    for (int cc = 0; cc < d; ++cc) {
      double cn = 0.0;
      for (int k = 0; k < d; ++k) {
        const double v = layer_T[(size_t)k * d + cc];
        cn += v * v;
        tmax = std::fmax(tmax, std::fabs(v));
      }
      tf2 += cn;
      cmin = std::fmin(cmin, cn);
      cmax = std::fmax(cmax, cn);
    }
    if (!std::isfinite(tf2) || !(tmax > 0.0) || !(tmax < 1e100) || !(cmin > 0.0)) return 0;
    // zeta against the binary16 term of Delta: g sqrt(d) cond(T) < 2^-11, or the uncertainty band of the filter more than
    // doubles (T = eigenvectors x diag: the column norms are its singular values)
    if (g * std::sqrt((double)d) * std::sqrt(cmax / cmin) * 2048.0 > 1.0) return 0;
    const double st = pow2_scale(tmax, 8);
    const double tf = std::sqrt(tf2);
    std::vector<uint16_t> ttf(prep4_ttf_count(dp));
    const double et = prep4_t_fragments(layer_T, d, dp, st, ttf.data());
    c.zt = f32_up((g * tf + et / st) * (1.0 + 1e-12));
    c.zt_abs = f32_up(tf * std::sqrt((double)kdim) * std::ldexp(1.0, -25) / sx * (1.0 + 1e-12));
    c.inv_st_sx = (float)(1.0 / (st * sx));
    if (!(c.inv_st_sx > 0.0f) || !std::isfinite(1.0f / c.inv_st_sx)) return 0;
    if (int rc = upload(r->p4_TtF, ttf.data(), ttf.size() * sizeof(uint16_t), s)) return rc;
    std::vector<double> t64((size_t)64 * 64, 0.0);   // for the exact whitening inside the re-check
    for (int k = 0; k < d; ++k)
      for (int c2 = 0; c2 < d; ++c2) t64[(size_t)k * 64 + c2] = layer_T[(size_t)k * d + c2];
    if (int rc = upload(r->lay_T64, t64.data(), t64.size() * sizeof(double), s)) return rc;
    // Same quadratic form?  E = sym(A) - T T^T in binary64 (the reference's einsum sees delta^T A delta = delta^T sym(A) delta);
    // |delta^T E delta| <= |E|_F |delta|^2 joins eps.  The residue is computed with rounding errors of its own: every entry of
    // T T^T is a d-term dot product (error <= d 2^-53 (|T| |T|^T)_ij, in the Frobenius norm <= d 2^-53 |T|_F^2), the
    // symmetrisation and the difference add 2^-52 |A|_F.  Accepted while the enlarged eps stays below twice the old one.
    if (ell_invcov) {
      double e2 = 0.0;
      for (int i = 0; i < d; ++i)
        for (int j = 0; j <= i; ++j) {
          double pij = 0.0;
          for (int cc = 0; cc < d; ++cc) pij += layer_T[(size_t)i * d + cc] * layer_T[(size_t)j * d + cc];
          const double e = 0.5 * (ell_invcov[(size_t)i * d + j] + ell_invcov[(size_t)j * d + i]) - pij;
          e2 += (i == j ? 1.0 : 2.0) * e * e;
        }
      const double afro = std::sqrt(fro2);
      const double e_bound = (std::sqrt(e2) + (d + 4.0) * std::ldexp(1.0, -52) * (tf2 + afro)) * (1.0 + 1e-12);
      if (std::isfinite(e_bound) && e_bound <= std::ldexp(1.0, -34) * afro) {
        Prep4Consts &q = r->p4c_same;
        q = c;
        q.y0n = 0.0f;                                   // the whitening chain starts at zero
        q.lf = f32_up(tf * (1.0 + 1e-12));              // eta = g |T|_F |delta| + |E_T|_F |delta| / s_T + |T|_F sqrt(K) 2^-25 / s_x
        q.el = f32_up(et / st * (1.0 + 1e-12));
        q.l_abs = c.zt_abs;
        q.s0n = 0.0f;                                   // (same_centres)
        q.eps_scale = f32_up((std::ldexp(1.0, -34) * afro + e_bound) * (1.0 + 1e-12));
        q.inv_sl_sx = c.inv_st_sx;                      // accumulator of the whitening chain -> T^T delta
        r->same_matrix = true;
      }
    }
  }
  if (!arena_active()) CK(hipStreamSynchronize(s));
  r->p4_ready = true;
  return region_prep4_centres(r, s);
}

// whiten `n` cube-space rows already on the device with the region's own layer (same kernels and
// arithmetic as for proposals, so a live point is at distance exactly 0 from itself)
int region_whiten_rows(mlf_region *r, const double *d_u, size_t n, double *d_t, hipStream_t s) {
  if (r->layer_kind == 0 && r->dp <= 64) {   // a few thousand rows at most: the wave-per-8-rows form of the same chain
    CK(launch_whiten_rows(d_u, (long long)n, r->d, r->dp, r->lay_ctr.as<double>(), r->lay_T8.as<double>(), (r->dp + 7) / 8 * 8,
                          r->has_wrap ? r->wrap.as<double>() : nullptr, d_t, r->d, s));
  } else if (r->layer_kind == 0) {
    PrepArgs pa{};
    pa.pts = d_u;
    pa.np = (long long)n;
    pa.d = r->d;
    pa.do_tr = 1;
    pa.lay_ctr = r->lay_ctr.as<double>();
    pa.lay_Tt = r->lay_mat.as<double>();
    pa.wrap_shift = r->has_wrap ? r->wrap.as<double>() : nullptr;
    pa.t_out = d_t;
    pa.ldt = r->d;
    CK(launch_prep(r->dp, pa, s));
  } else {
    launch_scaling_transform(d_u, (long long)n, r->d, r->lay_ctr.as<double>(), r->lay_mat.as<double>(),
                             r->has_wrap ? r->wrap.as<double>() : nullptr, nullptr, d_t, r->d, s);
    CK(hipGetLastError());
  }
  return 0;
}

int region_inside_enqueue(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                          hipStream_t s, hipEvent_t *ev /* 4 events or null */,
                          long long *d_idx = nullptr, const uint8_t *pregate = nullptr) {
  if (np == 0) return 0;
  if (d_idx && !r->use_scan) return fail_arg(MLF_E_STATE, "region has no live points set");
  if (r->use_scan)
    if (int rc = filter_refresh_refs(r->filter, r->refR.as<double>(), r->n, r->d, r->dp, s)) return rc;
  CK(r->gate.reserve(np));
  uint8_t *gate = r->use_scan ? r->gate.as<uint8_t>() : d_mask;
  const bool use_filter = r->use_scan && filter_applies(r->filter, (long long)np, r->r2);
  // fused stage (coalesced staging, coordinate-major output, optional quantisation) for affine layers
  const bool fused = r->layer_kind == 0 && opt(r->filter, OPT_FUSED_PREP) && prep3_usable(r->d) && r->chol_ready;
  long long ldq = r->d, ldk = 1;
  const bool bounded = fused && opt(r->filter, OPT_PREP_BOUNDED) && r->p4_ready && !r->has_wrap && (use_filter || !r->use_scan) &&
                       np < (size_t)0x7fffffff && (reinterpret_cast<uintptr_t>(d_pts) & 15) == 0;   // 16-byte pieces
  ExactSrc xsrc{};
  if (r->use_scan && !bounded) CK(r->tq.reserve(np * (size_t)r->d * sizeof(double)));
  if (ev) CK(hipEventRecord(ev[0], s));
  // The batch sizes of a real run (ndraw 128 ... 65536): per-proposal stage, sweep, re-check and answers in ONE launch
  // (mlf_mid.hip); the exact-scan launch behind it only works if a proposal was routed to it.
  const bool mid = bounded && r->use_scan && use_filter && !pregate && !d_idx && r->filter.ks <= 4 && mid_usable(r->dp) &&
                   (long long)np <= opt(r->filter, OPT_MID_MAX);
  if (mid) {
    FilterCtx &f = r->filter;
    if (int rc = misc_reserve(f)) return rc;
    CK(f.route.reserve(np));
    CK(f.counters.reserve(4 * sizeof(unsigned)));
    const long long ngroups = ((long long)np + 31) / 32, nsets = (ngroups + 3) / 4;
    const int ny = mid_range_quads(ngroups, f.ntiles32), R = 4 * ny;
    CK(f.mid_rec.reserve((size_t)nsets * R * 3 * sizeof(unsigned long long)));
    CK(f.mid_meta.reserve((size_t)nsets * 4 * sizeof(unsigned long long)));
    {
      const size_t before = f.mid_arrive.cap;
      CK(f.mid_arrive.reserve((size_t)nsets * sizeof(unsigned)));
      // they return to zero by themselves afterwards -- unless a launch failed part-way: then the next batch starts clean
      if (f.mid_arrive.cap != before || f.mid_dirty) CK(hipMemsetAsync(f.mid_arrive.p, 0, f.mid_arrive.cap, s));
      if (f.mid_dirty && f.misc.p) CK(hipMemsetAsync(f.misc.p, 0, 8 * sizeof(unsigned), s));
      f.mid_dirty = false;
    }
    MidArgs ma{};
    ma.pts = d_pts;
    ma.np = (long long)np;
    ma.d = r->d;
    ma.dp = r->dp;
    ma.ks = f.ks;
    ma.LtF = r->p4_LtF.p;
    ma.y0 = r->p4_y0.as<float>();
    ma.TtF = r->p4_TtF.p;
    ma.lay_ctr = r->lay_ctr.as<double>();
    ma.c = r->p4c;
    ma.c.enl_lo = f32_dn(r->enlarge);
    ma.c.enl_hi = f32_up(r->enlarge);
    ma.stats = f.stats.as<double>();
    ma.r2 = r->r2;
    ma.ell_ctr = r->ell_ctr.as<double>();
    ma.ell_L = r->ell_L.as<double>();
    ma.ell_A = r->ell_A.as<double>();
    ma.ell_eps_scale = r->ell_eps_scale;
    ma.enlarge = r->enlarge;
    ma.chol_ok = r->chol_ok ? 1 : 0;
    const bool use_m = f.ordered && opt(f, OPT_ORDER);
    ma.refF = use_m ? f.refFm.p : f.refF.p;
    ma.ntiles32 = f.ntiles32;
    ma.refR = use_m ? f.refRm.as<double>() : r->refR.as<double>();
    ma.n = r->n;
    ma.T64 = r->lay_T64.as<double>();
    ma.rec = f.mid_rec.as<unsigned long long>();
    ma.meta = f.mid_meta.as<unsigned long long>();
    ma.arrive = f.mid_arrive.as<unsigned>();
    ma.mask = d_mask;
    ma.route = f.route.as<uint8_t>();
    f.batch_parity ^= 1u;
    ma.scan_flag = f.misc.as<unsigned>() + 2 + f.batch_parity;
    ma.counters = f.counters.as<unsigned>();
    if (opt(f, OPT_TIME_LAUNCHES)) {   // diagnostics: stage stamps of workgroup (0, 0) (mlf_region_debug_stats)
      CK(f.segcnt.reserve(16 * sizeof(unsigned)));
      ma.stamps = f.segcnt.as<unsigned>();
    }
    f.mid_last = ma.stamps != nullptr;
    if (hipError_t le = launch_inside_mid(ma, ny, s); le != hipSuccess) {
      f.mid_dirty = true;
      CK(le);
    }
    if (ev) {
      CK(hipEventRecord(ev[1], s));
      CK(hipEventRecord(ev[2], s));
    }
    ScanArgs a{};   // proposals the pre-filter cannot take (route 2): the exact scan, idle unless the batch's flag is up
    a.refT = r->refT.as<double>();
    a.n = r->n;
    a.npad = r->npad;
    a.ntiles = r->npad / kWave;
    a.q = d_pts;
    a.ldq = r->d;
    a.ldk = 1;
    a.nq = (long long)np;
    a.d = r->d;
    a.r2 = r->r2;
    a.mode = SCAN_MASK;
    a.out_mask = d_mask;
    a.only_gated = 1;
    a.raw_ctr = r->lay_ctr.as<double>();
    a.raw_T8 = r->lay_T8.as<double>();
    a.raw_ldt = (r->dp + 7) / 8 * 8;
    a.route = f.route.as<uint8_t>();
    a.counters = f.counters.as<unsigned>();
    a.any_flag = f.misc.as<unsigned>() + 2 + f.batch_parity;
    a.fin_reset = f.misc.as<unsigned>() + 2 + (f.batch_parity ^ 1u);
    if (hipError_t le = launch_scan(r->dp, a, s); le != hipSuccess) {
      f.mid_dirty = true;
      CK(le);
    }
    f.last_nsegs = 0;
    if (ev) CK(hipEventRecord(ev[3], s));
    return 0;
  }
  if (bounded) {   // matrix cores (split binary16): bounded ellipsoid test + approximate whitening straight into the filter operand
    FilterCtx &f = r->filter;
    if (int rc = misc_reserve(f)) return rc;
    CK(f.ell_list.reserve(np * sizeof(int)));
    Prep4Args pa{};
    pa.pts = d_pts;
    pa.np = (long long)np;
    pa.d = r->d;
    pa.dp = r->dp;
    pa.LtF = r->p4_LtF.p;
    pa.y0 = r->p4_y0.as<float>();
    pa.TtF = r->p4_TtF.p;
    pa.lay_ctr = r->use_scan ? r->lay_ctr.as<double>() : r->ell_ctr.as<double>();
    pa.c = r->p4c;
    pa.c.enl_lo = f32_dn(r->enlarge);
    pa.c.enl_hi = f32_up(r->enlarge);
    pa.gate = gate;
    pa.do_tr = r->use_scan ? 1 : 0;
    pa.ell_count = f.misc.as<unsigned>();
    pa.ell_list = f.ell_list.as<int>();
    pa.ell_cap = (unsigned)np;
    if (r->use_scan) {
      unsigned cap = 0;
      if (int rc = filter_reserve(f, (long long)np, &cap)) return rc;
      pa.stats = f.stats.as<double>();
      pa.r2 = r->r2;
      pa.qF = f.qF.p;
      pa.tlo = f.tlo.as<float>();
      pa.thi = f.thi.as<float>();
      pa.route = f.route.as<uint8_t>();
      pa.best = f.best.as<int>();
      pa.counters = f.counters.as<unsigned>();
      f.batch_parity ^= 1u;
      pa.scan_flag = f.misc.as<unsigned>() + 2 + f.batch_parity;
      pa.ks = f.ks;
      pa.nqpad = ((long long)np + 31) / 32 * 32;
      xsrc.pts = d_pts;
      xsrc.lay_ctr = r->lay_ctr.as<double>();
      xsrc.T8 = r->lay_T8.as<double>();
      xsrc.ldt = (r->dp + 7) / 8 * 8;
      xsrc.T64 = r->lay_T64.as<double>();
    }
    // large batches on the min-only path: the per-proposal stage runs inside the first sweep launch (mlf_fused.hip)
    const bool defer_prep = r->use_scan && use_filter && !d_idx && opt(f, OPT_FUSED_FIRST) && fused_usable(r->dp) &&
                            filter_takes_min_path(f, (long long)np);
    static thread_local Prep4Args deferred;
    static thread_local Prep4Consts deferred_same;
    if (defer_prep) {
      deferred = pa;
      xsrc.prep = &deferred;
      if (pregate) {   // the fused first launch reads the pre-gate out of the gate array and writes its own verdict over it
        CK(hipMemcpyAsync(gate, pregate, np, hipMemcpyDeviceToDevice, s));
        xsrc.pregated = true;
        pregate = nullptr;   // honoured there: no k_apply_pregate, and the ellipsoid band rides in the re-check launch as usual
      }
      if (r->same_matrix && r->same_centres) {
        deferred_same = r->p4c_same;
        deferred_same.enl_lo = pa.c.enl_lo;
        deferred_same.enl_hi = pa.c.enl_hi;
        xsrc.same = &deferred_same;
      }
    } else {
      CK(launch_prep4(pa, s));
    }
    EllExactArgs ea{};
    ea.count = f.misc.as<unsigned>();
    ea.done = f.misc.as<unsigned>() + 1;
    ea.last = f.misc.as<unsigned>() + 4;
    ea.list = f.ell_list.as<int>();
    ea.cap = (unsigned)np;
    ea.pts = d_pts;
    ea.d = r->d;
    ea.dp = r->dp;
    ea.ell_ctr = r->ell_ctr.as<double>();
    ea.ell_Lt = r->ell_Lt.as<double>();
    ea.ell_L = r->ell_L.as<double>();
    ea.ell_A = r->ell_A.as<double>();
    ea.eps_scale = r->ell_eps_scale;
    ea.enlarge = r->enlarge;
    ea.chol_ok = r->chol_ok ? 1 : 0;
    ea.gate = gate;
    ea.route = r->use_scan ? f.route.as<uint8_t>() : nullptr;
    if (r->use_scan && !pregate) {   // decided by the tail of the re-check launch (k_recheck_whiten)
      f.ell_args = ea;
      f.ell_pending = true;
    } else {
      launch_ell_exact(ea, s);
      CK(hipGetLastError());
    }
  } else if (fused) {   // FP64 matrix-core version of the fused stage (v_mfma_f64_16x16x4_f64: the reference's FMA chain bit for bit)
    Prep3Args pa{};
    pa.pts = d_pts;
    pa.np = (long long)np;
    pa.d = r->d;
    pa.dp = r->dp;
    pa.nk = (r->d + 3) / 4;
    pa.ell_ctr = r->ell_ctr.as<double>();
    pa.ell_A = r->ell_A.as<double>();
    pa.lda = r->dp;
    pa.LtF = r->ell_LtF.as<double>();
    pa.ell_eps_scale = r->ell_eps_scale;
    pa.chol_ok = r->chol_ok ? 1 : 0;
    pa.enlarge = r->enlarge;
    pa.gate = gate;
    if (r->use_scan) {
      pa.do_tr = 1;
      pa.lay_ctr = r->lay_ctr.as<double>();
      pa.TtF = r->lay_TtF.as<double>();
      pa.wrap_shift = r->has_wrap ? r->wrap.as<double>() : nullptr;
      pa.t_out = r->tq.as<double>();
      pa.t_ldq = 1;
      pa.t_ldk = (long long)np;
      ldq = 1;
      ldk = (long long)np;
      if (use_filter) {   // the exact re-check gathers single proposals: rows of d contiguous doubles
        pa.t_ldq = ldq = (long long)r->d;
        pa.t_ldk = ldk = 1;
      }
      if (use_filter) {
        unsigned cap = 0;
        FilterCtx &f = r->filter;
        if (int rc = filter_reserve(f, (long long)np, &cap)) return rc;
        pa.qF = f.qF.p;
        pa.tlo = f.tlo.as<float>();
        pa.thi = f.thi.as<float>();
        pa.route = f.route.as<uint8_t>();
        pa.best = f.best.as<int>();
        pa.counters = f.counters.as<unsigned>();
        pa.stats = f.stats.as<double>();
        pa.r2 = r->r2;
        pa.ks = f.ks;
        pa.nqpad = ((long long)np + 31) / 32 * 32;
      }
    }
    CK(launch_prep3(pa, s));
  } else if (r->layer_kind == 0 && prep64_usable(r->d) && r->chol_ready && r->chol_ok && opt(r->filter, OPT_FUSED_PREP)) {
    // 65 ... 128 dimensions: the bounded quadratic form and the whitening chain on the FP64 matrix cores, the matrices streamed
    // from L2 (mlf_prep64.hip; rounds 1-4: the vector kernel below, 4-6 ms per 10^6 x 100)
    Prep64Args pa{};
    pa.pts = d_pts;
    pa.np = (long long)np;
    pa.d = r->d;
    pa.dp = r->dp;
    pa.ell_ctr = r->ell_ctr.as<double>();
    pa.ell_L = r->ell_L.as<double>();
    pa.ell_A = r->ell_A.as<double>();
    pa.lda = r->dp;
    pa.ell_eps_scale = r->ell_eps_scale;
    pa.chol_ok = 1;
    pa.enlarge = r->enlarge;
    pa.gate = gate;
    if (r->use_scan) {
      pa.do_tr = 1;
      pa.lay_ctr = r->lay_ctr.as<double>();
      pa.T8 = r->lay_T8.as<double>();
      pa.ldt8 = (r->dp + 7) / 8 * 8;
      pa.wrap_shift = r->has_wrap ? r->wrap.as<double>() : nullptr;
      pa.t_out = r->tq.as<double>();
      pa.ldt = r->d;
    }
    CK(launch_prep64(pa, s));
  } else {
    PrepArgs pa{};
    pa.pts = d_pts;
    pa.np = (long long)np;
    pa.d = r->d;
    pa.do_ell = 1;
    pa.ell_ctr = r->ell_ctr.as<double>();
    pa.ell_A = r->ell_A.as<double>();
    pa.enlarge = r->enlarge;
    pa.mask = gate;
    pa.q_out = nullptr;
    if (r->use_scan) {
      pa.t_out = r->tq.as<double>();
      pa.ldt = r->d;
      if (r->layer_kind == 0) {
        pa.do_tr = 1;
        pa.lay_ctr = r->lay_ctr.as<double>();
        pa.lay_Tt = r->lay_mat.as<double>();
        pa.wrap_shift = r->has_wrap ? r->wrap.as<double>() : nullptr;
      }
    }
    CK(launch_prep(r->dp, pa, s));
    if (r->use_scan && r->layer_kind == 1) {
      launch_scaling_transform(d_pts, (long long)np, r->d, r->lay_ctr.as<double>(), r->lay_mat.as<double>(),
                               r->has_wrap ? r->wrap.as<double>() : nullptr, gate, r->tq.as<double>(),
                               r->d, s);
      CK(hipGetLastError());
    }
  }
  if (pregate) {   // proposals rejected before the region test (outside the unit cube): never scanned
    const bool ff = fused && use_filter;
    launch_apply_pregate(pregate, (long long)np, gate, ff ? r->filter.route.as<uint8_t>() : nullptr,
                         ff ? r->filter.tlo.as<float>() : nullptr, ff ? r->filter.thi.as<float>() : nullptr, s);
    CK(hipGetLastError());
  }
  if (ev) CK(hipEventRecord(ev[1], s));
  if (use_filter) {
    if (int rc = filter_run(r->filter, r->refT.as<double>(), r->refR.as<double>(), r->n, r->npad, r->d, r->dp,
                            r->tq.as<double>(), ldq, ldk, (long long)np, r->r2, gate, d_idx ? nullptr : d_mask,
                            d_idx, s, fused, ev ? ev[2] : nullptr, bounded ? &xsrc : nullptr))
      return rc;
  } else if (r->use_scan) {
    ScanArgs a{};
    a.refT = r->refT.as<double>();
    a.n = r->n;
    a.npad = r->npad;
    a.ntiles = r->npad / kWave;
    a.q = r->tq.as<double>();
    a.ldq = ldq;
    a.ldk = ldk;
    a.nq = (long long)np;
    a.d = r->d;
    a.r2 = r->r2;
    a.mode = d_idx ? SCAN_FIRST : SCAN_MASK;
    a.gate = gate;
    a.out_mask = d_mask;
    a.out_idx = d_idx;
    CK(launch_scan(r->dp, a, s));
    if (ev) CK(hipEventRecord(ev[2], s));
    if (d_idx) {
      launch_mark_gated(gate, (long long)np, d_idx, s);
      CK(hipGetLastError());
    }
  }
  if (ev) {
    if (!r->use_scan) CK(hipEventRecord(ev[2], s));
    CK(hipEventRecord(ev[3], s));
  }
  return 0;
}

}  // namespace

namespace mlf {
int ctx_ensure() { return ensure_ctx(); }
hipStream_t ctx_stream() { return g_ctx.stream; }
int ctx_fail_hip(hipError_t e, const char *what, const char *file, int line) {
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
  g_err = buf;
  return -(int)e;
}
int ctx_fail_arg(int code, const char *msg) { return fail_arg(code, msg); }
}  // namespace mlf

extern "C" {

int mlf_abi_version(void) { return MLF_ABI_VERSION; }

const char *mlf_last_error(void) { return g_err.c_str(); }

int mlf_device_count(int *count) {
  if (!count) return fail_arg(MLF_E_BADARG, "null pointer");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  *count = (e == hipSuccess) ? c : 0;
  return 0;
}

int mlf_set_device(int device) {
  if (g_ctx.ready && device != g_ctx.device)
    return fail_arg(MLF_E_STATE, "mlf_set_device must be called before the first compute call");
  g_ctx.device = device;
  return 0;
}

int mlf_device_name(char *buf, size_t buflen) {
  if (!buf || !buflen) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, g_ctx.device));
  snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  return 0;
}

int mlf_set_option(const char *name, long long value) {
  if (!name) return fail_arg(MLF_E_BADARG, "null pointer");
  const int id = opt_id(name);
  if (id < 0) return fail_arg(MLF_E_BADARG, "unknown option");
  g_opt[id] = opt_clamp(id, value);
  return 0;
}

int mlf_get_option(const char *name, long long *value) {
  if (!name || !value) return fail_arg(MLF_E_BADARG, "null pointer");
  const int id = opt_id(name);
  if (id < 0) return fail_arg(MLF_E_BADARG, "unknown option");
  *value = g_opt[id];
  return 0;
}

int mlf_region_get_option(mlf_region *r, const char *name, long long *value) {
  if (!r || !name || !value) return fail_arg(MLF_E_BADARG, "null pointer");
  const int id = opt_id(name);
  if (id < 0) return fail_arg(MLF_E_BADARG, "unknown option");
  *value = opt(r->filter, id);
  return 0;
}

int mlf_debug_forget_grants(unsigned long long *grants_so_far) {
  if (grants_so_far) *grants_so_far = g_grant_calls.load(std::memory_order_relaxed);
  g_grant_epoch.fetch_add(1u, std::memory_order_acq_rel);
  return 0;
}

int mlf_option_name(int index, char *buf, size_t buflen) {
  if (!buf || !buflen || index < 0 || index >= OPT_COUNT) return fail_arg(MLF_E_BADARG, "no such option");
  snprintf(buf, buflen, "%s", kOptNames[index]);
  return 0;
}

int mlf_synchronize(void) {
  if (int rc = ensure_ctx()) return rc;
  CK(hipDeviceSynchronize());
  return 0;
}

// ---- device memory for callers that keep arrays resident between calls (ultranest_amd.device_rebuild) -------------------
int mlf_dev_alloc(size_t bytes, void **out) {
  if (!out) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  CK(hipMalloc(out, bytes ? bytes : 1));
  return 0;
}

int mlf_dev_free(void *p) {
  if (!p) return 0;
  if (int rc = ensure_ctx()) return rc;
  CK(hipFree(p));
  return 0;
}

// dst / src: host or device (unified addressing picks the direction); ordered on the library's stream; `sync` != 0 waits
// for it (needed before a host destination is read or a host source is re-used)
int mlf_dev_copy(void *dst, const void *src, size_t bytes, int sync) {
  if (bytes && (!dst || !src)) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  if (bytes) CK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, c.stream));
  if (sync) CK(hipStreamSynchronize(c.stream));
  return 0;
}

// lo[c], hi[c] = extents of column c of the (n, d) row-major array `pts` (host or device); lo / hi on the host
int mlf_col_extent(const double *pts, size_t n, size_t d, double *lo, double *hi) {
  if (int rc = check_dims(d)) return rc;
  if (d > 128) return fail_arg(MLF_E_DIM, "mlf_col_extent covers up to 128 columns (the device-resident rebuild it serves: d <= 64)");
  if (!pts || !lo || !hi || n == 0) return fail_arg(MLF_E_BADARG, "null pointer or no rows");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const double *dp = pts;
  if (!is_device_pointer(pts)) {
    if (int rc = upload(c.src, pts, n * d * sizeof(double), c.stream)) return rc;
    dp = c.src.as<double>();
  }
  CK(c.small2.reserve((size_t)(kExtentScratchBlocks + 1) * 2 * d * sizeof(double)));
  double *part = c.small2.as<double>();
  double *out = part + (size_t)kExtentScratchBlocks * 2 * d;
  launch_col_extent(dp, (int)n, (int)d, part, out, c.stream);
  CK(hipGetLastError());
  std::vector<double> h(2 * d);
  CK(hipMemcpyAsync(h.data(), out, 2 * d * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  memcpy(lo, h.data(), d * sizeof(double));
  memcpy(hi, h.data() + d, d * sizeof(double));
  return 0;
}

// ------------------------------------------------------------------------------ K1 / K2 ----
int mlf_find_nearby(const double *apts, size_t na, const double *bpts, size_t nb, size_t d,
                    double radiussq, int64_t *out) {
  return scan_host(apts, na, bpts, nb, d, radiussq, SCAN_FIRST, out);
}

int mlf_count_nearby(const double *apts, size_t na, const double *bpts, size_t nb, size_t d,
                     double radiussq, int64_t *out) {
  return scan_host(apts, na, bpts, nb, d, radiussq, SCAN_COUNT, out);
}

// ------------------------------------------------------------------------------ K3 ---------
int mlf_subtract_nearby(const double *pts, size_t n, size_t d, double radiussq, double *out) {
  if (int rc = check_dims(d)) return rc;
  if (n == 0) return 0;
  if (!pts || !out) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  const int npad = round_up((int)n, kWave);
  const int ntiles = npad / kWave;
  if (int rc = stage_live_points(pts, n, d, dp, npad, false)) return rc;
  CK(c.flags.reserve(n * (size_t)ntiles * sizeof(unsigned long long)));
  CK(c.out.reserve(n * d * sizeof(double)));
  ScanArgs a{};
  a.refT = c.refT.as<double>();
  a.n = (int)n;
  a.npad = npad;
  a.ntiles = ntiles;
  a.q = c.src.as<double>();
  a.ldq = (long long)d;
  a.nq = (long long)n;
  a.d = (int)d;
  a.r2 = radiussq;
  a.mode = SCAN_FLAGS;
  a.out_flags = c.flags.as<unsigned long long>();
  CK(launch_scan(dp, a, c.stream));
  launch_subtract_accum(c.src.as<double>(), (int)n, (int)d, c.flags.as<unsigned long long>(), ntiles,
                        c.out.as<double>(), c.stream);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(out, c.out.p, n * d * sizeof(double), hipMemcpyDefault, c.stream));   // host or device destination
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

// ------------------------------------------------------------------------------ H1 ---------
// Friends-of-friends labels of update_clusters (mlfriends.pyx:275-343) from ONE all-pairs pass: the hit ballots of
// every point against every point (k_scan, mode FLAGS: the adjacency matrix as n x ntiles 64-bit words) come back to
// the host, where the reference's growth rounds -- members of the current cluster against the unlabelled points, a new
// cluster seeded when nothing joins -- are replayed on the bit rows.  Distances are symmetric bit for bit, so a round
// joins exactly the points the reference's find_nearby call reports; labels, their numbering and the carried-over
// seeds (previous ids) are the reference's.
int mlf_adjacency_bits(const double *pts, size_t n, size_t d, double radiussq, const unsigned long long **adj_out) {
  if (int rc = check_dims(d)) return rc;
  if (!pts || !adj_out || n == 0) return fail_arg(MLF_E_BADARG, "null pointer or no points");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  const int npad = round_up((int)n, kWave);
  const int ntiles = npad / kWave;
  if (int rc = stage_live_points(pts, n, d, dp, npad, false)) return rc;
  CK(c.flags.reserve(n * (size_t)ntiles * sizeof(unsigned long long)));
  ScanArgs a{};
  a.refT = c.refT.as<double>();
  a.n = (int)n;
  a.npad = npad;
  a.ntiles = ntiles;
  a.q = c.src.as<double>();
  a.ldq = (long long)d;
  a.nq = (long long)n;
  a.d = (int)d;
  a.r2 = radiussq;
  a.mode = SCAN_FLAGS;
  a.out_flags = c.flags.as<unsigned long long>();
  CK(launch_scan(dp, a, c.stream));
  const size_t nwords = n * (size_t)ntiles;
  if (c.pin_adj_cap < nwords) {   // pinned landing buffer for the bit matrix (2 MB at n = 4000), kept for the next call
    if (c.pin_adj) (void)hipHostFree(c.pin_adj);
    c.pin_adj = nullptr;
    c.pin_adj_cap = 0;
    CK(hipHostMalloc(reinterpret_cast<void **>(&c.pin_adj), nwords * sizeof(unsigned long long), hipHostMallocDefault));
    c.pin_adj_cap = nwords;
  }
  CK(hipMemcpyAsync(c.pin_adj, c.flags.p, nwords * sizeof(unsigned long long), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  *adj_out = c.pin_adj;
  return 0;
}

// Host only (no device, no library state: may run on another thread next to device calls).
int mlf_host_cluster_replay(const unsigned long long *adj, size_t n, const int64_t *previous, int64_t *labels,
                            int64_t *nclusters) {
  if (!adj || !labels || !nclusters || n == 0) return fail_arg(MLF_E_BADARG, "null pointer or no points");
  const int ntiles = (int)((n + 63) / 64);
  // reach = union of the adjacency rows of all members of the current cluster (each member's row is OR-ed in once,
  // when it joins): a growth round joins the unlabelled points inside `reach` -- the points the reference's
  // find_nearby(members, unlabelled) call reports -- and then adds their rows
  std::vector<unsigned long long> reach_v((size_t)ntiles, 0ull), unl((size_t)ntiles, 0ull);
  unsigned long long *__restrict__ reach = reach_v.data();
  std::vector<size_t> fresh;
  for (size_t i = 0; i < n; ++i) {
    labels[i] = 0;
    unl[i >> 6] |= 1ull << (i & 63);
  }
  size_t nlabelled = 0;
  auto seed_for = [&](int64_t cid, size_t fallback) {
    if (previous)
      for (size_t i = 0; i < n; ++i)
        if (previous[i] == cid) return i;
    return fallback;
  };
  auto add_member = [&](size_t i) {
    const unsigned long long *__restrict__ row = adj + i * (size_t)ntiles;
#pragma clang loop vectorize(enable)
    for (int t = 0; t < ntiles; ++t) reach[t] |= row[t];
  };
  int64_t current = 1;
  auto plant = [&](size_t i) {   // a carried-over seed may already wear an earlier label: it is re-labelled, as in the reference
    if (labels[i] == 0) {
      ++nlabelled;
      unl[i >> 6] &= ~(1ull << (i & 63));
    }
    labels[i] = current;
    for (int t = 0; t < ntiles; ++t) reach[t] = 0ull;
    add_member(i);
  };
  plant(seed_for(current, 0));
  while (nlabelled < n) {
    fresh.clear();
    for (int t = 0; t < ntiles; ++t) {
      unsigned long long w = reach[t] & unl[t];
      while (w) {
        const int b = __builtin_ctzll(w);
        w &= w - 1ull;
        fresh.push_back((size_t)t * 64 + (size_t)b);
      }
    }
    if (!fresh.empty()) {
      for (size_t j : fresh) {
        labels[j] = current;
        unl[j >> 6] &= ~(1ull << (j & 63));
      }
      nlabelled += fresh.size();
      for (size_t j : fresh) add_member(j);
    } else {
      ++current;
      size_t first = 0;
      while (first < n && labels[first] != 0) ++first;
      plant(seed_for(current, first));
    }
  }
  std::vector<char> seen((size_t)current + 1, 0);   // number of DISTINCT labels (a re-labelled seed can empty a cluster)
  int64_t distinct = 0;
  for (size_t i = 0; i < n; ++i)
    if (!seen[(size_t)labels[i]]) {
      seen[(size_t)labels[i]] = 1;
      ++distinct;
    }
  *nclusters = distinct;
  return 0;
}

int mlf_cluster_labels(const double *tpts, size_t n, size_t d, double radiussq, const int64_t *previous, int64_t *labels,
                       int64_t *nclusters) {
  const unsigned long long *adj = nullptr;
  if (int rc = mlf_adjacency_bits(tpts, n, d, radiussq, &adj)) return rc;
  return mlf_host_cluster_replay(adj, n, previous, labels, nclusters);
}

// ------------------------------------------------------------------------------ K4 ---------
int mlf_maxradiussq_bootstrap(const double *pts, size_t n, size_t d, const uint8_t *selected,
                              size_t B, double *maxd_out, uint8_t *skipped_out) {
  return mlf_maxradiussq_bootstrap_rows(pts, n, d, selected, B, 0, n, maxd_out, skipped_out);
}

// One rank's share of K4 under ROW-BLOCK sharding: all B rounds, every live point i, but only the rows j in
// [row_lo, row_hi) as the left-out point -- the rank does 1/W of the pair distances (sharding by rounds would repeat all
// of them on every rank: k_boot computes a distance once for all 32 rounds of a pass).
int mlf_maxradiussq_bootstrap_rows(const double *pts, size_t n, size_t d, const uint8_t *selected, size_t B,
                                   size_t row_lo, size_t row_hi, double *maxd_out, uint8_t *skipped_out) {
  if (int rc = check_dims(d)) return rc;
  if (B == 0) return 0;
  if (!pts || !selected || !maxd_out || n == 0) return fail_arg(MLF_E_BADARG, "null pointer or no points");
  if (row_lo > row_hi || row_hi > n) return fail_arg(MLF_E_BADARG, "row range outside [0, n]");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  const int npad = round_up((int)n, kWave);
  if (int rc = stage_live_points(pts, n, d, dp, npad, true)) return rc;
  if (int rc = upload_any(c.selbytes, selected, B * n, c.stream)) return rc;
  CK(c.sel.reserve((size_t)npad * sizeof(unsigned)));
  CK(c.selmask.reserve((size_t)(npad + 1) * kBootGroup * sizeof(unsigned)));   // k_boot requests one row past its last
  CK(c.M.reserve((size_t)kBootGroup * npad * sizeof(unsigned long long)));
  CK(c.small0.reserve(B * sizeof(double)));
  CK(c.small1.reserve(B));
  // live-point chunks: one round of the 2048 waves the chip holds at two per SIMD (k_boot's register budget), equal shares
  // (a wave pays ~1 us of set-up for its own 64 rows: no chunk below 8 live points)
  const int blk0 = (int)(row_lo / kWave);
  const int rowblocks = row_hi > row_lo ? (int)((row_hi + kWave - 1) / kWave) - blk0 : 0;
  // a rank's share of a sharded pass is short: one wave per SIMD runs this kernel as fast as two, and half the
  // waves means half the atomicMin traffic on M, which does not shrink with the shard
  const int target_waves = 2 * rowblocks <= npad / kWave ? 1024 : 2048;
  int want_chunks = rowblocks ? target_waves / rowblocks : 1;
  if (want_chunks < 1) want_chunks = 1;
  int chunk = ((int)n + want_chunks - 1) / want_chunks;
  if (chunk < 8) chunk = 8;
  const int nchunks = ((int)n + chunk - 1) / chunk;
  const double init = 1e300;
  unsigned long long init_bits;
  memcpy(&init_bits, &init, sizeof init_bits);
  for (size_t b0 = 0; b0 < B; b0 += kBootGroup) {
    const int nb = (int)((B - b0) < (size_t)kBootGroup ? (B - b0) : (size_t)kBootGroup);
    launch_pack_selection(c.selbytes.as<uint8_t>(), (int)n, npad, (int)b0, nb, c.sel.as<unsigned>(),
                          c.stream, c.selmask.as<unsigned>());
    launch_fill_u64(c.M.as<unsigned long long>(), (long long)kBootGroup * npad, init_bits, c.stream);
    BootArgs a{};
    a.refT = c.refT.as<double>();
    a.refR = c.refR.as<double>();
    a.sel = c.sel.as<unsigned>();
    a.selmask = c.selmask.as<unsigned>();
    a.n = (int)n;
    a.npad = npad;
    a.chunk = chunk;
    a.M = c.M.as<unsigned long long>();
    a.blk0 = blk0;
    if (row_lo == 0 && row_hi == n && dp <= 64 && (g_opt[OPT_BOOT_SYM] == 2 || (g_opt[OPT_BOOT_SYM] == 1 && boot_sym_usable(dp, npad)))) {
      CK(launch_boot_sym(dp, a, c.stream));   // every pair distance once (a rank's row share keeps k_boot: its minima must be complete)
    } else {
      CK(launch_boot(dp, a, nchunks, c.stream, rowblocks));
    }
    launch_boot_final(c.M.as<unsigned long long>(), c.sel.as<unsigned>(), (int)n, npad, nb,
                      c.small0.as<double>() + b0, c.small1.as<uint8_t>() + b0, c.stream, (int)row_lo, (int)row_hi);
    CK(hipGetLastError());
  }
  CK(hipMemcpyAsync(maxd_out, c.small0.p, B * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  std::vector<uint8_t> sk(B);
  CK(hipMemcpyAsync(sk.data(), c.small1.p, B, hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  if (skipped_out) memcpy(skipped_out, sk.data(), B);
  return 0;
}

// ------------------------------------------------------------------------------ K5 ---------
int mlf_pair_dist2_lower(const double *pts, size_t n, size_t d, double *dist2_out) {
  if (int rc = check_dims(d)) return rc;
  if (n < 2) return 0;
  if (!pts || !dist2_out) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const size_t npairs = n * (n - 1) / 2;
  if (int rc = upload(c.src, pts, n * d * sizeof(double), c.stream)) return rc;
  CK(c.out.reserve(npairs * sizeof(double)));
  launch_pair_dist2_lower(c.src.as<double>(), (int)n, (int)d, c.out.as<double>(), c.stream);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(dist2_out, c.out.p, npairs * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

// ------------------------------------------------------------------------------ H3 / T1 ----
int mlf_inside_ellipsoid(const double *pts, size_t np, size_t d, const double *ctr,
                         const double *invcov, double sqradius, uint8_t *mask, double *q_out) {
  if (int rc = check_dims(d)) return rc;
  if (np == 0) return 0;
  if (!pts || !ctr || !invcov || !mask) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  if (int rc = prep_consts(c.small0, c.small1, ctr, invcov, (int)d, dp, false, c.stream)) return rc;
  if (int rc = upload(c.q, pts, np * d * sizeof(double), c.stream)) return rc;
  CK(c.mask.reserve(np));
  CK(c.out.reserve(np * sizeof(double)));
  PrepArgs a{};
  a.pts = c.q.as<double>();
  a.np = (long long)np;
  a.d = (int)d;
  a.do_ell = 1;
  a.ell_ctr = c.small0.as<double>();
  a.ell_A = c.small1.as<double>();
  a.enlarge = sqradius;
  a.mask = c.mask.as<uint8_t>();
  a.q_out = c.out.as<double>();
  CK(launch_prep(dp, a, c.stream));
  CK(hipMemcpyAsync(mask, c.mask.p, np, hipMemcpyDeviceToHost, c.stream));
  if (q_out) CK(hipMemcpyAsync(q_out, c.out.p, np * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int mlf_affine_transform(const double *pts, size_t np, size_t d, const double *ctr, const double *T,
                         const double *wrap_shift, double *out) {
  if (int rc = check_dims(d)) return rc;
  if (np == 0) return 0;
  if (!pts || !ctr || !T || !out) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  if (int rc = prep_consts(c.small0, c.small1, ctr, T, (int)d, dp, true, c.stream)) return rc;
  if (wrap_shift) {
    std::vector<double> w = pad_vector(wrap_shift, (int)d, dp, NAN);
    if (int rc = upload(c.small2, w.data(), w.size() * sizeof(double), c.stream)) return rc;
    CK(hipStreamSynchronize(c.stream));
  }
  // device in: read in place; device out: written in place (the device-resident rebuild whitens its live points twice)
  const bool in_dev = is_device_pointer(pts), out_dev = is_device_pointer(out);
  const double *src = pts;
  if (!in_dev) {
    if (int rc = upload(c.q, pts, np * d * sizeof(double), c.stream)) return rc;
    src = c.q.as<double>();
  }
  double *dst = out;
  if (!out_dev) {
    CK(c.out.reserve(np * d * sizeof(double)));
    dst = c.out.as<double>();
  }
  if (dp <= 64) {
    // wave-per-8-rows form of the same chain (k_whiten_rows: bit for bit what k_prep computes, 36 -> 5 us for 4000 rows)
    const int dp8 = (dp + 7) / 8 * 8;
    std::vector<double> t8((size_t)dp * dp8, 0.0);
    for (size_t k = 0; k < d; ++k)
      for (size_t cc = 0; cc < d; ++cc) t8[k * dp8 + cc] = T[k * d + cc];
    if (int rc = upload(c.small3, t8.data(), t8.size() * sizeof(double), c.stream)) return rc;
    CK(launch_whiten_rows(src, (long long)np, (int)d, dp, c.small0.as<double>(), c.small3.as<double>(), dp8,
                          wrap_shift ? c.small2.as<double>() : nullptr, dst, (long long)d, c.stream));
  } else {
    PrepArgs a{};
    a.pts = src;
    a.np = (long long)np;
    a.d = (int)d;
    a.do_tr = 1;
    a.lay_ctr = c.small0.as<double>();
    a.lay_Tt = c.small1.as<double>();
    a.wrap_shift = wrap_shift ? c.small2.as<double>() : nullptr;
    a.t_out = dst;
    a.ldt = (long long)d;
    CK(launch_prep(dp, a, c.stream));
  }
  if (!out_dev) CK(hipMemcpyAsync(out, c.out.p, np * d * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));   // the host vectors above leave scope; a host destination is read next
  return 0;
}

// ------------------------------------------------------------- bootstrap ellipsoid stats ----
int mlf_bootstrap_moments(const double *u, size_t n, size_t d, const uint8_t *selected, size_t B,
                          double *mean_out, double *cov_out) {
  if (int rc = check_dims(d)) return rc;
  if (B == 0) return 0;
  if (!u || !selected || !mean_out || !cov_out || n == 0) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  if (int rc = upload(c.src, u, n * d * sizeof(double), c.stream)) return rc;
  if (int rc = upload(c.selbytes, selected, B * n, c.stream)) return rc;
  CK(c.small0.reserve(B * d * sizeof(double)));
  CK(c.small1.reserve(B * sizeof(int)));
  CK(c.out.reserve(B * d * d * sizeof(double)));
  CK(c.small2.reserve(B * n * sizeof(int)));
  launch_boot_moments(c.src.as<double>(), (int)n, (int)d, c.selbytes.as<uint8_t>(), (int)B,
                      c.small0.as<double>(), c.small1.as<int>(), c.out.as<double>(), c.small2.as<int>(), c.stream);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(mean_out, c.small0.p, B * d * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipMemcpyAsync(cov_out, c.out.p, B * d * d * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int mlf_bootstrap_factor(const double *u, size_t n, size_t d, const uint8_t *selected, size_t B, double scale,
                         double *f_out) {
  if (int rc = check_dims(d)) return rc;
  if (B == 0) return 0;
  if (!u || !selected || !f_out || n == 0) return fail_arg(MLF_E_BADARG, "null pointer");
  if (d > 64) return fail_arg(MLF_E_DIM, "mlf_bootstrap_factor covers d <= 64 (use the moments + quadratic-form calls above that)");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  if (int rc = upload(c.src, u, n * d * sizeof(double), c.stream)) return rc;
  if (int rc = upload_any(c.selbytes, selected, B * n, c.stream)) return rc;
  CK(c.small0.reserve(B * d * sizeof(double)));
  CK(c.small1.reserve(B * sizeof(int)));
  CK(c.out.reserve(B * d * d * sizeof(double)));
  CK(c.small2.reserve(B * n * sizeof(int)));
  CK(c.small3.reserve(B * sizeof(unsigned long long)));
  launch_boot_moments(c.src.as<double>(), (int)n, (int)d, c.selbytes.as<uint8_t>(), (int)B, c.small0.as<double>(),
                      c.small1.as<int>(), c.out.as<double>(), c.small2.as<int>(), c.stream);
  CK(hipGetLastError());
  CK(hipMemsetAsync(c.small3.p, 0, B * sizeof(unsigned long long), c.stream));
  CK(c.M.reserve(boot_cholmax_scratch_bytes((int)d, (int)B)));
  CK(launch_boot_cholmax(c.src.as<double>(), (int)n, (int)d, c.selbytes.as<uint8_t>(), (int)B, c.small0.as<double>(),
                         c.out.as<double>(), scale, c.small3.as<unsigned long long>(), c.M.p, c.stream));
  std::vector<unsigned long long> bits(B);
  CK(hipMemcpyAsync(bits.data(), c.small3.p, B * sizeof(unsigned long long), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  for (size_t b = 0; b < B; ++b) {
    double v;
    if (bits[b] == ~0ull) {
      v = std::numeric_limits<double>::quiet_NaN();
    } else {
      memcpy(&v, &bits[b], sizeof v);
    }
    f_out[b] = v;
  }
  return 0;
}

int mlf_bootstrap_quadform_max(const double *u, size_t n, size_t d, const uint8_t *selected,
                               size_t B, const double *ctr, const double *invcov, double *f_out) {
  if (int rc = check_dims(d)) return rc;
  if (B == 0) return 0;
  if (!u || !selected || !ctr || !invcov || !f_out || n == 0) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  const int dp = pick_dp((int)d);
  if (int rc = upload(c.src, u, n * d * sizeof(double), c.stream)) return rc;
  if (int rc = upload(c.selbytes, selected, B * n, c.stream)) return rc;
  // all B padded centres / matrices in one upload
  std::vector<double> pc((size_t)B * dp, 0.0), pm((size_t)B * d * dp, 0.0);
  for (size_t b = 0; b < B; ++b) {
    for (size_t k = 0; k < d; ++k) pc[b * dp + k] = ctr[b * d + k];
    for (size_t r = 0; r < d; ++r)
      for (size_t k = 0; k < d; ++k) pm[(b * d + r) * dp + k] = invcov[(b * d + r) * d + k];
  }
  if (int rc = upload(c.small0, pc.data(), pc.size() * sizeof(double), c.stream)) return rc;
  if (int rc = upload(c.small1, pm.data(), pm.size() * sizeof(double), c.stream)) return rc;
  const size_t nblk = wide_dims(dp) ? (size_t)quadmax_blocks_wide((int)n, (int)d) : (n + 255) / 256;
  CK(c.small2.reserve(B * nblk * sizeof(double)));
  QuadMaxArgs qa{};
  qa.u = c.src.as<double>();
  qa.n = (int)n;
  qa.d = (int)d;
  qa.selected = c.selbytes.as<uint8_t>();
  qa.ctr = c.small0.as<double>();
  qa.invcov = c.small1.as<double>();
  qa.part = c.small2.as<double>();
  CK(launch_boot_quadmax(dp, qa, (int)B, c.stream));
  std::vector<double> part(B * nblk);
  CK(hipMemcpyAsync(part.data(), c.small2.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  for (size_t b = 0; b < B; ++b) {
    double m = -INFINITY;
    for (size_t k = 0; k < nblk; ++k) {
      const double o = part[b * nblk + k];
      m = (o > m || o != o) ? o : m;
    }
    f_out[b] = m;
  }
  return 0;
}

// ------------------------------------------------------------------------------ region -----
int mlf_region_create(mlf_region **out) {
  if (!out) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  *out = new mlf_region();
  return 0;
}

int mlf_region_destroy(mlf_region *r) {
  if (!r) return 0;
  DevBuf *bufs[] = {&r->refT, &r->refR, &r->lay_ctr, &r->lay_mat, &r->lay_T8, &r->ell_Lt, &r->ell_LtF, &r->lay_TtF, &r->wrap, &r->ell_ctr,
                    &r->ell_A, &r->tq,  &r->gate,    &r->pts,     &r->mask, &r->row, &r->p4_LtF, &r->p4_TtF, &r->p4_y0, &r->lay_T64, &r->ell_L,
                    &r->gen, &r->gen2, &r->cube, &r->smask, &r->blk, &r->sout, &r->ax_zero, &r->ax_mat,
                    &r->s_invT, &r->s_lo, &r->s_hi, &r->s_thin, &r->s_count, &r->rf_p, &r->rf_L, &r->rf_out, &r->rf_aux,
                    &r->rf_keep, &r->ax_pad, &r->s_invT_pad, &r->s_tc, &r->s_wc, &r->s_thc, &r->s_gate};
  for (DevBuf *b : bufs) b->release();
  for (hipEvent_t e : r->events) (void)hipEventDestroy(e);
  r->filter.release();
  if (r->arena.p) (void)hipHostFree(r->arena.p);
  delete r;
  return 0;
}

int mlf_region_set(mlf_region *r, const double *unormed, size_t n, size_t d, int live_space,
                   int layer_kind, const double *layer_ctr, const double *layer_T,
                   const double *wrap_shift, const double *ell_center, const double *ell_invcov,
                   double enlarge, double radiussq, int use_scan) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (int rc = check_dims(d)) return rc;
  if (!ell_center || !ell_invcov) return fail_arg(MLF_E_BADARG, "null ellipsoid");
  if (use_scan && (!unormed || !layer_ctr || !layer_T || n == 0))
    return fail_arg(MLF_E_BADARG, "scan regions need live points and a layer");
  if (layer_kind != 0 && layer_kind != 1) return fail_arg(MLF_E_BADARG, "layer_kind must be 0 or 1");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  if (!r->arena.p) {   // pinned staging of the constants (kept with the handle; handles are recycled)
    if (hipHostMalloc(reinterpret_cast<void **>(&r->arena.p), kArenaBytes, hipHostMallocMapped) == hipSuccess &&
        hipHostGetDevicePointer(reinterpret_cast<void **>(&r->arena.p_dev), r->arena.p, 0) == hipSuccess)
      r->arena.cap = kArenaBytes;
    else
      (void)hipGetLastError();
  }
  r->arena.used = 0;
  r->arena.npending = 0;
  struct ArenaScope {   // every exit path of this call drops the arena
    explicit ArenaScope(HostArena *a) { g_arena = a->p ? a : nullptr; }
    ~ArenaScope() { g_arena = nullptr; }
  } arena_scope(&r->arena);
  r->ready = false;
  r->axes_ready = r->sampling_ready = false;   // a handle may be set again for another region (kernels.DeviceRegion recycles them)
  r->n = (int)n;
  r->d = (int)d;
  r->dp = pick_dp((int)d);
  r->npad = round_up((int)n, kWave);
  r->layer_kind = layer_kind;
  r->live_space = live_space ? 1 : 0;
  r->use_scan = use_scan ? 1 : 0;
  r->enlarge = enlarge;
  r->r2 = radiussq;
  r->has_wrap = wrap_shift != nullptr;
  const int dp = r->dp;
  if (int rc = prep_consts(r->ell_ctr, r->ell_A, ell_center, ell_invcov, (int)d, dp, false, c.stream))
    return rc;
  std::vector<double> L((size_t)d * d, 0.0);
  double fro_sq = 0.0;
  {  // Cholesky factor + Frobenius norm of the ellipsoid matrix for the bounded H3 evaluation
    std::vector<double> Lt((size_t)dp * dp, 0.0);
    bool ok = true;
    double fro = 0.0;
    for (size_t e = 0; e < d * d; ++e) fro += ell_invcov[e] * ell_invcov[e];
    for (size_t j = 0; j < d && ok; ++j) {
      double diag = ell_invcov[j * d + j];
      for (size_t k = 0; k < j; ++k) diag -= L[j * d + k] * L[j * d + k];
      if (!(diag > 0.0) || !std::isfinite(diag)) {
        ok = false;
        break;
      }
      const double ljj = std::sqrt(diag);
      L[j * d + j] = ljj;
      for (size_t i = j + 1; i < d; ++i) {
        double v = 0.5 * (ell_invcov[i * d + j] + ell_invcov[j * d + i]);
        for (size_t k = 0; k < j; ++k) v -= L[i * d + k] * L[j * d + k];
        L[i * d + j] = v / ljj;
      }
    }
    // the bound assumes a symmetric matrix: an asymmetric one takes the exact path
    for (size_t i = 0; i < d && ok; ++i)
      for (size_t j = 0; j < i; ++j)
        if (std::fabs(ell_invcov[i * d + j] - ell_invcov[j * d + i]) >
            1e-14 * (std::fabs(ell_invcov[i * d + i]) + std::fabs(ell_invcov[j * d + j])))
          ok = false;
    if (ok)
      for (size_t k = 0; k < d; ++k)
        for (size_t j = 0; j < d; ++j) Lt[k * dp + j] = L[j * d + k];
    fro_sq = fro;
    r->chol_ok = ok && std::isfinite(fro);
    r->ell_eps_scale = std::ldexp(1.0, -34) * std::sqrt(fro);
    if (int rc = upload(r->ell_Lt, Lt.data(), Lt.size() * sizeof(double), c.stream)) return rc;
    if (prep64_usable((int)d) && ok) {   // 65 ... 128 dimensions: the factor itself, row-major (mlf_prep64.hip reads its rows)
      std::vector<double> lrm((size_t)dp * dp, 0.0);
      for (size_t j = 0; j < d; ++j)
        for (size_t k = 0; k <= j; ++k) lrm[j * dp + k] = L[j * d + k];
      if (int rc = upload(r->ell_L, lrm.data(), lrm.size() * sizeof(double), c.stream)) return rc;
    }
    if (prep3_usable((int)d)) {   // the same factor as 16 x 4 matrix-core fragments: (row kb, k j) = L[j][kb]
      std::vector<double> frag(prep3_fragment_count((int)d));
      prep3_fragments(L.data(), (int)d, true, true, frag.data());
      if (int rc = upload(r->ell_LtF, frag.data(), frag.size() * sizeof(double), c.stream)) return rc;
    }
    if (!arena_active()) CK(hipStreamSynchronize(c.stream));
    r->chol_ready = true;
  }
  if (use_scan) {
    if (layer_kind == 0) {
      if (int rc = prep_consts(r->lay_ctr, r->lay_mat, layer_ctr, layer_T, (int)d, dp, true, c.stream))
        return rc;
      const int dp8 = (dp + 7) / 8 * 8;
      std::vector<double> t8((size_t)dp * dp8, 0.0);
      for (size_t k = 0; k < d; ++k)
        for (size_t cc = 0; cc < d; ++cc) t8[k * dp8 + cc] = layer_T[k * d + cc];
      if (int rc = upload(r->lay_T8, t8.data(), t8.size() * sizeof(double), c.stream)) return rc;
      if (prep3_usable((int)d)) {   // (row c, k) = T[k][c]
        std::vector<double> frag(prep3_fragment_count((int)d));
        prep3_fragments(layer_T, (int)d, true, false, frag.data());
        if (int rc = upload(r->lay_TtF, frag.data(), frag.size() * sizeof(double), c.stream)) return rc;
      }
      if (!arena_active()) CK(hipStreamSynchronize(c.stream));
    } else {
      if (int rc = upload(r->lay_ctr, layer_ctr, d * sizeof(double), c.stream)) return rc;
      if (int rc = upload(r->lay_mat, layer_T, d * sizeof(double), c.stream)) return rc;
    }
    if (wrap_shift) {
      std::vector<double> w = pad_vector(wrap_shift, (int)d, dp, NAN);
      if (int rc = upload(r->wrap, w.data(), w.size() * sizeof(double), c.stream)) return rc;
      if (!arena_active()) CK(hipStreamSynchronize(c.stream));  // w goes out of scope
    }
    if (int rc = upload(c.src, unormed, n * d * sizeof(double), c.stream)) return rc;
    const double *rows = c.src.as<double>();
    if (int rc = arena_flush(c.stream)) return rc;   // the layer constants, in front of the kernels that read them
    if (r->live_space) {  // rows are cube-space live points: whiten them on the device
      CK(c.tq.reserve(n * d * sizeof(double)));
      if (int rc = region_whiten_rows(r, c.src.as<double>(), n, c.tq.as<double>(), c.stream)) return rc;
      rows = c.tq.as<double>();
    }
    CK(r->refT.reserve((size_t)r->npad * dp * sizeof(double)));
    CK(r->refR.reserve((size_t)r->npad * dp * sizeof(double)));
    launch_build_layouts(rows, (int)n, (int)d, dp, r->npad, r->refT.as<double>(), r->refR.as<double>(),
                         c.stream);
    CK(hipGetLastError());
    if (int rc = filter_prepare_refs(r->filter, r->refR.as<double>(), (int)n, (int)d, dp, c.stream, true))
      return rc;
  }
  {
    size_t n_for_scale = n;
    const double *live_host = (use_scan && live_space) ? unormed : nullptr;
    std::vector<double> back;
    const double hint = r->live_extent_hint;
    r->live_extent_hint = -1.0;
    if (live_host && hint > 0.0 && std::isfinite(hint)) {   // the caller knows the extent: one fictitious row carries it
      back.assign(d, 0.0);
      for (size_t k = 0; k < d; ++k) back[k] = layer_ctr[k];
      back[0] = layer_ctr[0] + hint;
      live_host = back.data();
      n_for_scale = 1;
    } else if (live_host && is_device_pointer(live_host)) {   // the operand scale is found on the host: fetch the rows once
      back.resize(n * d);
      CK(hipMemcpyAsync(back.data(), unormed, n * d * sizeof(double), hipMemcpyDeviceToHost, c.stream));
      CK(hipStreamSynchronize(c.stream));
      live_host = back.data();
    }
    if (int rc = region_prep4_setup(r, L, fro_sq, ell_center, layer_ctr, layer_T, live_host, n_for_scale, c.stream,
                                    ell_invcov))
      return rc;
  }
  if (int rc = arena_flush(c.stream)) return rc;
  CK(hipStreamSynchronize(c.stream));
  r->ready = true;
  return 0;
}

// Pinned, device-mapped staging of the single-launch membership call (proposals in, mask and completion word out) and
// of the row replacements: created on first use.
constexpr size_t kSmallStagingBytes = (size_t)kSmallMaxPoints * kSmallMaxDim * sizeof(double);
static int small_staging(Ctx &c) {
  if (c.pin_pts) return 0;
  CK(hipHostMalloc(reinterpret_cast<void **>(&c.pin_pts), kSmallStagingBytes, hipHostMallocMapped));
  CK(hipHostMalloc(reinterpret_cast<void **>(&c.pin_mask), (size_t)kSmallMaxPoints + 64, hipHostMallocMapped));
  memset(c.pin_mask, 0, (size_t)kSmallMaxPoints + 64);
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c.pin_pts_dev), c.pin_pts, 0));
  CK(hipHostGetDevicePointer(reinterpret_cast<void **>(&c.pin_mask_dev), c.pin_mask, 0));
  CK(c.small_words.reserve(2 * kSmallMaxPoints * sizeof(unsigned)));
  CK(hipMemsetAsync(c.small_words.p, 0, 2 * kSmallMaxPoints * sizeof(unsigned), c.stream));
  return 0;
}

int mlf_region_set_option(mlf_region *r, const char *name, long long value, int inherit) {
  if (!r) return fail_arg(MLF_E_BADARG, "null pointer");
  if (!name) {   // every option of the handle back to the process defaults (a recycled handle starts clean)
    if (!inherit) return fail_arg(MLF_E_BADARG, "null pointer");
    if (r->filter.ov.set[OPT_ORDER] && r->filter.refs_ready) r->filter.refs_dirty = true;
    r->filter.ov = OptOverrides{};
    return 0;
  }
  const int id = opt_id(name);
  if (id < 0) return fail_arg(MLF_E_BADARG, "unknown option");
  r->filter.ov.set[id] = !inherit;
  r->filter.ov.v[id] = inherit ? 0 : opt_clamp(id, value);
  if (id == OPT_ORDER && r->filter.refs_ready) r->filter.refs_dirty = true;   // the next batch builds (or drops) the ordered operand
  return 0;
}

int mlf_region_hint_live_extent(mlf_region *r, double amax) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  r->live_extent_hint = amax;
  return 0;
}

int mlf_region_update_points(mlf_region *r, size_t count, const int64_t *rows, const double *live_rows) {
  if (!r || (count && (!rows || !live_rows))) return fail_arg(MLF_E_BADARG, "null pointer");
  if (!r->ready || !r->use_scan) return fail_arg(MLF_E_STATE, "region has no live points set");
  for (size_t k = 0; k < count; ++k)
    if (rows[k] < 0 || rows[k] >= (int64_t)r->n) return fail_arg(MLF_E_BADARG, "row out of range");
  if (count == 0) return 0;
  Ctx &c = g_ctx;
  // one buffer: [count x d new rows | count x d whitened rows | count indices]
  const size_t block = count * (size_t)r->d;
  CK(r->row.reserve((2 * block + count) * sizeof(double)));
  double *raw = r->row.as<double>(), *white = raw + block;
  long long *index = reinterpret_cast<long long *>(raw + 2 * block);
  const double *src = raw;
  const long long *index_src = index;
  if (int rc = small_staging(c)) return rc;
  if ((block + count) * sizeof(double) <= kSmallStagingBytes) {
    // the usual few rows: no copy on the stream, the kernels read the pinned staging buffer (free: a single-launch
    // membership call has its mask back before it returns, and this call ends with a synchronisation)
    memcpy(c.pin_pts, live_rows, block * sizeof(double));
    memcpy(c.pin_pts + block, rows, count * sizeof(int64_t));
    src = c.pin_pts_dev;
    index_src = reinterpret_cast<const long long *>(c.pin_pts_dev + block);
  } else {
    CK(hipMemcpyAsync(raw, live_rows, block * sizeof(double), hipMemcpyHostToDevice, c.stream));
    CK(hipMemcpyAsync(index, rows, count * sizeof(int64_t), hipMemcpyHostToDevice, c.stream));
  }
  if (r->live_space) {
    if (int rc = region_whiten_rows(r, src, count, white, c.stream)) return rc;
    src = white;
  }
  launch_update_rows(src, (int)count, r->d, r->dp, r->npad, index_src, r->refT.as<double>(), r->refR.as<double>(), c.stream);
  CK(hipGetLastError());
  // centre / scale / norms of the pre-filter operands depend on every row: requantised (four small kernels) by the next
  // batch that uses them -- the 1-10 point calls between two replacements (mlf_small.hip) never do
  if (r->filter.refs_ready) r->filter.refs_dirty = true;
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int mlf_region_update_point(mlf_region *r, size_t row, const double *unormed_row) {
  if (!unormed_row) return fail_arg(MLF_E_BADARG, "null pointer");
  const int64_t index = (int64_t)row;
  if (r && row >= (size_t)r->n && r->ready && r->use_scan) return fail_arg(MLF_E_BADARG, "row out of range");
  return mlf_region_update_points(r, 1, &index, unormed_row);
}

int mlf_region_set_thresholds(mlf_region *r, double enlarge, double radiussq) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  r->enlarge = enlarge;
  r->r2 = radiussq;
  return 0;
}

int mlf_region_set_ellipsoid_center(mlf_region *r, const double *ell_center) {
  if (!r || !ell_center) return fail_arg(MLF_E_BADARG, "null pointer");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region not set");
  Ctx &c = g_ctx;
  std::vector<double> pc = pad_vector(ell_center, r->d, r->dp);
  if (int rc = upload(r->ell_ctr, pc.data(), pc.size() * sizeof(double), c.stream)) return rc;
  CK(hipStreamSynchronize(c.stream));
  if (r->p4_ready) {
    r->h_ell_ctr.assign(ell_center, ell_center + r->d);
    if (!r->use_scan) r->h_lay_ctr = r->h_ell_ctr;   // no layer: the proposals are centred on the ellipsoid itself
    if (int rc = region_prep4_centres(r, c.stream)) return rc;
  }
  return 0;
}

// One launch, no copies on the stream: the proposals go through a pinned staging buffer the kernel reads directly,
// the mask comes back the same way.
static int region_inside_small(mlf_region *r, const double *pts, size_t np, uint8_t *mask) {
  Ctx &c = g_ctx;
  if (int rc = small_staging(c)) return rc;
  memcpy(c.pin_pts, pts, np * (size_t)r->d * sizeof(double));
  SmallArgs a{};
  a.pts = c.pin_pts_dev;
  a.np = (int)np;
  a.d = r->d;
  a.dp = r->dp;
  a.ell_ctr = r->ell_ctr.as<double>();
  a.ell_A = r->ell_A.as<double>();
  a.ell_Lt = r->ell_Lt.as<double>();
  a.eps_scale = r->ell_eps_scale;
  a.enlarge = r->enlarge;
  a.chol_ok = (r->chol_ready && r->chol_ok) ? 1 : 0;
  a.use_scan = r->use_scan;
  a.layer_kind = r->layer_kind;
  a.wpp = 1;
  if (r->use_scan) {
    a.lay_ctr = r->lay_ctr.as<double>();
    if (r->layer_kind == 0) {
      a.lay_T8 = r->lay_T8.as<double>();
      a.ldt8 = (r->dp + 7) / 8 * 8;
    } else {
      a.lay_std = r->lay_mat.as<double>();
    }
    a.wrap = r->has_wrap ? r->wrap.as<double>() : nullptr;
    a.refT = r->refT.as<double>();
    a.n = r->n;
    a.npad = r->npad;
    a.r2 = r->r2;
    // enough workgroups per proposal to spread its live points over the chip, but no more than ~512 in all
    const int shares = (r->n + 255) / 256;
    int wpp = 512 / (int)np;
    if (wpp > shares) wpp = shares;
    a.wpp = wpp < 1 ? 1 : wpp;
  }
  a.state = c.small_words.as<unsigned>();
  a.finished = c.small_words.as<unsigned>() + kSmallMaxPoints;
  a.mask = c.pin_mask_dev;
  a.flag = reinterpret_cast<unsigned *>(c.pin_mask_dev + kSmallMaxPoints);
  a.seq = ++c.small_seq;
  if (a.seq == 0u) a.seq = ++c.small_seq;   // 0 is the initial content of the word
  launch_inside_small(a, c.stream);
  CK(hipGetLastError());
  {   // spin on the completion word; a kernel that does not report within ~40 ms is left to the stream and its error path
    volatile unsigned *flag = reinterpret_cast<volatile unsigned *>(c.pin_mask + kSmallMaxPoints);
    bool seen = false;
    for (long spin = 0; spin < 4000000; ++spin) {
      if (*flag == a.seq) {
        seen = true;
        break;
      }
      __builtin_ia32_pause();
    }
    if (!seen) {
      CK(hipStreamSynchronize(c.stream));
      if (*flag != a.seq) return fail_arg(MLF_E_STATE, "single-launch membership kernel did not complete");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
  }
  memcpy(mask, c.pin_mask, np);
  return 0;
}

int mlf_region_inside(mlf_region *r, const double *pts, size_t np, uint8_t *mask) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region used before mlf_region_set");
  if (np == 0) return 0;
  if (!pts || !mask) return fail_arg(MLF_E_BADARG, "null pointer");
  if (opt(r->filter, OPT_SMALL_PATH) && np <= (size_t)kSmallMaxPoints && r->d <= kSmallMaxDim) return region_inside_small(r, pts, np, mask);
  Ctx &c = g_ctx;
  const size_t row_bytes = (size_t)r->d * sizeof(double);
  CK(r->mask.reserve(np));
  // Large host batches (what integrator.py:1776-1804 hands over: 400 MB at 10^6 x 50) are sent in chunks on a copy
  // stream; the kernels of chunk k run while chunk k + 1 crosses PCIe, so that only the last chunk's kernels and the
  // mask's way back are not hidden behind the transfer (round 2: upload, then 0.55 ms of kernels, then the mask).
  constexpr size_t kChunkRows = 131072;
  if (np >= 3 * kChunkRows && !is_device_pointer(pts)) {
    if (!c.copy_stream) {
      CK(hipStreamCreateWithFlags(&c.copy_stream, hipStreamNonBlocking));
      CK(hipEventCreateWithFlags(&c.copy_event, hipEventDisableTiming));
    }
    CK(r->pts.reserve(np * row_bytes));
    CK(hipStreamSynchronize(c.stream));   // the staging buffer may still be read by an earlier call's kernels
    for (size_t row0 = 0; row0 < np; row0 += kChunkRows) {
      const size_t rows = np - row0 < kChunkRows ? np - row0 : kChunkRows;
      double *dst = r->pts.as<double>() + row0 * (size_t)r->d;
      CK(hipMemcpyAsync(dst, pts + row0 * (size_t)r->d, rows * row_bytes, hipMemcpyHostToDevice, c.copy_stream));
      CK(hipEventRecord(c.copy_event, c.copy_stream));
      CK(hipStreamWaitEvent(c.stream, c.copy_event, 0));
      if (int rc = region_inside_enqueue(r, dst, rows, r->mask.as<uint8_t>() + row0, c.stream, nullptr)) return rc;
    }
  } else {
    if (int rc = upload(r->pts, pts, np * row_bytes, c.stream)) return rc;
    if (int rc = region_inside_enqueue(r, r->pts.as<double>(), np, r->mask.as<uint8_t>(), c.stream, nullptr))
      return rc;
  }
  CK(hipMemcpyAsync(mask, r->mask.p, np, hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int mlf_region_inside_dev(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                          void *stream) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region used before mlf_region_set");
  if (np && (!d_pts || !d_mask)) return fail_arg(MLF_E_BADARG, "null pointer");
  return region_inside_enqueue(r, d_pts, np, d_mask, (hipStream_t)stream, nullptr);
}

int mlf_region_find_nearby_dev(mlf_region *r, const double *d_tpts, size_t np, int64_t *d_idx,
                               void *stream) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (!r->ready || !r->use_scan) return fail_arg(MLF_E_STATE, "region has no live points set");
  if (np == 0) return 0;
  if (!d_tpts || !d_idx) return fail_arg(MLF_E_BADARG, "null pointer");
  ScanArgs a{};
  a.refT = r->refT.as<double>();
  a.n = r->n;
  a.npad = r->npad;
  a.ntiles = r->npad / kWave;
  a.q = d_tpts;
  a.ldq = r->d;
  a.nq = (long long)np;
  a.d = r->d;
  a.r2 = r->r2;
  a.mode = SCAN_FIRST;
  a.out_idx = reinterpret_cast<long long *>(d_idx);
  CK(launch_scan(r->dp, a, (hipStream_t)stream));
  return 0;
}

int mlf_region_set_axes(mlf_region *r, const double *axes_T) {
  if (!r || !axes_T) return fail_arg(MLF_E_BADARG, "null pointer");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region not set");
  Ctx &c = g_ctx;
  const int d = r->d, dp = r->dp;
  std::vector<double> zero((size_t)dp, 0.0);
  // k_prep computes (x - ctr) . T from T^T rows: with T = axes_T the staged matrix is axes itself
  std::vector<double> m = pad_matrix(axes_T, d, dp, true);
  if (int rc = upload(r->ax_zero, zero.data(), zero.size() * sizeof(double), c.stream)) return rc;
  if (int rc = upload(r->ax_mat, m.data(), m.size() * sizeof(double), c.stream)) return rc;
  std::vector<double> ap;   // k_generate_ellipsoid's copy: element (j, k) = axes_T[j][k], rows padded to 4 x chunk outputs
  if (d <= 128) {
    const int ldk = 4 * generate_ellipsoid_chunk(d);
    ap.assign((size_t)d * ldk, 0.0);
    for (int j = 0; j < d; ++j)
      for (int k = 0; k < d; ++k) ap[(size_t)j * ldk + k] = axes_T[(size_t)j * d + k];
    if (int rc = upload(r->ax_pad, ap.data(), ap.size() * sizeof(double), c.stream)) return rc;
  }
  CK(hipStreamSynchronize(c.stream));
  r->axes_ready = true;
  return 0;
}

int mlf_region_set_sampling_data(mlf_region *r, const double *invT, const double *bbox_lo, const double *bbox_hi) {
  if (!r || !invT || !bbox_lo || !bbox_hi) return fail_arg(MLF_E_BADARG, "null pointer");
  if (!r->ready || !r->use_scan) return fail_arg(MLF_E_STATE, "region has no live points set");
  if (r->layer_kind != 0) return fail_arg(MLF_E_STATE, "t-space sampling needs an affine layer");
  Ctx &c = g_ctx;
  const size_t d = (size_t)r->d;
  if (int rc = upload(r->s_invT, invT, d * d * sizeof(double), c.stream)) return rc;
  std::vector<double> ip;   // k_rows_affine's copy: rows padded to 4 x chunk outputs
  if (d <= 128) {
    const size_t ldk = 4 * (size_t)generate_ellipsoid_chunk((int)d);
    ip.assign(d * ldk, 0.0);
    for (size_t j = 0; j < d; ++j)
      for (size_t k = 0; k < d; ++k) ip[j * ldk + k] = invT[j * d + k];
    if (int rc = upload(r->s_invT_pad, ip.data(), ip.size() * sizeof(double), c.stream)) return rc;
  }
  if (int rc = upload(r->s_lo, bbox_lo, d * sizeof(double), c.stream)) return rc;
  if (int rc = upload(r->s_hi, bbox_hi, d * sizeof(double), c.stream)) return rc;
  CK(hipStreamSynchronize(c.stream));
  r->sampling_ready = true;
  return 0;
}

namespace {

// neighbour test of t-space points against the resident live points (MFMA pre-filter when it applies)
int region_scan_mask(mlf_region *r, const double *d_t, long long np, uint8_t *d_mask, hipStream_t s) {
  if (int rc = filter_refresh_refs(r->filter, r->refR.as<double>(), r->n, r->d, r->dp, s)) return rc;
  if (filter_applies(r->filter, np, r->r2))
    return filter_run(r->filter, r->refT.as<double>(), r->refR.as<double>(), r->n, r->npad, r->d, r->dp, d_t,
                      (long long)r->d, 1, np, r->r2, nullptr, d_mask, nullptr, s, false);
  ScanArgs a{};
  a.refT = r->refT.as<double>();
  a.n = r->n;
  a.npad = r->npad;
  a.ntiles = r->npad / kWave;
  a.q = d_t;
  a.ldq = r->d;
  a.nq = np;
  a.d = r->d;
  a.r2 = r->r2;
  a.mode = SCAN_MASK;
  a.out_mask = d_mask;
  CK(launch_scan(r->dp, a, s));
  return 0;
}

// w = t . invT + ctr (+ unwrap) and the cube flags of `n` rows
int region_untransform(mlf_region *r, const double *t, long long n, double *w, uint8_t *in_cube, hipStream_t s) {
  const double *wrap = r->has_wrap ? r->wrap.as<double>() : nullptr;
  if (r->d <= 128 && r->s_invT_pad.p)
    CK(launch_rows_affine(t, n, r->d, r->s_invT_pad.as<double>(), r->lay_ctr.as<double>(), wrap, w, in_cube, s));
  else
    launch_untransform_rows(t, n, r->d, r->s_invT.as<double>(), r->lay_ctr.as<double>(), wrap, w, in_cube, s);
  CK(hipGetLastError());
  return 0;
}

// wrapping-ellipsoid test alone (H3, mlfriends.pyx:882-912) of `np` cube-space rows: the bounded matrix-core form with the exact
// test behind its band where the region has it (what a region without a neighbour scan runs), else the binary64 kernel
int region_ellipsoid_gate(mlf_region *r, const double *d_pts, size_t np, uint8_t *gate, hipStream_t s) {
  if (np == 0) return 0;
  FilterCtx &f = r->filter;
  const bool bounded = r->layer_kind == 0 && opt(f, OPT_FUSED_PREP) && opt(f, OPT_PREP_BOUNDED) && prep3_usable(r->d) && r->chol_ready &&
                       r->p4_ready && !r->has_wrap && np < (size_t)0x7fffffff && (reinterpret_cast<uintptr_t>(d_pts) & 15) == 0;
  if (!bounded) {
    PrepArgs pa{};
    pa.pts = d_pts;
    pa.np = (long long)np;
    pa.d = r->d;
    pa.do_ell = 1;
    pa.ell_ctr = r->ell_ctr.as<double>();
    pa.ell_A = r->ell_A.as<double>();
    pa.enlarge = r->enlarge;
    pa.mask = gate;
    CK(launch_prep(r->dp, pa, s));
    return 0;
  }
  if (int rc = misc_reserve(f)) return rc;
  CK(f.ell_list.reserve(np * sizeof(int)));
  Prep4Args pa{};
  pa.pts = d_pts;
  pa.np = (long long)np;
  pa.d = r->d;
  pa.dp = r->dp;
  pa.LtF = r->p4_LtF.p;
  pa.y0 = r->p4_y0.as<float>();
  pa.TtF = r->p4_TtF.p;
  pa.lay_ctr = r->use_scan ? r->lay_ctr.as<double>() : r->ell_ctr.as<double>();   // the centre the chain's start values belong to
  pa.c = r->p4c;
  pa.c.enl_lo = f32_dn(r->enlarge);
  pa.c.enl_hi = f32_up(r->enlarge);
  pa.gate = gate;
  pa.do_tr = 0;
  pa.ell_count = f.misc.as<unsigned>();
  pa.ell_list = f.ell_list.as<int>();
  pa.ell_cap = (unsigned)np;
  CK(launch_prep4(pa, s));
  EllExactArgs ea{};
  ea.count = f.misc.as<unsigned>();
  ea.done = f.misc.as<unsigned>() + 1;
  ea.last = f.misc.as<unsigned>() + 4;
  ea.list = f.ell_list.as<int>();
  ea.cap = (unsigned)np;
  ea.pts = d_pts;
  ea.d = r->d;
  ea.dp = r->dp;
  ea.ell_ctr = r->ell_ctr.as<double>();
  ea.ell_Lt = r->ell_Lt.as<double>();
  ea.ell_L = r->ell_L.as<double>();
  ea.ell_A = r->ell_A.as<double>();
  ea.eps_scale = r->ell_eps_scale;
  ea.enlarge = r->enlarge;
  ea.chol_ok = r->chol_ok ? 1 : 0;
  ea.gate = gate;
  ea.route = nullptr;
  launch_ell_exact(ea, s);
  CK(hipGetLastError());
  return 0;
}

// methods 2 and 3 of MLFriends.sample: proposals are born in t-space
int region_sample_tspace(mlf_region *r, int method, long long n, uint64_t seed, uint64_t offset, double *out,
                         size_t capacity, size_t *naccepted, uint64_t *next_offset, bool fetch) {
  Ctx &c = g_ctx;
  hipStream_t s = c.stream;
  const int d = r->d;
  const int nblk = (int)((n + 255) / 256);
  CK(r->gen.reserve((size_t)n * d * sizeof(double)));
  CK(r->gen2.reserve((size_t)n * d * sizeof(double)));
  CK(r->smask.reserve((size_t)n));
  CK(r->cube.reserve((size_t)n));
  CK(r->blk.reserve(((size_t)nblk + 1) * sizeof(unsigned)));
  double *t = r->gen.as<double>();
  uint8_t *mask = r->smask.as<uint8_t>();
  if (method == 2) {
    launch_generate_tbox(t, n, d, r->s_lo.as<double>(), r->s_hi.as<double>(), std::sqrt(r->r2), seed, offset, s);
    CK(hipGetLastError());
    *next_offset = offset + (uint64_t)((n * d + 1) / 2);
    if (int rc = region_scan_mask(r, t, n, mask, s)) return rc;
  } else {   // method 3
    // Reference order (:1072-1094, :1154-1160): multiplicity of every proposal -> thinning -> untransform -> cube and ellipsoid
    // tests.  Every one of these is a function of the proposal alone (the thinning uniform is drawn with it), so the accepted set
    // does not depend on their order: the cheap tests run FIRST, on the whole batch (untransform + cube 0.2 ms, ellipsoid 0.15 ms
    // per 2^20 x 50), and the multiplicity -- the exact count over all live points, 22 ms for the whole batch, the one stage that
    // cannot stop at the first hit -- is taken of their survivors only (a few per cent at C5).
    CK(r->s_thin.reserve((size_t)n * sizeof(double)));
    CK(r->s_gate.reserve((size_t)n));
    launch_generate_around_points(t, r->s_thin.as<double>(), n, d, r->refR.as<double>(), r->n, r->dp, r->r2, seed,
                                  offset, s);
    CK(hipGetLastError());
    *next_offset = offset + (uint64_t)n * (uint64_t)((d + 1) / 2 + 2);
    double *wall = r->gen2.as<double>();
    if (int rc = region_untransform(r, t, n, wall, r->cube.as<uint8_t>(), s)) return rc;
    if (int rc = region_ellipsoid_gate(r, wall, (size_t)n, r->s_gate.as<uint8_t>(), s)) return rc;
    launch_mask_and(r->s_gate.as<uint8_t>(), r->cube.as<uint8_t>(), n, s);
    launch_mask_offsets(r->s_gate.as<uint8_t>(), n, r->blk.as<unsigned>(), s);
    CK(hipGetLastError());
    unsigned k0 = 0;
    CK(hipMemcpyAsync(&k0, r->blk.as<unsigned>() + nblk, sizeof k0, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    *naccepted = 0;
    if (k0 == 0) return 0;
    CK(r->s_tc.reserve((size_t)k0 * d * sizeof(double)));
    CK(r->s_wc.reserve((size_t)k0 * d * sizeof(double)));
    CK(r->s_thc.reserve((size_t)k0 * sizeof(double)));
    CK(r->s_count.reserve((size_t)k0 * sizeof(long long)));
    // (the offsets of this mask are in blk already: launch_mask_offsets above)
    launch_scatter(t, r->s_gate.as<uint8_t>(), n, d, r->blk.as<unsigned>(), r->s_tc.as<double>(), k0, s);
    launch_scatter(wall, r->s_gate.as<uint8_t>(), n, d, r->blk.as<unsigned>(), r->s_wc.as<double>(), k0, s);
    launch_scatter(r->s_thin.as<double>(), r->s_gate.as<uint8_t>(), n, 1, r->blk.as<unsigned>(), r->s_thc.as<double>(), k0, s);
    ScanArgs a{};   // multiplicity: how many balls contain the proposal (no early exit, reference :1087-1088)
    a.refT = r->refT.as<double>();
    a.n = r->n;
    a.npad = r->npad;
    a.ntiles = r->npad / kWave;
    a.q = r->s_tc.as<double>();
    a.ldq = d;
    a.nq = k0;
    a.d = d;
    a.r2 = r->r2;
    a.mode = SCAN_COUNT;
    a.out_idx = r->s_count.as<long long>();
    CK(launch_scan(r->dp, a, s));
    launch_thin_by_multiplicity(r->s_count.as<long long>(), r->s_thc.as<double>(), k0, mask, s);
    CK(hipGetLastError());
    CK(r->sout.reserve(capacity * (size_t)d * sizeof(double)));
    const unsigned cap3 = capacity > 0xffffffffu ? 0xffffffffu : (unsigned)capacity;
    launch_compact(r->s_wc.as<double>(), mask, k0, d, r->blk.as<unsigned>(), r->sout.as<double>(), cap3, s);
    CK(hipGetLastError());
    unsigned count3 = 0;
    const int nblk0 = (int)((k0 + 255) / 256);
    CK(hipMemcpyAsync(&count3, r->blk.as<unsigned>() + nblk0, sizeof count3, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    const size_t take3 = count3 < cap3 ? count3 : cap3;
    if (take3 && fetch) {
      CK(hipMemcpyAsync(out, r->sout.p, take3 * (size_t)d * sizeof(double), hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
    }
    *naccepted = take3;
    return 0;
  }
  // survivors of the neighbour test, compacted; everything after works on those rows only
  launch_compact(t, mask, n, d, r->blk.as<unsigned>(), r->gen2.as<double>(), (unsigned)n, s);
  CK(hipGetLastError());
  unsigned k1 = 0;
  CK(hipMemcpyAsync(&k1, r->blk.as<unsigned>() + nblk, sizeof k1, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  *naccepted = 0;
  if (k1 == 0) return 0;
  double *w = r->gen.as<double>();   // the t-space batch is not needed any more
  if (int rc = region_untransform(r, r->gen2.as<double>(), k1, w, r->cube.as<uint8_t>(), s)) return rc;
  if (int rc = region_ellipsoid_gate(r, w, k1, mask, s)) return rc;
  launch_mask_and(mask, r->cube.as<uint8_t>(), k1, s);
  CK(r->sout.reserve(capacity * (size_t)d * sizeof(double)));
  const unsigned cap = capacity > 0xffffffffu ? 0xffffffffu : (unsigned)capacity;
  launch_compact(w, mask, k1, d, r->blk.as<unsigned>(), r->sout.as<double>(), cap, s);
  CK(hipGetLastError());
  unsigned count = 0;
  const int nblk1 = (int)((k1 + 255) / 256);
  CK(hipMemcpyAsync(&count, r->blk.as<unsigned>() + nblk1, sizeof count, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  const size_t take = count < cap ? count : cap;
  if (take && fetch) {
    CK(hipMemcpyAsync(out, r->sout.p, take * (size_t)d * sizeof(double), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
  }
  *naccepted = take;
  return 0;
}

}  // namespace

// fetch = false leaves the accepted rows in r->sout (device) for mlf_region_refill
// masked_ok (with fetch = false): methods 0 and 1 may leave the batch where it was drawn (r->gen, all `nsamples` rows) with the
// membership mask in r->smask instead of compacting the accepted rows into r->sout; *masked tells which of the two happened
static int region_sample_impl(mlf_region *r, int method, size_t nsamples, uint64_t seed, uint64_t offset, double *out,
                              size_t capacity, size_t *naccepted, uint64_t *next_offset, bool fetch, bool masked_ok = false,
                              bool *masked = nullptr);

int mlf_region_sample(mlf_region *r, int method, size_t nsamples, uint64_t seed, uint64_t offset, double *out,
                      size_t capacity, size_t *naccepted, uint64_t *next_offset) {
  if (!out) return fail_arg(MLF_E_BADARG, "null pointer");
  return region_sample_impl(r, method, nsamples, seed, offset, out, capacity, naccepted, next_offset, true);
}

static int region_sample_impl(mlf_region *r, int method, size_t nsamples, uint64_t seed, uint64_t offset,
                      double *out, size_t capacity, size_t *naccepted, uint64_t *next_offset, bool fetch, bool masked_ok,
                      bool *masked) {
  if (masked) *masked = false;
  if (!r || !naccepted || !next_offset) return fail_arg(MLF_E_BADARG, "null pointer");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region used before mlf_region_set");
  if (method < 0 || method > 3)
    return fail_arg(MLF_E_BADARG, "method must be 0 (cube), 1 (wrapping ellipsoid), 2 (t-space box) or 3 (live points)");
  if (method == 1 && !r->axes_ready) return fail_arg(MLF_E_STATE, "mlf_region_set_axes not called");
  if (method >= 2 && (!r->use_scan || !r->sampling_ready))
    return fail_arg(MLF_E_STATE, "mlf_region_set_sampling_data not called (or region without live points)");
  *naccepted = 0;
  *next_offset = offset;
  if (nsamples == 0 || capacity == 0) return 0;
  if (method >= 2)
    return region_sample_tspace(r, method, (long long)nsamples, seed, offset, out, capacity, naccepted, next_offset, fetch);
  Ctx &c = g_ctx;
  hipStream_t s = c.stream;
  const long long n = (long long)nsamples;
  const int d = r->d;
  const int nblk = (int)((n + 255) / 256);
  CK(r->gen.reserve((size_t)n * d * sizeof(double)));
  CK(r->smask.reserve((size_t)n));
  CK(r->blk.reserve(((size_t)nblk + 1) * sizeof(unsigned)));
  CK(r->sout.reserve(capacity * (size_t)d * sizeof(double)));
  const uint8_t *pregate = nullptr;
  if (method == 0) {
    launch_generate_cube(r->gen.as<double>(), n * d, seed, offset, s);
    *next_offset = offset + (uint64_t)((n * d + 1) / 2);
  } else {
    CK(r->cube.reserve((size_t)n));
    *next_offset = offset + (uint64_t)n * (uint64_t)((d + 1) / 2 + 1);
    if (d <= 128 && r->ax_pad.p) {   // draws, axes product, centre and cube test in one launch: the batch is written once
      CK(launch_generate_ellipsoid(r->gen.as<double>(), n, d, r->enlarge, r->ax_pad.as<double>(), r->ell_ctr.as<double>(),
                                   r->cube.as<uint8_t>(), seed, offset, s));
    } else {
      CK(r->gen2.reserve((size_t)n * d * sizeof(double)));
      launch_generate_ball(r->gen2.as<double>(), n, d, r->enlarge, seed, offset, s);
      PrepArgs pa{};
      pa.pts = r->gen2.as<double>();
      pa.np = n;
      pa.d = d;
      pa.do_tr = 1;
      pa.lay_ctr = r->ax_zero.as<double>();
      pa.lay_Tt = r->ax_mat.as<double>();
      pa.t_out = r->gen.as<double>();
      pa.ldt = d;
      CK(launch_prep(r->dp, pa, s));
      launch_center_and_cube(r->gen.as<double>(), n, d, r->ell_ctr.as<double>(), r->cube.as<uint8_t>(), s);
    }
    pregate = r->cube.as<uint8_t>();
  }
  CK(hipGetLastError());
  if (int rc = region_inside_enqueue(r, r->gen.as<double>(), nsamples, r->smask.as<uint8_t>(), s, nullptr,
                                     nullptr, pregate))
    return rc;
  const unsigned cap = capacity > 0xffffffffu ? 0xffffffffu : (unsigned)capacity;
  unsigned count = 0;
  if (masked_ok && !fetch && capacity >= nsamples) {
    // the refill works on the batch where it is: count the accepted rows; only a thin batch (under a quarter accepted) is
    // worth compacting before the prior transform and the likelihood run over it
    launch_mask_offsets(r->smask.as<uint8_t>(), n, r->blk.as<unsigned>(), s);
    CK(hipGetLastError());
    CK(hipMemcpyAsync(&count, r->blk.as<unsigned>() + nblk, sizeof count, hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
    if ((size_t)count * 4 >= nsamples) {
      *naccepted = count;
      *masked = true;
      return 0;
    }
    if (count == 0) return 0;
  }
  launch_compact(r->gen.as<double>(), r->smask.as<uint8_t>(), n, d, r->blk.as<unsigned>(), r->sout.as<double>(),
                 cap, s);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(&count, r->blk.as<unsigned>() + nblk, sizeof count, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  const size_t take = count < cap ? count : cap;
  if (take && fetch) {
    CK(hipMemcpyAsync(out, r->sout.p, take * (size_t)d * sizeof(double), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
  }
  *naccepted = take;
  return 0;
}

int mlf_region_refill(mlf_region *r, int method, size_t nsamples, uint64_t seed, uint64_t offset, double Lmin, int tkind,
                      double ta, double tb, int lkind, const double *aux, double sigma, double *out_u, double *out_p,
                      double *out_L, size_t capacity, size_t *nevaluated, size_t *nkept, uint64_t *next_offset) {
  if (!r || !out_u || !out_p || !out_L || !nevaluated || !nkept || !next_offset)
    return fail_arg(MLF_E_BADARG, "null pointer");
  if (tkind < 0 || tkind > 2 || lkind < 0 || lkind > 3) return fail_arg(MLF_E_BADARG, "unknown transform / likelihood kind");
  if (lkind == 0 && !aux) return fail_arg(MLF_E_BADARG, "the Gaussian likelihood needs its centres");
  *nevaluated = 0;
  *nkept = 0;
  size_t nacc = 0;
  bool masked = false;
  if (int rc = region_sample_impl(r, method, nsamples, seed, offset, nullptr, nsamples, &nacc, next_offset, false, true, &masked))
    return rc;
  *nevaluated = nacc;
  if (nacc == 0 || capacity == 0) return 0;
  Ctx &c = g_ctx;
  hipStream_t s = c.stream;
  const int d = r->d;
  // masked: the whole batch as drawn (r->gen) with its membership mask (r->smask) -- the prior transform and the likelihood run
  // over all rows (a rejected row costs a wasted evaluation, no copy), the threshold cut keeps accepted rows only, and the
  // rows that pass it are compacted once; rows, order and values are those of the compacted route (r->sout, dense)
  const double *rows = masked ? r->gen.as<double>() : r->sout.as<double>();
  const uint8_t *member = masked ? r->smask.as<uint8_t>() : nullptr;
  const long long n = masked ? (long long)nsamples : (long long)nacc;
  const int nblk = (int)((n + 255) / 256);
  CK(r->rf_p.reserve((size_t)n * d * sizeof(double)));
  CK(r->rf_L.reserve((size_t)n * sizeof(double)));
  CK(r->rf_out.reserve(capacity * (2 * (size_t)d + 1) * sizeof(double)));
  CK(r->rf_keep.reserve((size_t)n));
  CK(r->blk.reserve(((size_t)nblk + 1) * sizeof(unsigned)));
  if (aux)
    if (int rc = upload(r->rf_aux, aux, (size_t)d * sizeof(double), s)) return rc;
  // prior transform + likelihood on the accepted proposals, where they are (reference _refill_samples,
  // integrator.py:1789-1804); only the points above the threshold travel to the host
  const double *prow = rows;   // identity transform: the parameters ARE the cube coordinates, no copy
  if (tkind != 0) {
    launch_elementwise_affine(rows, n * d, tkind, ta, tb, r->rf_p.as<double>(), s);
    prow = r->rf_p.as<double>();
  }
  launch_loglike(lkind, prow, d, n, r->rf_aux.as<double>(), sigma, r->rf_L.as<double>(), s);
  uint8_t *keep = r->rf_keep.as<uint8_t>();
  launch_mask_greater(r->rf_L.as<double>(), n, Lmin, keep, s, member);
  const unsigned cap = capacity > 0xffffffffu ? 0xffffffffu : (unsigned)capacity;
  double *ou = r->rf_out.as<double>(), *op = ou + capacity * (size_t)d, *oL = op + capacity * (size_t)d;
  launch_mask_offsets(keep, n, r->blk.as<unsigned>(), s);   // one count + scan for the three arrays
  launch_scatter(rows, keep, n, d, r->blk.as<unsigned>(), ou, cap, s);
  launch_scatter(prow, keep, n, d, r->blk.as<unsigned>(), op, cap, s);
  launch_scatter(r->rf_L.as<double>(), keep, n, 1, r->blk.as<unsigned>(), oL, cap, s);
  CK(hipGetLastError());
  unsigned count = 0;
  CK(hipMemcpyAsync(&count, r->blk.as<unsigned>() + nblk, sizeof count, hipMemcpyDeviceToHost, s));
  CK(hipStreamSynchronize(s));
  const size_t take = count < cap ? count : cap;
  if (take) {
    CK(hipMemcpyAsync(out_u, ou, take * (size_t)d * sizeof(double), hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(out_p, op, take * (size_t)d * sizeof(double), hipMemcpyDeviceToHost, s));
    CK(hipMemcpyAsync(out_L, oL, take * sizeof(double), hipMemcpyDeviceToHost, s));
    CK(hipStreamSynchronize(s));
  }
  *nkept = take;
  return 0;
}

int mlf_debug_philox(uint64_t seed, unsigned stream, size_t nblocks, uint32_t *out) {
  if (!out) return fail_arg(MLF_E_BADARG, "null pointer");
  if (nblocks == 0) return 0;
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  CK(c.out.reserve(nblocks * 4 * sizeof(unsigned)));
  launch_philox_words(seed, stream, (long long)nblocks, c.out.as<unsigned>(), c.stream);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(out, c.out.p, nblocks * 4 * sizeof(unsigned), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int mlf_region_first_index_dev(mlf_region *r, const double *d_pts, size_t np, int64_t *d_idx,
                               void *stream) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region used before mlf_region_set");
  if (np && (!d_pts || !d_idx)) return fail_arg(MLF_E_BADARG, "null pointer");
  return region_inside_enqueue(r, d_pts, np, nullptr, (hipStream_t)stream, nullptr,
                               reinterpret_cast<long long *>(d_idx));
}

int mlf_region_inside_dev_timed(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                                void *stream) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region used before mlf_region_set");
  if (!np || !d_pts || !d_mask) return fail_arg(MLF_E_BADARG, "bad argument");
  while (r->events.size() < r->events_used + 4) {
    hipEvent_t e;
    CK(hipEventCreate(&e));
    r->events.push_back(e);
  }
  hipEvent_t *ev = r->events.data() + r->events_used;
  r->events_used += 4;
  return region_inside_enqueue(r, d_pts, np, d_mask, (hipStream_t)stream, ev);
}

int mlf_region_timing_collect(mlf_region *r, int *ncalls, double *ms_prep, double *ms_scan,
                              double *ms_rest) {
  if (!r || !ncalls || !ms_prep || !ms_scan || !ms_rest) return fail_arg(MLF_E_BADARG, "null pointer");
  double prep = 0.0, scan = 0.0, rest = 0.0;
  const size_t calls = r->events_used / 4;
  for (size_t i = 0; i < calls; ++i) {
    hipEvent_t *ev = r->events.data() + 4 * i;
    CK(hipEventSynchronize(ev[3]));
    float a = 0.f, b = 0.f, c2 = 0.f;
    CK(hipEventElapsedTime(&a, ev[0], ev[1]));
    CK(hipEventElapsedTime(&b, ev[1], ev[2]));
    CK(hipEventElapsedTime(&c2, ev[2], ev[3]));
    prep += a;
    scan += b;
    rest += c2;
  }
  r->events_used = 0;
  *ncalls = (int)calls;
  *ms_prep = prep;
  *ms_scan = scan;
  *ms_rest = rest;
  return 0;
}

int mlf_region_timing_filter_launches(mlf_region *r, int *nlaunches, double *ms_total) {
  if (!r || !nlaunches || !ms_total) return fail_arg(MLF_E_BADARG, "null pointer");
  FilterCtx &f = r->filter;
  double total = 0.0;
  for (size_t i = 0; i + 1 < f.kev_used; i += 2) {
    CK(hipEventSynchronize(f.kev[i + 1]));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, f.kev[i], f.kev[i + 1]));
    total += ms;
  }
  *nlaunches = (int)(f.kev_used / 2);
  *ms_total = total;
  f.kev_used = 0;
  return 0;
}

int mlf_region_timing_filter_launch_ms(mlf_region *r, double *ms, int cap, int *nlaunches) {
  if (!r || !nlaunches || (cap > 0 && !ms)) return fail_arg(MLF_E_BADARG, "null pointer");
  FilterCtx &f = r->filter;
  int n = 0;
  for (size_t i = 0; i + 1 < f.kev_used; i += 2, ++n) {
    if (n >= cap) continue;
    CK(hipEventSynchronize(f.kev[i + 1]));
    float t = 0.f;
    CK(hipEventElapsedTime(&t, f.kev[i], f.kev[i + 1]));
    ms[n] = t;
  }
  *nlaunches = n;
  return 0;
}

int mlf_region_filter_info(mlf_region *r, size_t np, int *active, int *kdim, int *ntiles32) {
  if (!r || !active || !kdim || !ntiles32) return fail_arg(MLF_E_BADARG, "null pointer");
  *active = (r->ready && r->use_scan && filter_applies(r->filter, (long long)np, r->r2)) ? 1 : 0;
  *kdim = r->filter.ks * 16;
  *ntiles32 = r->filter.ntiles32;
  return 0;
}

int mlf_region_debug_fused_stamps(mlf_region *r, int block, unsigned long long *out, int cap) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  FilterCtx &f = r->filter;
  if (out && cap > 0) {
    for (int i = 0; i < cap; ++i) out[i] = 0ull;
    if (f.fstamps.p && f.stamp_block >= 0) {
      CK(hipStreamSynchronize(g_ctx.stream));
      CK(hipMemcpy(out, f.fstamps.p, (size_t)(cap < 16 ? cap : 16) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    }
  }
  f.stamp_block = block;
  return 0;
}

int mlf_region_debug_stats(mlf_region *r, unsigned long long *out, int cap) {
  // counters of the LAST filtered batch of this region (after a synchronisation): [0] proposals in the binary32
  // ellipsoid band, [1] queries whitened in the reference arithmetic, [2] uncertain pairs listed, [3] largest list
  // segment, [4] list segments, [5] 32-query groups left for the second live-point range
  if (!r || !out || cap < 6) return fail_arg(MLF_E_BADARG, "bad argument");
  FilterCtx &f = r->filter;
  for (int i = 0; i < cap; ++i) out[i] = 0;
  CK(hipDeviceSynchronize());
  if (f.misc.p) {
    unsigned m[8];
    CK(hipMemcpy(m, f.misc.p, sizeof m, hipMemcpyDeviceToHost));
    out[0] = m[4];
  }
  if (f.segcnt.p && f.segcnt.cap >= sizeof(unsigned)) {
    const size_t n = f.last_nsegs;
    std::vector<unsigned> c(n);
    if (n) CK(hipMemcpy(c.data(), f.segcnt.p, n * sizeof(unsigned), hipMemcpyDeviceToHost));
    unsigned long long sum = 0, mx = 0;
    for (unsigned v : c) {
      sum += v;
      mx = v > mx ? v : mx;
    }
    out[2] = sum;
    out[3] = mx;
    out[4] = n;
  }
  if (f.png.p) {
    unsigned g[6];
    CK(hipMemcpy(g, f.png.p, sizeof g, hipMemcpyDeviceToHost));
    out[5] = g[0];
    if (cap > 6) out[6] = g[1];   // queries of the last min-only batch whose minimum ended in the band (uncertain set)
    if (cap > 7) out[7] = g[5];   // three ranges: groups that entered the third
  }
  if (cap > 17) {
    out[16] = (unsigned long long)f.last_cut[0];
    out[17] = (unsigned long long)f.last_cut[1];
  }
  if (cap > 18) out[18] = f.last_same_form ? 1ull : 0ull;   // the last fused launch took the ellipsoid form from the whitening chain
  {
    if (cap >= 16 && f.mid_last && f.segcnt.p) {   // k_inside_mid, workgroup (0, 0): stage boundaries
      unsigned st[8];
      CK(hipMemcpy(st, f.segcnt.p, sizeof st, hipMemcpyDeviceToHost));
      for (int i = 0; i < 8; ++i) out[8 + i] = st[i];
    } else if (cap >= 16 && f.last_nsegs == (size_t)uncertain_blocks() && f.segcnt.cap >= (f.last_nsegs + 8) * sizeof(unsigned)) {
      unsigned st[8];   // shader-clock stamps of k_uncertain's workgroup 0 (stage boundaries of its first set)
      CK(hipMemcpy(st, f.segcnt.as<unsigned>() + uncertain_stamp_base(), sizeof st, hipMemcpyDeviceToHost));
      for (int i = 0; i < 8; ++i) out[8 + i] = st[i];
    }
  }
  return 0;
}

int mlf_bench_fp64_valu(double *tflops) {
  if (!tflops) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  CK(c.out.reserve(1 << 20));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int blocks = 256 * 8, iters = 2000;
  double ops = launch_fp64_probe(c.out.as<double>(), blocks, iters, c.stream);  // warm-up
  CK(hipEventRecord(e0, c.stream));
  ops = launch_fp64_probe(c.out.as<double>(), blocks, iters, c.stream);
  CK(hipEventRecord(e1, c.stream));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0));
  CK(hipEventDestroy(e1));
  *tflops = ops / (ms * 1e-3) / 1e12;
  return 0;
}

int mlf_region_time_inside_dev(mlf_region *r, const double *d_pts, size_t np, uint8_t *d_mask,
                               void *stream, int reps, float *ms_total, float *ms_scan) {
  if (!r) return fail_arg(MLF_E_BADARG, "null region");
  if (!r->ready) return fail_arg(MLF_E_STATE, "region used before mlf_region_set");
  if (!d_pts || !d_mask || !ms_total || !ms_scan || reps <= 0 || np == 0)
    return fail_arg(MLF_E_BADARG, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t ev[4];
  for (auto &e : ev) CK(hipEventCreate(&e));
  double tot = 0.0, scan = 0.0;
  for (int i = 0; i < reps; ++i) {
    if (int rc = region_inside_enqueue(r, d_pts, np, d_mask, s, ev)) return rc;
    CK(hipEventSynchronize(ev[3]));
    float a = 0.f, b = 0.f;
    CK(hipEventElapsedTime(&a, ev[0], ev[3]));
    CK(hipEventElapsedTime(&b, ev[1], ev[3]));
    tot += a;
    scan += b;
  }
  for (auto &e : ev) CK(hipEventDestroy(e));
  *ms_total = (float)(tot / reps);
  *ms_scan = (float)(scan / reps);
  return 0;
}

// ------------------------------------------------------------------------------ likelihoods -
static int loglike_host(int kind, const double *params, size_t d, size_t n, const double *aux,
                        double sigma, double *like) {
  if (d == 0) return fail_arg(MLF_E_BADARG, "dimensionality must be positive");
  if (n == 0) return 0;
  if (!params || !like || (kind == 0 && !aux)) return fail_arg(MLF_E_BADARG, "null pointer");
  if (int rc = ensure_ctx()) return rc;
  Ctx &c = g_ctx;
  if (int rc = upload(c.q, params, n * d * sizeof(double), c.stream)) return rc;
  if (aux)
    if (int rc = upload(c.small0, aux, d * sizeof(double), c.stream)) return rc;
  CK(c.out.reserve(n * sizeof(double)));
  launch_loglike(kind, c.q.as<double>(), (int)d, (long long)n, aux ? c.small0.as<double>() : nullptr,
                 sigma, c.out.as<double>(), c.stream);
  CK(hipGetLastError());
  CK(hipMemcpyAsync(like, c.out.p, n * sizeof(double), hipMemcpyDeviceToHost, c.stream));
  CK(hipStreamSynchronize(c.stream));
  return 0;
}

int mlf_loglike_gauss(const double *params, size_t d, size_t n, const double *centers, double sigma,
                      double *like) {
  return loglike_host(0, params, d, n, centers, sigma, like);
}
int mlf_loglike_eggbox(const double *params, size_t d, size_t n, double *like) {
  return loglike_host(1, params, d, n, nullptr, 0.0, like);
}
int mlf_loglike_eggbox2(const double *params, size_t d, size_t n, double *like) {
  return loglike_host(2, params, d, n, nullptr, 0.0, like);
}
int mlf_loglike_rosenbrock(const double *params, size_t d, size_t n, double *like) {
  return loglike_host(3, params, d, n, nullptr, 0.0, like);
}

int mlf_loglike_dev(int kind, const double *d_params, size_t d, size_t n, const double *d_aux,
                    double sigma, double *d_like, void *stream) {
  if (kind < 0 || kind > 3 || d == 0) return fail_arg(MLF_E_BADARG, "bad likelihood kind / dimension");
  if (n == 0) return 0;
  if (!d_params || !d_like || (kind == 0 && !d_aux)) return fail_arg(MLF_E_BADARG, "null pointer");
  launch_loglike(kind, d_params, (int)d, (long long)n, d_aux, sigma, d_like, (hipStream_t)stream);
  CK(hipGetLastError());
  return 0;
}

}  // extern "C"

namespace mlf {

int pick_dp(int d) {
#define X(D) \
  if (d <= D) return D;
  MLF_FOR_EACH_DP(X)
#undef X
  if (d <= MLF_MAX_DIM) return (d + 15) / 16 * 16;   // above 128: the run-time kernels of mlf_wide.hip, coordinates padded to 16
  return -1;
}

}  // namespace mlf
