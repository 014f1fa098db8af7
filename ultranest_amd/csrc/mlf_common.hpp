// mlf_common.hpp -- shared declarations of the gfx950 MLFriends kernels (internal; the public
// boundary is include/mlfriends_hip.h).
//
// Data layouts in HBM (all float64):
//   refT  : live points, COORDINATE-major  [DP][npad]   (npad = n rounded up to 64, zero padded)
//           -> a wave reads 64 consecutive live points of one coordinate in one 512-byte load
//   refR  : live points, row-major         [npad][DP]   (zero padded), for LDS tile staging
//   query : row-major (nq, ldq) as handed over by the caller
// DP is the compile-time padded dimensionality of a kernel instance (d <= DP, DP even).  Zero
// padding is exact: (0-0)^2 adds +0.0 to a non-negative accumulator.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

namespace mlf {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) belongs to (function, DEVICE): a grant made while device 0 was current
// says nothing about device 1.  One DeviceGrant per kernel instance: a bit per device ordinal, set after the first
// successful grant on that device; `ensure(f)` runs f() (the hipFuncSetAttribute calls of the instance) once per device.
// mlf_debug_forget_grants() (tests: a second mlf_set_device on a 1-GPU box must re-grant) starts a new epoch.
extern std::atomic<unsigned> g_grant_epoch;
extern std::atomic<unsigned long long> g_grant_calls;   // grants actually issued (statistics for the test)
struct DeviceGrant {
  std::atomic<unsigned long long> bits[4];
  std::atomic<unsigned> epoch;
  constexpr DeviceGrant() : bits{}, epoch(0u) {}
  template <class F>
  hipError_t ensure(F &&grant) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned now = g_grant_epoch.load(std::memory_order_acquire);
    if (epoch.load(std::memory_order_acquire) != now) {
      for (auto &w : bits) w.store(0ull, std::memory_order_relaxed);
      epoch.store(now, std::memory_order_release);
    }
    const bool tracked = dev >= 0 && dev < 256;
    const unsigned long long bit = 1ull << (dev & 63);
    if (tracked && (bits[dev >> 6].load(std::memory_order_acquire) & bit)) return hipSuccess;
    e = grant();
    if (e != hipSuccess) return e;
    g_grant_calls.fetch_add(1ull, std::memory_order_relaxed);
    if (tracked) bits[dev >> 6].fetch_or(bit, std::memory_order_release);
    return hipSuccess;
  }
};

constexpr int kWave = 64;
constexpr int kScanQB = 64;        // queries staged in LDS per scan workgroup
constexpr int kScanThreads = 256;  // 4 waves; wave w owns live-point tiles w, w+4, ...
constexpr int kBootGroup = 32;     // bootstrap rounds handled per launch (one selection bit each)
constexpr int kNone = 0x7fffffff;  // "no neighbour found yet"

enum ScanMode : int {
  SCAN_FIRST = 0,  // K1: lowest live-point index within r2, else -1        (int64 out)
  SCAN_COUNT = 1,  // K2: number of live points within r2                   (int64 out)
  SCAN_FLAGS = 2,  // K3 pass 1: 64-bit hit ballots per (query, tile)       (u64 out)
  SCAN_MASK = 3    // R3: byte mask "gate && any live point within r2"     (u8 out)
};

// exact ellipsoid test of the proposals k_prep4 could not decide (bounded binary64 form first, the reference's
// summation order only inside that form's own band)
struct EllExactArgs {
  unsigned *count;         // reset to 0 by the last workgroup
  unsigned *done;          // workgroups finished (returns to 0)
  unsigned *last;          // optional: receives the count before it is reset
  const int *list;
  unsigned cap;
  const double *pts;
  int d, dp;
  const double *ell_ctr;   // [dp]
  const double *ell_Lt;    // [dp][dp]  Lt[k][j] = L[j][k]
  const double *ell_L;     // [dp][dp]  L[j][k] row-major (the same factor, for the wave-per-proposal form)
  const double *ell_A;     // [d][dp]
  double eps_scale, enlarge;
  int chol_ok;
  uint8_t *gate;
  uint8_t *route;          // may be null
};

struct ScanArgs {
  const double *refT;
  int n, npad, ntiles;
  const double *q;        // query element (j, k) at q[j*ldq + k*ldk]
  long long ldq;
  long long ldk;          // 0 or 1: row-major; > 1: coordinate-major
  long long nq;
  int d;
  double r2;
  int mode;
  const uint8_t *gate;            // optional, per query: 0 = do not scan (result: none)
  int only_gated;                 // 1: write outputs only for gated-in queries; idle workgroups exit early
  // optional: q holds the proposals as handed over (not whitened); the workgroup whitens the rows it has staged in the
  // arithmetic of k_prep (delta_k = x_k - c_k, k-ascending FMA chain) before scanning -- the second-stage launch
  // behind k_prep4, which stores no whitened coordinates
  const double *raw_ctr;          // [>= d] layer centre; nullptr = q is already whitened
  const double *raw_T8;           // row-major layer matrix, element (k, c) at raw_T8[k * raw_ldt + c]
  int raw_ldt;
  const unsigned *any_flag;       // optional (device): 0 and no list overflow (counters[1]) = nothing to scan, the scanning
                                  // workgroups return at once
  // optional: the gate comes from the pre-filter's routing instead of a byte array -- a query is scanned if
  // route == 2, or route == 1 and the uncertain-pair list overflowed (counters[1]); `gate` is ignored then
  const uint8_t *route;
  const unsigned *counters;
  // optional tail of the launch (workgroups behind the scanning ones): answers of the FILTERED queries from best[]
  // (k_filter_finalize's job, riding in this launch to save a kernel boundary)
  const int *fin_best;            // nullptr = no tail
  unsigned fin_grid0;             // first workgroup of the tail (filled in by the launcher)
  unsigned *fin_reset;            // optional word zeroed by the tail
  unsigned *fin_slots;            // optional: slot counter of the fused compaction -- *fin_groups = ceil(it / 32), then zeroed
  unsigned *fin_groups;
  unsigned *fin_slots2;           // optional: slot counter of the uncertain set (mlf_sweepmin.hip) -- fin_groups[1] = its value, then zeroed
  unsigned *fin_slots3;           // optional: slot counter of the middle set of three ranges -- fin_groups[5] = ceil(it / 32), then zeroed
  long long *out_idx;             // SCAN_FIRST / SCAN_COUNT
  unsigned long long *out_flags;  // SCAN_FLAGS  [nq][ntiles]
  uint8_t *out_mask;              // SCAN_MASK
};

struct BootArgs {
  const double *refT;   // [DP][npad]
  const double *refR;   // [npad][DP]
  const unsigned *sel;  // [npad] bit b set = live point selected in bootstrap round b of this group
  const unsigned *selmask;  // [npad][kBootGroup] 0 if the live point is selected in the round, 0xffffffff otherwise
  int n, npad;
  int chunk;            // live points per blockIdx.y
  unsigned long long *M;  // [kBootGroup][npad] running minima as ordered bit patterns
  int blk0;             // first 64-row block of the launch (row-block sharding over ranks: blockIdx.x + blk0)
};

struct PrepArgs {
  const double *pts;
  long long np;
  int d;
  int do_ell;
  const double *ell_ctr;  // [DP] zero padded
  const double *ell_A;    // [d][DP] rows zero padded
  double enlarge;
  uint8_t *mask;          // out (do_ell) : inside ellipsoid
  double *q_out;          // optional
  int do_tr;
  const double *lay_ctr;      // [DP]
  const double *lay_Tt;       // [d][DP]: row c holds T[:, c], zero padded
  const double *wrap_shift;   // [DP] (NaN = not wrapped) or nullptr
  double *t_out;
  long long ldt;
};

struct QuadMaxArgs {
  const double *u;          // (n, d) row-major points
  int n, d;
  const uint8_t *selected;  // (B, n)
  const double *ctr;        // [B][DP]
  const double *invcov;     // [B][d][DP]
  double *part;             // [B][ceil(n/256)] per-workgroup maxima
};

// smallest instantiated DP >= d; above 128: d rounded up to 16 (the run-time kernels of mlf_wide.hip); -1 above MLF_MAX_DIM
int pick_dp(int d);

// ---- dimensionalities above 128 (mlf_wide.hip): k-chunked forms with the dimensionality at run time, same arithmetic
bool wide_dims(int dp);
hipError_t launch_scan_wide(int dp, const ScanArgs &a, hipStream_t s);
hipError_t launch_boot_wide(int dp, const BootArgs &a, int d, int nchunks, hipStream_t s, int nblocks);
hipError_t launch_prep_wide(int dp, const PrepArgs &a, hipStream_t s);
int quadmax_blocks_wide(int n, int d);   // workgroups per round (= partial maxima per round) of launch_quadmax_wide
hipError_t launch_quadmax_wide(int dp, const QuadMaxArgs &a, int B, hipStream_t s);
void launch_subtract_wide(const double *pts, int n, int d, const unsigned long long *flags, int ntiles, double *out, hipStream_t s);
void launch_boot_mean_cov_wide(const double *u, int n, int d, const int *idx, const int *count, int B, double *mean, double *cov, hipStream_t s);
void launch_update_rows_wide(const double *rows, int count, int d, int dp, int npad, const long long *index, double *refT, double *refR, hipStream_t s);

hipError_t launch_scan(int dp, const ScanArgs &a, hipStream_t s);
hipError_t launch_boot(int dp, const BootArgs &a, int nchunks, hipStream_t s, int nblocks = -1);
bool boot_sym_usable(int dp, int npad);   // k_boot_sym: every pair distance once (whole-range passes with >= 1024 tiles)
hipError_t launch_boot_sym(int dp, const BootArgs &a, hipStream_t s);
hipError_t launch_prep(int dp, const PrepArgs &a, hipStream_t s);
hipError_t launch_whiten_rows(const double *pts, long long n, int d, int dp, const double *lay_ctr, const double *T8, int ldt8,
                              const double *wrap_shift, double *t_out, long long ldt, hipStream_t s);
hipError_t launch_boot_quadmax(int dp, const QuadMaxArgs &a, int B, hipStream_t s);

// instantiated dimensionalities: every even value up to 32, every 4th up to 64 (+50, the
// headline configuration), every 16th up to 128
#define MLF_FOR_EACH_DP(X)                                                                   \
  X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
  X(36) X(40) X(44) X(48) X(50) X(52) X(56) X(60) X(64) X(80) X(96) X(112) X(128)

}  // namespace mlf
