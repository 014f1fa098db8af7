// mlf_filter.hpp -- MFMA pre-filter for the neighbour scan (see mlf_filter.hip)
#pragma once
#include "mlf_common.hpp"
#include "mlf_prep4.hpp"

#define MLF_FILTER_MAXD 128

constexpr int kFilterMaxSplit = 16;   // tile ranges (grid.y) of a single-sweep launch over a small batch

namespace mlf {

struct FilterArgs {
  const void *refF;      // f16 fragments of the live points  [ntiles32][KS][64][8]
  const void *qF;        // f16 fragments of the queries      [ngroups][KS][64][8]
  const float *tlo, *thi;
  int ntiles32;
  long long ngroups;     // query groups of 32
  long long nq;
  int *best;             // per query: first certain/confirmed live index, kNone if none
  unsigned long long *list;   // [nwaves][seg_cap] wave-private segments of uncertain pairs
  unsigned seg_cap;
  unsigned *seg_count;        // [nwaves]
  unsigned *counters;         // [1] overflow flag
  // phased sweep (see filter_run): live-point tiles [tile0, tile1) only; queries may be a compacted
  // subset: qmap[slot] = original query (-1 = padding), ngroups_dev = number of slot groups
  int tile0, tile1;
  const int *qmap;
  const unsigned *ngroups_dev;
  // or: nslots_dev = number of SLOTS of the compacted set, straight from the compacting launch's counter (the last group
  // may be partly filled: slots past the count are ignored; nobody has padded it)
  const unsigned *nslots_dev;
  int append;                 // 1: keep the list entries of the earlier phases (cursor starts at seg_count)
  // fused compaction (non-final phase): every wave appends its still undecided queries (route 1, no certain
  // hit) to a fresh fragment set straight from its registers; slots come from one atomic per wave on *ccount,
  // so the slot order varies from run to run (results do not: every slot carries its query id in cmap)
  void *cq;                   // [ngroups][KS][64][8] f16 fragments of the compacted set, nullptr = off
  float *ctlo, *cthi;
  int *cmap;
  unsigned *ccount;
  unsigned ccap;              // slots the compacted set can hold
  const uint8_t *route;
  // list segments [seg_first_extra, seg_first_extra + seg_extra) are zeroed by the first waves of a non-appending launch
  long long seg_first_extra, seg_extra;
  // tile ranges of a non-compacting launch (grid.y); list segment of (wave, range) = wave + range * 4 * grid.x
  int split;
  // mask-mode sweep behind the bounded per-proposal stage (k_sweep): every wave re-checks the list segment it has just
  // written itself (rw.pts != nullptr), and the first launch of a batch carries the ellipsoid band (rw.ell.count != nullptr)
  // in kEllWaves leading waves -- no separate re-check launch
  RecheckWArgs rw;
};
// ---- min-only phased sweep (mlf_sweepmin.hip) ----
struct MinArgs {
  const void *refF;      // f16 fragments of the live points  [ntiles32][KS][64][8]
  int ntiles32;
  int tile0, tile1;      // live-point tiles of this range
  // the set that is swept: slot = query (qmap == nullptr) or a compacted set of nslots_dev[0] slots
  const void *qF;
  const float *tlo, *thi;
  const int *qmap;
  const int *qmin;       // minima (bit patterns) the slots bring along from earlier ranges, nullptr = none
  long long ngroups;     // groups of 32 slots the grid is sized for
  const unsigned *nslots_dev;
  long long nq;
  int *best;             // certain hit: best[query] = 0
  // the set that is written: slots from one atomic per wave on *ccount
  void *cq;
  float *ctlo, *cthi;
  int *cmap;
  int *cmin;             // nullptr when last
  unsigned *ccount;
  unsigned ccap;
  int last;              // 0: keeps the queries without a certain hit; 1: keeps the uncertain ones (minimum in the band)
  unsigned long long *stamps;   // diagnostics (mlf_region_debug_fused_stamps, blocks >= 1 000 000): stage stamps of wave 0 of ...
  unsigned stamp_block;         // ... this workgroup's first pass
};
constexpr unsigned kUncertainListCap = 4096;   // band pairs of one set of 128 uncertain queries (expected: ~140)
struct UncertainArgs {
  const void *refF;
  int ntiles32;
  const void *qF;        // the uncertain set (compacted by the last k_sweep_min launch)
  const float *thi;
  const int *qmap;
  const unsigned *nslots_dev;
  // exact side: the proposals as handed over, the layer, the whitened live points
  const double *pts;
  int d, dp;
  const double *lay_ctr;
  const double *T8;      // row-major layer matrix, row stride ldt8, zero padded to dp rows
  int ldt8;
  const double *refR;    // [npad][dp]
  int n;
  double r2;
  int *best;
  unsigned *counters;    // [1] overflow flag
  unsigned *seg_count;   // [uncertain_blocks()]: pairs listed per workgroup (statistics)
  EllExactArgs ell;      // ell.count != nullptr: the band proposals of k_prep4, in kEllWaves / 8 trailing workgroups
  unsigned nsweepblk;    // filled in by the launcher
};
hipError_t launch_sweep_min(int ks, int qw, const MinArgs &a, hipStream_t s);
hipError_t launch_uncertain(int ks, const UncertainArgs &a, hipStream_t s);
long long uncertain_blocks();
__host__ __device__ constexpr unsigned uncertain_stamp_base() { return 256u; }   // = uncertain_blocks(): 8 diagnostic words behind the per-workgroup counts

// ---- one launch for the batch sizes of a real run (mlf_mid.hip) ----
struct MidArgs {
  // per-proposal stage (k_prep4's)
  const double *pts;      // (np, d) row-major proposals, 16-byte aligned
  long long np;
  int d, dp, ks;
  const void *LtF;
  const float *y0;
  const void *TtF;
  const double *lay_ctr;
  Prep4Consts c;
  const double *stats;    // [0] sigma, [1] namax, [8 + c] centre of the whitened live points
  double r2;
  // exact ellipsoid test of the band proposals
  const double *ell_ctr, *ell_L, *ell_A;
  double ell_eps_scale, enlarge;
  int chol_ok;
  // sweep + re-check
  const void *refF;
  int ntiles32;
  const double *refR;     // [npad][dp]
  int n;
  const double *T64;
  // hand-off between the tile ranges of a set
  unsigned long long *rec;    // [nsets][R][3]
  unsigned long long *meta;   // [nsets][4]
  unsigned *arrive;           // [nsets], zero between batches
  // answers
  uint8_t *mask;
  uint8_t *route;             // 2 = left to the exact scan launch that follows
  unsigned *scan_flag;
  unsigned *counters;
  unsigned *stamps;           // optional: 8 diagnostic words
};
#define MLF_FOR_EACH_DP_MID(X)                                                                \
  X(2) X(4) X(6) X(8) X(10) X(12) X(14) X(16) X(18) X(20) X(22) X(24) X(26) X(28) X(30) X(32) \
  X(36) X(40) X(44) X(48) X(50) X(52) X(56)
bool mid_usable(int dp);
int mid_range_quads(long long ngroups, int ntiles32);
hipError_t launch_inside_mid(const MidArgs &a, int ny, hipStream_t s);

// ---- per-proposal stage + first range of the min-only sweep in one launch (mlf_fused.hip) ----
struct FusedArgs {
  Prep4Args p;           // qF / tlo / thi are not written (the operand never leaves the registers)
  const void *refF;
  int ntiles32;
  int tile0, tile1;
  void *cq;              // the proposals without a certain hit: fragments, thresholds, query, minimum
  float *ctlo, *cthi;
  int *cmap, *cmin;
  unsigned *ccount;
  unsigned ccap;
  unsigned long long *stamps;   // optional diagnostics: 16 shader-clock stamps of wave 0 of workgroup `stamp_block`
  unsigned stamp_block;
  unsigned variant;             // "fused_variant" option: bit 0 matrix fragments by LDS-DMA (default) / by a load-store loop; bit 1 same
                                // quadratic form (the caller put that form's constants into p.c); bit 2 (set by the caller, not an
                                // option): p.gate[] holds a pre-gate on entry, 0 = outside whatever the ellipsoid test says
};
bool fused_usable(int dp);
hipError_t launch_prep_sweep(const FusedArgs &a, hipStream_t s, int waves = 8);

// after a compacting launch: group count of the compacted set, padding of its last group, counter reset
void launch_phase_finish(void *cq, float *ctlo, float *cthi, int *cmap, unsigned *ccount, unsigned *ngroups_dst,
                         int ks, hipStream_t s);

struct RecheckArgs {
  const unsigned long long *list;
  unsigned seg_cap;
  const unsigned *seg_count;
  const double *refR;    // [npad][dp]
  int n, d, dp;
  const double *q;
  long long ldq, ldk, nq;   // query element (j, k) at q[j*ldq + k*ldk]
  double r2;
  int *best;
};

// scratch: (64 * 128 + 2) doubles, zero-initialised once (the last two words are running maxima, reset by the launch)
// keys (optional, [n] u64): bit patterns of |a_i - c|^2, the sort key of the mask-mode operand's order
void launch_ref_stats(const double *refR, int n, int d, int dp, double *stats, double *scratch, hipStream_t s,
                      unsigned long long *keys = nullptr);
// refF: the storage-order operand; refFm (optional): the mask-mode operand, slot i = storage row perm[i], with the rows in that
// order in rows_out [nrows_out][dp] -- both in one launch
void launch_quant_refs(const double *refR, int n, int npad32, int d, int dp, int ks,
                       const double *stats, void *refF, hipStream_t s, void *refFm = nullptr, const int *perm = nullptr,
                       double *rows_out = nullptr, int nrows_out = 0);
// perm[slot] = storage row, slots in ascending key order (ties: ascending row)
void launch_ref_rank(const unsigned long long *keys, int n, int *perm, hipStream_t s);
void launch_quant_queries(const double *q, long long ldq, long long nq, long long nqpad, int d_src, int d, int ks,
                          const double *stats, double r2, const uint8_t *gate, void *qF, float *tlo,
                          float *thi, uint8_t *route, int *best, unsigned *counters, hipStream_t s);
hipError_t launch_filter(int ks, const FilterArgs &a, bool first, hipStream_t s, int narrow = 0);
// mask-mode sweep (mlf_sweep.hip): qw query groups per wave
hipError_t launch_sweep(int ks, int qw, const FilterArgs &a, hipStream_t s);
int filter_groups_per_wave(int ks, int narrow);
void launch_recheck(const RecheckArgs &a, long long nwaves, hipStream_t s);
long long filter_wave_count(int ks, long long ngroups, int narrow = 0);
// tile ranges a single-sweep launch over `ntiles` tiles should use for a batch of `ngroups` query groups (1 ... 4)
int filter_tile_split(int ks, long long ngroups, int ntiles, int target_waves = 2048);  // waves (= list segments) of a k_filter launch
void launch_filter_finalize(const uint8_t *route, const int *best, const unsigned *counters, long long nq,
                            uint8_t *out_mask, long long *out_idx, uint8_t *exact_gate, hipStream_t s,
                            unsigned *reset_word = nullptr);
void launch_route_gate(const uint8_t *route, const unsigned *counters, long long nq, int which,
                       uint8_t *gate, hipStream_t s);

}  // namespace mlf
