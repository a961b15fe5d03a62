// Kernel argument blocks and launcher prototypes shared by kernels.hip and api.cpp.
#pragma once
#include <hip/hip_runtime.h>

#include "tdtk_internal.h"

namespace tdtk {

struct Mat4 {
  double m[16];
};

constexpr int SEARCH_LAZY_MAX = 8;     // longest chain of queued scan moves a search launch carries out itself (kernels.hip: LAZY_MAX)

struct TreeDev {
  const KdHot* hot;           // compact hot records (fp32 box), same indexing as nodes
  const KdFat* fat;           // a node's hot record together with its children's: two levels per round trip (big batches)
  float absmax;               // largest |coordinate| of the root box: scales the fp32 error bound
  const KdNode* nodes;
  const KdPoint* pts;
  const float4* grp;          // fp32 shadow groups of the (padded) buckets, 3 x float4 per 4 slots; null: buckets are not padded
  const LeafEntry* leaf_tab;  // non-null only in table mode
  const double* node_r;       // FindClosestAlongDir only
  uint32_t root_ref;
  uint32_t cb, cmask;
  uint32_t n_hot;             // records in `hot` (= internal nodes)
  uint32_t n_slots;           // point slots of the (padded) leaf-ordered array: a hit's position is < n_slots
  // 16-bit shadow of the (padded) buckets (round 5; kernels.hip, "bucket_scan_q16"): every slot's coordinates on one grid of
  // 65536 cells per axis over the root box, cell = largest extent / 65535, stored as int16 (grid index - 32768), two slots per
  // 12 bytes { (x0,y0), (x1,y1), (z0,z1) }: a bucket of <= 20 points is 120 contiguous bytes = eight 16-byte loads (fifteen
  // for the fp32 groups).  null: not built (degenerate box, TDTK_BUCKET_Q16=0).
  const uint32_t* q16;
  double q_lo[3];             // grid origin = the root box's lower corner
  double q_scale;             // cells per unit
  // the split half of every internal node on its own, 16 bytes { splitval, c1, c2 } (round 5; kernels.hip, "the quick check
  // deferred"): what a visit needs when it does not make the quick check.  Same indexing as hot / nodes.
  const double2* split;
};

#ifdef TDTK_LAB
// ---- lab: the ICP loop that runs without the host (loop_dev.h, round 6; a measured negative, NEGATIVES.md) ----------
// Launch k of the loop (k = 0, 1, ...) is ONE search kernel whose every workgroup first does, redundantly, what the host used
// to do between two launches: adds up the rows of pair sums launch k-1 left (k_final's association, bit for bit), solves
// Horn's eigenproblem, applies the stopping rule -- and then either returns (the loop has ended) or fuses the transform it has
// just computed into its pass.  No second kernel, no grid-wide synchronisation, no host: a dependent launch costs ~3 us on
// this part and the smallest kernel ~4, which is what a separate solve kernel behind every search would add (built and
// measured first: NEGATIVES.md).  Workgroup 0 alone writes what outlives the launch:
struct IcpLoopDev {
  struct Hist { double ret, prev_ret; int stop, pad; } h[2];   // slot k & 1: the RMS history and the flag after launch k's solve
  double* rows_host;               // the record: a ring of row_cap rows of ICP_ROW doubles in pinned host memory
  int row_cap, pad;
};
// a row of the loop's record, one per iteration (the solve of iteration i is made -- and its row written -- by launch i + 1)
constexpr int ICP_LOOP_COLS = 17;  // = ACC_DD: the base block (n, sum, centroid sums, cross sums)
constexpr int ICP_LOOP_MAX_ROWS = 512;   // rows of pair sums every workgroup can afford to add up itself (two per thread)
enum { ICP_ROW_N = 0, ICP_ROW_RMS = 1, ICP_ROW_XF = 2, ICP_ROW_STATUS = 18, ICP_ROW_ACC = 19, ICP_ROW_READY = 36, ICP_ROW = 40 };
enum { ICP_ROW_CONTINUE = 1, ICP_ROW_CONVERGED = 2, ICP_ROW_LAST = 3, ICP_ROW_FEW_PAIRS = 4, ICP_ROW_NEED_HOST = 5 };
hipError_t launch_solve_once(const double* partials, int rows, const double shift[3], IcpLoopDev* st, hipStream_t s);
#endif

struct SearchArgs {
  TreeDev T;
  double *x, *y, *z;     // queries, SoA, spatially sorted; written when has_pending
  double *nx, *ny, *nz;  // normals / directions (nullable)
  size_t n;
  Mat4 pending;  // alignxf to apply in place before searching (ICP loop)
  Mat4 inv;      // M4inv(Source->dalignxf): world -> tree frame
  int has_pending, has_inv;
  double maxd2;
  // warm, the persistent-lane single pass only: tie > 0 -- a query that starts from a previous hit walks WITHOUT the quick check of
  // the divergent visits (16 bytes per visit instead of 48) and is searched again, cold and exactly, if it ever accepted a point
  // that improved its closest_d2 by no more than `tie` (kernels.hip, "the quick check deferred"); the warm radius is the previous
  // hit's d2 + 2 tie.  0: every visit makes the quick check.
  double tie;
  // warm: what the previous hit's d2 is widened by (twice this) to give the starting radius -- the rounding of the reference's
  // quick check and split test (api.cpp: search_margin), so that no node holding the previous hit, or a nearer point, can be cut
  // off by a rounding at a radius this tight.  Set whenever warm is; tie (above) is either 0 or equal to it.
  double margin;
  int warm;     // kpos holds the previous pass's hits of the SAME queries in the SAME tree: start each search with
                // a radius just above the distance to that point (see warm_radius in kernels.hip)
  int* kpos;    // out: position of the hit in the leaf-ordered point array, or -1
  double* d2;   // out, nullable
  double* ovf_m2;  // stack overflow area (nullable when max_depth-1 <= LDS depth)
  uint32_t* ovf_ref;
  unsigned long long* counters;  // COUNT instantiations only: internal nodes, buckets, bucket points visited
  int qpw;      // persistent-lane kernels: consecutive sorted queries per wave (set by launch_search)
  int slab;     // work-queue kernel: queries per draw (set by launch_search)
  uint32_t* q_ctr;       // work-queue kernel: 8 draw counters of THIS launch (zero on entry) ...
  uint32_t* q_ctr_next;  // ... and the 8 of the next launch on the same stream, zeroed by this one
  // fused retire-time accumulation of the base pair sums (k_search_refill<.., FUSE>): fuse != 0, A = Source->dalignxf,
  // shift as in AccumArgs, partials [search_fused_rows(n)][ACC_TOTAL]
  int pool_slab;     // persistent-lane kernel, > 0: a wave's slab is `qpw` queries and the rest of its XCD's region is a pool
                     // of pieces this long, drawn when a wave runs dry (q_ctr: one counter per XCD); region = queries per XCD
  size_t region;
  int trace;         // diagnostics (TDTK_WAVE_TRACE=<launch>): every wave prints its XCD, start and end time
  // persistent-lane kernel: buckets each query visited (saturating byte, written when it retires; nullptr: not wanted);
  // use_cost != 0: the bytes of the PREVIOUS pass of the same queries are valid, and a wave hands out each piece of its
  // slab with the expensive queries first
  unsigned char* cost;
  int use_cost;
  int side_by_side;  // > 1: one of that many whole-scan passes running concurrently on streams of their own (set by the caller)
  int phases;   // persistent-lane kernel: a wave's slab is handed out in this many pieces (see k_search_refill)
  int fuse;
  Mat4 A;
  double shift[3];
  double* partials;
  // Lazy scan moves (several-links launch only, k_search_refill_multi): nmoves != 0 -- before it searches, every wave moves
  // its own slab: the unmoved points from sx / sy / sz, the `nmoves` in-place transforms queued on the scan since it was
  // last read (Scan::transformToEuler of the graph-SLAM rounds since, two per round) applied in order -- the arithmetic
  // of k_transform_chain_batch, bit for bit --, the result stored into x / y / z above, which are then this link's own:
  // the scan's spare arrays for the ONE link of the launch that owns the scan's update (the host swaps them in behind
  // the launch; that link also gets nx / ny / nz and moves the normals in place), a scratch copy for any other link of
  // the launch that reads the same scan.
  // single-pass launch (k_search_refill): wave w of the launch owns the queries [bounds[w], bounds[w + 1]) -- slabs of equal
  // cost by the previous pass's cost bytes (launch_slab_bounds); nullptr: slabs of qpw queries
  const uint32_t* bounds;
  const double *sx, *sy, *sz;
  const Mat4* moves;
  int nmoves;        // <= SEARCH_LAZY_MAX
  // -R (rnd > 1, searchTree.cc:118): one byte per query in sorted order, non-zero = "not drawn this pass": the point still moves
  // (has_pending) but is no candidate -- kpos = -1, no search.  nullptr: every query is a candidate.
  const unsigned char* skip;
#ifdef TDTK_LAB
  // the small-batch kernels inside the host-free ICP loop (non-null; see IcpLoopDev): this is launch `loop_iter`; the rows of
  // pair sums of the launch in front (loop_rows of them; this launch writes `partials`, the other buffer of the pair) and the
  // loop's stopping rule.  has_pending / pending above are not looked at: the transform is the one the prologue solves for.
  IcpLoopDev* loop;
  const double* loop_prev;
  int loop_rows, loop_iter, loop_max_iter, loop_pad;
  double loop_eps;
#endif
};

// internal bit beside the public TDTK_WANT_* ones: no centroid / cross-covariance columns (see k_accum)
constexpr unsigned ACC_WANT_NO_CROSS = 0x100u;

// accumulator columns
enum {
  ACC_N = 0,
  ACC_SUM = 1,
  ACC_SM = 2,    // 3: sum (m - shift)
  ACC_SD = 5,    // 3: sum (d - shift)
  ACC_P = 8,     // 9: sum (m - shift)_a (d - shift)_b
  ACC_DD = 17,   // 6: sum (d - shift)_a (d - shift)_b, upper       [APX]
  ACC_NA = 23,   // 21: sum v v^T upper, v = [(d-shift) x n ; n]    [NAPX]
  ACC_NB = 44,   // 6: sum v
  ACC_NS = 50,   // 1: sum ((p1-p2).n)^2
  ACC_L = 51,    // 15: lum6DEuler sums                              [LUM]
  ACC_LSS = 66,  // 1: residual^2 against D (second pass)
  ACC_MM = 67,   // 6: sum (m - shift)_a (m - shift)_b, upper       [GAPX / MOM2, with ACC_DD]
  ACC_LU = 73,   // 1: sum u.delta, u = (p1+p2)/2, delta = p1-p2     [LUM; lum6DQuat's MZ(4)]
  ACC_TOTAL = 74
};

struct AccumArgs {
  TreeDev T;
  const double *x, *y, *z;
  const double *nx, *ny, *nz;
  const int* kpos;
  size_t n;
  Mat4 A;    // Source->dalignxf
  Mat4 inv;  // its inverse (pairing mode 1 keeps the rotated normal)
  double shift[3];
  double D[6];
  int has_D;
  double* partials;  // [grid][ACC_TOTAL]
};

struct BinArgs {
  const double* q;    // [n][3]
  const double* dir;  // [n][3] nullable
  size_t n;
  double lo[3], scale[3];
  uint32_t* hist;  // 32768
  uint32_t* cell;  // [n]
  double *sx, *sy, *sz, *sdx, *sdy, *sdz;
  int32_t* order;  // sorted position -> original index
};

struct PairListArgs {
  TreeDev T;
  const double *x, *y, *z, *nx, *ny, *nz;  // resident (sorted) scan; normals nullable
  const int* kpos;
  const int32_t* order;    // sorted position -> caller index
  const uint32_t* slot;    // exclusive scan of the found flags, caller order
  size_t n;
  Mat4 A, inv;
  double *p1, *p2, *pn;    // [pairs][3], nullable
};

uint32_t search_grid(size_t n);
size_t search_max_lanes(size_t n);     // most lanes any search kernel launches for n queries (stack overflow area)
bool search_uses_queue(size_t n);      // does it get the work-queue kernel (needs q_ctr / q_ctr_next)?
bool search_can_fuse(size_t n);
bool search_fuse_after_last_pays(size_t n);   // FUSE 3 by default for a batch of this size?
int search_fuse_kind(size_t n);   // 0 no, 1 persistent-lane FUSE modes (on request), 2 chunk epilogue of the small-batch kernels        // does a batch of n queries get the kernel that can fuse the base sums?
uint32_t search_fused_rows(size_t n, int side_by_side = 1);  // rows of partials the fused kernel writes
#ifdef TDTK_LAB
size_t slab_bounds_bytes(size_t n);
hipError_t launch_slab_bounds(const unsigned char* cost, size_t n, void* buf, const uint32_t** bounds_out, hipStream_t s);
#endif
hipError_t launch_bandwidth(int kind, void* a, void* b, size_t bytes, double* moved_bytes, hipStream_t s);
hipError_t launch_make_hot(const KdNode* nodes, size_t n, KdHot* hot, hipStream_t s, double2* split = nullptr);
hipError_t launch_make_fat(const KdNode* nodes, size_t n, KdFat* fat, hipStream_t s);
// bucket groups (kernels.hip, "bucket groups"): mark -> exclusive scan of ng_at[0..M] (launch_scan_u32) -> fill
hipError_t launch_pad_mark(const KdNode* nodes, size_t n_internal, const LeafEntry* leaf_tab, uint32_t cb, uint32_t cmask, uint32_t* ng_at,
                           size_t M, hipStream_t s);
hipError_t launch_pad_fill(KdNode* nodes, size_t n_internal, LeafEntry* leaf_tab, uint32_t cb, uint32_t cmask, const uint32_t* g_at,
                           const KdPoint* pts, KdPoint* ptsP, float4* grp, hipStream_t s, uint32_t* q16 = nullptr, const double* q_lo = nullptr,
                           double q_scale = 0.0);
hipError_t launch_final(const double* partials, uint32_t rows, double* d_out, hipStream_t s, int ncols = ACC_TOTAL);   // columns >= ncols: +0.0
// several batches (the link passes of a graph-SLAM round) in one launch: see k_search_refill_multi in kernels.hip
struct FinalDesc { const double* partials; double* out; int rows, pad; };
uint32_t search_multi_prepare(SearchArgs& a, int links_in_launch, bool long_slabs = false);   // sets the slab fields, returns the batch's workgroups (x8)
int search_multi_thresh(size_t n);
int search_multi_class(size_t n);   // the kernel family a batch of n queries gets (20 / 4 / 10); 0: not available in this form
hipError_t launch_search_multi(const SearchArgs* d_args, const uint32_t* d_base, int nbatch, uint32_t total_blocks, int cls, int thresh,
                               bool count, hipStream_t s, bool ordered = false, bool lum_sums = false);
hipError_t launch_accum_multi(const AccumArgs* d_args, const uint32_t* d_base, int nbatch, uint32_t total_blocks, unsigned want,
                              const FinalDesc* d_final, hipStream_t s, bool rows_done = false);
int search_lds_depth();
int search_block();
uint32_t accum_grid(size_t n);

hipError_t launch_search(const SearchArgs& a, uint32_t grid, int dirmode, bool count, hipStream_t s);
hipError_t launch_accum(const AccumArgs& a, uint32_t grid, unsigned want, int pmode, double* d_out,
                        hipStream_t s);
hipError_t launch_transform(double* x, double* y, double* z, double* nx, double* ny, double* nz,
                            size_t n, const Mat4& A, hipStream_t s);
hipError_t launch_bin(const BinArgs& b, hipStream_t s);
hipError_t launch_split_soa(const double* q, size_t n, double* x, double* y, double* z, hipStream_t s);
// keep-mask of an -R pass, one bit per query in the CALLER's order (bit set = drawn) -> one byte per query in sorted order (1 = skip)
hipError_t launch_skip_from_mask(const unsigned char* mask_bits, const int32_t* order, size_t n, unsigned char* skip, hipStream_t s);
hipError_t launch_idx_hash(const int* kpos, const int32_t* order, const KdPoint* pts, size_t n, unsigned long long* out, hipStream_t s);
hipError_t launch_scatter_idx(const int* kpos, const double* d2s, const int32_t* order,
                              const KdPoint* pts, size_t n, int32_t* idx_out, double* d2_out,
                              hipStream_t s);

struct DevBuildResult {
  hipError_t err;
  bool degenerate;
  KdNode* nodes;
  double* node_r;
  LeafEntry* leaf_tab;
  KdPoint* pts;
  uint32_t root_ref;
  int cb;
  bool table_mode;
  uint32_t n_internal, n_leaves, max_depth, max_leaf;
  bool respeculated;   // the speculative build failed its check (or fell over) and the in-order build took its place
};
// level-synchronous construction of the reference's kd-tree on the device (build.hip)
size_t device_build_arena_bytes(size_t M, bool with_background_chain = true);
// `side` (nullable): a second stream and two events of the caller's for the build's background chain -- with it the exact
// centroid sums of the big nodes run beside the levels below them (see "speculative splits" in build.hip)
struct BuildSide { hipStream_t s2, s3; hipEvent_t e1, e2, e3; void* h_pin; hipStream_t s4; hipEvent_t e4; };   // s2 null: no background streams (s3 / e3, s4 / e4 nullable: fewer);
                                                                                    // h_pin: 64 KB of pinned host memory for the build's looks at the device, or null
// no_finish != 0: every level by its own launches (the subtrees are not handed to single workgroups; see k_fin_subtrees)
DevBuildResult device_build_tree(const double* d_xyz, size_t M, int bucket, void* arena, hipStream_t s,
                                 const BuildSide* side = nullptr, int no_finish = 0);

// ---- normals: the ANN kd-tree (one point per leaf, sliding midpoint) + approximate k-NN + PCA (ann.hip) -------
struct AnnNode {          // 32 B: one splitting node (ANNkd_split: cut_val, cd_bnds[2], child[2], cut_dim)
  double cut_val, lo, hi;
  uint32_t c0, c1;        // child references: bit 29 = leaf, bits 0..28 = node index / point position; c0 bits 30..31 = cut_dim
};
struct AnnBuildResult {
  hipError_t err;
  bool degenerate;
  uint32_t root_ref, max_depth, levels;
};
size_t ann_build_arena_bytes(size_t M);
AnnBuildResult ann_build_tree(const double* d_xyz, size_t M, void* arena, AnnNode* nodes, KdPoint* pts, double* bb,
                              hipStream_t s);
uint32_t ann_search_threads(size_t n);
size_t ann_spill_entries(size_t n, uint32_t max_depth);
hipError_t launch_ann_normals(const AnnNode* nodes, uint32_t root_ref, const KdPoint* pts, size_t n, int k, double eps,
                              const double* d_bb, const double rPos[3], uint32_t* spill_ref, double* spill_bd,
                              uint32_t max_depth, double* d_normals, int32_t* d_knn, unsigned long long* d_cnt,
                              hipStream_t s);

hipError_t launch_pp_error(const AccumArgs& a, uint32_t grid, double scale, double* d_partial, double* d_out, hipStream_t s);
hipError_t launch_found_flags(const int* kpos, const int32_t* order, size_t n, uint32_t* flags, hipStream_t s);
hipError_t launch_pair_list(const PairListArgs& a, int pmode, hipStream_t s);
size_t scan_u32_temp_bytes(size_t n);
hipError_t launch_scan_u32(const uint32_t* in, uint32_t* out, size_t n, void* tmp, size_t tmp_bytes, hipStream_t s);

// exclusive scan of 64-bit words that hold two 27-bit counts, in ONE launch (sort.hip); see there for `state` / `epoch`
size_t scan_pair27_state_bytes(size_t n);
hipError_t launch_scan_pair27(const unsigned long long* in, unsigned long long* out, size_t n, void* state, uint32_t epoch,
                              uint32_t* err, hipStream_t s, const uint32_t* gate = nullptr);   // gate: non-null and *gate == 0 -> nothing runs

size_t morton_sort_temp_bytes(size_t n);
hipError_t launch_morton_order(const double* d_xyz, size_t n, const double* d_box, uint32_t* keys_a, uint32_t* idx_a, uint32_t* keys_b, uint32_t* idx_b, void* d_tmp,
                               size_t tmp_bytes, hipStream_t s);
// one resident scan moved by a chain of consecutive in-place transforms (Scan::transformToEuler is two of them per
// graph-SLAM round; a scan whose moves were queued over several rounds gets them all in one pass over its points)
struct XfChainDesc {
  double *x, *y, *z, *nx, *ny, *nz;
  size_t n;
  const Mat4* mats;   // device memory, applied mats[0], mats[1], ...
  int nm, pad;
};
hipError_t launch_transform_chain_batch(const XfChainDesc* d_desc, int count, size_t max_n, hipStream_t s);

// reduce.hip: bounding box + octree-centre reduction
struct OctRoot {
  double center[3];
  double size;   // half edge of the root cube (largest half extent + 1.0)
  int depth;     // halvings until size <= voxelSize (>= 1)
};
size_t bbox_temp_bytes();
hipError_t launch_bbox(const double* d_xyz, size_t n, double* d_partial, double* d_box, hipStream_t s);
size_t oct_sort_temp_bytes(size_t n);
hipError_t launch_oct_keys_sorted(const double* d_xyz, size_t n, const OctRoot& R, uint64_t* keys_a,
                                  uint64_t* keys_b, void* tmp, size_t tmp_bytes, hipStream_t s);
hipError_t launch_oct_heads(const uint64_t* keys, size_t n, uint32_t* flags, hipStream_t s);
// random modes of the octree reduction: the reference's within-leaf point order (reduce.hip)
size_t scan_u64_temp_bytes(size_t n);
hipError_t launch_oct_leaf_order(const uint64_t* keys, const uint64_t* sorted, size_t n, int depth, uint32_t* perm, uint32_t* work,
                                 uint64_t* flags, uint64_t* P, void* tmp, size_t tmp_bytes, hipStream_t s);
hipError_t launch_oct_leaf_starts(const uint32_t* flags, const uint32_t* slot, size_t n, uint32_t* starts, hipStream_t s);
hipError_t launch_oct_gather(const double* xyz, const uint32_t* perm, const uint32_t* sel, size_t m, double* out, hipStream_t s);
hipError_t launch_oct_centres(const uint64_t* keys, const uint32_t* flags, const uint32_t* slot, size_t n,
                              const OctRoot& R, double* out, hipStream_t s);
hipError_t launch_unsort_aos(const double* x, const double* y, const double* z, const int32_t* order, size_t n,
                             double* out, hipStream_t s);
hipError_t launch_gather_soa(const double* d_src, const uint32_t* order, size_t n, double* x, double* y, double* z,
                             hipStream_t s);

}  // namespace tdtk
