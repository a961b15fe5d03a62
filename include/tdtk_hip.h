/*
 * tdtk_hip.h -- C ABI of lib3dtk_hip.so, the MI355X (gfx950) implementation of
 * 3DTK's slam6D ICP correspondence + alignment hot path.
 *
 * Plain C types only; caller owns every host buffer; the library never frees
 * caller memory; no exception crosses this boundary: every function returns
 * TDTK_OK (0) or a negative TDTK_E* code and tdtk_last_error() gives the text
 * (per calling thread).  Handles are opaque; functions are re-entrant across
 * handles.  All floating point is fp64 with FMA contraction off, so that
 * correspondence indices are bit-exact with the reference KDtree.
 *
 * 4x4 matrices are column-major ("OpenGL order", translation in [12..14]) as
 * everywhere in the reference (include/slam6d/globals.icc:298-328).
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the JMUWRobotics/3DTK checkout).  The reference-side bindings that call
 * these are shown in INTEGRATION.md.
 */
#ifndef TDTK_HIP_H
#define TDTK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TDTK_OK 0
#define TDTK_EINVAL (-1)   /* bad argument (NULL, zero points, bucket < 1 ...)        */
#define TDTK_EDEVICE (-2)  /* no usable gfx950 device / HIP runtime error             */
#define TDTK_ENOMEM (-3)   /* host or device allocation failed                        */
#define TDTK_ESOLVE (-4)   /* minimizer could not be solved (Cholesky failed, ...)    */
#define TDTK_EUNSUP (-5)   /* valid in the reference but not supported on this entry point */
#define TDTK_EPEER (-6)    /* another rank of the communicator failed; the collective result is void on every rank */

typedef struct tdtk_tree tdtk_tree; /* model-scan search tree, resident in HBM   */
typedef struct tdtk_scan tdtk_scan; /* data-scan points (+normals), resident in HBM */

/* include/slam6d/pairingMode.h:4-8 */
enum {
  TDTK_CLOSEST_POINT = 0,
  TDTK_CLOSEST_POINT_ALONG_NORMAL_SIMPLE = 1,
  TDTK_CLOSEST_PLANE_SIMPLE = 2
};

/* minimizer ids == the -a values of src/slam6d/slam6D.cc:696-727.  QUAT / SVD / APX / NAPX are the ones
 * an OpenMP build of the reference can run (Align_Parallel, icp6D.cc:201-219); the other six exist only
 * as serial Align (icp6Dortho.cc, icp6Ddual.cc, icp6Dhelix.cc, icp6Dlumeuler.cc, icp6Dlumquat.cc,
 * icp6Dquatscale.cc) and are computed here from the second-moment block TDTK_WANT_MOM2.      */
enum {
  TDTK_ALGO_QUAT = 1, TDTK_ALGO_SVD = 2, TDTK_ALGO_ORTHO = 3, TDTK_ALGO_DUAL = 4, TDTK_ALGO_HELIX = 5,
  TDTK_ALGO_APX = 6, TDTK_ALGO_LUMEULER = 7, TDTK_ALGO_LUMQUAT = 8, TDTK_ALGO_QUAT_SCALE = 9,
  TDTK_ALGO_NAPX = 10
};

/* which accumulator blocks tdtk_scan_pairs / tdtk_get_pt_pairs should fill */
#define TDTK_WANT_BASE 0u  /* n, sum, centroids, Si: always                              */
#define TDTK_WANT_APX 1u   /* A[6], B[3] of icp6D_APX (src/slam6d/icp6Dapx.cc:203-245)   */
#define TDTK_WANT_NAPX 2u  /* A[21], B[6], sum of icp6D_NAPX (icp6Dnapx.cc:53-95)        */
#define TDTK_WANT_LUM 4u   /* the 15 sums + ss of lum6DEuler::covarianceEuler            */
#define TDTK_WANT_GAPX 8u  /* the per-link blocks of gapx6D::genBArotForLinkedPair (alone) */
#define TDTK_WANT_MOM2 16u /* centred second moments of p1 and of p2 (alone): with n, centroids and
                            * Si they are the full second-moment matrix of (p1 ; p2), from which
                            * every minimizer that is quadratic in the pairs follows            */

/* Merged per-call sums: what T OpenMP threads' (n, sum, centroid_m, centroid_d, Si)
 * of src/slam6d/icp6D.cc:129-192 add up to; feed slot 0 of Align_Parallel with it. */
typedef struct tdtk_pair_sums {
  uint64_t n_queries;    /* queries issued                                             */
  uint64_t n;            /* pairs found (pairs.size())                                  */
  double sum;            /* sum |p1-p2|^2              searchTree.cc:172-177            */
  double centroid_m[3];  /* mean of p1 (model, world)  scan.cc:1253-1259 (normalised)   */
  double centroid_d[3];  /* mean of p2 (data)                                           */
  double Si[9];          /* sum (p1-cm)[a]*(p2-cd)[b] at [a*3+b]   icp6D.cc:170-191     */
  double apx_A[6];       /* A00 A01 A02 A11 A12 A22 about centroid_d (icp6Dapx.cc:203)  */
  double apx_B[3];
  double napx_A[21];     /* upper triangle row-major, about centroid_d                  */
  double napx_B[6];
  double napx_sum;       /* sum ((p1-p2).n)^2                                           */
  double lum[15];        /* sx sy sz xpy xpz ypz xy xz yz MZ[0..5]  lum6Deuler.cc:143-175 */
  double lum_sumd2;      /* sum |p1-p2|^2 (== sum), kept for the ss identity check      */
  /* gapx6D::genBArotForLinkedPair (src/slam6d/gapx6D.cc:153-310), both points centred on
   * centroid_m, including the literal `p1x*p2x + p1y + p2y` diagonal terms (gapx6D.cc:208-210) */
  double gapx_MkMkt[9], gapx_DkDkt[9], gapx_MkDkt[9], gapx_DkMkt[9];   /* row-major 3x3 */
  double gapx_Ak1[3], gapx_Ak2[3];
  double mom_mm[6];      /* sum (p1-cm)(p1-cm)^T: xx xy xz yy yz zz        [TDTK_WANT_MOM2]   */
  double mom_dd[6];      /* sum (p2-cd)(p2-cd)^T                                              */
  double lum_udot;       /* sum u.delta (lum6DQuat's MZ(4), lum6Dquat.cc:163)     [TDTK_WANT_LUM]  */
} tdtk_pair_sums;

typedef struct tdtk_tree_info {
  uint64_t n_points, n_internal, n_leaves;
  uint32_t max_depth, max_leaf_points;
  uint64_t device_bytes;
  double build_ms, upload_ms;
} tdtk_tree_info;

typedef struct tdtk_icp_params {
  int algo;                /* TDTK_ALGO_*                           slam6D.cc:696-727   */
  int pairing_mode;        /* TDTK_CLOSEST_*                        slam6D.cc:756-763   */
  int max_num_iterations;  /* -i                                    icp6D.cc:122        */
  double max_dist_match2;  /* sqr(-d)                               icp6D.cc:80         */
  double epsilon_icp;      /* --epsICP                              icp6D.cc:266-267    */
  int quiet;               /* 0: print the reference's per-iteration RMS line           */
} tdtk_icp_params;

typedef struct tdtk_icp_result {
  int iterations;          /* value icp6D::match returns (icp6D.cc:284)                 */
  int converged;           /* 1 if the epsilon test fired, 0 if the cap / too few pairs */
  uint64_t last_pairs;     /* nr_pointPair                                              */
  double last_rms;         /* last `ret`                                                */
  double total_ms;         /* wall time of the loop (the reference's "TIME" line)       */
  double nn_ms;            /* of which: device time in the correspondence kernel (0 unless tdtk_kernel_timing) */
  double sums_ms;          /* and in the pair-sum kernels behind it (k_final alone when the sums are fused) */
} tdtk_icp_result;

/* ---- library ------------------------------------------------------------ */
const char* tdtk_last_error(void);
int tdtk_device_count(void);
/* Device memory of destroyed trees and scans is kept for the next handle of about that size (up to TDTK_POOL_MB megabytes
 * per device, default 1024; 0: every array goes straight back to the driver).  tdtk_pool_trim gives what is kept back
 * now and returns the number of bytes released.                                                                    */
size_t tdtk_pool_trim(void);
/* Diagnostics: how many device tree builds of this process had to be redone in order because a node cut at the plain
 * parallel sum of its points would have been cut elsewhere at the exact serial sum (build.hip, "speculative splits";
 * the tree that comes out is the same either way).  Expected to stay 0 outside the test that forces it.             */
uint64_t tdtk_build_respeculated(void);
const char* tdtk_version(void);

/* ---- model tree: replaces KDtree::KDtree(double**, int, int) (src/slam6d/kd.cc:46-49,
 * KDTreeImpl::create include/slam6d/kdTreeImpl.h:82-201) as created by
 * BasicScan::createSearchTreePrivate (src/slam6d/basicScan.cc:702-728).
 * xyz = "xyz reduced original" [M][3] row-major in the tree frame; it is COPIED
 * (the reference tree borrows it).  Builds the identical tree (same split rule,
 * same partition order), lays it out breadth-first and uploads it.             */
int tdtk_tree_create(const double* xyz, size_t M, int bucket_size, int device, tdtk_tree** out);
/* The same tree over the points of a resident scan as they are now, in the caller's order: what BasicScan
 * builds over "xyz reduced original" (src/slam6d/basicScan.cc:702-728) when asked before the scan has been moved;
 * the points never visit the host.                                                                    */
int tdtk_tree_create_from_scan(const tdtk_scan* scan, int bucket_size, tdtk_tree** out);
/* KDtreeMetaManaged (src/slam6d/kdMeta.cc:34-134) for a MetaScan: one tree over the CURRENT points of several
 * resident scans, concatenated in the order given (prepareTempIndices, kdMeta.cc:60-79); indices returned by
 * searches on it count through that concatenation.                                                    */
int tdtk_tree_create_from_scans(tdtk_scan* const* scans, int nscans, int bucket_size, tdtk_tree** out);
void tdtk_tree_destroy(tdtk_tree* t);
int tdtk_tree_get_info(const tdtk_tree* t, tdtk_tree_info* info);
/* diagnostic: rebuild the tree with the host builder (kd_build.cpp; this is its only use -- trees are always built on
 * the device) and compare it with the resident one.  mismatches = {node records, node radii, points, structure}. */
int tdtk_tree_verify(const tdtk_tree* t, uint64_t mismatches[4]);

/* ---- batched KDtree::FindClosest (kd.cc:78-87; _FindClosest kdTreeImpl.h:345-383).
 * q [K][3] in the tree frame, host memory.  idx[k] = index into the xyz given to
 * tdtk_tree_create, or -1 (the reference returns NULL).  d2 nullable.           */
int tdtk_find_closest(const tdtk_tree* t, const double* q, size_t K, double maxdist2,
                      int32_t* idx, double* d2);
/* same with device pointers (inputs already resident in HBM); stream = hipStream_t or NULL.
 * presorted != 0 promises that q is already spatially ordered (skips the binning pass).
 * With a caller's stream the call returns with the work queued on it; the kernels use the calling thread's
 * workspaces, so later tdtk calls of this thread are ordered behind them (stream-wait on an event), but the
 * caller must not issue a second tdtk_find_closest_dev on ANOTHER stream before the first has finished. */
int tdtk_find_closest_dev(const tdtk_tree* t, const double* d_q, size_t K, double maxdist2,
                          int32_t* d_idx, double* d_d2, int presorted, void* stream);

/* batched KDtree::FindClosestAlongDir (kd.cc:89-100; kdTreeImpl.h:390-425) */
int tdtk_find_closest_along_dir(const tdtk_tree* t, const double* q, const double* dir, size_t K,
                                double maxdist2, int32_t* idx, double* d2);

/* ---- SearchTree::getPtPairs, DataXYZ overload (src/slam6d/searchTree.cc:92-189), fused
 * with the per-thread Si pass of icp6D::match (icp6D.cc:170-191) and the APX/NAPX/LUM
 * pair loops.  Host buffers.  xyz_r = Target "xyz reduced" [*][3]; normal_r nullable
 * unless pairing_mode != 0 or TDTK_WANT_NAPX.  rnd > 1 draws the reference's keep-mask on the
 * host (one std::rand() per candidate in index order, i.e. serial-build semantics; SURVEY N-d)
 * and sends only the kept queries.  idx_out (nullable) [end-start]: model index per query or -1.
 * p1_out/p2_out/pn_out (nullable, [end-start][3]): compact pair list in query order, the
 * PtPair(s, t, normal) the reference pushes (for unmodified minimizers).
 * sums is overwritten (the reference accumulates into sum/centroids; callers add).     */
int tdtk_get_pt_pairs(const tdtk_tree* t, const double source_alignxf[16], const double* xyz_r,
                      const double* normal_r, size_t start, size_t end, int rnd, int pairing_mode,
                      double max_dist_match2, uint32_t want, const double* lum_D /*[6] or NULL*/,
                      int32_t* idx_out, double* p1_out, double* p2_out, double* pn_out,
                      tdtk_pair_sums* sums);

/* ---- device-resident data scan: replaces the "xyz reduced"/"normal reduced" arrays of
 * BasicScan (src/slam6d/basicScan.cc:532-668) for the duration of matching.  Points are
 * copied, spatially reordered once (results are reported in the caller's order).        */
int tdtk_scan_create(const double* xyz_reduced, const double* normal_reduced /*nullable*/, size_t N,
                     int device, tdtk_scan** out);
void tdtk_scan_destroy(tdtk_scan* s);
size_t tdtk_scan_size(const tdtk_scan* s);
/* Scan::transformReduced (src/slam6d/scan.cc:851-875): in-place transform3 of every point
 * (and transform3normal of every normal), incremental, same arithmetic.                 */
int tdtk_scan_transform(tdtk_scan* s, const double alignxf[16]);
/* copy the current points (caller's order) back, e.g. after matching */
int tdtk_scan_download(const tdtk_scan* s, double* xyz_out, double* normal_out /*nullable*/);

/* "xyz reduced original" (BasicScan::copyReducedToOriginal, src/slam6d/basicScan.cc:739-757): after
 * tdtk_scan_mark_original the points as they are now count as the original; the first call that moves the scan
 * (tdtk_scan_transform, tdtk_icp_match, the pose updates) first saves them on the device, so that
 * tdtk_tree_create_from_scan / tdtk_scan_download_original keep answering with the original.            */
int tdtk_scan_mark_original(tdtk_scan* s);
int tdtk_scan_download_original(const tdtk_scan* s, double* xyz_out);

/* Scan::getPtPairs (scan.cc:1220-1260) over a resident scan: whole-scan pass + sums. */
int tdtk_scan_pairs(const tdtk_tree* model, const double source_alignxf[16], tdtk_scan* data,
                    int pairing_mode, double max_dist_match2, uint32_t want,
                    const double* lum_D /*[6] or NULL*/, int32_t* idx_out /*host, nullable*/,
                    tdtk_pair_sums* sums);

/* icp6D::Point_Point_Error (src/slam6d/icp6D.cc:293-367): the closest-point pairs of the scan within
 * max_dist_match of the model, error = -0.39894228 * mean exp(|p1-p2|^2 * log(scale_max) / max_dist_match^2). */
int tdtk_point_point_error(const tdtk_tree* model, const double source_alignxf[16], tdtk_scan* data,
                           double max_dist_match, double scale_max, uint64_t* np_out /*nullable*/, double* error_out);

/* ---- minimizers: icp6Dminimizer::Align_Parallel with the merged sums in slot 0
 * (icp6Dquat.cc:515-634, icp6Dsvd.cc:170-280, icp6Dapx.cc:136-307, icp6Dnapx.cc:34-149),
 * serial-Align semantics (S normalised by 1/n; SVD reflection fix), and the serial-only
 * ORTHO / DUAL / HELIX / LUMEULER / LUMQUAT / QUAT_SCALE from sums filled with TDTK_WANT_MOM2.
 * alignxf is in/out: LUMEULER and LUMQUAT read the current scan's transMat from it
 * (icp6D.cc:237-241); the others ignore the input.  Returns the RMS the reference returns
 * (`ret`) through *rms.                                                                  */
int tdtk_align(int algo, const tdtk_pair_sums* sums, double alignxf[16], double* rms);

/* ---- icp6D::match (src/slam6d/icp6D.cc:104-285), device-resident loop.
 * model_dalignxf = PreviousScan->dalignxf.  data is moved in place exactly like
 * CurrentScan->transform(alignxf, ...) per iteration; data_transMat / data_dalignxf
 * (in/out) get alignxf premultiplied per iteration (Scan::transformMatrix, scan.cc:878-898).
 * trace (nullable, capacity trace_cap rows of 18 doubles): per iteration
 * {pairs, rms, alignxf[16]}.                                                             */
int tdtk_icp_match(const tdtk_tree* model, const double model_dalignxf[16], tdtk_scan* data,
                   double data_transMat[16], double data_dalignxf[16], const tdtk_icp_params* prm,
                   tdtk_icp_result* res, double* trace, int trace_cap);

/* The same loop with `-R <rnd>` (icp6D's rnd, handed to getPtPairs: "take about 1/rnd-th of the numbers only",
 * src/slam6d/searchTree.cc:116-118): per iteration one std::rand() per point of the data scan in its index order decides
 * whether the point is a candidate of that iteration (globals.icc:607-610), drawn on the host at the moment the iteration
 * starts -- the process's random stream is consumed exactly as a serial build of the reference consumes it -- and sent to
 * the device as one bit per point; every point still moves with every alignxf.  rnd <= 1: tdtk_icp_match.            */
int tdtk_icp_match_rnd(const tdtk_tree* model, const double model_dalignxf[16], tdtk_scan* data,
                       double data_transMat[16], double data_dalignxf[16], const tdtk_icp_params* prm, int rnd,
                       tdtk_icp_result* res, double* trace, int trace_cap);

/* ---- lum6DEuler::covarianceEuler (src/slam6d/lum6Deuler.cc:94-251) for one link:
 * first = model tree + its dalignxf, second = resident data scan.  C[36] row-major, CD[6].
 * Returns the pair count through *m; C/CD are zero if m <= 2 or ss < 1e-13.              */
int tdtk_lum_link(const tdtk_tree* first, const double first_dalignxf[16], tdtk_scan* second,
                  double max_dist_match2, double C[36], double CD[6], uint64_t* m, double* ss);

/* lum6DEuler::FillGB3D's link loop (src/slam6d/lum6Deuler.cc:265-303) for a batch of links on one
 * device: all correspondence passes and reductions are enqueued back to back and synchronised
 * once.  first[i] / first_dalignxf[i*16] / second[i] describe link i.  ss is evaluated from the
 * normal equations (sum|d|^2 - D.MZ, exact for the solved D) instead of a second pass over the
 * pairs; tdtk_lum_link keeps the reference's two-pass form.  Outputs are per link.          */
int tdtk_lum_links(int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                   tdtk_scan* const* second, double max_dist_match2, double* C /*[nlinks][36]*/,
                   double* CD /*[nlinks][6]*/, uint64_t* m /*[nlinks]*/, double* ss /*[nlinks]*/);

/* Batched Scan::getPtPairs over a list of links (first[i] = model tree + dalignxf, second[i] =
 * resident data scan) with any accumulator blocks: one sync for the whole batch.  sums[nlinks]. */
int tdtk_links_pair_sums(int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                         tdtk_scan* const* second, double max_dist_match2, uint32_t want,
                         tdtk_pair_sums* sums);

/* ---- graph-SLAM back-ends by their -G id (src/slam6d/slam6D.cc:784-804): lum6DEuler 1 (lum6Deuler.cc),
 * lum6DQuat 2 (lum6Dquat.cc), ghelix6DQ2 3 (ghelix6DQ2.cc), gapx6D 4 (gapx6D.cc).
 * tdtk_graph_link_blocks: the per-link quantities of this rank's links (covarianceEuler / covarianceQuat /
 *   genBBdForLinkedPair / genBArotForLinkedPair), tdtk_graph_block_doubles(backend) doubles per link; device
 *   passes batched, one sync.
 * tdtk_graph_solve_update: given the blocks of ALL links (after the all-reduce; zeros for links nobody could
 *   fill), the scatter in link order (FillGB3D and its counterparts), the solve, and the pose update of scans
 *   1..nscans-1: transMat / dalignxf [nscans][16], rPos / rPosTheta [nscans][3] are updated in place, resident
 *   scans (scans[i] may be NULL) are moved, xf_out [nscans][32] (nullable) receives the one or two transforms
 *   applied per scan.  state: ghelix6DQ2's B | bd ((6n)^2 + 6n doubles, zeroed once per doGraphSlam6D call),
 *   gapx6D's T (3n, likewise); NULL for the LUM back-ends.  *ret = what doGraphSlam6D's loop tests.       */
enum { TDTK_GRAPH_LUMEULER = 1, TDTK_GRAPH_LUMQUAT = 2, TDTK_GRAPH_GHELIX = 3, TDTK_GRAPH_GAPX = 4 };
int tdtk_graph_block_doubles(int backend);
int tdtk_graph_link_blocks(int backend, int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                           tdtk_scan* const* second, double max_dist_match2, double* blocks);
int tdtk_graph_solve_update(int backend, int nlinks, const int32_t* from, const int32_t* to, const double* blocks,
                            int nscans, double* transMat, double* dalignxf, double* rPos, double* rPosTheta,
                            tdtk_scan* const* scans, double* state, double* xf_out, double* ret);

/* ---- multi-GPU: one process per GPU, links sharded, ONE all-reduce (sum, fp64) of the per-link blocks per global
 * iteration over RCCL / xGMI (the reference's unit of parallelism is the same: `omp parallel for` over
 * links in FillGB3D, lum6Deuler.cc:270-283).  RCCL is bound at run time; a single-GPU user never touches it.
 * tdtk_comm_unique_id: rank 0 draws the id (ncclGetUniqueId) and hands the 128 bytes to the other ranks by whatever
 *   the host program has (MPI, a file, torch.distributed); tdtk_comm_create: ncclCommInitRank on `device`.
 * tdtk_graph_exchange: blocks[n] <- sum over ranks, in place (host buffer; staged through the communicator's own
 *   device buffer and stream).
 * tdtk_graph_deal_links: owner[l] = rank that evaluates link l -- round-robin / (from+to) % world for scans of equal
 *   size, longest-processing-time-first by the point count of the link's second scan otherwise.
 * tdtk_graph_iteration: link blocks of this rank's links (mine[] = their indices in the link list; first / second /
 *   first_dalignxf describe them in that order) -> exchange -> tdtk_graph_solve_update, all inside the library.
 * Failure semantics with more than one rank: arguments are validated before the link passes; a rank whose link passes
 *   fail still takes part in the collective (a status slot rides behind the blocks), and then EVERY rank returns an
 *   error -- the failing rank its own code, the others TDTK_EPEER -- so nobody is left waiting in ncclAllReduce.  A
 *   failure that cannot be carried through the collective (no staging memory, a failed copy) aborts the communicator
 *   (ncclCommAbort): the peers' collective returns an error and the communicator is dead on all ranks.  The staging
 *   buffers are allocated by tdtk_comm_create.                                                                       */
#define TDTK_COMM_ID_BYTES 128
typedef struct tdtk_comm tdtk_comm;
int tdtk_comm_unique_id(char id[TDTK_COMM_ID_BYTES]);
int tdtk_comm_create(const char id[TDTK_COMM_ID_BYTES], int rank, int world, int device, tdtk_comm** out);
void tdtk_comm_destroy(tdtk_comm* c);
int tdtk_comm_info(const tdtk_comm* c, int* rank, int* world, uint64_t* n_allreduce);
int tdtk_graph_exchange(tdtk_comm* c, double* blocks, size_t n);
/* the number of ranks RCCL itself reports for the communicator (ncclCommCount); tdtk_comm_create fails unless it equals
 * the `world` it was given */
int tdtk_comm_rccl_world(const tdtk_comm* c);
int tdtk_graph_deal_links(int nlinks, const int32_t* from, const int32_t* to, const uint64_t* scan_points /*[nscans] or NULL*/,
                          int nscans, int world, int32_t* owner);
int tdtk_graph_iteration(int backend, tdtk_comm* comm /*nullable*/, int nlinks, const int32_t* from, const int32_t* to,
                         int n_mine, const int32_t* mine, const tdtk_tree* const* first, const double* first_dalignxf,
                         tdtk_scan* const* second, double max_dist_match2, int nscans, double* transMat, double* dalignxf,
                         double* rPos, double* rPosTheta, tdtk_scan* const* scans, double* state, double* xf_out, double* ret);

/* ---- ELCH loop closing (-L 1), host control flow: elch6D::graph_balancer (src/slam6d/elch6D.cc:186-279) on an
 * undirected weighted graph given as an edge list -- weights[f] = 0, weights[l] = 1, every vertex on a shortest path
 * between two junctions gets the distance-proportional value, branches inherit (weights[] entries of vertices the
 * balancer never reaches are left as they were).  tdtk_pair_sums_merge: the base block (n, sum, centroids, Si) of the
 * union of several whole-scan passes, for a MetaScan as the data scan of icp6D::match (scan.cc:1305-1327).       */
int tdtk_elch_graph_balancer(int nvertices, int nedges, const int32_t* from, const int32_t* to, const double* w,
                             int first, int last, double* weights);
int tdtk_pair_sums_merge(int count, const tdtk_pair_sums* parts, tdtk_pair_sums* out);
/* Graph::Graph(int nodes, double cldist2, int loopsize) (src/slam6d/graph.cc:107-130), the graph matchGraph6Dautomatic rebuilds
 * before every LUM round (slam6D.cc:525-532): the chain i -> i + 1, then every pair (j, k) with k - j > loopsize whose scanner
 * positions rPos [nscans][3] are closer than sqrt(cldist2), j-major.  *nlinks = the number of links; at most cap are written. */
int tdtk_graph_links(int nscans, const double* rPos, double cldist2, int loopsize, int32_t* from, int32_t* to, int cap, int* nlinks);

/* Scan::transform for many resident scans at once: scan i is moved in place by A1[i] and then by A2[i]
 * (A2 nullable), e.g. Scan::transformToEuler / transformToQuat (scan.cc:1061-1104) = M4inv(transMat) then
 * the new pose; one kernel launch for all of them.                                               */
int tdtk_scans_transform2(int count, tdtk_scan* const* scans, const double* A1 /*[count][16]*/,
                          const double* A2 /*[count][16] or NULL*/);

/* FillGB3D's scatter-add of every link's (C, CD) into the dense G (6(n-1))^2 / B 6(n-1), in link order
 * (lum6Deuler.cc:285-300), then graphSlam6D::solveSparseCholesky (graphSlam6D.cc:345-379) -> X.
 * from/to are scan numbers (0 = the fixed scan).  With links sharded over ranks, all-reduce the
 * per-link blocks (42 doubles each, zeros for links a rank does not own) and call this on every
 * rank: the result does not depend on the number of ranks.  G_out / B_out nullable.            */
int tdtk_lum_assemble_solve(int nlinks, const int32_t* from, const int32_t* to, const double* C /*[nlinks][36]*/,
                            const double* CD /*[nlinks][6]*/, int nscans, double* X /*[6(nscans-1)]*/,
                            double* G_out, double* B_out);

/* Pose update of lum6DEuler::doGraphSlam6D (lum6Deuler.cc:378-473) for scans 1..n-1:
 * result = Ha^-1 * X_i, pose -= result, Scan::transformToEuler (scan.cc:1061-1083: transform by
 * M4inv(transMat), then by EulerToMatrix4(new pose)).  transMat/dalignxf/rPos/rPosTheta are
 * [n][16]/[n][16]/[n][3]/[n][3], updated in place; scans[i] (nullable) are moved on the device;
 * xf_out (nullable, [n][32]) receives the two matrices applied to scan i so that a caller can
 * queue them for scans that are not resident.  *ret = sum |dxyz| / n  (lum6Deuler.cc:470-473). */
int tdtk_lum_update_poses(int nscans, const double* X, double* transMat, double* dalignxf,
                          double* rPos, double* rPosTheta, tdtk_scan* const* scans, double* xf_out,
                          double* ret);

/* graphSlam6D::solveSparseCholesky(GraphMatrix*, B) (src/slam6d/graphSlam6D.cc:345-379,
 * 477-503): dense SPD solve of G x = B (G row-major n x n, entries with |v| <= 1e-5
 * dropped like convertToCS does).  x may alias B.                                       */
int tdtk_solve_spd(const double* G, const double* B, int n, double* x);
/* graphSlam6D::solveSparseCholesky(const Matrix&, B) (graphSlam6D.cc:302-343) as gapx6D calls it on
 * a matrix that is not exactly symmetric: CSparse's cs_cholsol reads the UPPER triangle only.   */
int tdtk_solve_chol_upper(const double* G, const double* B, int n, double* x);
/* dense inverse (newmat `.i()`), row-major n x n */
int tdtk_invert(const double* A, int n, double* Ainv);

/* ---- octree reduction, "-r <voxelSize>" with the default centre mode: replaces
 * Scan::calcReducedPoints (src/slam6d/scan.cc:577-603, reduction_nrpts == 0) = BOctTree<double>(pts,
 * n, voxelSize) (include/slam6d/Boctree.h:222-270) + GetOctTreeCenter (:928-948).  out_xyz has room
 * for n points; *n_out = number of occupied leaf cells; centres come out in the reference's
 * depth-first child order (that order feeds the kd-tree build, so it matters).               */
int tdtk_reduce_octree(const double* xyz, size_t n, double voxel_size, int device, double* out_xyz,
                       size_t* n_out);
/* The same with `-O <nrpts>` (reduction_nrpts, scan.cc:586-596): 0 = the centres above; 1 = one random point per occupied
 * leaf (BOctTree::GetOctTreeRandom, Boctree.h:985-1018); N > 1 = up to N random points per leaf (Boctree.h:1020-1062 with
 * rm_scatter == false).  Leaves, their order and the order of the points inside a leaf (the reference's in-place
 * partitions, Boctree.h:1737-1816) come from the device; the draws are std::rand() on the host, leaf by leaf, as the
 * reference makes them (seed with std::srand; reproducible against a serial reference build).  TDTK_EUNSUP for
 * nrpts == -1 (GetOctTreeAvg sums into uninitialised memory, Boctree.h:961-964) and for rm_scatter != 0 (the reference's
 * loop walks a stale child pointer there, Boctree.h:1032-1040).  out_xyz has room for n points.                       */
int tdtk_reduce_octree_nrpts(const double* xyz, size_t n, double voxel_size, int nrpts, int rm_scatter, int device,
                             double* out_xyz, size_t* n_out);

/* ---- point normals, "-z" / "-a 10" / pairing modes 1 and 2: replaces Scan::calcNormals
 * (src/slam6d/scan.cc:398-427) = calculateNormalsApxKNN(normals, points, k, rPos, eps)
 * (src/slam6d/normals.cc:35-111; the reference calls it with k = 10, eps = 1.0).  For every point: its k
 * (1+eps)-approximate nearest neighbours exactly as the ANN 1.1.1 kd-tree (bucket size 1, sliding midpoint)
 * returns them -- the same tree is built and walked in the same order on the device --, their mean and
 * covariance, the eigenvector of the smallest eigenvalue (newmat tred2/tql2 arithmetic), flipped so that it
 * points away from rPos... i.e. n . (p - rPos) >= 0, normalised.  Lists and normals are bit-identical to the
 * library's.  normals_out [n][3]; knn_out (nullable) [n][k] neighbour indices in list order (nearest first).
 * Errors: n == 0 ("XYZ data is empty", scan.cc:408), k > n (ANN aborts: kd_search.cpp:103), k > 32,
 * non-finite coordinates.                                                                     */
int tdtk_normals_apx_knn(const double* xyz, size_t n, int k, const double rPos[3], double eps, int device,
                         double* normals_out, int32_t* knn_out);
/* the same for a resident scan, from its current points; the result becomes its "normal reduced" */
int tdtk_scan_calc_normals(tdtk_scan* s, int k, const double rPos[3], double eps);

/* ---- instrumentation: per-kernel device time of the last call on this thread (ms) and
 * traversal counters of the last counting run.                                          */
/* HIP events around the search and pair-sum kernels of every pass: a profiling aid, off by default (process-wide;
 * also TDTK_KERNEL_TIMING=1 in the environment) because the events themselves cost ~10 us per ICP iteration.  While
 * it is off tdtk_icp_result.nn_ms / sums_ms, tdtk_last_kernel_ms and out[0], out[1] of tdtk_last_timings are 0.
 * Returns the previous setting.  (The reference has nothing of the kind: icp6D::match prints its wall time only,
 * icp6D.cc:279-283 -- tdtk_icp_result.total_ms.) */
/* Deferred scan moves.  The batched pose update of a graph-SLAM round (tdtk_graph_solve_update / tdtk_graph_iteration,
 * tdtk_scans_transform2, tdtk_lum_update_poses) returns while the move of the resident scans is still running on the device; a
 * process-wide fence makes the next library call of ANY host thread on that device wait for it before it touches a scan
 * or a tree (every entry point that takes a tree / scan / device argument, including the destroy and mark_original
 * calls).  The read-outs in this section (tdtk_kernel_timing, tdtk_last_kernel_ms, tdtk_last_timings, the visit
 * counters) deliberately do NOT wait -- they touch no scan and must not end the overlap.  TDTK_SYNC_MOVES=1 in the
 * environment restores "the scans have moved when the call returns". */
int tdtk_kernel_timing(int on);
int tdtk_last_kernel_ms(double* nn_ms);
/* out[0] = search kernel, out[1] = pair-sum kernels of the last pass on this thread, out[2] = the k-NN + PCA kernel
 * of the last calcNormals (HIP events on the stream the kernels were launched on), out[3] = wall time of the last
 * device tree build (a chain of ~40 launches) */
int tdtk_last_timings(double out[4]);
/* on != 0: every FindClosest pass of the calling thread on `device` runs the instrumented instantiation of the
 * kernel it would have used (identical traversal and results, slower) and adds to four counters, zeroed here;
 * tdtk_visit_counters reads {internal nodes, buckets, bucket points visited, queries issued} -- the exact
 * n_int / n_pts of SURVEY 8(d)'s algorithmic bytes for exactly the launches that ran (warm radius included) --
 * and, for calcNormals, out[4..6] = {ANN splitting nodes, leaf points visited, points processed}; out[7] = queries of
 * repeated passes that were searched a second time with every quick check (a warm query walks without the quick check of
 * its divergent visits and is searched again, cold, if it accepted a point that improved closest_d2 by a rounding's
 * worth: DESIGN.md section 4, "the quick check deferred").
 * on == 2: while counting, every search also starts cold (no warm start, no deferred quick check): the counters then hold
 * the walk of the reference's _FindClosest (kdTreeImpl.h:345-383) over the same queries -- SURVEY 8(d)'s n_int / n_pts,
 * "properties of (tree, query, maxdist2), independent of implementation"; the results, and so a loop's path, are unchanged. */
int tdtk_visit_counting(int device, int on);
int tdtk_visit_counters(int device, uint64_t out[8]);
/* measured roofline denominators: kind 0 = HBM stream copy over `bytes` (read + written per pass), kind 1 =
 * repeated reads of an L2-resident buffer (bytes <= 16 MB); best of `reps` passes in GB/s */
int tdtk_measure_bandwidth(int device, int kind, size_t bytes, int reps, double* gbs);
int tdtk_count_visits(const tdtk_tree* t, const double* q, size_t K, double maxdist2,
                      uint64_t counters[3] /* internal nodes, leaves, leaf points */);
/* Correspondence hashes of the resident loop (off by default, process-wide; returns the previous setting): while on,
 * every iteration of tdtk_icp_match also reduces its correspondences to one word on the device -- the XOR over the found
 * queries of (model index * 1315423911 + query index), both in the caller's numbering, i.e. the hash of the index array
 * SearchTree::getPtPairs' loop walks (searchTree.cc:118-147) -- so that a test can compare the loop's INDICES with the
 * reference's iteration by iteration at a million points without downloading them.  tdtk_icp_index_hashes returns the
 * previous setting; a negative argument only asks.  tdtk_icp_last_hashes: the words of the calling thread's last
 * tdtk_icp_match, on whatever device it ran (*n_out = how many passes it ran, at most 1024 are kept; 0 while switched off). */
int tdtk_icp_index_hashes(int on);
int tdtk_icp_last_hashes(uint64_t* out, int cap, int* n_out);

/* ---- on-disk formats either side of the path (host only): uos ASCII scans with the -m/-M range
 * filter (src/scanio/helper.cc:564-880, src/slam6d/pointfilter.cc:162-188), .pose files
 * (helper.cc:192-234) and .frames files (src/slam6d/basicScan.cc:902-917).                  */
int tdtk_io_read_uos(const char* path, double range_max, double range_min, double** xyz_out,
                     size_t* n_out);
void tdtk_io_free(void* p);
int tdtk_io_read_pose(const char* path, double rPos[3], double rPosTheta[3]);
int tdtk_io_write_frames(const char* path, const double* transMats, const int* types, size_t count,
                         int append);

/* ---- host-only diagnostics (no device needed; used by the CPU test tier) ----------------
 * tdtk_host_tree_layout: run the host tree builder only.  perm_out [M] = caller indices in
 * leaf order (== the reference's post-build pointer order); stats = {internal nodes, leaves,
 * max depth, max leaf points}.  tdtk_host_m4inv / tdtk_host_mmult: the bit-exact M4inv /
 * MMult (globals.icc:762-785, 298-328) used for every query transform.                      */
int tdtk_host_tree_layout(const double* xyz, size_t M, int bucket_size, int32_t* perm_out,
                          uint64_t stats[4]);
int tdtk_host_m4inv(const double in[16], double out[16]);
void tdtk_host_mmult(const double a[16], const double b[16], double out[16]);
/* the pose conversions the pose updates use: EulerToMatrix4 / Matrix4ToEuler (globals.icc:501-576),
 * QuatToMatrix4 / Matrix4ToQuat (globals.icc:988-1075) */
void tdtk_host_euler_to_matrix4(const double rPos[3], const double rPosTheta[3], double out[16]);
void tdtk_host_matrix4_to_euler(const double in[16], double rPosTheta[3], double rPos[3]);
void tdtk_host_quat_to_matrix4(const double quat[4], const double t[3], double out[16]);
void tdtk_host_matrix4_to_quat(const double in[16], double quat[4], double t[3]);

#ifdef __cplusplus
}
#endif
#endif /* TDTK_HIP_H */
