// Internal structures of lib3dtk_hip.so (not part of the C ABI).
#pragma once
#include <cstdlib>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "tdtk_hip.h"

namespace tdtk {

// ---- flattened kd-tree, breadth-first -------------------------------------------------
// One 64-byte record per INTERNAL node (two per 128-byte cache line).  Leaves have no
// record: a child reference that points to a leaf carries (start,count) of the leaf's run
// in the permuted point array, so visiting a leaf costs no dependent node load.
//
// child reference (32 bit):  bit31 = leaf flag, bit30 = one bit of the split axis
// (c1 carries axis bit 0, c2 carries axis bit 1), bits[29:0] = value.
//   internal: value = index of the node record
//   leaf    : value = (start << cb) | count          (packed mode, cb = count bits)
//             value = leaf id into leaf_tab[]        (table mode, when it does not fit)
struct alignas(64) KdNode {
  double cx, cy, cz;  // box centre      node.center   (kdTreeImpl.h:128-130)
  double hx, hy, hz;  // half extents    node.dx/dy/dz (kdTreeImpl.h:132-134)
  double splitval;    //                 node.splitval (kdTreeImpl.h:170)
  uint32_t c1, c2;    // child refs      node.child1/child2
};
static_assert(sizeof(KdNode) == 64, "KdNode must be 64 bytes");

// model point in leaf order (the reference's post-partition pointer order), 32 bytes so a
// 128-byte line holds exactly four points and a point never straddles lines.
struct alignas(32) KdPoint {
  double x, y, z;
  int32_t orig;  // index into the caller's xyz
  int32_t pad;
};
static_assert(sizeof(KdPoint) == 32, "KdPoint must be 32 bytes");

// The part of a node every visit needs, with the box in fp32: 48 bytes (three 16-byte loads instead of four) and a
// box test at full VALU rate.  The fp32 box only ever decides what it can decide rigorously (see visit_node_hot in
// kernels.hip); the rest falls back to the fp64 box of KdNode, so the visits stay the reference's, node for node.
struct alignas(16) KdHot {
  float cx, cy, cz, hx;
  float hy, hz;
  uint32_t axis;      // the split axis again, in the clear (0 / 1 / 2): saves the persistent-lane kernel the bit fiddling
  uint32_t pad1;
  double splitval;
  uint32_t c1, c2;    // child references, WITH the axis bits (REF_AXIS) like KdNode's
};
static_assert(sizeof(KdHot) == 48, "KdHot must be 48 bytes");
// (One record per 64-byte line instead -- no record straddling two lines -- was measured again in round 3 with the buckets
// down to one round trip: 0.1954 .. 0.1991 ms against 0.1970 .. 0.1974, i.e. nothing; gpurun_out/r3f.)

// Two levels per round trip (round 3): the hot part of a node TOGETHER with the hot parts of its two children, 128 bytes,
// one cache line.  A lane that arrives at node X tests X's box, picks the near child N and -- when N is an internal node
// that X's record already describes -- tests N's box and descends again, all behind ONE fetch.  Same visits, same order,
// same decisions; half the dependent round trips of a walk.  Fields of a child that is a leaf are zero.
struct alignas(128) KdFat {
  float cx, cy, cz, hx;            // quad 0: X box
  float hy, hz; uint32_t c1, c2;   // quad 1: X box, X child references (with X's axis bits)
  double splitval, a_split;        // quad 2: split values of X and of child 1 ("A")
  double b_split; uint32_t a_c1, a_c2;   // quad 3: split value of child 2 ("B"), A's child references
  float a_cx, a_cy, a_cz, a_hx;    // quad 4: A box
  float a_hy, a_hz, b_hy, b_hz;    // quad 5
  float b_cx, b_cy, b_cz, b_hx;    // quad 6: B box
  uint32_t b_c1, b_c2, pad0, pad1; // quad 7: B's child references
};
static_assert(sizeof(KdFat) == 128, "KdFat must be 128 bytes");

// pool.cpp: hipMalloc / hipFree for the arrays of trees and resident scans, with freed blocks kept for the next request
// (declared with plain types so that host-only sources can include this header)
int pool_malloc_raw(void** out, size_t bytes);   // 0 = success, else the hipError_t value
// Product and lab.  The default build (`make`, lib3dtk_hip.so) holds what a slam6D run uses and reads thirteen environment
// switches; `make LAB=1` (lib3dtk_hip_lab.so, -DTDTK_LAB) adds the kernels and policies that were
// built, measured and lost (NEGATIVES.md) with the switches that select them -- the tests that compare those variants with
// the product path load that library.  lab_env() is getenv() in the lab build and nothing in the product.
#ifdef TDTK_LAB
constexpr bool kLab = true;
inline const char* lab_env(const char* name) { return getenv(name); }
#else
constexpr bool kLab = false;
inline const char* lab_env(const char*) { return nullptr; }
#endif
void pool_free(void* p);
size_t pool_trim();                              // gives every shelved block back to the driver; returns the bytes

constexpr uint32_t REF_LEAF = 0x80000000u;
constexpr uint32_t REF_AXIS = 0x40000000u;
constexpr uint32_t REF_VAL = 0x3FFFFFFFu;
constexpr uint32_t REF_DONE = 0xFFFFFFFFu;
constexpr uint32_t REF_STAGE1 = 0xFFFFFFFEu;   // persistent-lane kernel, pipelined hand-out: the lane's query has been requested, not yet set up

struct LeafEntry {
  int32_t start, count;
};

struct HostTree {
  std::vector<KdNode> nodes;
  std::vector<double> node_r;  // bounding-sphere radius per node (FindClosestAlongDir)
  std::vector<KdPoint> pts;    // leaf order
  std::vector<LeafEntry> leaf_tab;  // always filled (table mode uses it on the device)
  uint32_t root_ref = 0;
  int cb = 0;              // count bits in packed leaf refs
  bool table_mode = false; // leaf refs index leaf_tab
  double bbmin[3], bbmax[3];
  uint64_t n_internal = 0, n_leaves = 0;
  uint32_t max_depth = 0, max_leaf_points = 0;
};

// Builds the identical tree to KDTreeImpl::create (kdTreeImpl.h:82-201).  Returns false on
// M == 0 (the reference throws).  threads <= 1 builds serially.
bool build_tree(const double* xyz, size_t M, int bucket, HostTree& out, std::string& err);

// ---- host 4x4 helpers (bit-exact restatements of globals.icc formulas) -----------------
int m4inv(const double* Min, double* Mout);                      // globals.icc:762-785
void mmult(const double* M1, const double* M2, double* Mout);    // globals.icc:298-328
void m4identity(double* M);
void euler_to_matrix4(const double* rPos, const double* rPosTheta, double* alignxf);  // globals.icc:501-531
void matrix4_to_euler(const double* alignxf, double* rPosTheta, double* rPos);        // globals.icc:541-576

// ---- minimizers ------------------------------------------------------------------------
int align_from_sums(int algo, const tdtk_pair_sums& s, double alignxf[16], double* rms,
                    std::string& err);
bool invert_dense(int n, const double* A, double* Ainv);  // LU with partial pivoting
bool solve_spd_dense(int n, const double* G, const double* B, double* x, double drop);
// Cholesky of a matrix already in skyline storage (row i: columns first[i] .. i at sky[off[i] - first[i] + k]) + the two substitutions
bool skyline_solve(int n, const int* first, const size_t* off, double* sky, const double* B, double* y, double* x);
// the calling host thread's context stream on `device` (a hipStream_t; api.cpp), for library code outside api.cpp
int ctx_stream(int device, void** stream_out);

void helix_compute_rt(const double ccs[6], double* alignxf);                 // icp6Dhelix.cc:144-206
void matrix4_to_quat_t(const double* mat, double quat[4], double t[3]);       // globals.icc:1032-1075
void quat_to_matrix4(const double quat[4], const double t[3], double* mat);   // globals.icc:988-1022
void apx_compute_rt(const double x[3], const double dx[3], double* alignxf);  // icp6Dapx.cc:310-335

void set_error(const std::string& s);

}  // namespace tdtk
