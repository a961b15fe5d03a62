// Host-side construction of the model-scan search tree.
//
// The tree must be *the same tree* the reference builds (KDTreeImpl::create,
// include/slam6d/kdTreeImpl.h:82-201) -- same split axis rule, same mean split value, same
// in-place partition order -- because nearest-neighbour ties (the bundled scans contain exact
// duplicate points) are broken by visiting order (SURVEY N-b).  What is ours is the shape of
// the computation and of the result: an iterative breadth-first work list instead of
// recursion + new, emitting 64-byte internal-node records in BFS order, leaves as (start,
// count) runs of one permuted 32-byte point array, and child references that embed the leaf
// run so the GPU never chases a pointer to learn where a bucket lives.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>

#include "tdtk_internal.h"

namespace tdtk {
namespace {

struct Work {
  uint32_t start, n;
  int32_t parent;  // node index or -1 for the root
  uint8_t side;    // 0 -> parent's c1, 1 -> c2
  uint32_t depth;
};

struct Extent {
  double lo[3], hi[3], mean[3];
};

// bbox + mean of a run, in run order (the mean is an order-dependent fp64 sum:
// kdTreeImpl.h:94-111 starts from the first point and adds the rest left to right).
inline void measure(const double* xyz, const uint32_t* run, uint32_t n, Extent& e)
{
  const double* p0 = xyz + 3 * (size_t)run[0];
  for (int a = 0; a < 3; a++) e.lo[a] = e.hi[a] = e.mean[a] = p0[a];
  for (uint32_t i = 1; i < n; i++) {
    const double* p = xyz + 3 * (size_t)run[i];
    for (int a = 0; a < 3; a++) {
      const double v = p[a];
      e.lo[a] = (v < e.lo[a]) ? v : e.lo[a];
      e.hi[a] = (e.hi[a] < v) ? v : e.hi[a];
      e.mean[a] += v;
    }
  }
  for (int a = 0; a < 3; a++) e.mean[a] /= n;
}

// Hoare partition of kdTreeImpl.h:172-182: "< splitval" to the left, ">=" to the right,
// scanning inwards and swapping the stopped pair.  Returns the size of the left part.
inline uint32_t split_run(const double* xyz, uint32_t* run, uint32_t n, int axis, double splitval)
{
  uint32_t* l = run;
  uint32_t* r = run + n - 1;
  for (;;) {
    while (xyz[3 * (size_t)*l + axis] < splitval) ++l;
    while (xyz[3 * (size_t)*r + axis] >= splitval) --r;
    if (r < l) break;
    std::swap(*l, *r);
  }
  return (uint32_t)(l - run);
}

inline int bits_for(uint64_t v)
{
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b;
}

}  // namespace

bool build_tree(const double* xyz, size_t M, int bucket, HostTree& T, std::string& err)
{
  if (!xyz || M == 0) { err = "cannot create kdtree with zero points"; return false; }
  if (bucket < 1) { err = "bucket size must be >= 1"; return false; }
  if (M > (size_t)REF_VAL || M * sizeof(KdPoint) >= (1ull << 32)) {
    err = "model scan too large (30-bit references / 32-bit byte offsets: < 2^27 points)";
    return false;
  }

  // the reference recurses on an empty side and throws for NaN / inf input; here it is an error up front
  for (size_t i = 0; i < 3 * M; i++)
    if (!std::isfinite(xyz[i])) { err = "degenerate split (non-finite coordinates?)"; return false; }

  std::vector<uint32_t> perm(M);
  std::iota(perm.begin(), perm.end(), 0u);

  T = HostTree();
  T.nodes.reserve(M / (size_t)std::max(4, bucket / 2) + 16);
  std::vector<Work> fifo;
  fifo.reserve(1024);
  fifo.push_back({0u, (uint32_t)M, -1, 0, 1});

  auto attach = [&](const Work& w, uint32_t ref) {
    if (w.parent < 0) { T.root_ref = ref; return; }
    KdNode& p = T.nodes[(size_t)w.parent];
    if (w.side == 0) p.c1 = (p.c1 & REF_AXIS) | ref;
    else p.c2 = (p.c2 & REF_AXIS) | ref;
  };
  auto make_leaf = [&](const Work& w) {
    const uint32_t id = (uint32_t)T.leaf_tab.size();
    T.leaf_tab.push_back({(int32_t)w.start, (int32_t)w.n});
    T.max_leaf_points = std::max(T.max_leaf_points, w.n);
    attach(w, REF_LEAF | id);  // leaf id for now; re-encoded below
  };

  Extent e;
  for (size_t head = 0; head < fifo.size(); head++) {
    const Work w = fifo[head];
    uint32_t* run = perm.data() + w.start;
    T.max_depth = std::max(T.max_depth, w.depth);
    measure(xyz, run, w.n, e);
    if (head == 0)
      for (int a = 0; a < 3; a++) { T.bbmin[a] = e.lo[a]; T.bbmax[a] = e.hi[a]; }

    if (w.n <= (uint32_t)bucket) { make_leaf(w); continue; }  // kdTreeImpl.h:114-123

    KdNode nd;
    nd.cx = 0.5 * (e.lo[0] + e.hi[0]);
    nd.cy = 0.5 * (e.lo[1] + e.hi[1]);
    nd.cz = 0.5 * (e.lo[2] + e.hi[2]);
    nd.hx = 0.5 * (e.hi[0] - e.lo[0]);
    nd.hy = 0.5 * (e.hi[1] - e.lo[1]);
    nd.hz = 0.5 * (e.hi[2] - e.lo[2]);
    // longest half extent, ties resolved as kdTreeImpl.h:138-150 does
    int axis;
    if (nd.hx > nd.hy) axis = (nd.hx > nd.hz) ? 0 : 2;
    else axis = (nd.hy > nd.hz) ? 1 : 2;
    // points measured very closely together stay in one bucket (kdTreeImpl.h:153-162)
    if (std::fabs(std::max(std::max(nd.hx, nd.hy), nd.hz)) < 0.01) { make_leaf(w); continue; }

    nd.splitval = e.mean[axis];
    nd.c1 = (axis & 1) ? REF_AXIS : 0u;
    nd.c2 = (axis & 2) ? REF_AXIS : 0u;
    // the inward scans of split_run stop only at a point on the other side of the split value: one
    // "< splitval" and one ">= splitval" point must exist (NaN / inf coordinates break exactly this)
    if (!(nd.splitval > e.lo[axis] && nd.splitval <= e.hi[axis]) || !std::isfinite(nd.hx + nd.hy + nd.hz)) {
      err = "degenerate split (non-finite coordinates?)";
      return false;
    }
    const uint32_t nleft = split_run(xyz, run, w.n, axis, nd.splitval);
    if (nleft == 0 || nleft == w.n) {
      // cannot happen for finite input with extent >= 0.01 (mean lies strictly inside);
      // the reference would recurse on an empty side and throw.
      err = "degenerate split (non-finite coordinates?)";
      return false;
    }
    const int32_t me = (int32_t)T.nodes.size();
    if ((uint32_t)me >= REF_VAL) { err = "too many tree nodes"; return false; }
    T.nodes.push_back(nd);
    T.node_r.push_back(std::sqrt(nd.hx * nd.hx + nd.hy * nd.hy + nd.hz * nd.hz));  // :135
    attach(w, (uint32_t)me);
    fifo.push_back({w.start, nleft, me, 0, w.depth + 1});
    fifo.push_back({w.start + nleft, w.n - nleft, me, 1, w.depth + 1});
  }
  T.n_internal = T.nodes.size();
  T.n_leaves = T.leaf_tab.size();

  // points in leaf order
  T.pts.resize(M);
  for (size_t k = 0; k < M; k++) {
    const double* p = xyz + 3 * (size_t)perm[k];
    T.pts[k] = {p[0], p[1], p[2], (int32_t)perm[k], 0};
  }

  // re-encode leaf references: packed (start << cb | count) when it fits in 30 bits
  T.cb = bits_for(T.max_leaf_points);
  T.table_mode = (bits_for(M) + T.cb) > 30;
  if (!T.table_mode) {
    auto pack = [&](uint32_t ref) -> uint32_t {
      if (!(ref & REF_LEAF)) return ref;
      const LeafEntry& le = T.leaf_tab[ref & REF_VAL];
      return (ref & (REF_LEAF | REF_AXIS)) | ((uint32_t)le.start << T.cb) | (uint32_t)le.count;
    };
    T.root_ref = pack(T.root_ref);
    for (KdNode& nd : T.nodes) { nd.c1 = pack(nd.c1); nd.c2 = pack(nd.c2); }
  }
  return true;
}

}  // namespace tdtk
