// The orchestration around the two plug points, in C++ and on the C ABI only: what bin/slam6D's own driver code does
// between the matches -- matchGraph6Dautomatic (src/slam6d/slam6D.cc:387-548: sequential ICP, loop detection by pose
// distance, loop closing, rounds of global relaxation), Graph(nodes, cldist2, loopsize) (src/slam6d/graph.cc:108-131),
// elch6Deuler::close_loop (src/slam6d/elch6Deuler.cc:44-138) and icp6D::match between two MetaScans (what close_loop
// asks for; Scan::getPtPairsParallel over the member scans, scan.cc:1305-1327) -- arranged the way this library wants
// it driven: the next scans made resident and their trees built on worker threads while the current one is matched
// (HipPrefetcher, icp_glue.h), all covariance passes of a loop closing in ONE batched device call, a MetaScan's
// members moved by one launch.  Like icp_glue.h and graph_slam_glue.h it needs nothing from the reference but the SHAPE
// of its Scan interface; adapters/harness/slam_glue_harness.cc instantiates it with a minimal scan type and is EXECUTED
// on the GPU box, where tests/test_gpu_parity.py::test_slam_glue_executes compares every pose it ends with, bit for
// bit, with the Python mirror (3dtk_amd/slam6d.py: matchGraph6Dautomatic, elch6Deuler, Graph) on the same scans.
//
// ScanT needs, beyond what icp_glue.h and graph_slam_glue.h ask for:
//   const double* get_rPos(), get_rPosTheta();  int hipBucket();
//   void transformToEuler(const double rP[3], const double rPT[3], int type, int islum)   (scan.cc:1061-1083; must move
//        a resident copy too -- adapters/reference.patch makes Scan::transformReduced do that)
// and one hook for the frame bookkeeping of a MetaScan's transform (scan.cc:962-975), which is the reference's own:
//   static void metaFrames(const std::vector<ScanT*>& members, int type)
#ifndef __SLAM6D_GLUE_H__
#define __SLAM6D_GLUE_H__

#include <algorithm>
#include <cmath>
#include <utility>

#include "graph_slam_glue.h"
#include "icp_glue.h"

struct HipScanTypes { int invalid, icp, lum, elch; };      // Scan::INVALID, Scan::ICP, Scan::LUM, Scan::ELCH as ints

// Graph(int nodes, double cldist2, int loopsize) (graph.cc:108-131): the chain, then every (j, k) with k - j > loopsize
// whose poses are closer than cldist2, row by row
struct HipClGraph {
  int nrScans = 0;
  std::vector<int> frm, to;
  int getNrScans() const { return nrScans; }
  int getNrLinks() const { return (int)frm.size(); }
  int getLink(int i, int fromTo) const { return fromTo == 0 ? frm[i] : to[i]; }
};
template <class ScanT>
HipClGraph hip_make_graph(int nodes, double cldist2, int loopsize, const std::vector<ScanT*>& allScans)
{
  HipClGraph g;
  g.nrScans = nodes;
  for (int i = 0; i + 1 < nodes; i++) { g.frm.push_back(i); g.to.push_back(i + 1); }
  for (int j = 0; j < nodes; j++)
    for (int k = j + 1; k < nodes; k++) {
      if (!(k - j > loopsize)) continue;
      const double* a = allScans[j]->get_rPos();
      const double* b = allScans[k]->get_rPos();
      const double dx = b[0] - a[0], dy = b[1] - a[1], dz = b[2] - a[2];
      if (dx * dx + dy * dy + dz * dz < cldist2) { g.frm.push_back(j); g.to.push_back(k); }
    }
  return g;
}

// Scan::transform of a MetaScan (scan.cc:920-926, metaScan.cc): every member moved -- the resident copies by one launch
// -- and its matrices updated without frames of its own; the frames `islum == 0` asks for are the hook's business
template <class ScanT>
void hip_meta_transform(const std::vector<ScanT*>& members, const double alignxf[16], int type, int islum, const HipScanTypes& ty)
{
  std::vector<tdtk_scan*> hs;
  std::vector<double> A;
  for (ScanT* m : members) {
    hs.push_back(m->hipResident());
    A.insert(A.end(), alignxf, alignxf + 16);
  }
  if (!hs.empty() && tdtk_scans_transform2((int)hs.size(), hs.data(), A.data(), nullptr) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  for (ScanT* m : members) m->transformMatrixAndFrames(alignxf, ty.invalid, -1);
  if (type != ty.invalid && islum == 0) ScanT::metaFrames(members, type);
}

// icp6D::match(MetaScan* model, MetaScan* data) (icp6D.cc:104-285 with Scan::getPtPairsParallel's MetaScan branch):
// one search tree over the model members' CURRENT points (KDtreeMetaManaged, kdMeta.cc:34-134: concatenation order,
// the first member's bucket size; a MetaScan's own dalignxf is the identity), per iteration one batched pass of every
// data member through it, the base blocks merged (Align_Parallel's merge, icp6Dquat.cc:533-588), one Align, every data
// member moved.  -a 1 / -a 2 only (what merges from the base block).  Returns the iteration count.
template <class ScanT>
int hip_meta_match(const std::vector<ScanT*>& model, const std::vector<ScanT*>& data, const HipIcpSettings& cfg,
                   const HipScanTypes& ty, unsigned int* nr_pointPair)
{
  if (cfg.algo != TDTK_ALGO_QUAT && cfg.algo != TDTK_ALGO_SVD)
    throw std::runtime_error("a MetaScan as data scan is matched with -a 1 or -a 2 only");
  const double id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  hip_meta_transform(data, id, ty.icp, 0, ty);                       // icp6D.cc:109
  if (cfg.max_num_iterations == 0) return 0;
  std::vector<tdtk_scan*> ms;
  for (ScanT* m : model) ms.push_back(m->hipResident());
  tdtk_tree* tree = nullptr;
  if (tdtk_tree_create_from_scans(ms.data(), (int)ms.size(), model[0]->hipBucket(), &tree) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  const int nl = (int)data.size();
  std::vector<const tdtk_tree*> first(nl, tree);
  std::vector<tdtk_scan*> second;
  std::vector<double> dal;
  for (ScanT* d : data) { second.push_back(d->hipResident()); dal.insert(dal.end(), id, id + 16); }
  double ret = 0.0, prev_ret = 0.0, prev_prev_ret = 0.0;
  int it = 0;
  try {
    for (it = 0; it < cfg.max_num_iterations; it++) {
      prev_prev_ret = prev_ret; prev_ret = ret;
      std::vector<tdtk_pair_sums> parts(nl);
      tdtk_pair_sums merged;
      if (tdtk_links_pair_sums(nl, first.data(), dal.data(), second.data(), cfg.max_dist_match2, 0, parts.data()) != TDTK_OK ||
          tdtk_pair_sums_merge(nl, parts.data(), &merged) != TDTK_OK)
        throw std::runtime_error(tdtk_last_error());
      if (nr_pointPair) *nr_pointPair = (unsigned int)merged.n;
      if (!(merged.n > 3)) break;                                       // icp6D.cc:235-245
      double alignxf[16];
      if (tdtk_align(cfg.algo, &merged, alignxf, &ret) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
      hip_meta_transform(data, alignxf, ty.icp, it == 0 ? 0 : -1, ty);  // icp6D.cc:246-252, anim = -1
      if ((std::fabs(ret - prev_ret) < cfg.epsilonICP && std::fabs(ret - prev_prev_ret) < cfg.epsilonICP) ||
          it == cfg.max_num_iterations - 1) {
        hip_meta_transform(data, id, ty.icp, 0, ty);                    // write end pose
        break;
      }
    }
  } catch (...) {
    tdtk_tree_destroy(tree);
    throw;
  }
  tdtk_tree_destroy(tree);
  return it;
}

// elch6Deuler::close_loop (-L 1, elch6Deuler.cc:44-138).  g: the loop-optimisation graph as (from, to) edges over scans
// 0 .. n-1 (slam6D.cc:416-428 adds (i-1, i) for every matched scan and (first, last) after every closed loop).  Every
// edge's covarianceEuler pass runs in ONE batched call; the six balancer runs and the pose distribution are host work.
// delta_out (nullable): the six pose differences the meta match found ("Delta:" line of the reference).
template <class ScanT>
void hip_elch_close_loop_euler(std::vector<ScanT*>& allScans, int first, int last, const std::vector<std::pair<int, int>>& g,
                               const HipIcpSettings& cfg, const HipScanTypes& ty, double* delta_out = nullptr)
{
  int n = 0;
  for (const auto& e : g) n = std::max(n, std::max(e.first, e.second) + 1);   // num_vertices(g)
  const int ne = (int)g.size();
  std::vector<const tdtk_tree*> firsts(ne);
  std::vector<tdtk_scan*> seconds(ne);
  std::vector<double> dal(16 * (size_t)ne), blocks(42 * (size_t)ne);
  std::vector<int32_t> from(ne), to(ne);
  for (int e = 0; e < ne; e++) {
    from[e] = g[e].first; to[e] = g[e].second;
    firsts[e] = allScans[g[e].first]->hipTree();
    seconds[e] = allScans[g[e].second]->hipResident();
    std::memcpy(&dal[16 * (size_t)e], allScans[g[e].first]->getDAlign(), 16 * sizeof(double));
  }
  if (tdtk_graph_link_blocks(TDTK_GRAPH_LUMEULER, ne, firsts.data(), dal.data(), seconds.data(), cfg.max_dist_match2,
                             blocks.data()) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  // edge weights: |diag(C^-1)| per pose component (elch6Deuler.cc:60-65)
  std::vector<std::vector<double>> wts(6, std::vector<double>(ne));
  for (int e = 0; e < ne; e++) {
    double Cinv[36];
    if (tdtk_invert(&blocks[42 * (size_t)e], 6, Cinv) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
    for (int j = 0; j < 6; j++) wts[j][e] = std::fabs(Cinv[7 * j]);
  }
  std::vector<std::vector<double>> weights(6, std::vector<double>(n, 0.0));
  for (int j = 0; j < 6; j++)
    if (tdtk_elch_graph_balancer(n, ne, from.data(), to.data(), wts[j].data(), first, last, weights[j].data()) != TDTK_OK)
      throw std::runtime_error(tdtk_last_error());
  for (int i = last - 2; i <= last; i++)
    for (int j = 0; j < 6; j++) weights[j][i] = 0.0;
  const std::vector<ScanT*> start = {allScans[first], allScans[first + 1], allScans[first + 2]};
  const std::vector<ScanT*> end = {allScans[last - 2], allScans[last - 1], allScans[last]};
  double delta[6];
  for (int k = 0; k < 3; k++) { delta[k] = allScans[last]->get_rPos()[k]; delta[3 + k] = allScans[last]->get_rPosTheta()[k]; }
  unsigned int pairs = 0;
  (void)hip_meta_match(start, end, cfg, ty, &pairs);
  for (int k = 0; k < 3; k++) {
    delta[k] = allScans[last]->get_rPos()[k] - delta[k];
    delta[3 + k] = allScans[last]->get_rPosTheta()[k] - delta[3 + k];
  }
  if (delta_out) std::memcpy(delta_out, delta, sizeof delta);
  for (int i = 1; i < n; i++) {
    double rP[3], rT[3];
    for (int k = 0; k < 3; k++) {
      rP[k] = allScans[i]->get_rPos()[k] + delta[k] * (weights[k][i] - weights[k][0]);
      rT[k] = allScans[i]->get_rPosTheta()[k] + delta[3 + k] * (weights[3 + k][i] - weights[3 + k][0]);
    }
    allScans[i]->transformToEuler(rP, rT, ty.elch, i == n - 1 ? 2 : 1);
  }
}

// ---- -L 2 .. 4: the loop closers with quaternion poses (round 4) ----------------------------------------------------------------
// ScanT needs in addition: void get_rPosQuat(double q[4]) (rQuat: Matrix4ToQuat of transMat, scan.cc:886),
//   void transformToQuat(const double rP[3], const double rPQ[4], int type, int islum) (scan.cc:1093-1104),
//   void transform(const double alignxf[16], int type, int islum) (scan.cc:918-1009; all three move a resident copy too).
static inline void hip_qmult(const double* q1, const double* q2, double* q3)      // QMult, globals.icc:1112-1117
{
  const double r[4] = {q1[0] * q2[0] - q1[1] * q2[1] - q1[2] * q2[2] - q1[3] * q2[3],
                       q1[0] * q2[1] + q1[1] * q2[0] + q1[2] * q2[3] - q1[3] * q2[2],
                       q1[0] * q2[2] - q1[1] * q2[3] + q1[2] * q2[0] + q1[3] * q2[1],
                       q1[0] * q2[3] + q1[1] * q2[2] - q1[2] * q2[1] + q1[3] * q2[0]};
  std::memcpy(q3, r, sizeof r);
}
static inline void hip_normalize4(double* q)                                     // Normalize4, globals.icc:267-275
{
  const double norm = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; k++) q[k] = q[k] / norm;
}
static inline void hip_slerp(const double* qa, const double* qb, double t, double* qm)   // slerp, globals.icc:1123-1166
{
  const double c = qa[0] * qb[0] + qa[1] * qb[1] + qa[2] * qb[2] + qa[3] * qb[3];
  if (std::fabs(c) >= 1.0) { std::memcpy(qm, qa, 4 * sizeof(double)); return; }
  const double half = std::acos(c), sn = std::sqrt(1.0 - c * c);
  if (std::fabs(sn) < 0.001) {
    for (int k = 0; k < 4; k++) qm[k] = qa[k] * 0.5 + qb[k] * 0.5;
    hip_normalize4(qm);
    return;
  }
  const double ra = std::sin((1 - t) * half) / sn, rb = std::sin(t * half) / sn;
  for (int k = 0; k < 4; k++) qm[k] = qa[k] * ra + qb[k] * rb;
  hip_normalize4(qm);
}

// What the three share (elch6Dquat.cc:52-72, elch6DunitQuat.cc:52-75, elch6Dslerp.cc:50-90): one
// lum6DQuat::covarianceQuat pass per edge of the loop graph -- all of them in ONE batched device call --, C = C.i(), edge
// weights from |diag C| (seven, or three translations + the sum of the four quaternion entries), one balancer run per
// weight set.  weights[j][vertex].
template <class ScanT>
int hip_elch_quat_weights(std::vector<ScanT*>& allScans, int first, int last, const std::vector<std::pair<int, int>>& g,
                          double max_dist_match2, bool combine, std::vector<std::vector<double>>& weights)
{
  int n = 0;
  for (const auto& e : g) n = std::max(n, std::max(e.first, e.second) + 1);
  const int ne = (int)g.size(), nw = combine ? 4 : 7;
  std::vector<const tdtk_tree*> firsts(ne);
  std::vector<tdtk_scan*> seconds(ne);
  std::vector<double> dal(16 * (size_t)ne), blocks(56 * (size_t)ne);
  std::vector<int32_t> from(ne), to(ne);
  for (int e = 0; e < ne; e++) {
    from[e] = g[e].first; to[e] = g[e].second;
    firsts[e] = allScans[g[e].first]->hipTree();
    seconds[e] = allScans[g[e].second]->hipResident();
    std::memcpy(&dal[16 * (size_t)e], allScans[g[e].first]->getDAlign(), 16 * sizeof(double));
  }
  if (tdtk_graph_link_blocks(TDTK_GRAPH_LUMQUAT, ne, firsts.data(), dal.data(), seconds.data(), max_dist_match2, blocks.data()) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  std::vector<std::vector<double>> wts(nw, std::vector<double>(ne));
  for (int e = 0; e < ne; e++) {
    double Cinv[49];
    if (tdtk_invert(&blocks[56 * (size_t)e], 7, Cinv) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
    if (combine) {
      for (int j = 0; j < 3; j++) wts[j][e] = std::fabs(Cinv[8 * j]);
      wts[3][e] = std::fabs(Cinv[8 * 3]) + std::fabs(Cinv[8 * 4]) + std::fabs(Cinv[8 * 5]) + std::fabs(Cinv[8 * 6]);
    } else {
      for (int j = 0; j < 7; j++) wts[j][e] = std::fabs(Cinv[8 * j]);
    }
  }
  weights.assign(nw, std::vector<double>(n, 0.0));
  for (int j = 0; j < nw; j++)
    if (tdtk_elch_graph_balancer(n, ne, from.data(), to.data(), wts[j].data(), first, last, weights[j].data()) != TDTK_OK)
      throw std::runtime_error(tdtk_last_error());
  return n;
}

// elch6Dquat::close_loop (-L 2, elch6Dquat.cc:44-148): every one of the seven pose components distributed linearly, the
// quaternion renormalised.  delta_out (nullable): position and quaternion difference the MetaScan match found.
template <class ScanT>
void hip_elch_close_loop_quat(std::vector<ScanT*>& allScans, int first, int last, const std::vector<std::pair<int, int>>& g,
                              const HipIcpSettings& cfg, const HipScanTypes& ty, double* delta_out = nullptr)
{
  std::vector<std::vector<double>> weights;
  const int n = hip_elch_quat_weights(allScans, first, last, g, cfg.max_dist_match2, false, weights);
  const std::vector<ScanT*> start = {allScans[first], allScans[first + 1], allScans[first + 2]};
  const std::vector<ScanT*> end = {allScans[last - 2], allScans[last - 1], allScans[last]};
  for (int i = last - 2; i <= last; i++)
    for (int j = 0; j < 7; j++) weights[j][i] = 0.0;
  double delta[7], q[4];
  allScans[last]->get_rPosQuat(q);
  for (int k = 0; k < 3; k++) delta[k] = allScans[last]->get_rPos()[k];
  for (int k = 0; k < 4; k++) delta[3 + k] = q[k];
  unsigned int pairs = 0;
  (void)hip_meta_match(start, end, cfg, ty, &pairs);
  allScans[last]->get_rPosQuat(q);
  for (int k = 0; k < 3; k++) delta[k] = allScans[last]->get_rPos()[k] - delta[k];
  for (int k = 0; k < 4; k++) delta[3 + k] = q[k] - delta[3 + k];
  if (delta_out) std::memcpy(delta_out, delta, sizeof delta);
  for (int i = 1; i < n; i++) {
    double rP[3], rQ[4], qi[4];
    allScans[i]->get_rPosQuat(qi);
    for (int k = 0; k < 3; k++) rP[k] = allScans[i]->get_rPos()[k] + delta[k] * (weights[k][i] - weights[k][0]);
    for (int k = 0; k < 4; k++) rQ[k] = qi[k] + delta[3 + k] * (weights[3 + k][i] - weights[3 + k][0]);
    hip_normalize4(rQ);
    allScans[i]->transformToQuat(rP, rQ, ty.elch, i == n - 1 ? 2 : 1);
  }
}

// elch6DunitQuat::close_loop (-L 3, elch6DunitQuat.cc:45-199): the rotation of the loop error as ONE unit quaternion,
// blended in per scan by its weight; the three matched scans are put back first, the chain is counter-rotated so that
// scan 0 stays.  delta_out (nullable): delta[3] then deltaQ[4].
template <class ScanT>
void hip_elch_close_loop_unitquat(std::vector<ScanT*>& allScans, int first, int last, const std::vector<std::pair<int, int>>& g,
                                  const HipIcpSettings& cfg, const HipScanTypes& ty, double* delta_out = nullptr)
{
  std::vector<std::vector<double>> weights;
  const int n = hip_elch_quat_weights(allScans, first, last, g, cfg.max_dist_match2, true, weights);
  const std::vector<ScanT*> start = {allScans[first], allScans[first + 1], allScans[first + 2]};
  const std::vector<ScanT*> end = {allScans[last - 2], allScans[last - 1], allScans[last]};
  double pOld[3][7];
  for (int k = 0; k < 3; k++) {                  // last, last - 1, last - 2
    ScanT* s = allScans[last - k];
    for (int c = 0; c < 3; c++) pOld[k][c] = s->get_rPos()[c];
    s->get_rPosQuat(&pOld[k][3]);
  }
  double delta[3], q1[4], q2[4], deltaQ[4];
  for (int k = 0; k < 3; k++) delta[k] = allScans[last]->get_rPos()[k];
  allScans[last]->get_rPosQuat(q1);
  q1[1] = -q1[1]; q1[2] = -q1[2]; q1[3] = -q1[3];
  unsigned int pairs = 0;
  (void)hip_meta_match(start, end, cfg, ty, &pairs);
  for (int k = 0; k < 3; k++) delta[k] = allScans[last]->get_rPos()[k] - delta[k];
  allScans[last]->get_rPosQuat(q2);
  hip_qmult(q2, q1, deltaQ);
  if (delta_out) { std::memcpy(delta_out, delta, sizeof delta); std::memcpy(delta_out + 3, deltaQ, sizeof deltaQ); }
  for (int k = 0; k < 3; k++) allScans[last - k]->transformToQuat(pOld[k], &pOld[k][3], ty.invalid, -1);
  double q0[4], pd[4], s0[4], counter[4];
  allScans[0]->get_rPosQuat(q0);
  const double w0 = weights[3][0];
  hip_qmult(deltaQ, q0, pd);
  s0[0] = (1 - w0) * q0[0] + pd[0] * w0;
  for (int k = 1; k < 4; k++) s0[k] = ((1 - w0) * q0[k] + pd[k] * w0) * -1.0;
  hip_normalize4(s0);
  hip_qmult(q0, s0, counter);
  for (int i = 1; i < n; i++) {
    double rP[3], qi[4], rot[4], tmp[4], rQ[4];
    for (int k = 0; k < 3; k++) rP[k] = allScans[i]->get_rPos()[k] + delta[k] * (weights[k][i] - weights[k][0]);
    allScans[i]->get_rPosQuat(qi);
    const double wi = weights[3][i];
    hip_qmult(deltaQ, qi, rot);
    for (int k = 0; k < 4; k++) tmp[k] = (1 - wi) * qi[k] + rot[k] * wi;
    hip_normalize4(tmp);
    hip_qmult(counter, tmp, rQ);
    hip_normalize4(rQ);
    allScans[i]->transformToQuat(rP, rQ, ty.elch, i == n - 1 ? 2 : 1);
  }
}

// elch6Dslerp::close_loop (-L 4, elch6Dslerp.cc:44-184): the loop error as a rigid motion in the frame of scan `first`,
// rotation interpolated on the sphere, translation scaled per axis; MetaScans first-2 .. first+2 and last-2 .. last.
// delta_out (nullable): deltaT[3] then deltaQ[4].
template <class ScanT>
void hip_elch_close_loop_slerp(std::vector<ScanT*>& allScans, int first, int last, const std::vector<std::pair<int, int>>& g,
                               const HipIcpSettings& cfg, const HipScanTypes& ty, double* delta_out = nullptr)
{
  std::vector<std::vector<double>> weights;
  const int n = hip_elch_quat_weights(allScans, first, last, g, cfg.max_dist_match2, true, weights);
  std::vector<ScanT*> start, end;
  for (int i = first - 2; i <= first + 2; i++) if (i >= 0) start.push_back(allScans[i]);
  for (int i = last - 2; i <= last && i < n; i++) end.push_back(allScans[i]);
  double Pl0[16], Pp0[16], Pf0[16], Pf0_inv[16], t1[16], t2[16], deltaf[16];
  std::memcpy(Pl0, allScans[last]->get_transMat(), sizeof Pl0);
  unsigned int pairs = 0;
  (void)hip_meta_match(start, end, cfg, ty, &pairs);
  std::memcpy(Pp0, allScans[last]->get_transMat(), sizeof Pp0);
  std::memcpy(Pf0, allScans[first]->get_transMat(), sizeof Pf0);
  tdtk_host_m4inv(Pf0, Pf0_inv);
  tdtk_host_mmult(Pf0_inv, Pl0, t1);
  tdtk_host_m4inv(t1, t2);
  tdtk_host_mmult(Pp0, t2, t1);
  tdtk_host_mmult(Pf0_inv, t1, deltaf);
  double deltaT[3], deltaQ[4];
  tdtk_host_matrix4_to_quat(deltaf, deltaQ, deltaT);
  if (delta_out) { std::memcpy(delta_out, deltaT, sizeof deltaT); std::memcpy(delta_out + 3, deltaQ, sizeof deltaQ); }
  const double idQ[4] = {1, 0, 0, 0};
  auto share = [&](int i, double* M) {       // the share of scan i of the loop error
    double rP[3], rQ[4];
    for (int k = 0; k < 3; k++) rP[k] = deltaT[k] * weights[k][i];
    hip_slerp(idQ, deltaQ, weights[3][i], rQ);
    tdtk_host_quat_to_matrix4(rQ, rP, M);
  };
  double delta0[16];
  share(0, t1);
  tdtk_host_m4inv(t1, t2);
  tdtk_host_mmult(Pf0, t2, delta0);
  for (int i = 1; i < n; i++) {
    double M[16];
    if (i >= last - 2 && i <= last) {
      tdtk_host_mmult(delta0, Pf0_inv, M);
    } else {
      share(i, t1);
      tdtk_host_mmult(delta0, t1, t2);
      tdtk_host_mmult(t2, Pf0_inv, M);
    }
    allScans[i]->transform(M, ty.elch, i == n - 1 ? 2 : 1);
  }
}

struct HipSlamSettings {
  HipIcpSettings icp;        // the sequential matches (and, with its own max_dist_match2 / iterations, the ELCH meta match)
  HipIcpSettings loop_icp;   // icp6D of the loop closer (loopSlam6D's own my_icp6D)
  bool use_elch;             // -L 1 .. 4 ...
  int elch_variant;          // ... which: 0 / 1 elch6Deuler, 2 elch6Dquat, 3 elch6DunitQuat, 4 elch6Dslerp
  int graph_backend;         // TDTK_GRAPH_* of -G, or -1: no global relaxation
  double cldist, mdml, epsilonSLAM, epsilonLUM;
  int loopsize, nrIt, prefetch;
  tdtk_comm* comm;           // nullable
  bool meta_icp;             // slam6D.cc:436-448: every scan is matched against a MetaScan of the scans before it ...
  int max_num_metascans;     // ... the last n of them (<= 0: all)
  double mdmll, graphDist;   // the closing pass of slam6D.cc:535-547 (-DlastSLAM / --graphDist): mdmll > 0 runs it, on
                             // Graph(n, graphDist^2, loopsize) with max_dist_match2_LUM = mdmll^2
};

// matchGraph6Dautomatic (slam6D.cc:387-548, meta_icp and the -DlastSLAM pass included): returns the number of global
// rounds it ran.  A scan's preparation (upload, ordering, tree build) runs up to `prefetch` scans ahead of the
// match on worker threads; scan i + 1 is not part of the graph over scans 0 .. i, so the relaxation never touches a
// scan that is being prepared, and every scan is made resident BEFORE its pose extrapolation, prepared ahead or not
// (see hip_do_icp) -- the result does not depend on `prefetch`.
template <class ScanT>
int hip_match_graph6d_automatic(std::vector<ScanT*>& allScans, const HipSlamSettings& cfg, const HipScanTypes& ty)
{
  const double cldist2 = cfg.cldist * cfg.cldist;
  const int n = (int)allScans.size();
  int loop_detection = 0, rounds = 0, first = 0, last = 0;
  double min_dist = -1.0;
  std::vector<std::pair<int, int>> g;
  std::vector<ScanT*> metas;          // slam6D.cc:407
  auto global_rounds = [&](int nodes, double graph_dist2, double max_dist_match2_LUM) {
    int j = 0;
    double ret;
    std::vector<ScanT*> sub(allScans.begin(), allScans.begin() + nodes);
    do {
      HipClGraph gr = hip_make_graph(nodes, graph_dist2, cfg.loopsize, allScans);
      ret = hip_graph_slam(cfg.graph_backend, gr, sub, 1, cfg.epsilonLUM, max_dist_match2_LUM, cfg.comm, ty.invalid, ty.lum);
      j++; rounds++;
    } while (j < cfg.nrIt && ret > cfg.epsilonSLAM);
  };
  auto close_loop = [&]() {
    switch (cfg.elch_variant) {
      case 2: hip_elch_close_loop_quat(allScans, first, last, g, cfg.loop_icp, ty); break;
      case 3: hip_elch_close_loop_unitquat(allScans, first, last, g, cfg.loop_icp, ty); break;
      case 4: hip_elch_close_loop_slerp(allScans, first, last, g, cfg.loop_icp, ty); break;
      default: hip_elch_close_loop_euler(allScans, first, last, g, cfg.loop_icp, ty); break;
    }
  };
  HipPrefetcher* pool = (cfg.prefetch > 0 && n > 2) ? new HipPrefetcher(cfg.prefetch) : nullptr;
  try {
    for (int i = 0; i < n; i++) {
      if (pool) {
        for (int j = i; j < n && j <= i + cfg.prefetch; j++) {
          ScanT* s = allScans[j];      // (the global rounds and the loop closer use every scan's own tree: built ahead also with meta_icp)
          pool->submit((size_t)j, [s] { (void)s->hipResident(); (void)s->hipTree(); });
        }
        if (i > 0) pool->wait((size_t)i - 1);
        pool->wait((size_t)i);
      }
      if (i == 0) continue;
      g.push_back({i - 1, i});
      (void)allScans[i]->hipResident();
      if (cfg.icp.eP) allScans[i]->mergeCoordinatesWithRoboterPosition(allScans[i - 1]);
      unsigned int pairs = 0;
      if (cfg.meta_icp) {              // slam6D.cc:436-448
        metas.push_back(allScans[i - 1]);
        if (cfg.max_num_metascans > 0)
          while (metas.size() > (size_t)cfg.max_num_metascans) metas.erase(metas.begin());
        (void)hip_icp_match_metascan(metas, allScans[i], cfg.icp, &pairs);
      } else {
        (void)hip_icp_match(allScans[i - 1], allScans[i], cfg.icp, &pairs);
      }
      if (loop_detection == 1) loop_detection = 2;
      for (int j = 0; j < i - cfg.loopsize; j++) {
        const double* a = allScans[j]->get_rPos();
        const double* b = allScans[i]->get_rPos();
        const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        const double dist = dx * dx + dy * dy + dz * dz;
        if (dist < cldist2) {
          loop_detection = 1;
          if (min_dist < 0 || dist < min_dist) { min_dist = dist; first = j; last = i; }
        }
      }
      if (loop_detection == 2) {
        loop_detection = 0;
        min_dist = -1.0;
        if (cfg.use_elch) {
          close_loop();
          g.push_back({first, last});
        }
        if (cfg.graph_backend >= 0 && cfg.mdml > 0) global_rounds(i + 1, cldist2, cfg.mdml * cfg.mdml);
      }
    }
  } catch (...) {
    delete pool;
    throw;
  }
  delete pool;
  if (loop_detection == 1 && cfg.use_elch) {
    close_loop();
    g.push_back({first, last});
  }
  if (cfg.graph_backend >= 0 && cfg.mdml > 0.0) global_rounds(n, cldist2, cfg.mdml * cfg.mdml);
  // slam6D.cc:535-547: set_mdmll(mdmll), then rounds on the graph of --graphDist
  if (cfg.graph_backend >= 0 && cfg.mdmll > 0.0) global_rounds(n, cfg.graphDist * cfg.graphDist, cfg.mdmll * cfg.mdmll);
  return rounds;
}

#endif
