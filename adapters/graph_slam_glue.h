// The part of the graph-SLAM binding that needs nothing from the reference but the SHAPE of its Scan / Graph
// interfaces: one global iteration = deal the links (tdtk_graph_deal_links), run this rank's links, exchange and
// solve inside the library (tdtk_graph_iteration: RCCL all-reduce of the per-link blocks over the library's own
// communicator), hand the poses back and replay the frame bookkeeping.  adapters/graphSlam6D_hip.h instantiates it
// with the reference's Scan and Graph; tests/test_host_logic.py compiles and links it with two minimal types that have
// the same member functions (no Boost needed for that).
#ifndef __GRAPH_SLAM_GLUE_H__
#define __GRAPH_SLAM_GLUE_H__

#include <cfloat>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "tdtk_hip.h"

// ScanT needs: get_transMat(), getDAlign(), get_rPos(), get_rPosTheta() -> const double*; size_t hipPoints();
//              tdtk_tree* hipTree(); tdtk_scan* hipResident(); tdtk_scan* hipResidentOrNull();
//              void transformMatrixAndFrames(const double*, int type, int islum)   (Scan::INVALID = 0, Scan::LUM = 3)
// GraphT needs: getNrScans(), getNrLinks(), getLink(i, 0 / 1)
template <class ScanT, class GraphT>
double hip_graph_slam(int backend, GraphT& gr, std::vector<ScanT*>& allScans, int nrIt, double epsilonLUM,
                      double max_dist_match2_LUM, tdtk_comm* comm, int type_invalid, int type_lum)
{
  const int nscans = gr.getNrScans(), nlinks = gr.getNrLinks(), n = nscans - 1;
  int rank = 0, world = 1;
  if (comm) tdtk_comm_info(comm, &rank, &world, 0);
  // what ghelix6DQ2 / gapx6D carry over the iterations of one call (ghelix6DQ2.cc:329-330, gapx6D.cc:356)
  std::vector<double> state(backend == TDTK_GRAPH_GHELIX ? (size_t)36 * n * n + 6 * n
                            : backend == TDTK_GRAPH_GAPX ? (size_t)3 * n : 0, 0.0);
  std::vector<int32_t> from(nlinks), to(nlinks), owner(nlinks);
  for (int i = 0; i < nlinks; i++) { from[i] = gr.getLink(i, 0); to[i] = gr.getLink(i, 1); }
  std::vector<uint64_t> npts(nscans);
  for (int i = 0; i < nscans; i++) npts[i] = allScans[i]->hipPoints();
  if (tdtk_graph_deal_links(nlinks, from.data(), to.data(), npts.data(), nscans, world, owner.data()) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  std::vector<int32_t> mine;
  for (int i = 0; i < nlinks; i++)
    if (owner[i] == rank) mine.push_back(i);
  double ret = DBL_MAX;
  for (int it = 0; it < nrIt && ret > epsilonLUM; it++) {
    std::vector<const tdtk_tree*> first(mine.size() + 1);
    std::vector<tdtk_scan*> second(mine.size() + 1);
    std::vector<double> dal(16 * (mine.size() + 1));
    for (size_t k = 0; k < mine.size(); k++) {
      ScanT* a = allScans[from[mine[k]]];
      ScanT* b = allScans[to[mine[k]]];
      first[k] = a->hipTree();
      second[k] = b->hipResident();
      std::memcpy(&dal[16 * k], a->getDAlign(), 16 * sizeof(double));
    }
    std::vector<double> tm(16 * nscans), da(16 * nscans), rp(3 * nscans), rt(3 * nscans), xf(32 * nscans);
    std::vector<tdtk_scan*> res(nscans);
    for (int i = 0; i < nscans; i++) {
      std::memcpy(&tm[16 * i], allScans[i]->get_transMat(), 16 * sizeof(double));
      std::memcpy(&da[16 * i], allScans[i]->getDAlign(), 16 * sizeof(double));
      std::memcpy(&rp[3 * i], allScans[i]->get_rPos(), 3 * sizeof(double));
      std::memcpy(&rt[3 * i], allScans[i]->get_rPosTheta(), 3 * sizeof(double));
      res[i] = allScans[i]->hipResidentOrNull();
    }
    if (tdtk_graph_iteration(backend, comm, nlinks, from.data(), to.data(), (int)mine.size(), mine.data(), first.data(),
                             dal.data(), second.data(), max_dist_match2_LUM, nscans, tm.data(), da.data(), rp.data(),
                             rt.data(), res.data(), state.empty() ? 0 : state.data(), xf.data(), &ret) != TDTK_OK)
      throw std::runtime_error(tdtk_last_error());
    // the resident points have moved on the GPU; replay the one or two transforms of every scan on its matrices and
    // frames only, the last scan with islum == 2 like the reference's loops (lum6Deuler.cc:451-455)
    const bool two = backend == TDTK_GRAPH_LUMEULER || backend == TDTK_GRAPH_LUMQUAT;
    for (int i = 1; i < nscans; i++) {
      if (two) allScans[i]->transformMatrixAndFrames(&xf[32 * i], type_invalid, -1);
      allScans[i]->transformMatrixAndFrames(&xf[32 * i + (two ? 16 : 0)], type_lum, i == nscans - 1 ? 2 : 1);
    }
  }
  return ret;
}

#endif
