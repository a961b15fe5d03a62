// graphSlam6D_hip -- reference-side binding for the graph-SLAM plug point
// (`virtual double doGraphSlam6D(Graph gr, vector<Scan*> scans, int nrIt) = 0`, include/slam6d/graphSlam6D.h:58).
// One class serves -G 1..4 (lum6DEuler, lum6DQuat, ghelix6DQ2, gapx6D): construct it wherever slam6D.cc
// constructs those (src/slam6d/slam6D.cc:784-804) when -t HipKD is selected, with the -G id as `backend`.
//
// Per iteration: this rank's links in one batched call (tdtk_graph_link_blocks: every whole-scan correspondence
// pass of lum6Deuler.cc:265-303 / lum6Dquat.cc:248-276 / ghelix6DQ2.cc:376-409 / gapx6D.cc:404-441 on the GPU),
// one RCCL all-reduce of the per-link blocks over xGMI (one process per GPU), then scatter + solve + pose update
// on every rank (tdtk_graph_solve_update) and the matrices handed back to the Scan objects.
//
// NOT compiled in this repository: graphSlam6D.h pulls in scan.h -> Boost and <cs.h> (SuiteSparse), which the
// build image lacks.  It relies on two small additions to the reference listed at the bottom.
#ifndef __GRAPHSLAM6D_HIP_H__
#define __GRAPHSLAM6D_HIP_H__

#include <cfloat>
#include <stdexcept>
#include <vector>

#include <rccl/rccl.h>

#include "slam6d/graphSlam6D.h"
#include "slam6d/hip_search_tree.h"
#include "tdtk_hip.h"

class graphSlam6D_hip : public graphSlam6D {
public:
  // rank / world / comm / stream: one process per GPU; world == 1 needs no communicator
  graphSlam6D_hip(int backend, icp6Dminimizer* m, double mdm, double mdml, int mni, bool quiet, bool meta, int rnd,
                  bool eP, int anim, double epsilonICP, int nns_method, double epsilonLUM, int rank = 0, int world = 1,
                  ncclComm_t comm = 0, hipStream_t stream = 0)
    : graphSlam6D(m, mdm, mdml, mni, quiet, meta, rnd, eP, anim, epsilonICP, nns_method, epsilonLUM),
      backend(backend), rank(rank), world(world), comm(comm), stream(stream) {}

  virtual double doGraphSlam6D(Graph gr, std::vector<Scan*> allScans, int nrIt)
  {
    const int nscans = gr.getNrScans(), nlinks = gr.getNrLinks(), n = nscans - 1;
    const int Bn = tdtk_graph_block_doubles(backend);
    // what ghelix6DQ2 / gapx6D carry over the iterations of one call (ghelix6DQ2.cc:329-330, gapx6D.cc:356)
    std::vector<double> state(backend == TDTK_GRAPH_GHELIX ? (size_t)36 * n * n + 6 * n
                              : backend == TDTK_GRAPH_GAPX ? (size_t)3 * n : 0, 0.0);
    std::vector<int32_t> from(nlinks), to(nlinks);
    for (int i = 0; i < nlinks; i++) { from[i] = gr.getLink(i, 0); to[i] = gr.getLink(i, 1); }
    double ret = DBL_MAX;
    for (int it = 0; it < nrIt && ret > epsilonLUM; it++) {
      // links dealt to ranks: chain links round-robin, closures by (from + to) % world (a link keeps its owner
      // when the graph grows between rounds)
      std::vector<int> mine;
      for (int i = 0; i < nlinks; i++)
        if (((to[i] == from[i] + 1) ? from[i] : from[i] + to[i]) % world == rank) mine.push_back(i);
      std::vector<const tdtk_tree*> first(mine.size());
      std::vector<tdtk_scan*> second(mine.size());
      std::vector<double> dal(16 * mine.size()), mb((size_t)Bn * mine.size());
      for (size_t k = 0; k < mine.size(); k++) {
        Scan* a = allScans[from[mine[k]]];
        Scan* b = allScans[to[mine[k]]];
        first[k] = static_cast<HipSearchTree*>(a->getSearchTree())->handle();
        second[k] = b->hipResident();                               // addition (2) below
        memcpy(&dal[16 * k], a->getDAlign(), 16 * sizeof(double));
      }
      if (tdtk_graph_link_blocks(backend, (int)mine.size(), first.data(), dal.data(), second.data(),
                                 max_dist_match2_LUM, mb.data()) != TDTK_OK)
        throw std::runtime_error(tdtk_last_error());
      // every link has one owner, the others contribute zeros: the sum is exact, X does not depend on `world`
      std::vector<double> blocks((size_t)Bn * nlinks, 0.0);
      for (size_t k = 0; k < mine.size(); k++) memcpy(&blocks[(size_t)Bn * mine[k]], &mb[(size_t)Bn * k], Bn * sizeof(double));
      if (world > 1) {
        double* d_blocks = hipScratch(blocks.size());               // device staging owned by this object
        hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(double), hipMemcpyHostToDevice, stream);
        ncclAllReduce(d_blocks, d_blocks, blocks.size(), ncclDouble, ncclSum, comm, stream);
        hipMemcpyAsync(blocks.data(), d_blocks, blocks.size() * sizeof(double), hipMemcpyDeviceToHost, stream);
        hipStreamSynchronize(stream);
      }
      // poses in, poses out
      std::vector<double> tm(16 * nscans), da(16 * nscans), rp(3 * nscans), rt(3 * nscans), xf(32 * nscans);
      std::vector<tdtk_scan*> res(nscans);
      for (int i = 0; i < nscans; i++) {
        memcpy(&tm[16 * i], allScans[i]->get_transMat(), 16 * sizeof(double));
        memcpy(&da[16 * i], allScans[i]->getDAlign(), 16 * sizeof(double));
        memcpy(&rp[3 * i], allScans[i]->get_rPos(), 3 * sizeof(double));
        memcpy(&rt[3 * i], allScans[i]->get_rPosTheta(), 3 * sizeof(double));
        res[i] = allScans[i]->hipResidentOrNull();
      }
      if (tdtk_graph_solve_update(backend, nlinks, from.data(), to.data(), blocks.data(), nscans, tm.data(), da.data(),
                                  rp.data(), rt.data(), res.data(), state.empty() ? 0 : state.data(), xf.data(),
                                  &ret) != TDTK_OK)
        throw std::runtime_error(tdtk_last_error());
      // the resident points have moved on the GPU; replay the one or two transforms of every scan on its
      // matrices and frames only (addition (1) of adapters/icp6D_hip.h), last scan with islum == 2
      const bool two = backend == TDTK_GRAPH_LUMEULER || backend == TDTK_GRAPH_LUMQUAT;
      for (int i = 1; i < nscans; i++) {
        if (two) allScans[i]->transformMatrixAndFrames(&xf[32 * i], Scan::INVALID, -1);
        allScans[i]->transformMatrixAndFrames(&xf[32 * i + (two ? 16 : 0)], Scan::LUM, i == nscans - 1 ? 2 : 1);
      }
    }
    return ret;
  }

private:
  int backend, rank, world;
  ncclComm_t comm;
  hipStream_t stream;
  std::vector<double*> scratch;
  double* hipScratch(size_t n)
  {
    double* p = 0;
    hipMalloc((void**)&p, n * sizeof(double));
    scratch.push_back(p);
    return p;
  }
};

// Additions this adapter needs in the reference (besides those listed in adapters/icp6D_hip.h):
//   (2) Scan: `tdtk_scan* hipResident()` -- the scan's "xyz reduced" uploaded once with tdtk_scan_create (+
//       tdtk_scan_mark_original) and kept as a member next to the search tree, `hipResidentOrNull()` returning it
//       only if it exists; Scan::get("xyz reduced") downloads with tdtk_scan_download when the host copy is stale.
#endif
