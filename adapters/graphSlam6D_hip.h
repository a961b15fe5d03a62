// graphSlam6D_hip -- reference-side binding for the graph-SLAM plug point
// (`virtual double doGraphSlam6D(Graph gr, vector<Scan*> scans, int nrIt) = 0`, include/slam6d/graphSlam6D.h:58).
// One class serves -G 1..4 (lum6DEuler, lum6DQuat, ghelix6DQ2, gapx6D): construct it wherever slam6D.cc
// constructs those (src/slam6d/slam6D.cc:784-804) when -t HipKD is selected, with the -G id as `backend` and -- for
// more than one process -- a communicator made with tdtk_comm_unique_id (rank 0) / tdtk_comm_create.
//
// Per iteration (adapters/graph_slam_glue.h): this rank's links in one batched call (every whole-scan
// correspondence pass of lum6Deuler.cc:265-303 / lum6Dquat.cc:248-276 / ghelix6DQ2.cc:376-409 / gapx6D.cc:404-441 on
// the GPU), one RCCL all-reduce of the per-link blocks over xGMI inside the library, then scatter + solve + pose
// update on every rank and the matrices handed back to the Scan objects.
//
// NOT compiled in this repository as a whole: graphSlam6D.h pulls in scan.h -> Boost and <cs.h> (SuiteSparse), which
// the build image lacks; the glue it instantiates IS (tests/test_host_logic.py::test_graph_slam_glue_compiles_and_links).
// It relies on the additions adapters/reference.patch makes to Scan (transformMatrixAndFrames, hipResident).
#ifndef __GRAPHSLAM6D_HIP_H__
#define __GRAPHSLAM6D_HIP_H__

#include "slam6d/graphSlam6D.h"
#include "slam6d/hip_search_tree.h"
#include "slam6d/graph_slam_glue.h"

// the reference's Scan seen through the member names the glue uses
struct HipScanView {
  Scan* s;
  const double* get_transMat() const { return s->get_transMat(); }
  const double* getDAlign() const { return s->getDAlign(); }
  const double* get_rPos() const { return s->get_rPos(); }
  const double* get_rPosTheta() const { return s->get_rPosTheta(); }
  size_t hipPoints() { return s->size<DataXYZ>("xyz reduced"); }
  tdtk_tree* hipTree() { return static_cast<HipSearchTree*>(s->getSearchTree())->handle(); }
  tdtk_scan* hipResident() { return s->hipResident(); }
  tdtk_scan* hipResidentOrNull() { return s->hipResidentOrNull(); }
  void transformMatrixAndFrames(const double* xf, int type, int islum) { s->transformMatrixAndFrames(xf, (Scan::AlgoType)type, islum); }
};

class graphSlam6D_hip : public graphSlam6D {
public:
  graphSlam6D_hip(int backend, icp6Dminimizer* m, double mdm, double mdml, int mni, bool quiet, bool meta, int rnd,
                  bool eP, int anim, double epsilonICP, int nns_method, double epsilonLUM, tdtk_comm* comm = 0)
    : graphSlam6D(m, mdm, mdml, mni, quiet, meta, rnd, eP, anim, epsilonICP, nns_method, epsilonLUM),
      backend(backend), comm(comm) {}

  virtual double doGraphSlam6D(Graph gr, std::vector<Scan*> allScans, int nrIt)
  {
    std::vector<HipScanView> views(allScans.size());
    std::vector<HipScanView*> ptrs(allScans.size());
    for (size_t i = 0; i < allScans.size(); i++) { views[i].s = allScans[i]; ptrs[i] = &views[i]; }
    return hip_graph_slam(backend, gr, ptrs, nrIt, epsilonLUM, max_dist_match2_LUM, comm, (int)Scan::INVALID, (int)Scan::LUM);
  }

private:
  int backend;
  tdtk_comm* comm;
};
#endif
