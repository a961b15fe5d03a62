// icp6D_hip -- reference-side binding for the icp6D plug points (include/slam6d/icp6D.h:51,
// `virtual int match(Scan*, Scan*, PairingMode)`, and `void doICP(vector<Scan*>, PairingMode)`).  Construct it wherever
// slam6D.cc constructs icp6D (src/slam6d/slam6D.cc:737,770,810,825) when -t HipKD is selected.  The whole loop of
// src/slam6d/icp6D.cc:104-285 then runs device-resident: the data scan is uploaded once (Scan::hipResident), one tiny
// D2H per iteration, the moved points are downloaded when the match is over.  doICP keeps every scan resident and
// prepares the next ones (upload, ordering, tree build) on worker threads while the current pair is matched.
//
// The bodies live in slam6d/icp_glue.h, templated on the scan type: that header is compiled, linked and EXECUTED on the
// GPU box with a minimal scan type (adapters/harness/icp_glue_harness.cc).  This file itself is NOT compiled in this
// repository: icp6D.h pulls in scan.h -> Boost, which the build image lacks.
#ifndef __ICP6D_HIP_H__
#define __ICP6D_HIP_H__

#include "slam6d/icp6D.h"
#include "slam6d/icp6Dlumeuler.h"
#include "slam6d/hip_search_tree.h"
#include "slam6d/icp_glue.h"
#include "tdtk_hip.h"

// the -a id tdtk_icp_match expects.  getAlgorithmID() is not unique in the reference (icp6D_LUMEULER
// answers 3 like icp6D_ORTHO, include/slam6d/icp6Dlumeuler.h:33), so the class decides.
static inline int hip_algo_id(icp6Dminimizer* m)
{
  if (dynamic_cast<icp6D_LUMEULER*>(m)) return TDTK_ALGO_LUMEULER;
  const int id = m->getAlgorithmID();      // 1 QUAT 2 SVD 3 ORTHO 4 DUAL 5 HELIX 6 APX 8 LUMQUAT 9 QUAT_SCALE 10 NAPX
  return (id >= 1 && id <= 10 && id != 7) ? id : 0;
}

// the reference's Scan seen through the member names the glue uses
struct HipIcpScanView {
  Scan* s;
  const double* get_transMat() const { return s->get_transMat(); }
  const double* getDAlign() const { return s->getDAlign(); }
  tdtk_tree* hipTree() { return static_cast<HipSearchTree*>(s->getSearchTree())->handle(); }
  tdtk_scan* hipResident() { return s->hipResident(); }
  int hipBucket() { return s->getBucketSize(); }    // KDtreeMetaManaged: kdMeta.cc:45-46 (the configured size: no tree is built to ask)
  void transformMatrixAndFrames(const double* xf, int type, int islum) { s->transformMatrixAndFrames(xf, (Scan::AlgoType)type, islum); }
  // Scan::transform moves the resident copy too (reference.patch, Scan::transformReduced)
  void mergeCoordinatesWithRoboterPosition(HipIcpScanView* prev) { s->mergeCoordinatesWithRoboterPosition(prev->s); }
};

class icp6D_hip : public icp6D {
public:
  using icp6D::icp6D;

  HipIcpSettings settings(PairingMode pairing_mode)
  {
    HipIcpSettings c = {hip_algo_id(my_icp6Dminimizer), (int)pairing_mode, max_num_iterations, max_dist_match2, epsilonICP,
                        quiet, anim, eP, (int)Scan::ICP, meta, max_num_metascans, rnd};
    return c;
  }
  bool on_device(Scan* model)
  {
    // the model tree lives in the scan (Scan::getSearchTree, scan.cc:268); it is a HipSearchTree when the scan was
    // configured with nns_type HipKD.  (-R draws std::rand() per candidate: the resident loop draws the mask on the host per
    // iteration and sends it as bits since round 5 -- tdtk_icp_match_rnd -- with a serial build's consumption of the stream.)
    return dynamic_cast<HipSearchTree*>(model->getSearchTree()) != 0 && hip_algo_id(my_icp6Dminimizer) != 0;
  }

  virtual int match(Scan* PreviousScan, Scan* CurrentScan, PairingMode pairing_mode = CLOSEST_POINT)
  {
    if (!on_device(PreviousScan)) return icp6D::match(PreviousScan, CurrentScan, pairing_mode);     // CPU path of the reference
    HipIcpScanView prev = {PreviousScan}, cur = {CurrentScan};
    unsigned int pairs = 0;      // (icp6D::nr_pointPair is an int, include/slam6d/icp6D.h:150)
    const int it = hip_icp_match(&prev, &cur, settings(pairing_mode), &pairs);
    nr_pointPair = (int)pairs;
    // a caller of match() may look at the points afterwards: hand the moved points back
    DataXYZ xyz(CurrentScan->get("xyz reduced"));
    if (xyz.size() && tdtk_scan_download(CurrentScan->hipResident(), xyz[0], 0) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
    return it;
  }

  // icp6D::doICP is not virtual in the reference; slam6D.cc holds an icp6D*, so reference.patch makes it virtual
  virtual void doICP(std::vector<Scan*> allScans, PairingMode pairing_mode = CLOSEST_POINT)
  {
    // --metascan (BASELINE config 1) stays on the device: hip_do_icp builds the MetaScan's tree over the resident copies of
    // the scans matched so far (tdtk_tree_create_from_scans = KDtreeMetaManaged, kdMeta.cc:34-134).  CAD matching keeps
    // the reference's loop.
    if (cad_matching || allScans.empty() || !on_device(allScans[0])) { icp6D::doICP(allScans, pairing_mode); return; }
    std::vector<HipIcpScanView> views(allScans.size());
    std::vector<HipIcpScanView*> ptrs(allScans.size());
    for (size_t i = 0; i < allScans.size(); i++) { views[i].s = allScans[i]; ptrs[i] = &views[i]; }
    unsigned int pairs = 0;
    hip_do_icp(ptrs, settings(pairing_mode), /*scans prepared ahead*/ 3, &pairs,
               [](size_t i, int) { std::cout << i << "*" << std::endl; });
    nr_pointPair = (int)pairs;
    // the host copies of "xyz reduced" once, at the end
    for (Scan* s : allScans) {
      DataXYZ xyz(s->get("xyz reduced"));
      if (xyz.size() && s->hipResidentOrNull() && tdtk_scan_download(s->hipResidentOrNull(), xyz[0], 0) != TDTK_OK)
        throw std::runtime_error(tdtk_last_error());
    }
  }
};
#endif
