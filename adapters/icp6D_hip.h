// icp6D_hip -- reference-side binding for the icp6D::match plug point (include/slam6d/icp6D.h:51,
// `virtual int match(Scan*, Scan*, PairingMode)`).  Construct it wherever slam6D.cc constructs
// icp6D (src/slam6d/slam6D.cc:737,770,810,825) when -t HipKD is selected.  The whole loop of
// src/slam6d/icp6D.cc:104-285 then runs device-resident: one upload of the data scan, one tiny
// D2H per iteration, one download at the end.
//
// NOT compiled in this repository: icp6D.h pulls in scan.h -> Boost, which the build image lacks.
#ifndef __ICP6D_HIP_H__
#define __ICP6D_HIP_H__

#include "slam6d/icp6D.h"
#include "slam6d/icp6Dlumeuler.h"
#include "slam6d/hip_search_tree.h"
#include "tdtk_hip.h"

// the -a id tdtk_icp_match expects.  getAlgorithmID() is not unique in the reference (icp6D_LUMEULER
// answers 3 like icp6D_ORTHO, include/slam6d/icp6Dlumeuler.h:33), so the class decides.
static inline int hip_algo_id(icp6Dminimizer* m)
{
  if (dynamic_cast<icp6D_LUMEULER*>(m)) return TDTK_ALGO_LUMEULER;
  const int id = m->getAlgorithmID();      // 1 QUAT 2 SVD 3 ORTHO 4 DUAL 5 HELIX 6 APX 8 LUMQUAT 9 QUAT_SCALE 10 NAPX
  return (id >= 1 && id <= 10 && id != 7) ? id : 0;
}

class icp6D_hip : public icp6D {
public:
  using icp6D::icp6D;

  virtual int match(Scan* PreviousScan, Scan* CurrentScan, PairingMode pairing_mode = CLOSEST_POINT)
  {
    // the model tree lives in the scan (Scan::getSearchTree, scan.cc:268); it is a HipSearchTree
    // when the scan was configured with nns_type HipKD
    HipSearchTree* hst = dynamic_cast<HipSearchTree*>(PreviousScan->getSearchTree());
    const int algo = hip_algo_id(my_icp6Dminimizer);
    if (!hst || rnd > 1 || algo == 0)
      return icp6D::match(PreviousScan, CurrentScan, pairing_mode);     // CPU path of the reference

    double id[16];
    M4identity(id);
    CurrentScan->transform(id, Scan::ICP, 0);                           // icp6D.cc:109
    if (max_num_iterations == 0) return 0;

    DataXYZ xyz(CurrentScan->get("xyz reduced"));
    DataNormal nrm(pairing_mode != CLOSEST_POINT ? CurrentScan->get("normal reduced") : DataPointer(0, 0));
    tdtk_scan* data = 0;
    if (tdtk_scan_create(xyz[0], nrm.size() ? nrm[0] : 0, xyz.size(), 0, &data) != TDTK_OK)
      throw std::runtime_error(tdtk_last_error());

    tdtk_icp_params prm = { algo, (int)pairing_mode, max_num_iterations, max_dist_match2, epsilonICP, quiet ? 1 : 0 };
    tdtk_icp_result res;
    std::vector<double> trace(18 * (size_t)max_num_iterations);
    double tm[16], da[16];                     // scratch: the Scan keeps its own matrices (below)
    memcpy(tm, CurrentScan->get_transMat(), sizeof tm);
    memcpy(da, CurrentScan->getDAlign(), sizeof da);
    int rc = tdtk_icp_match(hst->handle(), PreviousScan->getDAlign(), data, tm, da, &prm, &res,
                            trace.data(), max_num_iterations);
    if (rc != TDTK_OK) { tdtk_scan_destroy(data); throw std::runtime_error(tdtk_last_error()); }

    // hand the moved points back and replay the matrix / frame bookkeeping exactly as
    // Scan::transform would have done per iteration (scan.cc:918-1009)
    tdtk_scan_download(data, xyz[0], nrm.size() ? nrm[0] : 0);
    tdtk_scan_destroy(data);
    for (int i = 0; i <= res.iterations && i < max_num_iterations; i++)
      CurrentScan->transformMatrixAndFrames(&trace[18 * i + 2], Scan::ICP,
                                            (i == 0 && anim != -2) || (anim > 0 && i % anim == 0) ? 0 : -1);
    CurrentScan->transform(id, Scan::ICP, anim == -2 ? -1 : 0);          // write end pose
    nr_pointPair = (unsigned int)res.last_pairs;
    return res.iterations;
  }
};
// Two one-line additions this needs in the reference: `tdtk_tree* handle() const { return tree_; }`
// in HipSearchTree (present in our copy below when TDTK_EXPOSE_HANDLE is defined) and a public
// Scan::transformMatrixAndFrames(alignxf, type, islum) = Scan::transform without transformReduced
// (scan.cc:956-1008), because the points were already moved on the device.
#endif
