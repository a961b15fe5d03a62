// HipSearchTree -- the reference-side binding of lib3dtk_hip.so for the kd-tree plug point.
//
// Drop this pair of files into the reference tree (include/slam6d/ + src/slam6d/), add
// `HipKD` to `enum nns_type` (include/slam6d/scan.h:34-36) and one `case HipKD:` to the switch
// in BasicScan::createSearchTreePrivate (src/slam6d/basicScan.cc:706-727):
//
//     case HipKD:
//       kd = new HipSearchTree(ar.get(), xyz_orig.size(), searchtree_bucketsize);
//       break;
//
// and link slam6D against lib3dtk_hip.so.  `bin/slam6D -t <HipKD>` then routes every
// SearchTree::getPtPairs call (Scan::getPtPairs / getPtPairsParallel, scan.cc:1240,1336) through
// the GPU.  It compiles against the reference's own headers; this repository only
// syntax-checks it (tests/test_host_logic.py::test_adapter_compiles_against_reference_headers).
#ifndef __HIP_SEARCH_TREE_H__
#define __HIP_SEARCH_TREE_H__

#include <vector>

#include "slam6d/searchTree.h"
#include "tdtk_hip.h"

class HipSearchTree : public SearchTree {
public:
  // same signature as KDtree::KDtree (include/slam6d/kd.h); the points are copied to the GPU
  // (precedent for a tree that owns a private copy: ANNtree, BruteForceNotATree)
  HipSearchTree(double** pts, int n, int bucketSize = 20, int device = 0);
  virtual ~HipSearchTree();

  // single-query interface (scan_diff2d.cc:491, scan.cc:1138 ...): one-element batch
  virtual double* FindClosest(double* _p, double maxdist2, int threadNum = 0) const;
  virtual double* FindClosestAlongDir(double* _p, double* _dir, double maxdist2, int threadNum = 0) const;

  // the batch entry point ICP and graph-SLAM reach (searchTree.cc:92-189)
  virtual void getPtPairs(std::vector<PtPair>* pairs, double* source_alignxf, const DataXYZ& xyz_r,
                          const DataNormal& normal_r, unsigned int startindex, unsigned int endindex,
                          int thread_num, int rnd, double max_dist_match2, double& sum,
                          double* centroid_m, double* centroid_d, PairingMode pairing_mode = CLOSEST_POINT);
  // legacy pointer overload (searchTree.cc:31-90)
  virtual void getPtPairs(std::vector<PtPair>* pairs, double* source_alignxf, double* const* q_points,
                          unsigned int startindex, unsigned int endindex, int thread_num, int rnd,
                          double max_dist_match2, double& sum, double* centroid_m, double* centroid_d);

  tdtk_tree* handle() const { return tree_; }   // for icp6D_hip / lum6DEuler_hip
  int bucketSize() const { return bucket_; }    // a MetaScan's tree takes its first member's (kdMeta.cc:45-46)

private:
  tdtk_tree* tree_;
  int bucket_;
  std::vector<double*> index_to_ptr_;  // model index -> the caller's point (FindClosest returns these)
};

#endif
