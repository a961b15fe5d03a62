// STUB (see README.txt) of include/slam6d/searchTree.h:38-113, include/slam6d/ptpair.h, include/slam6d/pairingMode.h,
// include/slam6d/data_types.h: declarations only.
#ifndef STUB_SEARCHTREE_H
#define STUB_SEARCHTREE_H
#include <cstddef>
#include <vector>
enum PairingMode { CLOSEST_POINT, CLOSEST_POINT_ALONG_NORMAL_SIMPLE, CLOSEST_PLANE_SIMPLE };
class PtPair;
class DataPointer {};
class DataXYZ {
public:
  DataXYZ(const DataPointer&);
  size_t size() const;
  double* operator[](size_t i);
  const double* operator[](size_t i) const;
};
typedef DataXYZ DataNormal;
class SearchTree {
public:
  virtual ~SearchTree();
  virtual double* FindClosest(double* _p, double maxdist2, int threadNum = 0) const = 0;
  virtual double* FindClosestAlongDir(double* _p, double* _dir, double maxdist2, int threadNum = 0) const = 0;
  virtual void getPtPairs(std::vector<PtPair>* pairs, double* source_alignxf, const DataXYZ& xyz_r, const DataNormal& normal_r,
                          unsigned int startindex, unsigned int endindex, int thread_num, int rnd, double max_dist_match2,
                          double& sum, double* centroid_m, double* centroid_d, PairingMode pairing_mode = CLOSEST_POINT);
  virtual void getPtPairs(std::vector<PtPair>* pairs, double* source_alignxf, double* const* q_points, unsigned int startindex,
                          unsigned int endindex, int thread_num, int rnd, double max_dist_match2, double& sum,
                          double* centroid_m, double* centroid_d);
};
#endif
