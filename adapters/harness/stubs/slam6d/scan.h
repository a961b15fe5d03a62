// STUB (see README.txt) of include/slam6d/scan.h:114-360 plus the members adapters/reference.patch adds
// (transformMatrixAndFrames, hipResident, hipResidentOrNull): declarations only.
#ifndef STUB_SCAN_H
#define STUB_SCAN_H
#include <string>
#include <vector>
#include "slam6d/searchTree.h"
#include "tdtk_hip.h"
enum nns_type { simpleKD, ANNTree, BOCTree, HipKD };
class Scan {
public:
  enum AlgoType { INVALID, ICP, ICPINACTIVE, LUM, ELCH };
  static std::vector<Scan*> allScans;
  virtual ~Scan();
  const double* get_rPos() const;
  const double* get_rPosTheta() const;
  const double* get_rPosQuat() const;
  const double* get_transMat() const;
  const double* get_transMatOrg() const;
  const double* getDAlign() const;
  SearchTree* getSearchTree();
  int getBucketSize() const;
  virtual DataPointer get(const std::string& identifier) = 0;
  template <typename T> size_t size(const std::string& identifier) { return (T(get(identifier))).size(); }
  virtual void addFrame(AlgoType type) = 0;
  void mergeCoordinatesWithRoboterPosition(Scan* prevScan);
  void transform(const double alignxf[16], const AlgoType type, int islum = 0);
  void transformToEuler(double rP[3], double rPT[3], const AlgoType type, int islum = 0);
  void transformToQuat(double rP[3], double rPQ[4], const AlgoType type, int islum = 0);
  // adapters/reference.patch
  void transformMatrixAndFrames(const double alignxf[16], const AlgoType type, int islum);
  tdtk_scan* hipResident();
  tdtk_scan* hipResidentOrNull() const;
};
#endif
