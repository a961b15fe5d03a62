// STUB (see README.txt) of include/slam6d/icp6D.h:27-160 with adapters/reference.patch applied (doICP virtual):
// declarations only, member names and types as there (nr_pointPair is an int).
#ifndef STUB_ICP6D_H
#define STUB_ICP6D_H
#include <iostream>
#include <vector>
#include "slam6d/icp6Dminimizer.h"
#include "slam6d/scan.h"
using std::vector;
class icp6D {
public:
  icp6D(icp6Dminimizer* my_icp6Dminimizer, double max_dist_match = 25.0, int max_num_iterations = 50, bool quiet = false,
        bool meta = false, int rnd = 1, bool eP = true, int anim = -1, double epsilonICP = 0.0000001, int nns_method = simpleKD,
        bool cuda_enabled = false, bool cad_matching = false, int max_num_metascans = -1);
  virtual ~icp6D() {}
  virtual void doICP(vector<Scan*> allScans, PairingMode pairing_mode = CLOSEST_POINT);
  virtual int match(Scan* PreviousScan, Scan* CurrentScan, PairingMode pairing_mode = CLOSEST_POINT);
protected:
  bool quiet;
  int rnd;
  bool eP;
  bool meta;
  int nns_method;
  bool cuda_enabled;
  double max_dist_match2;
  int max_num_iterations;
  int anim;
  double epsilonICP;
  icp6Dminimizer* my_icp6Dminimizer;
  unsigned int max_scn_size;
  bool cad_matching;
  int cad_index;
  int nr_pointPair;
  int max_num_metascans;
};
#endif
