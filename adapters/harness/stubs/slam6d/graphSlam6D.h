// STUB (see README.txt) of include/slam6d/graphSlam6D.h:58-100 and include/slam6d/graph.h: declarations only.
#ifndef STUB_GRAPHSLAM6D_H
#define STUB_GRAPHSLAM6D_H
#include <vector>
#include "slam6d/icp6D.h"
class Graph {
public:
  int getNrScans();
  int getNrLinks();
  int getLink(int i, int fromTo);
};
class graphSlam6D {
public:
  graphSlam6D() {}
  graphSlam6D(icp6Dminimizer* my_icp6Dminimizer, double mdm, double max_dist_match, int max_num_iterations, bool quiet, bool meta,
              int rnd, bool eP, int anim, double epsilonICP, int nns_method, double epsilonLUM);
  virtual ~graphSlam6D();
  virtual double doGraphSlam6D(Graph gr, vector<Scan*> MetaScan, int nrIt) = 0;
  void set_mdmll(double mdmll);
protected:
  icp6D* my_icp;
  double epsilonLUM;
  double max_dist_match2_LUM;
  int nns_method;
  bool quiet;
};
#endif
