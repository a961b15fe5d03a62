// STUB (see README.txt) of include/slam6d/icp6Dlumeuler.h:22-40: declarations only.
#ifndef STUB_ICP6DLUMEULER_H
#define STUB_ICP6DLUMEULER_H
#include "slam6d/icp6Dminimizer.h"
class icp6D_LUMEULER : public icp6Dminimizer {
public:
  int getAlgorithmID();
};
#endif
