// STUB (see README.txt) of include/slam6d/icp6Dminimizer.h: declarations only.
#ifndef STUB_ICP6DMINIMIZER_H
#define STUB_ICP6DMINIMIZER_H
class icp6Dminimizer {
public:
  virtual ~icp6Dminimizer();
  virtual int getAlgorithmID() = 0;
};
#endif
