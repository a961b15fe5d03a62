// slam6D_hip -- reference-side binding for bin/slam6D's own orchestration: `matchGraph6Dautomatic` (the free function of
// src/slam6d/slam6D.cc:387-548 that main() calls for -G / -L runs, slam6D.cc:880-905) served by
// slam6d/slam6d_glue.h: sequential ICP with the next scans prepared ahead, loop detection, -L 1 loop closing with every
// covariance pass of the loop graph in one batched call and the MetaScan match on the device, -G 1..4 rounds through
// graph_slam_glue.h.  Call matchGraph6Dautomatic_hip where slam6D.cc calls matchGraph6Dautomatic when -t HipKD is
// selected and the run is one the glue serves (-L 0 .. 4, rnd <= 1; meta_icp and the -DlastSLAM pass included since round 4);
// otherwise keep the reference's call.
//
// The body lives in slam6d/slam6d_glue.h, templated on the scan type: that header is compiled, linked and EXECUTED on
// the GPU box with a minimal scan type (adapters/harness/slam_glue_harness.cc), where every pose it ends with equals the
// Python mirror's bit for bit.  This file itself is NOT compiled in this repository (scan.h -> Boost).
#ifndef __SLAM6D_HIP_H__
#define __SLAM6D_HIP_H__

#include "slam6d/graphSlam6D_hip.h"
#include "slam6d/icp6D_hip.h"
#include "slam6d/slam6d_glue.h"

// the reference's Scan seen through the member names the three glue headers use
struct HipSlamScanView {
  Scan* s;
  const double* get_transMat() const { return s->get_transMat(); }
  const double* get_transMatOrg() const { return s->get_transMatOrg(); }
  const double* getDAlign() const { return s->getDAlign(); }
  const double* get_rPos() const { return s->get_rPos(); }
  const double* get_rPosTheta() const { return s->get_rPosTheta(); }
  size_t hipPoints() { return s->size<DataXYZ>("xyz reduced"); }
  int hipBucket() { return s->getBucketSize(); }   // kdMeta.cc:45-46 (no tree is built to ask)
  tdtk_tree* hipTree() { return static_cast<HipSearchTree*>(s->getSearchTree())->handle(); }
  tdtk_scan* hipResident() { return s->hipResident(); }
  tdtk_scan* hipResidentOrNull() { return s->hipResidentOrNull(); }
  void transformMatrixAndFrames(const double* xf, int type, int islum) { s->transformMatrixAndFrames(xf, (Scan::AlgoType)type, islum); }
  // Scan::transform moves the resident copy too (reference.patch, Scan::transformReduced), so these are the reference's own
  void transformToEuler(const double rP[3], const double rPT[3], int type, int islum)
  {
    double p[3] = {rP[0], rP[1], rP[2]}, t[3] = {rPT[0], rPT[1], rPT[2]};
    s->transformToEuler(p, t, (Scan::AlgoType)type, islum);
  }
  void mergeCoordinatesWithRoboterPosition(HipSlamScanView* prev) { s->mergeCoordinatesWithRoboterPosition(prev->s); }
  // -L 2 .. 4 (the loop closers with quaternion poses)
  void get_rPosQuat(double q[4]) const { for (int k = 0; k < 4; k++) q[k] = s->get_rPosQuat()[k]; }
  void transformToQuat(const double rP[3], const double rPQ[4], int type, int islum)
  {
    double p[3] = {rP[0], rP[1], rP[2]}, q[4] = {rPQ[0], rPQ[1], rPQ[2], rPQ[3]};
    s->transformToQuat(p, q, (Scan::AlgoType)type, islum);
  }
  void transform(const double alignxf[16], int type, int islum) { s->transform(alignxf, (Scan::AlgoType)type, islum); }
  // the frames a MetaScan's transform writes for islum == 0 (scan.cc:962-975): members get `type`, the scans before the
  // first member ICPINACTIVE, the others INVALID
  static void metaFrames(const std::vector<HipSlamScanView*>& members, int type)
  {
    int found = 0;
    for (unsigned int i = 0; i < Scan::allScans.size(); i++) {
      bool member = false;
      for (HipSlamScanView* m : members) member = member || m->s == Scan::allScans[i];
      if (member) { found = i; Scan::allScans[i]->addFrame((Scan::AlgoType)type); }
      else Scan::allScans[i]->addFrame(found == 0 ? Scan::ICPINACTIVE : Scan::INVALID);
    }
  }
};

// returns the number of global rounds; `backend`: the -G id (1..4) or -1, `comm`: the library's communicator or 0
static inline int matchGraph6Dautomatic_hip(double cldist, int loopsize, std::vector<Scan*> allScans, icp6D_hip* my_icp6D,
                                            icp6D_hip* loop_icp6D /* 0: no -L */, int backend, int nrIt, double epsilonSLAM,
                                            double mdml, double epsilonLUM, int prefetch = 3, tdtk_comm* comm = 0,
                                            bool meta_icp = false, int max_num_metascans = -1, double mdmll = -1.0,
                                            double graphDist = 0.0, int elch_variant /* the -L id */ = 1)
{
  std::vector<HipSlamScanView> views(allScans.size());
  std::vector<HipSlamScanView*> ptrs(allScans.size());
  for (size_t i = 0; i < allScans.size(); i++) { views[i].s = allScans[i]; ptrs[i] = &views[i]; }
  HipSlamSettings cfg = HipSlamSettings();
  cfg.meta_icp = meta_icp; cfg.max_num_metascans = max_num_metascans; cfg.mdmll = mdmll; cfg.graphDist = graphDist;
  cfg.icp = my_icp6D->settings(CLOSEST_POINT);
  cfg.loop_icp = loop_icp6D ? loop_icp6D->settings(CLOSEST_POINT) : cfg.icp;
  cfg.use_elch = loop_icp6D != 0;
  cfg.elch_variant = elch_variant;
  cfg.graph_backend = backend;
  cfg.cldist = cldist; cfg.mdml = mdml; cfg.epsilonSLAM = epsilonSLAM; cfg.epsilonLUM = epsilonLUM;
  cfg.loopsize = loopsize; cfg.nrIt = nrIt; cfg.prefetch = prefetch; cfg.comm = comm;
  const HipScanTypes ty = {(int)Scan::INVALID, (int)Scan::ICP, (int)Scan::LUM, (int)Scan::ELCH};
  return hip_match_graph6d_automatic(ptrs, cfg, ty);
}
#endif
