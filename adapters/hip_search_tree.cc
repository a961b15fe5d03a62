// See hip_search_tree.h.  Error convention: the reference throws std::runtime_error from tree
// code (kdTreeImpl.h:86-88, basicScan.cc:723-726); the C ABI never throws, so its return codes are
// turned into exceptions here, on the reference's side of the boundary.
#include "slam6d/hip_search_tree.h"

#include <stdexcept>
#include <string>

static void tdtk_check(int rc)
{
  if (rc != TDTK_OK) throw std::runtime_error(std::string("lib3dtk_hip: ") + tdtk_last_error());
}

HipSearchTree::HipSearchTree(double** pts, int n, int bucketSize, int device) : tree_(0), bucket_(bucketSize)
{
  if (n <= 0) throw std::runtime_error("cannot create kdtree with zero points");
  std::vector<double> xyz(3 * (size_t)n);
  index_to_ptr_.resize(n);
  for (int i = 0; i < n; i++) {
    xyz[3 * i] = pts[i][0]; xyz[3 * i + 1] = pts[i][1]; xyz[3 * i + 2] = pts[i][2];
    index_to_ptr_[i] = pts[i];
  }
  tdtk_check(tdtk_tree_create(xyz.data(), (size_t)n, bucketSize, device, &tree_));
}

HipSearchTree::~HipSearchTree() { tdtk_tree_destroy(tree_); }

double* HipSearchTree::FindClosest(double* _p, double maxdist2, int) const
{
  int32_t idx = -1;
  tdtk_check(tdtk_find_closest(tree_, _p, 1, maxdist2, &idx, 0));
  return idx < 0 ? 0 : index_to_ptr_[idx];
}

double* HipSearchTree::FindClosestAlongDir(double* _p, double* _dir, double maxdist2, int) const
{
  int32_t idx = -1;
  tdtk_check(tdtk_find_closest_along_dir(tree_, _p, _dir, 1, maxdist2, &idx, 0));
  return idx < 0 ? 0 : index_to_ptr_[idx];
}

void HipSearchTree::getPtPairs(std::vector<PtPair>* pairs, double* source_alignxf, const DataXYZ& xyz_r,
                               const DataNormal& normal_r, unsigned int startindex, unsigned int endindex,
                               int, int rnd, double max_dist_match2, double& sum, double* centroid_m,
                               double* centroid_d, PairingMode pairing_mode)
{
  if (endindex <= startindex) return;
  const size_t n = endindex - startindex;
  // DataXYZ / DataNormal are views of one contiguous double[N][3] block (data_types.h:171-219)
  const double* xyz = xyz_r[0];
  const double* nrm = (pairing_mode != CLOSEST_POINT && normal_r.size() > 0) ? normal_r[0] : 0;
  std::vector<double> p1(3 * n), p2(3 * n), pn(3 * n);
  tdtk_pair_sums s;
  tdtk_check(tdtk_get_pt_pairs(tree_, source_alignxf, xyz, nrm, startindex, endindex, rnd, (int)pairing_mode,
                               max_dist_match2, TDTK_WANT_BASE, 0, 0, p1.data(), p2.data(), pn.data(), &s));
  // the reference accumulates un-normalised centroid sums and `sum` into its arguments
  // (searchTree.cc:165-177); Scan::getPtPairs* divides afterwards (scan.cc:1253-1259)
  sum += s.sum;
  for (int k = 0; k < 3; k++) {
    centroid_m[k] += s.centroid_m[k] * (double)s.n;
    centroid_d[k] += s.centroid_d[k] * (double)s.n;
  }
  pairs->reserve(pairs->size() + s.n);
  for (size_t k = 0; k < s.n; k++) {
    if (pairing_mode != CLOSEST_POINT) pairs->push_back(PtPair(&p1[3 * k], &p2[3 * k], &pn[3 * k]));
    else pairs->push_back(PtPair(&p1[3 * k], &p2[3 * k]));
  }
}

void HipSearchTree::getPtPairs(std::vector<PtPair>* pairs, double* source_alignxf, double* const* q_points,
                               unsigned int startindex, unsigned int endindex, int, int rnd,
                               double max_dist_match2, double& sum, double* centroid_m, double* centroid_d)
{
  if (endindex <= startindex) return;
  const size_t n = endindex - startindex;
  std::vector<double> q(3 * n), p1(3 * n), p2(3 * n);
  for (size_t i = 0; i < n; i++)
    for (int k = 0; k < 3; k++) q[3 * i + k] = q_points[startindex + i][k];
  tdtk_pair_sums s;
  tdtk_check(tdtk_get_pt_pairs(tree_, source_alignxf, q.data(), 0, 0, n, rnd, TDTK_CLOSEST_POINT, max_dist_match2,
                               TDTK_WANT_BASE, 0, 0, p1.data(), p2.data(), 0, &s));
  sum += s.sum;
  for (int k = 0; k < 3; k++) {
    centroid_m[k] += s.centroid_m[k] * (double)s.n;
    centroid_d[k] += s.centroid_d[k] * (double)s.n;
  }
  for (size_t k = 0; k < s.n; k++) pairs->push_back(PtPair(&p1[3 * k], &p2[3 * k]));
}
