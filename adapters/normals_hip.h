// normals_hip -- reference-side binding for Scan::calcNormals' worker: a function with the signature of
// calculateNormalsApxKNN (include/slam6d/normals.h:20-24, src/slam6d/normals.cc:35-111) that runs the ANN-tree build,
// the approximate k-NN search and the PCA on the GPU (tdtk_normals_apx_knn).  Same neighbour lists, same normals,
// bit for bit, so nothing downstream changes; call it from Scan::calcNormals (src/slam6d/scan.cc:419) instead of
// calculateNormalsApxKNN when -t HipKD is selected.
//
// Compiles against the reference's headers only (slam6d/point.h); not built in this repository's image because
// normals.h pulls in scan.h -> Boost.
#ifndef __NORMALS_HIP_H__
#define __NORMALS_HIP_H__

#include <stdexcept>
#include <vector>

#include "slam6d/point.h"
#include "tdtk_hip.h"

inline void calculateNormalsApxKNN_hip(std::vector<Point>& normals, const std::vector<Point>& points, const int k,
                                       const double _rPos[3], const double eps = 0.0, int device = 0)
{
  const size_t n = points.size();
  std::vector<double> xyz(3 * n), nrm(3 * n);
  for (size_t i = 0; i < n; i++) { xyz[3 * i] = points[i].x; xyz[3 * i + 1] = points[i].y; xyz[3 * i + 2] = points[i].z; }
  // the library reports what ANN answers with annError(..., ANNabort) (k > n) or what Scan::calcNormals throws
  // (no points) as an error code; keep the reference's exception type
  if (tdtk_normals_apx_knn(xyz.data(), n, k, _rPos, eps, device, nrm.data(), 0) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  normals.reserve(normals.size() + n);
  for (size_t i = 0; i < n; i++) normals.push_back(Point(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]));   // normals.cc:105
}

// Where the scan is already resident (adapters/graphSlam6D_hip.h, addition (2): Scan::hipResident()), skip the host
// round trip: tdtk_scan_calc_normals(scan->hipResident(), K_NEIGHBOURS, scan->get_rPos(), 1.0) computes the normals
// of the resident points in place and keeps them on the device as the scan's "normal reduced".
#endif
