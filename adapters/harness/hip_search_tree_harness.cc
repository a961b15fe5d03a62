// Executes the reference-side binding: links adapters/hip_search_tree.cc -- compiled against the reference's OWN
// headers (slam6d/searchTree.h, data_types.h, ptpair.h, point.h) -- with lib3dtk_hip.so and drives
// HipSearchTree::getPtPairs / FindClosest the way Scan::getPtPairs and the single-query callers do
// (src/slam6d/scan.cc:1240, :1138), with a real DataXYZ view over a double[N][3] block.  Compared against the C ABI
// called directly (tdtk_get_pt_pairs / tdtk_find_closest) and against a brute-force nearest neighbour.
//
// Built where a reference checkout exists (adapters/harness/build.sh puts the binary with the other build products
// that contain reference code, outside the history but inside what travels to the GPU box); run by tests/test_gpu_parity.py::test_reference_side_binding_executes.
//
// What is NOT the reference here: SearchTree's three out-of-line members live in src/slam6d/searchTree.cc, which
// includes scan.h -> Boost and so cannot be compiled in this image.  HipSearchTree overrides every one of them; the
// definitions below only give the linker the base class's vtable and trap if ever reached.  No reference behaviour
// is imitated by them.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "slam6d/hip_search_tree.h"

double* SearchTree::FindClosestAlongDir(double*, double*, double, int) const
{
  throw std::runtime_error("harness trap: SearchTree::FindClosestAlongDir (base)");
}
void SearchTree::getPtPairs(std::vector<PtPair>*, double*, double* const*, unsigned int, unsigned int, int, int, double,
                            double&, double*, double*)
{
  throw std::runtime_error("harness trap: SearchTree::getPtPairs (base, pointer overload)");
}
void SearchTree::getPtPairs(std::vector<PtPair>*, double*, const DataXYZ&, const DataNormal&, unsigned int, unsigned int,
                            int, int, double, double&, double*, double*, PairingMode)
{
  throw std::runtime_error("harness trap: SearchTree::getPtPairs (base, DataXYZ overload)");
}

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double urand(double lo, double hi)
{
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return lo + (hi - lo) * (double)(g_state >> 11) * (1.0 / 9007199254740992.0);
}

static int fail(const char* what)
{
  std::printf("HARNESS FAIL: %s\n", what);
  return 1;
}

int main(int argc, char** argv)
{
  const int M = argc > 1 ? std::atoi(argv[1]) : 200000;
  const int N = argc > 2 ? std::atoi(argv[2]) : 150000;
  const double maxd2 = 9.0;
  std::vector<double> model(3 * (size_t)M), data(3 * (size_t)N), nrm(3 * (size_t)N);
  for (double& v : model) v = urand(-200.0, 200.0);
  for (int i = 0; i < 2000 && i < M / 4; i++)              // exact duplicates: ties go to the first visited
    for (int k = 0; k < 3; k++) model[3 * (M - 1 - i) + k] = model[3 * i + k];
  // a pose for the model scan (dalignxf), data = model sample + noise mapped to the world
  const double th = 0.03, c = std::cos(th), s = std::sin(th);
  double A[16] = {c, s, 0, 0, -s, c, 0, 0, 0, 0, 1, 0, 3.0, -2.0, 1.5, 1};
  for (int i = 0; i < N; i++) {
    const double* p = &model[3 * (size_t)((uint64_t)i * 7919u % (uint64_t)M)];
    const double x = p[0] + urand(-0.6, 0.6), y = p[1] + urand(-0.6, 0.6), z = p[2] + urand(-0.6, 0.6);
    data[3 * i] = x * A[0] + y * A[4] + z * A[8] + A[12];
    data[3 * i + 1] = x * A[1] + y * A[5] + z * A[9] + A[13];
    data[3 * i + 2] = x * A[2] + y * A[6] + z * A[10] + A[14];
    double n[3] = {urand(-1, 1), urand(-1, 1), urand(-1, 1)};
    for (int k = 0; k < 3; k++) nrm[3 * i + k] = n[k];
  }
  // the caller's side exactly as BasicScan::createSearchTreePrivate does it (basicScan.cc:704-709)
  std::vector<double*> ptrs(M);
  for (int i = 0; i < M; i++) ptrs[i] = &model[3 * (size_t)i];
  try {
    HipSearchTree tree(ptrs.data(), M, 20);
    SearchTree* st = &tree;                              // every caller holds the base pointer

    DataXYZ xyz(DataPointer((unsigned char*)data.data(), data.size() * sizeof(double)));
    DataNormal normals(DataPointer((unsigned char*)nrm.data(), nrm.size() * sizeof(double)));
    DataNormal no_normals(DataPointer(0, 0));
    if (xyz.size() != (size_t)N) return fail("DataXYZ view has the wrong size");

    for (int mode = 0; mode <= 2; mode += 2) {
      std::vector<PtPair> pairs;
      double sum = 0, cm[3] = {0, 0, 0}, cd[3] = {0, 0, 0};
      st->getPtPairs(&pairs, A, xyz, mode ? normals : no_normals, 0, (unsigned)N, 0, 0, maxd2, sum, cm, cd,
                     mode ? CLOSEST_PLANE_SIMPLE : CLOSEST_POINT);
      // the C ABI directly
      std::vector<int32_t> idx(N);
      std::vector<double> p1(3 * (size_t)N), p2(3 * (size_t)N), pn(3 * (size_t)N);
      tdtk_pair_sums S;
      if (tdtk_get_pt_pairs(tree.handle(), A, data.data(), mode ? nrm.data() : 0, 0, N, 0, mode, maxd2, TDTK_WANT_BASE,
                            0, idx.data(), p1.data(), p2.data(), pn.data(), &S) != TDTK_OK)
        return fail(tdtk_last_error());
      if (pairs.size() != S.n) return fail("pair count differs between the adapter and the C ABI");
      for (size_t k = 0; k < pairs.size(); k++) {
        const PtPair& pp = pairs[k];
        if (pp.p1.x != p1[3 * k] || pp.p1.y != p1[3 * k + 1] || pp.p1.z != p1[3 * k + 2] || pp.p2.x != p2[3 * k] ||
            pp.p2.y != p2[3 * k + 1] || pp.p2.z != p2[3 * k + 2])
          return fail("PtPair differs from the C ABI pair list");
        if (mode && (pp.p2.nx != pn[3 * k] || pp.p2.ny != pn[3 * k + 1] || pp.p2.nz != pn[3 * k + 2]))
          return fail("PtPair normal differs from the C ABI pair list");
      }
      // the accumulators the reference passes in are ADDED to (searchTree.cc:165-177): un-normalised sums
      double s2 = 0, m2[3] = {0, 0, 0}, d2[3] = {0, 0, 0};
      for (const PtPair& pp : pairs) {
        const double dx = pp.p1.x - pp.p2.x, dy = pp.p1.y - pp.p2.y, dz = pp.p1.z - pp.p2.z;
        s2 += dx * dx + dy * dy + dz * dz;
        m2[0] += pp.p1.x; m2[1] += pp.p1.y; m2[2] += pp.p1.z;
        d2[0] += pp.p2.x; d2[1] += pp.p2.y; d2[2] += pp.p2.z;
      }
      if (std::fabs(s2 - sum) > 1e-9 * s2) return fail("sum does not match the pair list");
      for (int k = 0; k < 3; k++)
        if (std::fabs(m2[k] - cm[k]) > 1e-8 * (1.0 + std::fabs(m2[k])) || std::fabs(d2[k] - cd[k]) > 1e-8 * (1.0 + std::fabs(d2[k])))
          return fail("centroid sums do not match the pair list");
      // brute force on a sample of the queries (mode 0): same distance, and the same point unless duplicated
      if (mode == 0) {
        double inv[16];
        if (!tdtk_host_m4inv(A, inv)) return fail("M4inv");
        for (int i = 0; i < N; i += 1499) {
          const double* t = &data[3 * (size_t)i];
          const double qx = t[0] * inv[0] + t[1] * inv[4] + t[2] * inv[8] + inv[12];
          const double qy = t[0] * inv[1] + t[1] * inv[5] + t[2] * inv[9] + inv[13];
          const double qz = t[0] * inv[2] + t[1] * inv[6] + t[2] * inv[10] + inv[14];
          double best = maxd2; int bi = -1;
          for (int j = 0; j < M; j++) {
            const double dx = model[3 * (size_t)j] - qx, dy = model[3 * (size_t)j + 1] - qy, dz = model[3 * (size_t)j + 2] - qz;
            const double d = dx * dx + dy * dy + dz * dz;
            if (d < best) { best = d; bi = j; }
          }
          if ((bi < 0) != (idx[i] < 0)) return fail("brute force disagrees on found / not found");
          if (bi >= 0) {
            const double* g = &model[3 * (size_t)idx[i]];
            const double dx = g[0] - qx, dy = g[1] - qy, dz = g[2] - qz;
            if (dx * dx + dy * dy + dz * dz != best) return fail("brute force finds a nearer point");
          }
        }
        std::printf("mode 0: %zu pairs of %d queries, sum %.6f\n", pairs.size(), N, sum);
      } else {
        std::printf("mode 2: %zu pairs (point-to-plane projection, normals carried in PtPair)\n", pairs.size());
      }
      // appending semantics: a second call on a sub-range grows the same vector
      const size_t before = pairs.size();
      st->getPtPairs(&pairs, A, xyz, mode ? normals : no_normals, 1000, 3000, 0, 0, maxd2, sum, cm, cd,
                     mode ? CLOSEST_PLANE_SIMPLE : CLOSEST_POINT);
      size_t expect = 0;
      for (int i = 1000; i < 3000; i++) expect += idx[i] >= 0;
      if (pairs.size() != before + expect) return fail("sub-range call did not append the expected pairs");
    }

    // single-query interface (scan.cc:1138, scan_diff2d.cc:491): returns the CALLER's pointer
    int checked = 0;
    for (int i = 0; i < N && checked < 200; i += 733, checked++) {
      double inv[16];
      tdtk_host_m4inv(A, inv);
      const double* t = &data[3 * (size_t)i];
      double q[3] = {t[0] * inv[0] + t[1] * inv[4] + t[2] * inv[8] + inv[12], t[0] * inv[1] + t[1] * inv[5] + t[2] * inv[9] + inv[13],
                     t[0] * inv[2] + t[1] * inv[6] + t[2] * inv[10] + inv[14]};
      double* r = st->FindClosest(q, maxd2, 0);
      int32_t one = -1;
      if (tdtk_find_closest(tree.handle(), q, 1, maxd2, &one, 0) != TDTK_OK) return fail(tdtk_last_error());
      if ((r == 0) != (one < 0)) return fail("FindClosest: found / not found");
      if (r && r != &model[3 * (size_t)one]) return fail("FindClosest does not return the caller's pointer");
    }
    // legacy pointer overload (searchTree.cc:31-90)
    {
      std::vector<double*> qp(N);
      for (int i = 0; i < N; i++) qp[i] = &data[3 * (size_t)i];
      std::vector<PtPair> pairs;
      double sum = 0, cm[3] = {0, 0, 0}, cd[3] = {0, 0, 0};
      st->getPtPairs(&pairs, A, qp.data(), 0, 5000, 0, 0, maxd2, sum, cm, cd);
      std::vector<PtPair> pairs2;
      double sum2 = 0, cm2[3] = {0, 0, 0}, cd2[3] = {0, 0, 0};
      st->getPtPairs(&pairs2, A, xyz, no_normals, 0, 5000, 0, 0, maxd2, sum2, cm2, cd2, CLOSEST_POINT);
      if (pairs.size() != pairs2.size() || sum != sum2) return fail("pointer overload differs from the DataXYZ overload");
    }
    // error convention: exceptions on the reference's side of the boundary
    bool threw = false;
    try { HipSearchTree bad(ptrs.data(), 0, 20); } catch (const std::runtime_error&) { threw = true; }
    if (!threw) return fail("empty tree did not throw");
  } catch (const std::exception& e) {
    return fail(e.what());
  }
  std::printf("HARNESS OK\n");
  return 0;
}
