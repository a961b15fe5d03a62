// Executes adapters/icp_glue.h -- the body of icp6D_hip::match and the prefetching doICP -- on the GPU box, with a
// minimal scan type that has the member functions the glue asks of the reference's Scan (get_transMat, getDAlign,
// hipTree, hipResident, transformMatrixAndFrames, mergeCoordinatesWithRoboterPosition) and keeps the same books
// (Scan::transformMatrix, scan.cc:878-898; frames as Scan::transform appends them, scan.cc:956-1008).
//
// Checks: (1) hip_do_icp over a chain of scans == the same matches issued one by one through the C ABI
// (tdtk_icp_match + tdtk_host_mmult bookkeeping), matrices bit for bit; (2) prefetch depth 0 / 1 / 3 give identical
// poses, frame counts and resident points; (3) the pose extrapolation (eP) reaches the resident copy; (4) the number
// of frames written per match follows icp6D.cc:246-268 (anim = -1: the start pose, iteration 0 and the end pose).
// No reference header is needed (that is the point of the glue); built by adapters/harness/build_glue.sh in build()
// and in the CPU tier, run by tests/test_gpu_parity.py::test_icp_glue_executes.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../icp_glue.h"

static uint64_t g_state = 0x243F6A8885A308D3ull;
static double urand(double lo, double hi)
{
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return lo + (hi - lo) * (double)(g_state >> 11) * (1.0 / 9007199254740992.0);
}

struct MiniScan {
  std::vector<double> xyz;            // "xyz reduced": moves with every transform
  std::vector<double> orig;           // "xyz reduced original" (copyReducedToOriginal, basicScan.cc:739-757): what the search
                                      // tree is built over and dalignxf is relative to -- never moves
  double transMat[16], transMatOrg[16], dalignxf[16];
  tdtk_scan* res = nullptr;
  tdtk_tree* tree = nullptr;
  int frames = 0;                     // frames appended (islum != -1)
  ~MiniScan() { if (tree) tdtk_tree_destroy(tree); if (res) tdtk_scan_destroy(res); }

  const double* get_transMat() const { return transMat; }
  const double* get_transMatOrg() const { return transMatOrg; }
  const double* getDAlign() const { return dalignxf; }
  int hipBucket() const { return 20; }
  tdtk_scan* hipResident()
  {
    if (!res) {
      if (tdtk_scan_create(xyz.data(), nullptr, xyz.size() / 3, 0, &res) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
      tdtk_scan_mark_original(res);
    }
    return res;
  }
  tdtk_tree* hipTree()
  {
    // like BasicScan::createSearchTreePrivate (basicScan.cc:702-728): over the ORIGINAL points, wherever the scan is now
    if (!tree && tdtk_tree_create(orig.data(), orig.size() / 3, 20, 0, &tree) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
    return tree;
  }
  void transformMatrixAndFrames(const double* alignxf, int /*type*/, int islum)
  {
    double t[16];
    tdtk_host_mmult(alignxf, transMat, t); std::memcpy(transMat, t, sizeof t);      // scan.cc:883-884
    tdtk_host_mmult(alignxf, dalignxf, t); std::memcpy(dalignxf, t, sizeof t);      // scan.cc:896-897
    if (islum != -1) frames++;
  }
  void transform(const double* alignxf, int type, int islum)     // Scan::transform: the points too, host AND resident
  {
    for (size_t i = 0; i < xyz.size(); i += 3) {
      const double x = xyz[i], y = xyz[i + 1], z = xyz[i + 2];
      xyz[i] = x * alignxf[0] + y * alignxf[4] + z * alignxf[8] + alignxf[12];
      xyz[i + 1] = x * alignxf[1] + y * alignxf[5] + z * alignxf[9] + alignxf[13];
      xyz[i + 2] = x * alignxf[2] + y * alignxf[6] + z * alignxf[10] + alignxf[14];
    }
    if (res && tdtk_scan_transform(res, alignxf) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
    transformMatrixAndFrames(alignxf, type, islum);
  }
  void mergeCoordinatesWithRoboterPosition(MiniScan* prev)       // scan.cc:826-833
  {
    double inv[16], delta[16];
    tdtk_host_m4inv(prev->get_transMatOrg(), inv);
    tdtk_host_mmult(prev->get_transMat(), inv, delta);
    transform(delta, 0, 0);
  }
};

static void pose(double* M, double tx, double ty, double tz, double yaw)
{
  const double c = std::cos(yaw), s = std::sin(yaw);
  const double T[16] = {c, 0, -s, 0, 0, 1, 0, 0, s, 0, c, 0, tx, ty, tz, 1};
  std::memcpy(M, T, sizeof T);
}

// a chain of scans of one world cloud seen from slightly wrong poses
static void make_chain(std::vector<MiniScan>& scans, int nscans, int npts)
{
  const uint64_t keep = g_state;
  std::vector<double> world(3 * (size_t)npts);
  for (double& v : world) v = urand(-300.0, 300.0);
  scans.clear(); scans.resize(nscans);
  for (int k = 0; k < nscans; k++) {
    MiniScan& s = scans[k];
    double odo[16];
    pose(odo, 0.8 * k, 0.0, -0.5 * k, 0.004 * k);                 // what the "odometry" says
    std::memcpy(s.transMatOrg, odo, sizeof odo);
    std::memcpy(s.transMat, odo, sizeof odo);
    const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::memcpy(s.dalignxf, I, sizeof I);
    // the points as the scan holds them after loading: world + noise, displaced by the pose error of this scan
    double err[16];
    pose(err, k ? 1.5 : 0.0, k ? -0.7 : 0.0, k ? 0.9 : 0.0, k ? 0.003 : 0.0);
    s.xyz.resize(world.size());
    for (size_t i = 0; i < world.size(); i += 3) {
      const double x = world[i] + urand(-0.3, 0.3), y = world[i + 1] + urand(-0.3, 0.3), z = world[i + 2] + urand(-0.3, 0.3);
      s.xyz[i] = x * err[0] + y * err[4] + z * err[8] + err[12];
      s.xyz[i + 1] = x * err[1] + y * err[5] + z * err[9] + err[13];
      s.xyz[i + 2] = x * err[2] + y * err[6] + z * err[10] + err[14];
    }
    s.orig = s.xyz;
  }
  g_state = keep;   // the same chain every time it is made
}

static int fail(const char* what) { std::printf("ICP GLUE HARNESS FAIL: %s\n", what); return 1; }

int main(int argc, char** argv)
{
  const int nscans = argc > 1 ? std::atoi(argv[1]) : 5;
  const int npts = argc > 2 ? std::atoi(argv[2]) : 120000;
  HipIcpSettings cfg = {TDTK_ALGO_QUAT, 0, 40, 25.0, 1e-5, true, -1, true, /*Scan::ICP*/ 1};
  try {
    // (1) the glue, no prefetch
    std::vector<MiniScan> a;
    make_chain(a, nscans, npts);
    std::vector<MiniScan*> pa;
    for (MiniScan& s : a) pa.push_back(&s);
    unsigned int pairs_glue = 0;
    std::vector<int> its_glue;
    auto t0 = std::chrono::steady_clock::now();
    hip_do_icp(pa, cfg, 0, &pairs_glue, [&](size_t, int it) { its_glue.push_back(it); });
    const double ms0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    std::printf("doICP, nothing prepared ahead: %.2f ms per scan (upload + ordering + tree build + match, %d scans)\n", ms0 / nscans, nscans);

    // the same through the C ABI, written out
    std::vector<MiniScan> b;
    make_chain(b, nscans, npts);
    std::vector<int> its_abi;
    unsigned int pairs_abi = 0;
    for (int i = 1; i < nscans; i++) {
      MiniScan &prev = b[i - 1], &cur = b[i];
      (void)cur.hipResident();                       // uploaded where it was loaded, then extrapolated (see hip_do_icp)
      cur.mergeCoordinatesWithRoboterPosition(&prev);
      tdtk_icp_params prm = {cfg.algo, cfg.pairing_mode, cfg.max_num_iterations, cfg.max_dist_match2, cfg.epsilonICP, 1};
      tdtk_icp_result res;
      // tdtk_icp_match keeps the two matrices itself when they are handed in: this IS Scan::transformMatrix per iteration
      if (tdtk_icp_match(prev.hipTree(), prev.dalignxf, cur.hipResident(), cur.transMat, cur.dalignxf, &prm, &res, nullptr, 0) != TDTK_OK)
        return fail(tdtk_last_error());
      its_abi.push_back(res.iterations);
      pairs_abi = (unsigned int)res.last_pairs;
    }
    if (its_glue != its_abi) return fail("iteration counts differ between the glue and the C ABI sequence");
    if (pairs_glue != pairs_abi) return fail("nr_pointPair differs");
    for (int i = 0; i < nscans; i++) {
      if (std::memcmp(a[i].transMat, b[i].transMat, sizeof a[i].transMat) != 0) return fail("transMat differs from the C ABI sequence");
      if (std::memcmp(a[i].dalignxf, b[i].dalignxf, sizeof a[i].dalignxf) != 0) return fail("dalignxf differs from the C ABI sequence");
    }
    // frames: the scan's start pose, iteration 0 and the end pose per match when anim = -1 (plus the eP transform)
    for (int i = 1; i < nscans; i++)
      if (a[i].frames != 4) { std::printf("scan %d wrote %d frames\n", i, a[i].frames); return fail("frame bookkeeping"); }
    // the poses were wrong by (1.5, -0.7, 0.9): matching must have moved every scan close to its predecessor's frame
    for (int i = 1; i < nscans; i++)
      if (std::fabs(a[i].dalignxf[12] + 1.5) > 0.2 || std::fabs(a[i].dalignxf[13] - 0.7) > 0.2 || std::fabs(a[i].dalignxf[14] + 0.9) > 0.2) {
        std::printf("scan %d: dalignxf t = %.3f %.3f %.3f\n", i, a[i].dalignxf[12], a[i].dalignxf[13], a[i].dalignxf[14]);
        return fail("ICP did not recover the pose error");
      }
    std::printf("glue == C ABI sequence on %d scans of %d points: iterations", nscans, npts);
    for (int v : its_glue) std::printf(" %d", v);
    std::printf(", last pairs %u\n", pairs_glue);

    // (2) prefetch depth must not change anything
    for (int depth : {1, 3}) {
      std::vector<MiniScan> c;
      make_chain(c, nscans, npts);
      std::vector<MiniScan*> pc;
      for (MiniScan& s : c) pc.push_back(&s);
      unsigned int pr = 0;
      std::vector<int> its;
      auto t1 = std::chrono::steady_clock::now();
      hip_do_icp(pc, cfg, depth, &pr, [&](size_t, int it) { its.push_back(it); });
      const double ms1 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
      std::printf("doICP, %d scan(s) prepared ahead: %.2f ms per scan\n", depth, ms1 / nscans);
      if (its != its_glue || pr != pairs_glue) return fail("prefetch changed the iteration counts");
      std::vector<double> pa_pts(3 * (size_t)npts), pc_pts(3 * (size_t)npts);
      for (int i = 0; i < nscans; i++) {
        if (std::memcmp(a[i].transMat, c[i].transMat, sizeof a[i].transMat) != 0) return fail("prefetch changed a pose");
        if (a[i].frames != c[i].frames) return fail("prefetch changed the frame count");
        if (tdtk_scan_download(a[i].hipResident(), pa_pts.data(), nullptr) != TDTK_OK ||
            tdtk_scan_download(c[i].hipResident(), pc_pts.data(), nullptr) != TDTK_OK)
          return fail(tdtk_last_error());
        if (std::memcmp(pa_pts.data(), pc_pts.data(), pa_pts.size() * sizeof(double)) != 0) return fail("prefetch changed the resident points");
      }
      std::printf("prefetch depth %d: identical poses, frames and resident points\n", depth);
    }
    // (3) a match that finds no pairs: breaks before anything is applied, no end pose (icp6D.cc:235-245)
    {
      std::vector<MiniScan> d;
      make_chain(d, 2, 20000);
      double far_away[16];
      pose(far_away, 5000.0, 0.0, 0.0, 0.0);
      d[1].transform(far_away, 0, -1);
      const int before = d[1].frames;
      unsigned int pr = 77;
      HipIcpSettings c2 = cfg; c2.eP = false;
      const int it = hip_icp_match(&d[0], &d[1], c2, &pr);
      if (it != 0 || pr != 0) return fail("a match without pairs must stop in iteration 0");
      if (d[1].frames != before + 1) return fail("a match without pairs writes the start pose only");
    }
  } catch (const std::exception& e) {
    return fail(e.what());
  }
  std::printf("ICP GLUE HARNESS OK\n");
  return 0;
}
