// Executes adapters/slam6d_glue.h on the GPU box: matchGraph6Dautomatic with sequential ICP (scans prepared ahead),
// loop detection, ELCH loop closing (-L 1: batched covariance passes, graph balancer, a MetaScan-against-MetaScan match)
// and rounds of global relaxation (-G 1 through adapters/graph_slam_glue.h), driven through a minimal scan type that
// keeps the books the way the reference's Scan does (transformMatrix, scan.cc:878-898; the frame rules of
// Scan::transform, scan.cc:945-1008; transformToEuler, scan.cc:1061-1083; a scan is loaded in its own frame and moved
// to its pose when first used, basicScan.cc:730-737).
//
// usage: slam_glue_harness <in.bin> <out.bin> [nscans] [npts] [prefetch] [variant]
// variant 0: sequential ICP against the predecessor; 1: meta_icp with max_num_metascans = 3 (slam6D.cc:436-448) and the
// closing -DlastSLAM pass (mdmll = 15, graphDist = 140; slam6D.cc:535-547); 2 / 3 / 4: like 0 with the loop closed by
// elch6Dquat / elch6DunitQuat / elch6Dslerp (-L 2 / 3 / 4) instead of elch6Deuler
// Writes the scans it made to <in.bin> (int32 nscans, int32 npts, then per scan rPos[3], rPosTheta[3], xyz[npts][3]) and
// what it ended with to <out.bin> (per scan transMat[16], then one int32 frame count per scan, int32 rounds), so that
// tests/test_gpu_parity.py::test_slam_glue_executes can run the Python mirror on the same scans and compare bit for bit.
// Also checks here: prefetch depth 0 and the given depth end in the same matrices.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../slam6d_glue.h"

enum { T_INVALID = 0, T_ICP = 1, T_ICPINACTIVE = 2, T_LUM = 3, T_ELCH = 4 };

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static double urand(double lo, double hi)
{
  g_state = g_state * 6364136223846793005ull + 1442695040888963407ull;
  return lo + (hi - lo) * (double)(g_state >> 11) * (1.0 / 9007199254740992.0);
}

struct MiniScan {
  static std::vector<MiniScan*> all;       // Scan::allScans
  std::vector<double> local;               // the points as loaded: scanner frame
  double in_rPos[3], in_rPosTheta[3];      // the pose the scan was created with (what the .pose file said)
  double rPos[3], rPosTheta[3];
  double transMat[16], transMatOrg[16], dalignxf[16];
  std::vector<std::vector<double>> queue;  // transforms applied while the scan was not resident yet
  tdtk_scan* res = nullptr;
  tdtk_tree* tree = nullptr;
  int frames = 0;
  ~MiniScan() { if (tree) tdtk_tree_destroy(tree); if (res) tdtk_scan_destroy(res); }

  void init(const double rP[3], const double rPT[3])
  {
    std::memcpy(in_rPos, rP, sizeof rPos); std::memcpy(in_rPosTheta, rPT, sizeof rPosTheta);
    tdtk_host_euler_to_matrix4(in_rPos, in_rPosTheta, transMatOrg);
    const double I[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    tdtk_host_mmult(transMatOrg, I, transMat);                     // basicScan.cc:188: transformMatrix(transMatOrg) from the
    tdtk_host_matrix4_to_euler(transMat, rPosTheta, rPos);         // identity (a -0.0 of the pose matrix becomes +0.0 here)
    std::memcpy(dalignxf, I, sizeof I);
  }
  const double* get_transMat() const { return transMat; }
  const double* get_transMatOrg() const { return transMatOrg; }
  const double* getDAlign() const { return dalignxf; }
  const double* get_rPos() const { return rPos; }
  const double* get_rPosTheta() const { return rPosTheta; }
  size_t hipPoints() const { return local.size() / 3; }
  int hipBucket() const { return 20; }
  tdtk_scan* hipResidentOrNull() { return res; }
  tdtk_scan* hipResident()
  {
    if (!res) {
      // calcReducedOnDemandPrivate (basicScan.cc:730-737): transformReduced(transMatOrg), then copyReducedToOriginal
      if (tdtk_scan_create(local.data(), nullptr, local.size() / 3, 0, &res) != TDTK_OK ||
          tdtk_scan_transform(res, transMatOrg) != TDTK_OK || tdtk_scan_mark_original(res) != TDTK_OK)
        throw std::runtime_error(tdtk_last_error());
      for (const std::vector<double>& A : queue)
        if (tdtk_scan_transform(res, A.data()) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
      queue.clear();
    }
    return res;
  }
  tdtk_tree* hipTree()
  {
    if (!tree && tdtk_tree_create_from_scan(hipResident(), hipBucket(), &tree) != TDTK_OK) throw std::runtime_error(tdtk_last_error());
    return tree;
  }
  void addFrame() { frames++; }
  // the frame rules of Scan::transform (scan.cc:945-1008)
  void addFrames(int type, int islum)
  {
    if (type == T_INVALID || islum == -1) return;
    bool registered = false;
    for (MiniScan* s : all) registered = registered || s == this;
    if (islum == 1 || !registered) { addFrame(); return; }
    int found = 0;
    if (islum == 0) {
      for (size_t i = 0; i < all.size(); i++) {
        if (all[i] == this) found = (int)i;
        all[i]->addFrame();            // this scan: `type`; before it ICPINACTIVE; after it INVALID -- a frame each
      }
    } else {
      for (size_t i = 0; i < all.size(); i++) {
        if (all[i] == this) { found = (int)i; addFrame(); all[0]->addFrame(); continue; }
        if (found != 0) all[i]->addFrame();
      }
    }
  }
  void transformMatrixAndFrames(const double* alignxf, int type, int islum)
  {
    double t[16];
    tdtk_host_mmult(alignxf, transMat, t); std::memcpy(transMat, t, sizeof t);
    tdtk_host_matrix4_to_euler(transMat, rPosTheta, rPos);
    tdtk_host_mmult(alignxf, dalignxf, t); std::memcpy(dalignxf, t, sizeof t);
    addFrames(type, islum);
  }
  void transform(const double* alignxf, int type, int islum)
  {
    if (res) { if (tdtk_scan_transform(res, alignxf) != TDTK_OK) throw std::runtime_error(tdtk_last_error()); }
    else queue.push_back(std::vector<double>(alignxf, alignxf + 16));
    transformMatrixAndFrames(alignxf, type, islum);
  }
  void transformToEuler(const double rP[3], const double rPT[3], int type, int islum)
  {
    double tinv[16], M[16];
    tdtk_host_m4inv(transMat, tinv);
    transform(tinv, T_INVALID, -1);
    tdtk_host_euler_to_matrix4(rP, rPT, M);
    transform(M, type, islum);
  }
  void get_rPosQuat(double q[4]) const { double t[3]; tdtk_host_matrix4_to_quat(transMat, q, t); }     // scan.cc:886
  void transformToQuat(const double rP[3], const double rPQ[4], int type, int islum)                     // scan.cc:1093-1104
  {
    double tinv[16], M[16];
    tdtk_host_m4inv(transMat, tinv);
    transform(tinv, T_INVALID, -1);
    tdtk_host_quat_to_matrix4(rPQ, rP, M);
    transform(M, type, islum);
  }
  void mergeCoordinatesWithRoboterPosition(MiniScan* prev)
  {
    double inv[16], delta[16];
    tdtk_host_m4inv(prev->get_transMatOrg(), inv);
    tdtk_host_mmult(prev->get_transMat(), inv, delta);
    transform(delta, T_INVALID, -1);
  }
  static void metaFrames(const std::vector<MiniScan*>& members, int /*type*/)
  {
    // scan.cc:962-975: members get `type`, the others ICPINACTIVE / INVALID -- a frame each if any member is registered
    bool any = false;
    for (MiniScan* s : all) for (MiniScan* m : members) any = any || s == m;
    if (!any) { for (MiniScan* m : members) m->addFrame(); return; }
    for (MiniScan* s : all) s->addFrame();
  }
};
std::vector<MiniScan*> MiniScan::all;

// a path around one world cloud that closes and then leaves: scans 0 .. 12 stand on a circle (scan 12 where scan 0
// stood), the scans behind them walk away from it; every scan sees the same points, and its "odometry" pose drifts away
// from the truth as k grows -- so scan 11 comes within the closing distance of scan 0, scan 12 nearly onto it, and the
// first scan with nothing near it (13) triggers the loop closing of (0, 12) and the relaxation (slam6D.cc:478-533)
static void make_loop(std::vector<MiniScan>& scans, int nscans, int npts)
{
  const uint64_t keep = g_state;
  std::vector<double> world(3 * (size_t)npts);
  for (double& v : world) v = urand(-400.0, 400.0);
  scans.clear(); scans.resize(nscans);
  for (int k = 0; k < nscans; k++) {
    const double ang = 2.0 * 3.14159265358979323846 * (k <= 12 ? k : 12) / 12.0;
    const double away = k <= 12 ? 0.0 : 150.0 * (k - 12);
    const double tP[3] = {120.0 * std::sin(ang), 3.0 * std::sin(2 * ang), 120.0 * (1.0 - std::cos(ang)) - away};
    const double tT[3] = {0.01 * std::sin(ang), 0.15 * std::sin(ang) + 0.0004 * away, 0.008 * std::cos(ang) - 0.008};
    double Tm[16], Ti[16];
    tdtk_host_euler_to_matrix4(tP, tT, Tm);
    tdtk_host_m4inv(Tm, Ti);
    MiniScan& s = scans[k];
    s.local.resize(world.size());
    for (size_t i = 0; i < world.size(); i += 3) {
      const double x = world[i] + urand(-0.2, 0.2), y = world[i + 1] + urand(-0.2, 0.2), z = world[i + 2] + urand(-0.2, 0.2);
      s.local[i] = x * Ti[0] + y * Ti[4] + z * Ti[8] + Ti[12];
      s.local[i + 1] = x * Ti[1] + y * Ti[5] + z * Ti[9] + Ti[13];
      s.local[i + 2] = x * Ti[2] + y * Ti[6] + z * Ti[10] + Ti[14];
    }
    const double drift = 0.35 * k;
    const double oP[3] = {tP[0] + drift, tP[1] - 0.3 * drift, tP[2] + 0.6 * drift};
    const double oT[3] = {tT[0], tT[1] + 0.0006 * k, tT[2]};
    s.init(oP, oT);
  }
  g_state = keep;
}

static int g_variant = 0;
static int run(std::vector<MiniScan>& scans, int prefetch, int* rounds)
{
  MiniScan::all.clear();
  std::vector<MiniScan*> ptrs;
  for (MiniScan& s : scans) { ptrs.push_back(&s); MiniScan::all.push_back(&s); }
  HipSlamSettings cfg{};
  if (g_variant == 1) { cfg.meta_icp = true; cfg.max_num_metascans = 3; cfg.mdmll = 15.0; cfg.graphDist = 140.0; }
  cfg.icp = {TDTK_ALGO_QUAT, 0, 30, 25.0 * 25.0, 1e-5, true, -1, true, T_ICP};
  cfg.loop_icp = {TDTK_ALGO_QUAT, 0, 30, 25.0 * 25.0, 1e-5, true, -1, true, T_ICP};
  cfg.use_elch = true;
  cfg.elch_variant = g_variant >= 2 ? g_variant : 1;
  cfg.graph_backend = TDTK_GRAPH_LUMEULER;
  cfg.cldist = 90.0; cfg.mdml = 25.0; cfg.epsilonSLAM = 0.05; cfg.epsilonLUM = 0.5;
  cfg.loopsize = 6; cfg.nrIt = 3; cfg.prefetch = prefetch; cfg.comm = nullptr;
  const HipScanTypes ty = {T_INVALID, T_ICP, T_LUM, T_ELCH};
  *rounds = hip_match_graph6d_automatic(ptrs, cfg, ty);
  return 0;
}

// `slam_glue_harness doicp <in.bin> <out.bin> <meta 0|1> <max_num_metascans> <mdm> <iterations> <epsilonICP> [rnd [seed]]`: icp6D::doICP
// (adapters/icp_glue.h, hip_do_icp) over scans read from <in.bin> -- int32 nscans, then per scan int32 npts, rPos[3],
// rPosTheta[3], xyz[npts][3] in the scanner's frame -- with -a 1; <out.bin>: per scan transMat[16], then int32 frames per
// scan.  BASELINE config 1 (`slam6D -m 500 -d 25.0 --metascan dat`) runs through this (test_config1_metascan_dat_through_the_cpp_glue).
static int run_doicp(int argc, char** argv)
{
  if (argc < 9) { std::printf("usage: %s doicp in.bin out.bin meta max_num_metascans mdm iterations epsilonICP\n", argv[0]); return 2; }
  FILE* f = std::fopen(argv[2], "rb");
  if (!f) { std::printf("SLAM GLUE HARNESS FAIL: cannot read %s\n", argv[2]); return 1; }
  int32_t nscans = 0;
  if (std::fread(&nscans, sizeof nscans, 1, f) != 1 || nscans < 1 || nscans > 4096) { std::fclose(f); std::printf("SLAM GLUE HARNESS FAIL: bad header\n"); return 1; }
  std::vector<MiniScan> scans((size_t)nscans);
  for (MiniScan& s : scans) {
    int32_t np = 0;
    double pose[6];
    if (std::fread(&np, sizeof np, 1, f) != 1 || np < 0 || std::fread(pose, sizeof pose, 1, f) != 1) { std::fclose(f); std::printf("SLAM GLUE HARNESS FAIL: short file\n"); return 1; }
    s.local.resize(3 * (size_t)np);
    if (np && std::fread(s.local.data(), sizeof(double), s.local.size(), f) != s.local.size()) { std::fclose(f); std::printf("SLAM GLUE HARNESS FAIL: short file\n"); return 1; }
    s.init(pose, pose + 3);
  }
  std::fclose(f);
  const double mdm = std::atof(argv[6]);
  // optional: <rnd> <seed> -- `-R <rnd>` with std::srand(seed) before each run (BASELINE config 1 is `-R 5`)
  const int rnd = argc > 9 ? std::atoi(argv[9]) : 1;
  const unsigned seed = argc > 10 ? (unsigned)std::atoi(argv[10]) : 1u;
  HipIcpSettings cfg = {TDTK_ALGO_QUAT, 0, std::atoi(argv[7]), mdm * mdm, std::atof(argv[8]), true, -1, true, T_ICP,
                        std::atoi(argv[4]) != 0, std::atoi(argv[5]), rnd};
  // Under -R the runs are seeded.  The HIP runtime itself draws from std::rand() while it initialises (first use of the device,
  // of a code object, of a thread's context: the next rand() behind the FIRST doICP of a process differs from run to run,
  // with or without -R), so a run whose draws are to be reproduced comes after one that has warmed the runtime up: under -R
  // a first, unseeded pass is made and thrown away.
  const int plan_rnd[] = {-1, 2, 0}, plan[] = {2, 0};
  for (int prefetch : (rnd > 1 ? std::vector<int>(plan_rnd, plan_rnd + 3) : std::vector<int>(plan, plan + 2))) {       // prepared ahead or not: the same poses (the second run's are written)
    const bool warmup = prefetch < 0;
    if (warmup) prefetch = 0;
    std::srand(seed);
    std::vector<MiniScan> run_scans((size_t)nscans);
    MiniScan::all.clear();
    std::vector<MiniScan*> ptrs;
    for (int k = 0; k < nscans; k++) {
      run_scans[k].local = scans[k].local;
      run_scans[k].init(scans[k].in_rPos, scans[k].in_rPosTheta);
      ptrs.push_back(&run_scans[k]); MiniScan::all.push_back(&run_scans[k]);
    }
    unsigned int pairs = 0;
    std::vector<int> its;
    hip_do_icp(ptrs, cfg, prefetch, &pairs, [&](size_t, int it) { its.push_back(it); });
    if (std::getenv("TDTK_HARNESS_DEBUG")) {
      std::printf("DEBUG prefetch %d: iterations", prefetch);
      for (int it : its) std::printf(" %d", it);
      std::printf(", last pairs %u, next rand %d\n", pairs, std::rand());
    }
    if (warmup) { MiniScan::all.clear(); continue; }
    if (prefetch == 2) {
      for (int k = 0; k < nscans; k++) std::memcpy(scans[k].transMat, run_scans[k].transMat, sizeof scans[k].transMat);
      continue;
    }
    for (int k = 0; k < nscans; k++)
      if (std::memcmp(scans[k].transMat, run_scans[k].transMat, sizeof scans[k].transMat) != 0) {
        std::printf("SLAM GLUE HARNESS FAIL: doICP scan %d differs between prefetch 2 and none\n", k);
        return 1;
      }
    FILE* o = std::fopen(argv[3], "wb");
    if (!o) { std::printf("SLAM GLUE HARNESS FAIL: cannot write %s\n", argv[3]); return 1; }
    for (MiniScan& s : run_scans) std::fwrite(s.transMat, sizeof s.transMat, 1, o);
    for (MiniScan& s : run_scans) { const int32_t fr = s.frames; std::fwrite(&fr, sizeof fr, 1, o); }
    for (int it : its) { const int32_t v = it; std::fwrite(&v, sizeof v, 1, o); }
    std::fclose(o);
    std::printf("SLAM GLUE HARNESS OK: doICP meta %d over %d scans, iterations", (int)cfg.meta, nscans);
    for (int it : its) std::printf(" %d", it);
    std::printf(", last pairs %u\n", pairs);
    MiniScan::all.clear();
  }
  return 0;
}

int main(int argc, char** argv)
{
  if (argc > 1 && std::strcmp(argv[1], "doicp") == 0) {
    try { return run_doicp(argc, argv); }
    catch (const std::exception& e) { std::printf("SLAM GLUE HARNESS FAIL: %s\n", e.what()); return 1; }
  }
  if (argc < 3) { std::printf("usage: %s in.bin out.bin [nscans] [npts] [prefetch]\n", argv[0]); return 2; }
  const int nscans = argc > 3 ? std::atoi(argv[3]) : 15;
  const int npts = argc > 4 ? std::atoi(argv[4]) : 30000;
  const int prefetch = argc > 5 ? std::atoi(argv[5]) : 3;
  g_variant = argc > 6 ? std::atoi(argv[6]) : 0;
  try {
    std::vector<MiniScan> a, b;
    make_loop(a, nscans, npts);
    {
      FILE* f = std::fopen(argv[1], "wb");
      if (!f) { std::printf("SLAM GLUE HARNESS FAIL: cannot write %s\n", argv[1]); return 1; }
      const int32_t hd[2] = {nscans, npts};
      std::fwrite(hd, sizeof hd, 1, f);
      for (MiniScan& s : a) {
        std::fwrite(s.in_rPos, sizeof s.in_rPos, 1, f); std::fwrite(s.in_rPosTheta, sizeof s.in_rPosTheta, 1, f);
        std::fwrite(s.local.data(), sizeof(double), s.local.size(), f);
      }
      std::fclose(f);
    }
    int rounds_a = 0, rounds_b = 0;
    run(a, prefetch, &rounds_a);
    make_loop(b, nscans, npts);
    run(b, 0, &rounds_b);
    if (rounds_a != rounds_b) { std::printf("SLAM GLUE HARNESS FAIL: rounds %d (prefetch %d) vs %d (none)\n", rounds_a, prefetch, rounds_b); return 1; }
    for (int k = 0; k < nscans; k++)
      if (std::memcmp(a[k].transMat, b[k].transMat, sizeof a[k].transMat) != 0 || a[k].frames != b[k].frames) {
        std::printf("SLAM GLUE HARNESS FAIL: scan %d differs between prefetch %d and none\n", k, prefetch);
        return 1;
      }
    FILE* f = std::fopen(argv[2], "wb");
    if (!f) { std::printf("SLAM GLUE HARNESS FAIL: cannot write %s\n", argv[2]); return 1; }
    for (MiniScan& s : a) std::fwrite(s.transMat, sizeof s.transMat, 1, f);
    for (MiniScan& s : a) { const int32_t fr = s.frames; std::fwrite(&fr, sizeof fr, 1, f); }
    const int32_t r = rounds_a;
    std::fwrite(&r, sizeof r, 1, f);
    std::fclose(f);
    double moved = 0.0;
    for (int k = 0; k < nscans; k++) moved = std::fmax(moved, std::fabs(a[k].transMat[12] - a[k].transMatOrg[12]));
    std::printf("SLAM GLUE HARNESS OK: variant %d, %d scans x %d points, %d global rounds, prefetch %d == none, largest pose correction %.3f\n",
                g_variant, nscans, npts, rounds_a, prefetch, moved);
  } catch (const std::exception& e) {
    std::printf("SLAM GLUE HARNESS FAIL: %s\n", e.what());
    return 1;
  }
  return 0;
}
