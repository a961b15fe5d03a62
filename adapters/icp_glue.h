// The part of the ICP binding that needs nothing from the reference but the SHAPE of its Scan interface: the body of
// icp6D_hip::match (icp6D::match, src/slam6d/icp6D.cc:104-285, run device-resident by tdtk_icp_match) and a doICP
// (icp6D.cc:374-437) that prepares the next scans -- upload, ordering, search-tree build -- on worker threads while the
// current one is matched, which is where the library's per-scan numbers come from (6.4 ms per 1M-point scan with three
// scans prepared ahead).  adapters/icp6D_hip.h instantiates both with the reference's Scan;
// adapters/harness/icp_glue_harness.cc instantiates them with a minimal scan type of the same member functions, is
// compiled and linked in the CPU tier and EXECUTED on the GPU box against tdtk_icp_match called directly.
#ifndef __ICP_GLUE_H__
#define __ICP_GLUE_H__

#include <condition_variable>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "tdtk_hip.h"

struct HipIcpSettings {
  int algo;                 // TDTK_ALGO_* of the minimizer (hip_algo_id in icp6D_hip.h)
  int pairing_mode;         // PairingMode of the call
  int max_num_iterations;   // icp6D::max_num_iterations
  double max_dist_match2;   // icp6D::max_dist_match2
  double epsilonICP;        // icp6D::epsilonICP
  bool quiet;               // icp6D::quiet
  int anim;                 // icp6D::anim
  bool eP;                  // icp6D::eP: extrapolate the pose before matching (doICP)
  int type_icp;             // Scan::ICP as an int (the glue does not see the enum)
  bool meta;                // icp6D::meta (--metascan): doICP matches every scan against a MetaScan of the scans before it
  int max_num_metascans;    // icp6D::max_num_metascans: only the last n of them (<= 0: all)
  int rnd;                  // icp6D::rnd (-R): > 1 = about every rnd-th point is a candidate of an iteration (tdtk_icp_match_rnd)
};

// ScanT needs: get_transMat(), getDAlign() -> const double*;  tdtk_tree* hipTree();  tdtk_scan* hipResident();
//              void transformMatrixAndFrames(const double*, int type, int islum);
//              void mergeCoordinatesWithRoboterPosition(ScanT* prev)     (doICP with eP; must also move a resident copy)
//              int hipBucket()                                           (doICP with meta: the MetaScan tree's bucket size)
//
// icp6D::match with the loop on the device.  Returns the iteration count icp6D::match returns; *nr_pointPair as the
// member of the same name.  The data scan stays resident (ScanT::hipResident): its host copy of "xyz reduced" is NOT
// refreshed here -- whoever needs it calls tdtk_scan_download on the resident handle (icp6D_hip::match does, doICP does
// once per scan at the end).
// (model: the search tree of the previous scan -- or of a MetaScan -- and the dalignxf that goes with it)
template <class ScanT>
int hip_icp_match_tree(const tdtk_tree* model_tree, const double* model_dalignxf, ScanT* cur, const HipIcpSettings& cfg,
                       unsigned int* nr_pointPair)
{
  const double id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  // icp6D.cc:109 `CurrentScan->transform(id, Scan::ICP, 0)`: the identity moves no point -- matrices and frame only
  cur->transformMatrixAndFrames(id, cfg.type_icp, 0);
  if (cfg.max_num_iterations == 0) return 0;
  tdtk_scan* data = cur->hipResident();
  tdtk_icp_params prm = {cfg.algo, cfg.pairing_mode, cfg.max_num_iterations, cfg.max_dist_match2, cfg.epsilonICP, cfg.quiet ? 1 : 0};
  tdtk_icp_result res;
  std::vector<double> trace(18 * (size_t)cfg.max_num_iterations);
  double tm[16], da[16];     // scratch: the Scan keeps its own matrices (replayed below)
  std::memcpy(tm, cur->get_transMat(), sizeof tm);
  std::memcpy(da, cur->getDAlign(), sizeof da);
  // -R: the keep-mask of every iteration is drawn inside the call (std::rand per point in index order, searchTree.cc:116-118)
  const int rc = cfg.rnd > 1 ? tdtk_icp_match_rnd(model_tree, model_dalignxf, data, tm, da, &prm, cfg.rnd, &res, trace.data(), cfg.max_num_iterations)
                             : tdtk_icp_match(model_tree, model_dalignxf, data, tm, da, &prm, &res, trace.data(), cfg.max_num_iterations);
  if (rc != TDTK_OK) throw std::runtime_error(tdtk_last_error());
  // The points have moved on the device; replay the matrix / frame bookkeeping of every iteration's
  // `CurrentScan->transform(alignxf, Scan::ICP, islum)` (icp6D.cc:246-252) and of the end pose (icp6D.cc:254-268).  The
  // loop ends either in the convergence / iteration-cap branch -- iteration `iterations` was applied and the end pose
  // written -- or because an iteration found at most 3 pairs, which breaks BEFORE that iteration is applied and writes
  // no end pose (icp6D.cc:235-245).
  const bool ended_in_the_cap_or_convergence_branch = res.last_pairs > 3;
  const int applied = res.iterations + (ended_in_the_cap_or_convergence_branch ? 1 : 0);
  for (int i = 0; i < applied && i < cfg.max_num_iterations; i++)
    cur->transformMatrixAndFrames(&trace[18 * (size_t)i + 2], cfg.type_icp,
                                  ((i == 0 && cfg.anim != -2) || (cfg.anim > 0 && i % cfg.anim == 0)) ? 0 : -1);
  if (ended_in_the_cap_or_convergence_branch)
    cur->transformMatrixAndFrames(id, cfg.type_icp, cfg.anim == -2 ? -1 : 0);     // write end pose
  if (nr_pointPair) *nr_pointPair = (unsigned int)res.last_pairs;
  return res.iterations;
}
template <class ScanT>
int hip_icp_match(ScanT* prev, ScanT* cur, const HipIcpSettings& cfg, unsigned int* nr_pointPair)
{
  return hip_icp_match_tree(prev->hipTree(), prev->getDAlign(), cur, cfg, nr_pointPair);
}

// icp6D::match(MetaScan* model, Scan* data) (icp6D.cc:104-285 with a MetaScan in front, what doICP's meta branch and
// matchGraph6Dautomatic's meta_icp ask for): ONE search tree over the members' CURRENT resident points, built on the
// device (tdtk_tree_create_from_scans = KDtreeMetaManaged, kdMeta.cc:34-134: concatenation order, the first member's
// bucket size); a MetaScan's own dalignxf is the identity (Scan base constructor, scan.cc:193).  The tree lives for
// this match, as the reference's MetaScan does.
template <class ScanT>
int hip_icp_match_metascan(const std::vector<ScanT*>& members, ScanT* cur, const HipIcpSettings& cfg, unsigned int* nr_pointPair)
{
  const double id[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (members.empty()) throw std::runtime_error("an empty MetaScan cannot be matched against");
  std::vector<tdtk_scan*> ms;
  for (ScanT* m : members) ms.push_back(m->hipResident());
  tdtk_tree* tree = nullptr;
  if (tdtk_tree_create_from_scans(ms.data(), (int)ms.size(), members[0]->hipBucket(), &tree) != TDTK_OK)
    throw std::runtime_error(tdtk_last_error());
  int it;
  try {
    it = hip_icp_match_tree(tree, id, cur, cfg, nr_pointPair);
  } catch (...) {
    tdtk_tree_destroy(tree);
    throw;
  }
  tdtk_tree_destroy(tree);
  return it;
}

// A few worker threads that run "prepare scan j" jobs in order; the library is thread-safe per handle and gives every
// host thread its own stream, so a preparation overlaps with the match the main thread is running.
class HipPrefetcher {
public:
  explicit HipPrefetcher(int threads)
  {
    for (int i = 0; i < threads; i++) workers.emplace_back([this] { run(); });
  }
  ~HipPrefetcher()
  {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv.notify_all();
    for (std::thread& t : workers) t.join();
  }
  // state[j]: 0 not started, 1 queued / running, 2 done, 3 failed (error[j] holds the text)
  void submit(size_t j, std::function<void()> job)
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      if (state.size() <= j) { state.resize(j + 1, 0); error.resize(j + 1); }
      if (state[j] != 0) return;
      state[j] = 1;
      jobs.push_back({j, std::move(job)});
    }
    cv.notify_one();
  }
  void wait(size_t j)
  {
    std::unique_lock<std::mutex> lk(mu);
    if (state.size() <= j || state[j] == 0) return;
    done.wait(lk, [&] { return state[j] >= 2; });
    if (state[j] == 3) throw std::runtime_error(error[j]);
  }

private:
  void run()
  {
    for (;;) {
      std::pair<size_t, std::function<void()>> job;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return stop || !jobs.empty(); });
        if (jobs.empty()) return;
        job = std::move(jobs.front());
        jobs.pop_front();
      }
      int st = 2;
      std::string err;
      try { job.second(); } catch (const std::exception& e) { st = 3; err = e.what(); }
      { std::lock_guard<std::mutex> lk(mu); state[job.first] = st; error[job.first] = err; }
      done.notify_all();
    }
  }
  std::mutex mu;
  std::condition_variable cv, done;
  std::deque<std::pair<size_t, std::function<void()>>> jobs;
  std::vector<int> state;
  std::vector<std::string> error;
  std::vector<std::thread> workers;
  bool stop = false;
};

// icp6D::doICP (icp6D.cc:374-437, without CAD matching): sequential matching of every scan against its predecessor or,
// with cfg.meta (--metascan, BASELINE config 1), against a MetaScan of the scans matched so far / the last
// cfg.max_num_metascans of them (icp6D.cc:396-434) -- with the next `prefetch` scans made resident (and, without meta,
// their search trees built) while the current pair is matched.  A scan's tree is built over "xyz reduced original", which no ICP step touches, and a scan is uploaded
// where it was loaded -- the pose extrapolation of scan j (eP) depends on the final pose of scan j-1, so it is applied
// after the match of j-1, to the resident copy (ScanT::mergeCoordinatesWithRoboterPosition must move both).  The result
// does not depend on `prefetch`, bit for bit (adapters/harness/icp_glue_harness.cc checks that).
// on_matched(i, iterations): called after scan i has been matched (the reference prints "i*" and the TIME line there).
template <class ScanT>
void hip_do_icp(std::vector<ScanT*>& allScans, const HipIcpSettings& cfg, int prefetch, unsigned int* nr_pointPair,
                const std::function<void(size_t, int)>& on_matched = nullptr)
{
  const size_t n = allScans.size();
  // -R (cfg.rnd > 1) draws from the process's std::rand() stream, which must be consumed as a serial reference consumes it;
  // worker threads that bring up HIP contexts draw from the same stream at times of their own (measured: the next rand()
  // behind a doICP differs from run to run as soon as scans are prepared ahead, with or without -R), so under -R nothing
  // is prepared ahead.
  if (cfg.rnd > 1) prefetch = 0;
  HipPrefetcher* pool = (prefetch > 0 && n > 2) ? new HipPrefetcher(prefetch) : nullptr;
  const bool meta = cfg.meta;
  std::vector<ScanT*> meta_scans;      // icp6D.cc:380-381, 421-433
  try {
    for (size_t i = 0; i < n; i++) {
      if (pool) {
        // scans i .. i + prefetch are (being) prepared; this iteration needs i - 1 (its tree) and i (its points).  With
        // meta the model tree is the MetaScan's, built per match: only the points are prepared ahead.
        for (size_t j = i; j < n && j <= i + (size_t)prefetch; j++) {
          ScanT* s = allScans[j];
          pool->submit(j, [s, meta] { (void)s->hipResident(); if (!meta) (void)s->hipTree(); });
        }
        if (i > 0) pool->wait(i - 1);
        pool->wait(i);
      }
      ScanT* cur = allScans[i];
      if (i > 0) {
        ScanT* prev = allScans[i - 1];
        // resident BEFORE the pose extrapolation, prepared ahead or not: the resident copy is ordered by where the points
        // are when they are uploaded, and the order of the pair sums (hence the last bits of a pose) follows from it
        (void)cur->hipResident();
        if (cfg.eP) cur->mergeCoordinatesWithRoboterPosition(prev);
        const int it = meta ? hip_icp_match_metascan(meta_scans, cur, cfg, nr_pointPair) : hip_icp_match(prev, cur, cfg, nr_pointPair);
        if (on_matched) on_matched(i, it);
      }
      if (meta && i != n - 1) {          // push processed scan; only keep the last n scans as metascans
        meta_scans.push_back(cur);
        if (cfg.max_num_metascans > 0)
          while (meta_scans.size() > (size_t)cfg.max_num_metascans) meta_scans.erase(meta_scans.begin());
      }
    }
  } catch (...) {
    delete pool;
    throw;
  }
  delete pool;
}

#endif
