// C ABI of lib3dtk_hip.so (include/tdtk_hip.h): handles, workspaces, host orchestration of
// the kernels in kernels.hip.  There is NO CPU fallback in this library: without a HIP device
// every compute entry point fails with TDTK_EDEVICE.
#include <sched.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>

#include "kernels.h"

using namespace tdtk;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local std::string g_err;
void tdtk::set_error(const std::string& s) { g_err = s; }

#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                          \
      return TDTK_EDEVICE;                                                                   \
    }                                                                                        \
  } while (0)

static std::atomic<int> g_ctx_live{0};             // host threads that hold a context right now (all devices)
static std::atomic<uint64_t> g_respeculated{0};   // tree builds whose speculative cuts failed the final check

static double now_ms()
{
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------------------
// per-thread, per-device context: stream, events, growable workspaces
// ------------------------------------------------------------------------------------------
// the arrays of trees and resident scans and the per-context workspaces below come from the pool (pool.cpp)
static inline hipError_t handle_malloc(void** p, size_t bytes) { return (hipError_t)pool_malloc_raw(p, bytes); }

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) pool_free(p); }     // (pooled like the handles' arrays: the contexts of a prefetch pool's threads come and go)
  int ensure(size_t bytes)
  {
    if (bytes <= cap) return TDTK_OK;
    if (p) pool_free(p);
    p = nullptr; cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = (hipError_t)pool_malloc_raw(&p, want);
    if (e != hipSuccess) { set_error(std::string("hipMalloc: ") + hipGetErrorString(e)); return TDTK_ENOMEM; }
    cap = want;
    return TDTK_OK;
  }
  template <class T> T* as() { return static_cast<T*>(p); }
};

enum { WS_KPOS, WS_D2, WS_PART, WS_OUT, WS_OVF_M2, WS_OVF_REF, WS_IDX, WS_QX, WS_QY, WS_QZ, WS_DX,
       WS_DY, WS_DZ, WS_ORDER, WS_CELL, WS_HIST, WS_TMPA, WS_TMPB, WS_CNT, WS_BOX, WS_ARENA, WS_COST, WS_MOVES, WS_BOUNDS, WS_COUNT };

// an auxiliary stream with the buffers one whole-scan pass needs: batches of links over small scans run several
// passes side by side (one pass of an 80K-point scan occupies a fraction of the machine and is latency-bound)
// draw counters of the work-queue search kernel: two sets of 8 that alternate from launch to launch on one stream
// (each launch zeroes the set of the next one, kernels.hip)
struct QueueCtr {
  DevBuf buf;
  int parity = 0;
  int attach(SearchArgs& a)
  {
    if (!buf.p) {
      int rc = buf.ensure(64 * sizeof(uint32_t));
      if (rc) return rc;
      if (hipMemset(buf.p, 0, 64 * sizeof(uint32_t)) != hipSuccess) { set_error("hipMemset failed"); return TDTK_EDEVICE; }
    }
    a.q_ctr = buf.as<uint32_t>() + 32 * parity;
    a.q_ctr_next = buf.as<uint32_t>() + 32 * (parity ^ 1);
    parity ^= 1;
    return TDTK_OK;
  }
};

struct Lane {
  hipStream_t s = nullptr;
  bool owns = true;     // lane 0 runs on the context's own stream
  QueueCtr qc;
  DevBuf kpos, part, ovf_m2, ovf_ref;
  DevBuf moved;      // batched link passes: this link's own copy of a scan another link of the launch is moving (lazy moves)
  // batched link passes: whose hits kpos holds (handle numbers of the tree and the scan, queries) -- the next pass of the SAME
  // link at this position starts every search from its previous hit (SearchArgs::warm)
  uint64_t k_tree = 0, k_scan = 0; size_t k_n = 0;
  ~Lane() { if (s && owns) (void)hipStreamDestroy(s); }
};

// batched link passes: what each query of the link at this position of the launch order cost in the previous pass (one
// byte per query: the next pass's hand-out order), and which link that was
struct LinkCost {
  DevBuf cost;
  const void* tree = nullptr; const void* scan = nullptr; size_t n = 0;
};

struct Ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;   // around the search kernel of the last pass
  hipEvent_t e2 = nullptr, e3 = nullptr;   // around the pair-sum kernels behind it (k_accum + k_final, or k_final alone)
  hipEvent_t e4 = nullptr, e5 = nullptr;   // around k_ann_normals of the last calcNormals
  hipEvent_t e_user = nullptr;             // fence between a caller's stream and this context's stream
  hipEvent_t e_defer = nullptr;            // behind the last batch of scan moves that was left running (defer_fence)
  hipStream_t stream_b = nullptr, stream_c = nullptr, stream_d = nullptr;   // the tree build's background chains (exact centroid sums beside the levels below)
  hipEvent_t e_b1 = nullptr, e_b2 = nullptr, e_b3 = nullptr, e_b4 = nullptr;
  DevBuf ws[WS_COUNT];
  double* h_pin = nullptr;  // pinned staging for the per-iteration sums: words [0, ACC_TOTAL); behind them two slots of the tree build
  // (both are filled by copies enqueued on `stream` and read only behind a synchronisation of that stream that was made after
  //  the copy was enqueued: one of the build's looks at the device for the box, tree_finish's own hipStreamSynchronize for the
  //  groups -- device_build_tree may return with its last kernels still running, see build.hip "no_last_look")
  static constexpr int PIN_BOX = 128;      // 6 doubles: the root bounding box (tree_from_device_points)
  static constexpr int PIN_GROUPS = 140;   // 1 uint32: groups of the padded layout (tree_pad_buckets -> tree_finish)
  void* h_build = nullptr;  // 64 KB, pinned: the tree build's looks at the device (BuildSide::h_pin)
  void* h_stage = nullptr;  // pinned staging for descriptor tables of batched launches (grows on demand)
  size_t h_stage_cap = 0;
  DevBuf d_mask, d_skip;                // -R passes: the keep-mask (bits, caller order) and what the search reads (bytes, sorted order)
  std::vector<unsigned char> h_mask;
  DevBuf d_loop;                        // lab, the host-free ICP loop: its IcpLoopDev block (kernels.h)
  double* h_loop = nullptr;             // ... and its record, pinned: ICP_LOOP_RING rows of ICP_ROW doubles
  DevBuf d_hash;                        // tdtk_icp_index_hashes: one 64-bit word per iteration of the last tdtk_icp_match
  std::vector<uint64_t> last_hashes;
  void* h_moves = nullptr;  // pinned staging of scans_settle's table (its own: a settle may precede a batched launch in one call)
  size_t h_moves_cap = 0;
  hipEvent_t e_moves = nullptr;   // behind the last copy out of h_moves
  bool moves_inflight = false;
  double last_nn_ms = 0.0, last_sums_ms = 0.0, last_normals_ms = 0.0, last_build_ms = 0.0;
  bool ev_pending = false, ev2_pending = false, ev4_pending = false;
  uint64_t counted_ann_queries = 0;
  // tdtk_visit_counting: every search of this thread runs its instrumented instantiation and adds to d_counters
  bool counting = false;
  // tdtk_visit_counting(device, 2): count the REFERENCE's walk -- every search while counting starts cold (no warm start, no
  // deferred quick check), i.e. kdTreeImpl.h:345-383 with radius maxdist2; results are the same, so a loop stays on its path
  bool count_cold = false;
  DevBuf d_counters;
  uint64_t counted_queries = 0;
  std::vector<std::unique_ptr<Lane>> lanes;
  std::vector<std::unique_ptr<Lane>> slots;   // per-link buffers of a several-links-in-one-launch batch (no streams)
  std::vector<std::unique_ptr<LinkCost>> link_costs;
  DevBuf multi_args;                          // its argument tables on the device
  std::vector<void*> free_later;              // see pool_free_later
  QueueCtr qc;       // for launches on `stream` (a caller's stream gets its launches ordered behind it, see run_search)
  // slabs of equal cost for the next pass of an ICP loop (launch_slab_bounds): valid for the loop's next scan_pass only
  const uint32_t* next_bounds = nullptr;
  size_t next_bounds_n = 0;
  // a context dies with its host thread (worker threads of a prefetch pool come and go): give everything back
  ~Ctx();
};

// Batched scan moves (the pose update of a graph-SLAM round: every resident scan of the rank, ~0.6 ms for 63 x 1M
// points) are left running when the call returns; whatever the host does next -- Python marshalling, building the
// next round's graph -- overlaps with them.  The fence is process-wide: the next library call of ANY host thread on
// that device waits for it in get_ctx before it touches a scan, so "the scans have moved when the call has returned"
// still holds for everything that can observe them.
struct Deferred { int device; hipEvent_t ev; Ctx* owner; };
static std::mutex g_defer_mu;
static std::atomic<int> g_defer_n{0};
static std::vector<Deferred> g_defer;

static void wait_deferred(int device, const Ctx* only_owner = nullptr)
{
  if (g_defer_n.load(std::memory_order_acquire) == 0) return;
  std::lock_guard<std::mutex> lk(g_defer_mu);
  for (size_t i = 0; i < g_defer.size();) {
    if (g_defer[i].device == device && (!only_owner || g_defer[i].owner == only_owner)) {
      (void)hipEventSynchronize(g_defer[i].ev);
      g_defer.erase(g_defer.begin() + (long)i);
    } else {
      ++i;
    }
  }
  g_defer_n.store((int)g_defer.size(), std::memory_order_release);
}

Ctx::~Ctx()
  {
    g_ctx_live.fetch_sub(1);
    if (device >= 0) (void)hipSetDevice(device);
    wait_deferred(device, this);
    if (e_defer) (void)hipEventDestroy(e_defer);
    for (void* q : free_later) pool_free(q);
    free_later.clear();
    if (h_stage) (void)hipHostFree(h_stage);
    if (h_moves) (void)hipHostFree(h_moves);
    if (e_moves) (void)hipEventDestroy(e_moves);
    lanes.clear();
    slots.clear();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e2) (void)hipEventDestroy(e2);
    if (e3) (void)hipEventDestroy(e3);
    if (e4) (void)hipEventDestroy(e4);
    if (e5) (void)hipEventDestroy(e5);
    if (e_user) (void)hipEventDestroy(e_user);
    if (e_b1) (void)hipEventDestroy(e_b1);
    if (e_b2) (void)hipEventDestroy(e_b2);
    if (e_b3) (void)hipEventDestroy(e_b3);
    if (e_b4) (void)hipEventDestroy(e_b4);
    if (stream_b) (void)hipStreamDestroy(stream_b);
    if (stream_c) (void)hipStreamDestroy(stream_c);
    if (stream_d) (void)hipStreamDestroy(stream_d);
    if (h_pin) (void)hipHostFree(h_pin);
    if (h_loop) (void)hipHostFree(h_loop);     // (lab)
    if (h_build) (void)hipHostFree(h_build);
    if (stream) (void)hipStreamDestroy(stream);
  }

static thread_local std::map<int, std::unique_ptr<Ctx>> g_ctx;

static int get_ctx(int device, Ctx** out, bool touches_scans = true)
{
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_error("no HIP device available (lib3dtk_hip has no CPU fallback)");
    return TDTK_EDEVICE;
  }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return TDTK_EINVAL; }
  HIPCHK(hipSetDevice(device));
  auto it = g_ctx.find(device);
  if (it == g_ctx.end()) {
    std::unique_ptr<Ctx> c(new Ctx);
    c->device = device;
    HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreate(&c->e0));
    HIPCHK(hipEventCreate(&c->e1));
    HIPCHK(hipEventCreate(&c->e2));
    HIPCHK(hipEventCreate(&c->e3));
    HIPCHK(hipEventCreate(&c->e4));
    HIPCHK(hipEventCreate(&c->e5));
    HIPCHK(hipEventCreateWithFlags(&c->e_user, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->e_defer, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->e_moves, hipEventDisableTiming));
    // (coherent, said explicitly: the host reads these words while the kernel that writes them is still running -- await_sums)
    HIPCHK(hipHostMalloc((void**)&c->h_pin, sizeof(double) * 256, hipHostMallocCoherent));
    HIPCHK(hipHostMalloc(&c->h_build, 65536, hipHostMallocDefault));
    it = g_ctx.emplace(device, std::move(c)).first;
    g_ctx_live.fetch_add(1);
  }
  *out = it->second.get();
  if (touches_scans) wait_deferred(device);    // (the timing / counter read-outs do not: they must not end the overlap)
  return TDTK_OK;
}

namespace tdtk {
int ctx_stream(int device, void** stream_out)
{
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  *stream_out = c->stream;
  return TDTK_OK;
}
}  // namespace tdtk

// leave what has been enqueued on c->stream running (see Deferred); TDTK_SYNC_MOVES=1 waits as before
static int defer_fence(Ctx* c)
{
  static const bool sync_moves = [] { const char* e = getenv("TDTK_SYNC_MOVES"); return e && e[0] == '1'; }();
  if (sync_moves) { HIPCHK(hipStreamSynchronize(c->stream)); return TDTK_OK; }
  // the event is re-recorded under the lock: another host thread may be inside hipEventSynchronize on this very event
  // (wait_deferred holds the lock while it waits), and re-recording an event somebody is waiting on is undefined
  std::lock_guard<std::mutex> lk(g_defer_mu);
  HIPCHK(hipEventRecord(c->e_defer, c->stream));
  bool have = false;
  for (const Deferred& d : g_defer) have = have || d.owner == c;
  if (!have) g_defer.push_back({c->device, c->e_defer, c});
  g_defer_n.store((int)g_defer.size(), std::memory_order_release);
  return TDTK_OK;
}

// blocks an enqueued kernel still reads: given back behind the next synchronisation of the context's stream
static void pool_free_later(Ctx* c, void* p) { if (p) c->free_later.push_back(p); }
static void flush_free_later(Ctx* c)
{
  for (void* p : c->free_later) pool_free(p);
  c->free_later.clear();
}

// pinned host staging that stays valid until the next library call on this thread (get_ctx has then waited for the
// copy that reads it)
static int stage_reserve(Ctx* c, size_t bytes)
{
  if (c->h_stage_cap < bytes) {
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_cap = 0;
    const size_t want = std::max<size_t>(bytes, 64 * 1024);
    if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return TDTK_ENOMEM; }
    c->h_stage_cap = want;
  }
  return TDTK_OK;
}
static int stage_pinned(Ctx* c, const void* src, size_t bytes, void** out)
{
  if (c->h_stage_cap < bytes) {
    if (c->h_stage) (void)hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_stage_cap = 0;
    const size_t want = std::max<size_t>(bytes, 64 * 1024);
    if (hipHostMalloc(&c->h_stage, want, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return TDTK_ENOMEM; }
    c->h_stage_cap = want;
  }
  std::memcpy(c->h_stage, src, bytes);
  *out = c->h_stage;
  return TDTK_OK;
}

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
// every tree / scan handle of the process has a number of its own: what "the same tree, the same scan as last time" is tested
// with where a stale answer would be an out-of-range read (a freed handle's address can come back)
static std::atomic<uint64_t> g_handle_uid{1};

struct tdtk_tree {
  const uint64_t uid = g_handle_uid.fetch_add(1, std::memory_order_relaxed);
  int device = 0;
  size_t M = 0;
  int bucket = 0;
  TreeDev dev{};
  void *d_nodes = nullptr, *d_pts = nullptr, *d_leaf = nullptr, *d_r = nullptr, *d_hot = nullptr, *d_grp = nullptr, *d_fat = nullptr;
  void* d_q16 = nullptr;     // 16-bit shadow of the padded buckets (TreeDev::q16), 6 bytes per slot + 128 of slack
  void* d_split = nullptr;   // { splitval, children } of every internal node, 16 bytes (TreeDev::split): inside d_hot's allocation
  double q_lo[3] = {0, 0, 0}, q_scale = 0.0;
  size_t Mp = 0;   // slots of d_pts: M, or 4 * groups once the buckets are padded to whole groups (tree_pad_buckets)
  double bbmin[3], bbmax[3], centre[3];
  tdtk_tree_info info{};
  tdtk_tree() = default;
  tdtk_tree(const tdtk_tree&) = delete;
  tdtk_tree& operator=(const tdtk_tree&) = delete;
  ~tdtk_tree()   // also the error paths of tdtk_tree_create: nothing stays allocated on the device
  {
    (void)hipSetDevice(device);
    void* p[] = {d_nodes, d_pts, d_leaf, d_r, d_hot, d_grp, d_fat, d_q16};
    for (void* q : p)
      if (q) pool_free(q);
  }
};

struct tdtk_scan {
  const uint64_t uid = g_handle_uid.fetch_add(1, std::memory_order_relaxed);
  int device = 0;
  size_t N = 0;
  double *x = nullptr, *y = nullptr, *z = nullptr, *nx = nullptr, *ny = nullptr, *nz = nullptr;
  int32_t* d_order = nullptr;    // sorted position -> caller index
  // "xyz reduced original" (basicScan.cc:739-757 copyReducedToOriginal): once tdtk_scan_mark_original has been
  // called, the first operation that moves the points first saves them here (a device-to-device copy), so the
  // scan's search tree can still be built later without the points ever visiting the host
  bool track_original = false;
  double *ox = nullptr, *oy = nullptr, *oz = nullptr;
  // Lazy moves.  The pose update of a graph-SLAM round does not touch the points: it queues its in-place transforms here
  // (oldest first), and whoever reads the scan next applies them -- the link passes of the next round in registers where a
  // lane takes a query (the link that owns the update stores the result into the spare arrays ax / ay / az, swapped in
  // behind the launch), every other entry point through scan_settle (one pass, all queued matrices in order).  A rank
  // never moves a scan none of its links reads.  The arithmetic is the one Scan::transform does point by point
  // (scan.cc:851-875), matrix after matrix: same bits as moving the scan every time.
  // Several host threads may hold the same scan (a prefetch pool, an OpenMP host): the queue, the spare arrays and the
  // swap are only touched under g_moves_mu; npend mirrors pending.size() so that the readers' fast path ("nothing
  // queued") takes no lock.  A settle issued while more than one context is live waits for its kernel before it
  // publishes npend == 0, so a reader on ANOTHER stream that finds nothing queued also finds the points moved.
  mutable std::vector<Mat4> pending;
  mutable std::atomic<uint32_t> npend{0};
  mutable double *ax = nullptr, *ay = nullptr, *az = nullptr;
  tdtk_scan() = default;
  tdtk_scan(const tdtk_scan&) = delete;
  tdtk_scan& operator=(const tdtk_scan&) = delete;
  ~tdtk_scan()
  {
    (void)hipSetDevice(device);
    double* p[] = {x, y, z, nx, ny, nz, ox, oy, oz, ax, ay, az};
    for (double* q : p)
      if (q) pool_free(q);
    if (d_order) pool_free(d_order);
  }
};

// ------------------------------------------------------------------------------------------
extern "C" {

const char* tdtk_last_error(void) { return g_err.c_str(); }
const char* tdtk_version(void) { return "3dtk_amd 0.1 (gfx950)"; }

size_t tdtk_pool_trim(void) { return pool_trim(); }
uint64_t tdtk_build_respeculated(void) { return g_respeculated.load(); }

int tdtk_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// copy-on-first-write of a tracked scan's original points; called by everything that moves a resident scan
static int scan_keep_original(Ctx* c, tdtk_scan* s)
{
  if (!s || !s->track_original || s->ox || s->N == 0) return TDTK_OK;
  const size_t b = s->N * sizeof(double);
  {   // all three or none (a partial set would pass the `s->ox` test above next time)
    void* p[3] = {nullptr, nullptr, nullptr};
    for (int k = 0; k < 3; k++)
      if (handle_malloc(&p[k], b) != hipSuccess) {
        for (int j = 0; j < k; j++) pool_free(p[j]);
        set_error("out of device memory (saved original of a scan)");
        return TDTK_ENOMEM;
      }
    s->ox = static_cast<double*>(p[0]); s->oy = static_cast<double*>(p[1]); s->oz = static_cast<double*>(p[2]);
  }
  HIPCHK(hipMemcpyAsync(s->ox, s->x, b, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(s->oy, s->y, b, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(s->oz, s->z, b, hipMemcpyDeviceToDevice, c->stream));
  return TDTK_OK;
}

// ---- lazy scan moves (see tdtk_scan::pending) ---------------------------------------------------
static std::recursive_mutex g_moves_mu;      // guards every scan's pending / npend / ax..az and the x <-> ax swap
static bool lazy_moves()
{
  const char* e = getenv("TDTK_LAZY_MOVES");     // 0: every queued move is carried out at once (the round-3 behaviour)
  return !(e && e[0] == '0');
}
// longest chain a scan may carry: a rank that never reads a scan (seven of eight ranks, for most scans) carries it out
// once per this many queued transforms -- one trip of the points through HBM per 16 rounds instead of one per round
constexpr size_t LAZY_CHAIN_MAX = 32;

// carry out what is queued on these scans: one launch, every scan's chain in order.  Enqueued on c->stream; the caller
// decides whether to wait (the entry points that go on to read the scan on the same stream need not).
static int scans_settle(Ctx* c, const tdtk_scan* const* scans, int count)
{
  {   // fast path without the lock: nothing queued on any of them
    bool any = false;
    for (int i = 0; i < count && !any; i++) any = scans[i] && scans[i]->npend.load(std::memory_order_acquire) != 0;
    if (!any) return TDTK_OK;
  }
  std::lock_guard<std::recursive_mutex> lk(g_moves_mu);
  size_t nmat = 0, max_n = 0;
  int nd = 0;
  for (int i = 0; i < count; i++) {
    const tdtk_scan* sc = scans[i];
    if (!sc || sc->pending.empty()) continue;
    bool dup = false;
    for (int j = 0; j < i && !dup; j++) dup = scans[j] == sc;
    if (dup) continue;
    if (!sc->N) { sc->pending.clear(); sc->npend.store(0, std::memory_order_release); continue; }
    if (sc->device != c->device) { set_error("resident scans of one call must live on one device"); return TDTK_EINVAL; }
    nmat += sc->pending.size(); nd++;
    max_n = std::max(max_n, sc->N);
  }
  if (!nd) {
    for (int i = 0; i < count; i++)
      if (scans[i] && scans[i]->pending.empty()) scans[i]->npend.store(0, std::memory_order_release);
    return TDTK_OK;
  }
  const size_t o_mat = ((sizeof(XfChainDesc) * (size_t)nd + 127) / 128) * 128, bytes = o_mat + nmat * sizeof(Mat4);
  int rc = c->ws[WS_MOVES].ensure(bytes);
  if (rc) return rc;
  if (c->moves_inflight) { HIPCHK(hipEventSynchronize(c->e_moves)); c->moves_inflight = false; }
  if (c->h_moves_cap < bytes) {
    if (c->h_moves) (void)hipHostFree(c->h_moves);
    c->h_moves = nullptr; c->h_moves_cap = 0;
    const size_t want = std::max<size_t>(bytes + bytes / 2, 64 * 1024);
    if (hipHostMalloc(&c->h_moves, want, hipHostMallocDefault) != hipSuccess) { set_error("hipHostMalloc failed"); return TDTK_ENOMEM; }
    c->h_moves_cap = want;
  }
  char* tab = static_cast<char*>(c->h_moves);
  std::memset(tab, 0, o_mat);
  XfChainDesc* hd = reinterpret_cast<XfChainDesc*>(tab);
  Mat4* hm = reinterpret_cast<Mat4*>(tab + o_mat);
  const Mat4* dm = reinterpret_cast<const Mat4*>(static_cast<char*>(c->ws[WS_MOVES].p) + o_mat);
  size_t k = 0;
  std::vector<const tdtk_scan*> moved;      // (each scan once, however often the caller's list names it)
  for (int i = 0; i < count; i++) {
    const tdtk_scan* sc = scans[i];
    if (!sc || sc->pending.empty() || std::find(moved.begin(), moved.end(), sc) != moved.end()) continue;
    XfChainDesc& e = hd[moved.size()];
    e.x = sc->x; e.y = sc->y; e.z = sc->z; e.nx = sc->nx; e.ny = sc->ny; e.nz = sc->nz; e.n = sc->N;
    e.mats = dm + k; e.nm = (int)sc->pending.size();
    for (const Mat4& m : sc->pending) hm[k++] = m;
    moved.push_back(sc);
  }
  // The queues are emptied -- chain and count together -- only once the chain kernel is on the stream: a copy, an event or a
  // launch that fails on the way returns with every move still queued (round-5 advice: they used to be lost).
  HIPCHK(hipMemcpyAsync(c->ws[WS_MOVES].p, tab, bytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipEventRecord(c->e_moves, c->stream));
  c->moves_inflight = true;
  HIPCHK(launch_transform_chain_batch(reinterpret_cast<const XfChainDesc*>(c->ws[WS_MOVES].p), (int)moved.size(), max_n, c->stream));
  for (const tdtk_scan* sc : moved) sc->pending.clear();
  // other contexts (host threads with streams of their own) may read these scans next: they must not find "nothing
  // queued" before the chain kernel has run.  A lone context orders everything on its one stream and need not wait.
  if (g_ctx_live.load() > 1) HIPCHK(hipStreamSynchronize(c->stream));
  for (int i = 0; i < count; i++)
    if (scans[i] && scans[i]->pending.empty()) scans[i]->npend.store(0, std::memory_order_release);
  return TDTK_OK;
}
static int scan_settle(Ctx* c, const tdtk_scan* s)
{
  if (!s || s->npend.load(std::memory_order_acquire) == 0) return TDTK_OK;
  return scans_settle(c, &s, 1);
}
// the spare coordinate arrays of a scan (tdtk_scan::ax / ay / az): all three or none -- a launch stores through all of
// them and swaps them in, so a partial set (one allocation of the three failed) must never be left on the handle
static int scan_ensure_spare(const tdtk_scan* sc)
{
  if (sc->ax && sc->ay && sc->az) return TDTK_OK;
  const size_t b = sc->N * sizeof(double);
  void* p[3] = {nullptr, nullptr, nullptr};
  for (int k = 0; k < 3; k++) {
    if (handle_malloc(&p[k], b) != hipSuccess) {
      for (int j = 0; j < k; j++) pool_free(p[j]);
      set_error("out of device memory (spare arrays of a moving scan)");
      return TDTK_ENOMEM;
    }
  }
  double* old[3] = {sc->ax, sc->ay, sc->az};
  for (double* q : old)
    if (q) pool_free(q);
  sc->ax = static_cast<double*>(p[0]); sc->ay = static_cast<double*>(p[1]); sc->az = static_cast<double*>(p[2]);
  return TDTK_OK;
}
// queue one in-place transform on a resident scan (the caller has saved "xyz reduced original" if it is tracked)
static void scan_queue_move(tdtk_scan* s, const double* A16)
{
  Mat4 m;
  std::memcpy(m.m, A16, sizeof m.m);
  std::lock_guard<std::recursive_mutex> lk(g_moves_mu);
  s->pending.push_back(m);
  s->npend.store((uint32_t)s->pending.size(), std::memory_order_release);
}

// ---- tree ------------------------------------------------------------------------------
// device construction (build.hip) over the [M][3] points already sitting in c->ws[WS_TMPA]
static int tree_from_device_points(Ctx* c, tdtk_tree* t, size_t M, int bucket_size, double t0)
{
  int rc;
  if ((rc = c->ws[WS_BOX].ensure(bbox_temp_bytes() + 8 * sizeof(double)))) return rc;
  double* d_box = c->ws[WS_BOX].as<double>();
  // root bounding box (binning of unsorted query batches, accumulation shift): min / max on the device
  // (read back behind the build: the build's own looks at the device are the next synchronisation points)
  HIPCHK(launch_bbox(c->ws[WS_TMPA].as<double>(), M, d_box + 8, d_box, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_pin + Ctx::PIN_BOX, d_box, 6 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  const double t1 = now_ms();
  t->info.upload_ms = t1 - t0;
  const bool alone = g_ctx_live.load() <= 2;
  if ((rc = c->ws[WS_ARENA].ensure(device_build_arena_bytes(M, alone)))) return rc;
  // The background chain of the build needs a stream of its own, and the runtime has four hardware queues for all the
  // streams of the process: when several host threads are at work -- a doICP that prepares
  // three scans ahead -- a sixth and seventh stream end up queued behind other threads' kernels, the root's chain (one
  // wave, 1.2 ms) in front of somebody's search, and ten 1M-point scans take 43.5 ms instead of 36.4.  So: beside at
  // most one other thread.
  if (alone && !c->stream_b) {
    // made when first needed: every stream of the process takes a share of the four hardware queues, used or not, and
    // the worker threads of a prefetch pool never build alone
    HIPCHK(hipStreamCreateWithFlags(&c->stream_b, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream_c, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->stream_d, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->e_b1, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->e_b2, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->e_b3, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&c->e_b4, hipEventDisableTiming));
  }
  const bool four = [] { const char* e = lab_env("TDTK_BUILD_STREAMS"); return !(e && e[0] == '3'); }();   // (lab: TDTK_BUILD_STREAMS=3: round 5's two side streams)
  const BuildSide side = {alone ? c->stream_b : nullptr, alone ? c->stream_c : nullptr, c->e_b1, c->e_b2, c->e_b3, c->h_build,
                          (alone && four) ? c->stream_d : nullptr, c->e_b4};
  DevBuildResult r = device_build_tree(c->ws[WS_TMPA].as<double>(), M, bucket_size, c->ws[WS_ARENA].p, c->stream, &side);
  if (r.respeculated) g_respeculated.fetch_add(1);
  if (r.err != hipSuccess) {
    set_error(r.degenerate ? std::string("degenerate split (non-finite coordinates?)")
                           : std::string("device tree build: ") + hipGetErrorString(r.err));
    return r.degenerate ? TDTK_EINVAL : TDTK_EDEVICE;
  }
  for (int a = 0; a < 3; a++) { t->bbmin[a] = c->h_pin[Ctx::PIN_BOX + a]; t->bbmax[a] = c->h_pin[Ctx::PIN_BOX + 3 + a]; }   // (the copy was enqueued in front of the build, and every path through device_build_tree synchronises c->stream at least once behind it: a look at the level loop's counters, or its last one)
  t->d_nodes = r.nodes; t->d_r = r.node_r; t->d_pts = r.pts;
  t->d_leaf = r.leaf_tab;   // non-null only in table mode
  t->dev.root_ref = r.root_ref;
  t->dev.cb = (uint32_t)r.cb;
  t->info.n_internal = r.n_internal; t->info.n_leaves = r.n_leaves;
  t->info.max_depth = r.max_depth; t->info.max_leaf_points = r.max_leaf;
  t->info.build_ms = now_ms() - t1;
  c->last_build_ms = t->info.build_ms;
  return TDTK_OK;
}

// Pad every bucket to whole groups of four slots and build the fp32 shadow groups the big-batch search filters buckets
// with (kernels.hip, "bucket groups").  The build's scratch arena is free again at this point and holds the two counter
// arrays and the scan's temporary.  Skipped (the tree stays as built, the search scans buckets in fp64 only) when the
// tree is a single bucket, when the padded positions would not fit the reference format, or with TDTK_BUCKET_GROUPS=0.
static int tree_pad_buckets(Ctx* c, tdtk_tree* t, size_t M)
{
  t->Mp = M;
  static const bool off = [] { const char* e = getenv("TDTK_BUCKET_GROUPS"); return e && e[0] == '0'; }();
  if (off || t->info.n_internal == 0) return TDTK_OK;
  const size_t n1 = M + 1;
  const size_t tmpb = scan_u32_temp_bytes(n1);
  const size_t need = 2 * n1 * sizeof(uint32_t) + tmpb + 256;
  int rc;
  if ((rc = c->ws[WS_ARENA].ensure(need))) return rc;
  uint32_t* ng_at = c->ws[WS_ARENA].as<uint32_t>();
  uint32_t* g_at = ng_at + n1;
  void* tmp = (void*)(((uintptr_t)(g_at + n1) + 127) & ~(uintptr_t)127);
  const uint32_t cb = t->dev.cb, cmask = (cb >= 32) ? 0xFFFFFFFFu : ((1u << cb) - 1u);
  KdNode* nodes = static_cast<KdNode*>(t->d_nodes);
  LeafEntry* leaf = static_cast<LeafEntry*>(t->d_leaf);
  HIPCHK(launch_pad_mark(nodes, t->info.n_internal, leaf, cb, cmask, ng_at, M, c->stream));
  HIPCHK(launch_scan_u32(ng_at, g_at, n1, tmp, tmpb, c->stream));
  // The number of groups is known on the device; every bucket is padded by at most three slots, so (M + 3 leaves) / 4 groups
  // are enough room.  When that bound passes the format checks below, the fill is enqueued right away and the exact count is
  // read with it -- one look at the device less (a small scan's tree is a few dozen microseconds of launches per look).
  // the 16-bit grid over the root box (TreeDev::q16): one cell size for the three axes, so that distances stay isotropic
  bool want_q16 = false;
  {
    static const bool q_off = [] { const char* e = getenv("TDTK_BUCKET_Q16"); return e && e[0] == '0'; }();
    double ext = 0.0;
    for (int a = 0; a < 3; a++) ext = std::max(ext, t->bbmax[a] - t->bbmin[a]);
    const double sc = 65535.0 / ext;
    if (!q_off && ext > 0.0 && std::isfinite(ext) && std::isfinite(sc) && sc > 0.0) {
      want_q16 = true;
      for (int a = 0; a < 3; a++) t->q_lo[a] = t->bbmin[a];
      t->q_scale = sc;
    }
  }
  uint32_t G = 0;
  const uint64_t G_bound = ((uint64_t)M + 3ull * t->info.n_leaves + 3ull) / 4ull;
  const bool bound_ok = (4ull * G_bound) * sizeof(KdPoint) < (1ull << 32) && (leaf || ((4ull * G_bound) << cb) <= (uint64_t)REF_VAL);
  if (bound_ok) {
    void *ptsB = nullptr, *grpB = nullptr;
    if (handle_malloc(&ptsB, 4ull * G_bound * sizeof(KdPoint)) == hipSuccess && handle_malloc(&grpB, (size_t)G_bound * 48) == hipSuccess) {
      void* q16B = nullptr;      // (no room for it: the fp32 groups alone)
      if (want_q16 && handle_malloc(&q16B, (size_t)G_bound * 24 + 128) != hipSuccess) { (void)hipGetLastError(); q16B = nullptr; }
      hipError_t e = launch_pad_fill(nodes, t->info.n_internal, leaf, cb, cmask, g_at, static_cast<const KdPoint*>(t->d_pts),
                                     static_cast<KdPoint*>(ptsB), static_cast<float4*>(grpB), c->stream, static_cast<uint32_t*>(q16B), t->q_lo, t->q_scale);
      if (e == hipSuccess) e = hipMemcpyAsync(c->h_pin + Ctx::PIN_GROUPS, g_at + M, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
      if (e != hipSuccess) { pool_free(ptsB); pool_free(grpB); if (q16B) pool_free(q16B); set_error(std::string("bucket groups: ") + hipGetErrorString(e)); return TDTK_EDEVICE; }
      pool_free_later(c, t->d_pts);
      t->d_pts = ptsB; t->d_grp = grpB; t->d_q16 = q16B;
      t->Mp = 0;                 // = 4 G, read in tree_finish behind its synchronisation
      return TDTK_OK;
    }
    (void)hipGetLastError();
    if (ptsB) pool_free(ptsB);
  }
  HIPCHK(hipMemcpyAsync(&G, g_at + M, sizeof G, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint64_t slots = 4ull * G;
  // the padded starts must fit where the starts fitted: 30-bit packed references (start << cb | count), 32-bit byte
  // offsets into the point and group arrays, int32 starts of the leaf table
  if (G == 0 || slots * sizeof(KdPoint) >= (1ull << 32) || (!leaf && (slots << cb) > (uint64_t)REF_VAL)) return TDTK_OK;
  // No room for the padded copy (a transient peak of twice the point array + 12 bytes per slot): the tree as it stands is
  // complete and searchable -- nothing has been rewritten yet --, so padding is skipped and the fp64-only bucket scan used.
  void *ptsP = nullptr, *grp = nullptr;
  if (handle_malloc(&ptsP, slots * sizeof(KdPoint)) != hipSuccess) { (void)hipGetLastError(); return TDTK_OK; }
  if (handle_malloc(&grp, (size_t)G * 48) != hipSuccess) { (void)hipGetLastError(); pool_free(ptsP); return TDTK_OK; }
  void* q16 = nullptr;
  if (want_q16 && handle_malloc(&q16, (size_t)G * 24 + 128) != hipSuccess) { (void)hipGetLastError(); q16 = nullptr; }
  hipError_t e = launch_pad_fill(nodes, t->info.n_internal, leaf, cb, cmask, g_at, static_cast<const KdPoint*>(t->d_pts),
                                 static_cast<KdPoint*>(ptsP), static_cast<float4*>(grp), c->stream, static_cast<uint32_t*>(q16), t->q_lo, t->q_scale);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e != hipSuccess) { pool_free(ptsP); pool_free(grp); if (q16) pool_free(q16); set_error(std::string("bucket groups: ") + hipGetErrorString(e)); return TDTK_EDEVICE; }
  pool_free(t->d_pts);
  t->d_pts = ptsP; t->d_grp = grp; t->d_q16 = q16; t->Mp = (size_t)slots;
  return TDTK_OK;
}

static int tree_finish(Ctx* c, tdtk_tree* t, size_t M)
{
  int prc = tree_pad_buckets(c, t, M);
  if (prc) return prc;
  for (int a = 0; a < 3; a++) t->centre[a] = 0.5 * (t->bbmin[a] + t->bbmax[a]);
  // the compact hot records (fp32 box + split value + children) the big-batch search kernel walks
  double am = 0.0;
  for (int a = 0; a < 3; a++) am = std::max(am, std::max(std::fabs(t->bbmin[a]), std::fabs(t->bbmax[a])));
  t->dev.absmax = (float)std::min(am * 1.0000002, 3.0e38);
  if (t->info.n_internal) {
    // (+ the split halves on their own behind them, 16 bytes per node, for the visits that defer the quick check: ONE allocation,
    // so that a lane picks between the two by an offset from the same base)
    HIPCHK(handle_malloc(&t->d_hot, t->info.n_internal * (sizeof(KdHot) + sizeof(double2))));
    t->d_split = static_cast<char*>(t->d_hot) + t->info.n_internal * sizeof(KdHot);
    HIPCHK(launch_make_hot(static_cast<const KdNode*>(t->d_nodes), t->info.n_internal, static_cast<KdHot*>(t->d_hot), c->stream,
                           static_cast<double2*>(t->d_split)));
#ifdef TDTK_LAB
    // ... and, on request only, the two-level records (a node with its children's hot parts): two tree levels per round
    // trip are a measured negative both for the persistent-lane kernel (TDTK_FAT_NODES=1) and for the lane-group kernels of
    // small batches (TDTK_FAT_SMALL=1); kernels.hip has the numbers
    static const bool want_fat = [] {
      const char *a = lab_env("TDTK_FAT_NODES"), *b = lab_env("TDTK_FAT_SMALL");
      return (a && a[0] == '1') || (b && b[0] == '1');
    }();
    if (want_fat) {
      HIPCHK(handle_malloc(&t->d_fat, t->info.n_internal * sizeof(KdFat)));
      HIPCHK(launch_make_fat(static_cast<const KdNode*>(t->d_nodes), t->info.n_internal, static_cast<KdFat*>(t->d_fat), c->stream));
    }
#endif
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (t->d_grp && t->Mp == 0) {        // the padded layout was filled without a look of its own: its size now
    if (!t->info.n_internal) HIPCHK(hipStreamSynchronize(c->stream));
    uint32_t G = 0;
    std::memcpy(&G, c->h_pin + Ctx::PIN_GROUPS, sizeof G);
    t->Mp = 4ull * G;
  }
  flush_free_later(c);
  t->dev.hot = static_cast<const KdHot*>(t->d_hot);
  t->dev.n_hot = (uint32_t)t->info.n_internal;
  t->dev.n_slots = (uint32_t)std::min<size_t>(t->Mp, 0xFFFFFFFFu);
  t->dev.fat = static_cast<const KdFat*>(t->d_fat);
  t->dev.nodes = static_cast<const KdNode*>(t->d_nodes);
  t->dev.pts = static_cast<const KdPoint*>(t->d_pts);
  t->dev.grp = static_cast<const float4*>(t->d_grp);
  t->dev.q16 = static_cast<const uint32_t*>(t->d_q16);
  t->dev.split = static_cast<const double2*>(t->d_split);
  for (int a = 0; a < 3; a++) t->dev.q_lo[a] = t->q_lo[a];
  t->dev.q_scale = t->q_scale;
  t->dev.leaf_tab = static_cast<const LeafEntry*>(t->d_leaf);
  t->dev.node_r = static_cast<const double*>(t->d_r);
  t->dev.cmask = (t->dev.cb >= 32) ? 0xFFFFFFFFu : ((1u << t->dev.cb) - 1u);
  t->info.n_points = M;
  t->info.device_bytes = t->info.n_internal * (sizeof(KdNode) + sizeof(KdHot) + sizeof(double2) + (t->d_fat ? sizeof(KdFat) : 0) + sizeof(double)) + t->Mp * sizeof(KdPoint) +
                         (t->d_grp ? t->Mp / 4 * 48 : 0) + (t->d_q16 ? t->Mp / 4 * 24 + 128 : 0) +
                         (t->d_leaf ? t->info.n_leaves * sizeof(LeafEntry) : 0);
  return TDTK_OK;
}

static int tree_check_args(size_t M, int bucket_size)
{
  if (bucket_size < 1) { set_error("bucket size must be >= 1"); return TDTK_EINVAL; }
  if (M > (size_t)REF_VAL || M * sizeof(KdPoint) >= (1ull << 32)) {
    set_error("model scan too large (30-bit references / 32-bit byte offsets: < 2^27 points)");
    return TDTK_EINVAL;
  }
  return TDTK_OK;
}

int tdtk_tree_create(const double* xyz, size_t M, int bucket_size, int device, tdtk_tree** out)
{
  if (!out) { set_error("out is NULL"); return TDTK_EINVAL; }
  *out = nullptr;
  if (!xyz || M == 0) { set_error("cannot create kdtree with zero points"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;

  const double t0 = now_ms();
  std::unique_ptr<tdtk_tree> t(new tdtk_tree);
  t->device = device; t->M = M; t->bucket = bucket_size;
  if ((rc = tree_check_args(M, bucket_size))) return rc;
  // device construction (build.hip): upload the points once, build level by level.  (The host builder, kd_build.cpp,
  // is reachable through tdtk_tree_verify only -- it is the cross-check of this path, not an alternative to it.)
  if ((rc = c->ws[WS_TMPA].ensure(3 * M * sizeof(double)))) return rc;
  HIPCHK(hipMemcpyAsync(c->ws[WS_TMPA].p, xyz, 3 * M * sizeof(double), hipMemcpyHostToDevice, c->stream));
  if ((rc = tree_from_device_points(c, t.get(), M, bucket_size, t0))) return rc;
  if ((rc = tree_finish(c, t.get(), M))) return rc;
  *out = t.release();
  return TDTK_OK;
}

// KDtree over the points of a resident scan as they are now, in the caller's order -- what BasicScan builds
// over "xyz reduced original" (basicScan.cc:702-728) when it is called before the scan has been moved: no trip
// of the points through the host.
int tdtk_tree_create_from_scan(const tdtk_scan* scan, int bucket_size, tdtk_tree** out)
{
  if (!out) { set_error("out is NULL"); return TDTK_EINVAL; }
  *out = nullptr;
  if (!scan || scan->N == 0) { set_error("cannot create kdtree with zero points"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(scan->device, &c);
  if (rc) return rc;
  const double t0 = now_ms();
  const size_t M = scan->N;
  std::unique_ptr<tdtk_tree> t(new tdtk_tree);
  t->device = scan->device; t->M = M; t->bucket = bucket_size;
  if ((rc = tree_check_args(M, bucket_size))) return rc;
  if ((rc = c->ws[WS_TMPA].ensure(3 * M * sizeof(double)))) return rc;
  const bool saved = scan->ox != nullptr;   // moved since tdtk_scan_mark_original: the saved points are the original
  if (!saved && (rc = scan_settle(c, scan))) return rc;
  HIPCHK(launch_unsort_aos(saved ? scan->ox : scan->x, saved ? scan->oy : scan->y, saved ? scan->oz : scan->z,
                           scan->d_order, M, c->ws[WS_TMPA].as<double>(), c->stream));
  if ((rc = tree_from_device_points(c, t.get(), M, bucket_size, t0))) return rc;
  if ((rc = tree_finish(c, t.get(), M))) return rc;
  *out = t.release();
  return TDTK_OK;
}

// KDtreeMetaManaged (src/slam6d/kdMeta.cc:34-134): one tree over the CURRENT points of several resident scans,
// concatenated in the order given, each scan in its caller's order (prepareTempIndices, kdMeta.cc:60-79)
int tdtk_tree_create_from_scans(tdtk_scan* const* scans, int nscans, int bucket_size, tdtk_tree** out)
{
  if (!out) { set_error("out is NULL"); return TDTK_EINVAL; }
  *out = nullptr;
  if (!scans || nscans <= 0 || !scans[0]) { set_error("cannot create kdtree with zero points"); return TDTK_EINVAL; }
  size_t M = 0;
  for (int i = 0; i < nscans; i++) {
    if (!scans[i]) { set_error("NULL scan"); return TDTK_EINVAL; }
    if (scans[i]->device != scans[0]->device) { set_error("scans live on different devices"); return TDTK_EINVAL; }
    M += scans[i]->N;
  }
  if (M == 0) { set_error("cannot create kdtree with zero points"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(scans[0]->device, &c);
  if (rc) return rc;
  const double t0 = now_ms();
  std::unique_ptr<tdtk_tree> t(new tdtk_tree);
  t->device = scans[0]->device; t->M = M; t->bucket = bucket_size;
  if ((rc = tree_check_args(M, bucket_size))) return rc;
  if ((rc = c->ws[WS_TMPA].ensure(3 * M * sizeof(double)))) return rc;
  if ((rc = scans_settle(c, scans, nscans))) return rc;
  size_t off = 0;
  for (int i = 0; i < nscans; i++) {
    const tdtk_scan* sc = scans[i];
    HIPCHK(launch_unsort_aos(sc->x, sc->y, sc->z, sc->d_order, sc->N, c->ws[WS_TMPA].as<double>() + 3 * off, c->stream));
    off += sc->N;
  }
  if ((rc = tree_from_device_points(c, t.get(), M, bucket_size, t0))) return rc;
  if ((rc = tree_finish(c, t.get(), M))) return rc;
  *out = t.release();
  return TDTK_OK;
}

void tdtk_tree_destroy(tdtk_tree* t)
{
  if (!t) return;
  // a batch of scan moves / link passes left running behind the fence may still read this tree: explicit wait, not
  // hipFree's implicit device synchronisation
  wait_deferred(t->device);
  delete t;   // ~tdtk_tree releases the device arrays
}

int tdtk_tree_get_info(const tdtk_tree* t, tdtk_tree_info* info)
{
  if (!t || !info) { set_error("NULL argument"); return TDTK_EINVAL; }
  *info = t->info;
  return TDTK_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------
// internal: one search pass over SoA queries
// ------------------------------------------------------------------------------------------
// the spill area behind the LDS stacks: [level][lane] for as many lanes as ANY search kernel may launch for this
// batch size (every kernel indexes it with its own grid, all of them <= search_max_lanes)
static int prepare_overflow_in(DevBuf& bm2, DevBuf& bref, const tdtk_tree* t, size_t nq, SearchArgs& a)
{
  const int need = (int)t->info.max_depth - 1 - search_lds_depth();
  a.ovf_m2 = nullptr; a.ovf_ref = nullptr;
  if (need > 0) {
    const size_t lanes = search_max_lanes(nq);
    int rc = bm2.ensure(lanes * need * sizeof(double));
    if (rc) return rc;
    rc = bref.ensure(lanes * need * sizeof(uint32_t));
    if (rc) return rc;
    a.ovf_m2 = bm2.as<double>();
    a.ovf_ref = bref.as<uint32_t>();
  }
  return TDTK_OK;
}
static int prepare_overflow(Ctx* c, const tdtk_tree* t, size_t nq, SearchArgs& a)
{
  return prepare_overflow_in(c->ws[WS_OVF_M2], c->ws[WS_OVF_REF], t, nq, a);
}

// HIP events around the search and the pair-sum kernels of every pass (tdtk_icp_result.nn_ms / sums_ms,
// tdtk_last_kernel_ms, tdtk_last_timings) are a profiling aid and OFF unless asked for (tdtk_kernel_timing(1) or
// TDTK_KERNEL_TIMING=1): four marker packets on the stream, two event waits and two read-outs per ICP iteration cost
// 10 us of it -- 4 % of a 1M-point iteration, 19 % of an 81K-point one (0.2602 -> 0.2504 ms, 54.9 -> 44.4 us).
static std::atomic<int> g_kernel_timing{-1};
static bool kernel_timing()
{
  int v = g_kernel_timing.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("TDTK_KERNEL_TIMING");
    v = (e && e[0] == '1') ? 1 : 0;
    g_kernel_timing.store(v, std::memory_order_relaxed);
  }
  return v != 0;
}

static int run_search(Ctx* c, const tdtk_tree* t, SearchArgs& a, int dirmode, bool count, hipStream_t s,
                      bool timed)
{
  timed = timed && kernel_timing();
  a.T = t->dev;
  const uint32_t grid = search_grid(a.n);
  int rc = prepare_overflow(c, t, a.n, a);
  if (rc) return rc;
  if (c->counting && dirmode == 0 && !count) {
    count = true;
    a.counters = c->d_counters.as<unsigned long long>();
    c->counted_queries += a.n;
  }
  if (dirmode == 0 && search_uses_queue(a.n)) {
    if (s != c->stream) {   // the counters belong to this context's stream: order a caller's stream behind it
      HIPCHK(hipEventRecord(c->e_user, c->stream));
      HIPCHK(hipStreamWaitEvent(s, c->e_user, 0));
    }
    if ((rc = c->qc.attach(a))) return rc;
  }
  if (timed) HIPCHK(hipEventRecord(c->e0, s));
  HIPCHK(launch_search(a, grid, dirmode, count, s));
  if (timed) { HIPCHK(hipEventRecord(c->e1, s)); c->ev_pending = true; }
  return TDTK_OK;
}

static int collect_ms(Ctx* c, double* ms, double* sums_ms = nullptr)
{
  if (c->ev_pending) {
    HIPCHK(hipEventSynchronize(c->e1));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, c->e0, c->e1));
    c->last_nn_ms = f;
    c->ev_pending = false;
  }
  if (c->ev2_pending) {
    HIPCHK(hipEventSynchronize(c->e3));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, c->e2, c->e3));
    c->last_sums_ms = f;
    c->ev2_pending = false;
  }
  if (c->ev4_pending) {
    HIPCHK(hipEventSynchronize(c->e5));
    float f = 0;
    HIPCHK(hipEventElapsedTime(&f, c->e4, c->e5));
    c->last_normals_ms = f;
    c->ev4_pending = false;
  }
  const bool on = kernel_timing();
  if (ms) *ms = on ? c->last_nn_ms : 0.0;
  if (sums_ms) *sums_ms = on ? c->last_sums_ms : 0.0;
  return TDTK_OK;
}

// Retire-time accumulation inside the search kernel (kernels.hip, FUSE 1) is OFF by default: its 34 accumulator
// registers take the kernel from 7 to 4 waves per SIMD and every retire waits for its own gather, so the search
// grows by about what the separate pair-sum pass costs (1M-vs-1M ICP: 0.2881 + 0.0072 ms fused against
// 0.2708 + 0.0280 ms; 4M: 1.179 + 0.016 against 0.965 + 0.058 -- gpurun_out/r2a/sweep.log).  TDTK_FUSE_SUMS=1 selects it.
// Batches below the persistent-lane kernel's size are another matter: their kernels are not short of issue slots, and
// the workgroup sums its own chunk once it is through with it (chunk_pair_sums): an 81K-point ICP iteration is two
// launches instead of three.  On by default there, TDTK_FUSE_SUMS=0 switches it off.
static int fuse_mode(size_t N)   // 0: separate k_accum, 1: at retire time, 3: by each wave after its last query,
{                                // 4: by each workgroup after its chunk (small batches)
  const char* e = lab_env("TDTK_FUSE_SUMS");
  const int v = e ? atoi(e) : -1;
  const int kind = search_fuse_kind(N);
  if (kind == 2) return v == 0 ? 0 : 4;
  // round 3: 3 is the default -- with the bucket-group kernel (four waves per SIMD either way) the waves that are done early
  // add up their own slabs while the launch waits for the slowest: 1M-vs-1M 0.2217 -> 0.2183 ms per iteration at the
  // driver's arguments, 0.2000 -> 0.1925 over 100 (the separate k_accum launch, 18-24 us, is gone; the search grows by ~10)
  if (kind == 1) return (v == 0 || v == 1 || v == 3) ? v : (search_fuse_after_last_pays(N) ? 3 : 0);   // (4M and more: k_accum)
  return 0;
}

// acc[ACC_TOTAL] (sums about `shift`) -> the reference's quantities
static void finish_sums(const double* acc, const double shift[3], size_t nq, unsigned want, tdtk_pair_sums* o)
{
  std::memset(o, 0, sizeof *o);
  o->n_queries = nq;
  const double n = acc[ACC_N];
  o->n = (uint64_t)(n + 0.5);
  o->sum = acc[ACC_SUM];
  o->lum_sumd2 = acc[ACC_SUM];
  if (o->n == 0) return;
  const double* Sm = acc + ACC_SM;
  const double* Sd = acc + ACC_SD;
  double wm[3], wd[3];
  for (int a = 0; a < 3; a++) {
    wm[a] = Sm[a] / n; wd[a] = Sd[a] / n;
    o->centroid_m[a] = shift[a] + wm[a];
    o->centroid_d[a] = shift[a] + wd[a];
  }
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) o->Si[a * 3 + b] = acc[ACC_P + a * 3 + b] - Sm[a] * Sd[b] / n;
  if (want & TDTK_WANT_APX) {
    double D2[3][3], Dc[3][3], E[3][3];
    const double* dd = acc + ACC_DD;
    D2[0][0] = dd[0]; D2[0][1] = D2[1][0] = dd[1]; D2[0][2] = D2[2][0] = dd[2];
    D2[1][1] = dd[3]; D2[1][2] = D2[2][1] = dd[4]; D2[2][2] = dd[5];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) {
        Dc[a][b] = D2[a][b] - Sd[a] * Sd[b] / n;
        // E[a][b] = sum (p1-p2)_a (p2-cd)_b
        E[a][b] = acc[ACC_P + a * 3 + b] - D2[a][b] - (Sm[a] - Sd[a]) * Sd[b] / n;
      }
    o->apx_A[0] = Dc[1][1] + Dc[2][2];
    o->apx_A[1] = -Dc[0][1];
    o->apx_A[2] = -Dc[0][2];
    o->apx_A[3] = Dc[0][0] + Dc[2][2];
    o->apx_A[4] = -Dc[1][2];
    o->apx_A[5] = Dc[0][0] + Dc[1][1];
    o->apx_B[0] = E[2][1] - E[1][2];
    o->apx_B[1] = E[0][2] - E[2][0];
    o->apx_B[2] = E[1][0] - E[0][1];
  }
  if (want & TDTK_WANT_NAPX) {
    // c_true = (p2 - cd) x n = u0 - w x n,  w = cd - shift  ->  v_true = L v0,
    // L = [[I, -W], [0, I]], W = [w]x
    double A0[6][6], L[6][6], T1[6][6];
    int q = 0;
    for (int r = 0; r < 6; r++)
      for (int s = r; s < 6; s++) { A0[r][s] = A0[s][r] = acc[ACC_NA + q]; q++; }
    for (int r = 0; r < 6; r++)
      for (int s = 0; s < 6; s++) L[r][s] = (r == s) ? 1.0 : 0.0;
    const double* w = wd;
    L[0][4] = w[2];  L[0][5] = -w[1];
    L[1][3] = -w[2]; L[1][5] = w[0];
    L[2][3] = w[1];  L[2][4] = -w[0];
    for (int r = 0; r < 6; r++)
      for (int s = 0; s < 6; s++) {
        double v = 0;
        for (int k = 0; k < 6; k++) v += L[r][k] * A0[k][s];
        T1[r][s] = v;
      }
    q = 0;
    for (int r = 0; r < 6; r++)
      for (int s = r; s < 6; s++) {
        double v = 0;
        for (int k = 0; k < 6; k++) v += T1[r][k] * L[s][k];
        o->napx_A[q++] = v;
      }
    for (int r = 0; r < 6; r++) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += L[r][k] * acc[ACC_NB + k];
      o->napx_B[r] = v;
    }
    o->napx_sum = acc[ACC_NS];
  }
  if (want & TDTK_WANT_LUM) {
    for (int k = 0; k < 15; k++) o->lum[k] = acc[ACC_L + k];
    o->lum_udot = acc[ACC_LU];
  }
  if (want & TDTK_WANT_MOM2) {
    // second moments about the respective centroids
    const double* mm = acc + ACC_MM;
    const double* dd = acc + ACC_DD;
    int q = 0;
    for (int a = 0; a < 3; a++)
      for (int b = a; b < 3; b++, q++) {
        o->mom_mm[q] = mm[q] - Sm[a] * Sm[b] / n;
        o->mom_dd[q] = dd[q] - Sd[a] * Sd[b] / n;
      }
  }
  if (want & TDTK_WANT_GAPX) {
    // a = p1 - cm, b = p2 - cm (BOTH about centroid_m, gapx6D.cc:185-191); w = cm - shift = Sm/n
    double MM[3][3], DD[3][3], Saa[3][3], Sbb[3][3], Sab[3][3], Sb[3];
    const double* mm = acc + ACC_MM;
    const double* dd = acc + ACC_DD;
    MM[0][0] = mm[0]; MM[0][1] = MM[1][0] = mm[1]; MM[0][2] = MM[2][0] = mm[2];
    MM[1][1] = mm[3]; MM[1][2] = MM[2][1] = mm[4]; MM[2][2] = mm[5];
    DD[0][0] = dd[0]; DD[0][1] = DD[1][0] = dd[1]; DD[0][2] = DD[2][0] = dd[2];
    DD[1][1] = dd[3]; DD[1][2] = DD[2][1] = dd[4]; DD[2][2] = dd[5];
    for (int i = 0; i < 3; i++) {
      Sb[i] = Sd[i] - Sm[i];
      for (int j = 0; j < 3; j++) {
        Saa[i][j] = MM[i][j] - n * wm[i] * wm[j];
        Sbb[i][j] = DD[i][j] - wm[i] * Sd[j] - wm[j] * Sd[i] + n * wm[i] * wm[j];
        Sab[i][j] = acc[ACC_P + i * 3 + j] - wm[i] * Sd[j];
      }
    }
    auto sym = [](const double S[3][3], double* o9) {
      o9[0] = S[1][1] + S[2][2]; o9[1] = -S[0][1]; o9[2] = -S[0][2];
      o9[3] = -S[0][1]; o9[4] = S[0][0] + S[2][2]; o9[5] = -S[1][2];
      o9[6] = -S[0][2]; o9[7] = -S[1][2]; o9[8] = S[0][0] + S[1][1];
    };
    sym(Saa, o->gapx_MkMkt);
    sym(Sbb, o->gapx_DkDkt);
    // sum(p1 - cm) is zero by construction; the literal "+ p1y + p2y" terms leave sum(p2 - cm)
    const double d11 = Sab[1][1] + Sb[2], d22 = Sab[0][0] + Sb[2], d33 = Sab[0][0] + Sb[1];
    double* A = o->gapx_MkDkt;
    A[0] = d11; A[1] = -Sab[1][0]; A[2] = -Sab[2][0];
    A[3] = -Sab[1][0]; A[4] = d22; A[5] = -Sab[2][1];
    A[6] = -Sab[2][0]; A[7] = -Sab[2][1]; A[8] = d33;
    double* Bm = o->gapx_DkMkt;
    Bm[0] = d11; Bm[1] = -Sab[0][1]; Bm[2] = -Sab[0][2];
    Bm[3] = -Sab[0][1]; Bm[4] = d22; Bm[5] = -Sab[1][2];
    Bm[6] = -Sab[0][2]; Bm[7] = -Sab[1][2]; Bm[8] = d33;
    o->gapx_Ak2[0] = Sab[2][1] - Sab[1][2];
    o->gapx_Ak2[1] = Sab[0][2] - Sab[2][0];
    o->gapx_Ak2[2] = Sab[1][0] - Sab[0][1];
    for (int k = 0; k < 3; k++) o->gapx_Ak1[k] = -o->gapx_Ak2[k];
  }
}

// The pair sums of a pass land in pinned host memory, one 8-byte store per sum (k_final).  Waiting for them with
// hipStreamSynchronize costs ~6 us more per round trip on this box than watching the words themselves
// (tools/micro/sync_cost.hip: 12.7 against 6.3 us for an empty kernel), which is a sixth of a small scan's ICP iteration:
// the host arms every word with a NaN no sum can produce, launches, and reads until none is left.  The stream is asked
// now and then, so a failed launch ends the wait with its error and a finished stream ends it whatever the words say.
// TDTK_POLL_SUMS=0: hipStreamSynchronize.
static const uint64_t SUMS_ARMED = 0x7FF8DEADBEEF0001ull;
static bool poll_sums()
{
  static const bool on = [] { const char* e = getenv("TDTK_POLL_SUMS"); return !(e && e[0] == '0'); }();
  return on;
}
static void arm_sums(double* h_pin)
{
  volatile uint64_t* w = reinterpret_cast<volatile uint64_t*>(h_pin);
  for (int k = 0; k < ACC_TOTAL; k++) w[k] = SUMS_ARMED;
}
// The wait is adaptive: the words are polled with a pause instruction between looks for TDTK_SPIN_US microseconds (default
// 30: an iteration over a small scan, 25-40 us, ends inside the window and keeps the 4 us it gains over
// hipStreamSynchronize); after that every look is followed by sched_yield(), so the 200 us kernel of a 1M-point
// iteration does not hold a core against the other threads of an OpenMP host (a yield with nobody waiting returns at
// once and costs the wait nothing).  The stream is asked now and then in both phases: a failed launch ends the wait
// with its error, a finished stream ends it whatever the words say.
static hipError_t await_sums(const double* h_pin, hipStream_t s)
{
  static const double spin_us = [] { const char* e = getenv("TDTK_SPIN_US"); const double v = e ? atof(e) : 30.0; return v < 0.0 ? 0.0 : v; }();
  const volatile uint64_t* w = reinterpret_cast<const volatile uint64_t*>(h_pin);
  const auto t0 = std::chrono::steady_clock::now();
  bool yielding = spin_us == 0.0;
  for (uint32_t spins = 1;; spins++) {
    bool all = true;
    for (int k = ACC_TOTAL - 1; k >= 0 && all; k--) all = w[k] != SUMS_ARMED;
    if (all) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
    if (yielding) {
      sched_yield();
      if ((spins & 0x3Fu) != 0) continue;
    } else {
      __builtin_ia32_pause();
      if ((spins & 0x3Fu) == 0 &&
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > spin_us) yielding = true;
      if ((spins & 0x3FFu) != 0) continue;
    }
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return hipSuccess;
    if (q != hipErrorNotReady) return q;
  }
}

// search + accumulate over a resident scan.  acc_out (host, ACC_TOTAL) receives raw columns.
// SearchArgs::tie (kernels.hip, "the quick check deferred"): the reference's quick check computes a = max_i(fabs(q_i - c_i) - h_i)
// with c = 0.5 * (min + max), h = 0.5 * (max - min) of the node's points; for a point p of the node and a query within the
// search radius R of it, a <= |q - p|_inf + E with E <= 8 * 2^-53 * (|min| + |max| + |q|) <= 2^-50 * (3 absmax + R), so a node
// the check cuts off (a * a >= closest_d2) holds only points with d2 >= closest_d2 - (2 E R + E^2).  Four times that, for the
// roundings of the comparison itself and of this expression; TDTK_DEFER_CHECK=0: every visit makes the check.
static double search_margin(const tdtk_tree* t, double maxd2)
{
  if (!(maxd2 > 0.0) || !std::isfinite(maxd2)) return 0.0;
  const double R = std::sqrt(maxd2), E = std::ldexp(3.0 * (double)t->dev.absmax + R, -50);
  const double m = 4.0 * (2.0 * E * R + E * E);
  return (std::isfinite(m) && m < 1e-3 * maxd2) ? m : 0.0;
}
// The margin always widens a warm radius (SearchArgs::margin); whether the walk also DEFERS the quick check (SearchArgs::tie > 0)
// is a performance policy of its own:
static double search_tie(const tdtk_tree* t, double maxd2)
{
  static const bool off = [] { const char* e = getenv("TDTK_DEFER_CHECK"); return e && e[0] == '0'; }();
  if (off || !t->d_split) return 0.0;
  // Only for trees that live in the caches (the Infinity Cache is 256 MB): a walk without the quick check visits a sixth more
  // buckets, and once a bucket is a trip to HBM that costs more than the node loads it saves.  ICP iteration, k_search with
  // every check / with the check deferred: 1M-point tree (57 MB) 0.1665 / 0.1570 ms, 2M (115 MB) 0.357 / 0.334, 4M (230 MB)
  // 0.706 / 0.656, 10M-point city (0.65 GB) 1.28 / 1.37 (tools/r5_nobox_ab.sh, r5_defer_sizes.sh, r5_c5_waves.sh)
  static const size_t max_mb = [] { const char* e = lab_env("TDTK_DEFER_MAX_MB"); return e ? (size_t)atol(e) : (size_t)256; }();
  if (t->info.device_bytes > (max_mb << 20)) return 0.0;
  return search_margin(t, maxd2);
}

static int scan_pass(Ctx* c, const tdtk_tree* model, const double* A16, tdtk_scan* data, int pmode,
                     double maxd2, unsigned want, const double* lum_D, const double* pending,
                     bool do_search, double* acc_out, double shift_out[3], bool warm = false, const unsigned char* skip = nullptr)
{
  hipStream_t s = c->stream;
  const size_t N = data->N;
  int rc = c->ws[WS_KPOS].ensure(N * sizeof(int));
  if (rc) return rc;
  if ((rc = scan_settle(c, data))) return rc;
  Mat4 A, inv;
  std::memcpy(A.m, A16, sizeof A.m);
  m4inv(A16, inv.m);  // searchTree.cc:110
  // sums are taken about the model box centre mapped to the world: keeps the raw second
  // moments small so the centred covariance survives the subtraction in fp64
  double sh[3];
  sh[0] = model->centre[0] * A16[0] + model->centre[1] * A16[4] + model->centre[2] * A16[8] + A16[12];
  sh[1] = model->centre[0] * A16[1] + model->centre[1] * A16[5] + model->centre[2] * A16[9] + A16[13];
  sh[2] = model->centre[0] * A16[2] + model->centre[1] * A16[6] + model->centre[2] * A16[10] + A16[14];
  for (int k = 0; k < 3; k++) shift_out[k] = sh[k];
  // the base block (n, sum, centroids, Si) of a closest-point pass comes out of the search kernel itself when the
  // batch is large enough for the persistent-lane kernel (retire-time accumulation, kernels.hip)
  const int fmode = (do_search && pmode == 0 && (want & ~TDTK_WANT_BASE) == 0 && !lum_D) ? fuse_mode(N) : 0;
  const bool fused = fmode != 0 && !(fmode == 4 && c->counting);   // (the instrumented small-batch kernels have no epilogue)
  if (do_search) {
    SearchArgs sa{};
    sa.x = data->x; sa.y = data->y; sa.z = data->z;
    sa.nx = data->nx; sa.ny = data->ny; sa.nz = data->nz;
    sa.n = N;
    sa.has_pending = pending ? 1 : 0;
    if (pending) std::memcpy(sa.pending.m, pending, sizeof sa.pending.m);
    sa.inv = inv; sa.has_inv = 1;
    sa.maxd2 = maxd2;
    sa.kpos = c->ws[WS_KPOS].as<int>();
    sa.skip = skip;
    sa.warm = (warm && pmode != 1 && !c->count_cold) ? 1 : 0;   // WS_KPOS still holds this scan's hits in this tree from the last pass
    sa.margin = sa.warm ? search_margin(model, maxd2) : 0.0;
    sa.tie = sa.warm ? search_tie(model, maxd2) : 0.0;
    {
      // ... and WS_COST how many buckets each of its queries visited then: the persistent-lane kernel hands a wave's slab
      // out with the expensive queries first (TDTK_COST_ORDER=0: in slab order)
      const char* co = lab_env("TDTK_COST_ORDER");
      if (pmode != 1 && !(co && co[0] == '0') && search_fuse_kind(N) == 1) {
        if ((rc = c->ws[WS_COST].ensure(N))) return rc;
        sa.cost = c->ws[WS_COST].as<unsigned char>();
        sa.use_cost = sa.warm;
      }
    }
    sa.d2 = nullptr;
#ifdef TDTK_LAB
    // lab (a measured negative, NEGATIVES.md): slabs of equal cost, cut by the previous pass of this loop behind its k_final;
    // TDTK_BALANCE=1 turns it on
    if (sa.use_cost && c->next_bounds && c->next_bounds_n == N) sa.bounds = c->next_bounds;
    c->next_bounds = nullptr;
    const bool cut_next = sa.cost != nullptr && search_fuse_after_last_pays(N) && [] { const char* e = lab_env("TDTK_BALANCE"); return e && e[0] == '1'; }();
    auto cut_slabs = [&]() -> int {     // enqueued behind the pass's last kernel: runs while the host solves for the pose
      if (!cut_next) return TDTK_OK;
      const void* before = c->ws[WS_BOUNDS].p;
      int r2 = c->ws[WS_BOUNDS].ensure(slab_bounds_bytes(N));
      if (r2) return r2;
      if (c->ws[WS_BOUNDS].p != before) HIPCHK(hipMemsetAsync(c->ws[WS_BOUNDS].p, 0, 256, s));
      const uint32_t* b = nullptr;
      HIPCHK(launch_slab_bounds(sa.cost, N, c->ws[WS_BOUNDS].p, &b, s));
      c->next_bounds = b; c->next_bounds_n = N;
      return TDTK_OK;
    };
#else
    auto cut_slabs = []() -> int { return TDTK_OK; };
#endif
    uint32_t rows = 0;
    if (fused) {
      rows = search_fused_rows(N);
      if ((rc = c->ws[WS_PART].ensure((size_t)rows * ACC_TOTAL * sizeof(double)))) return rc;
      sa.fuse = fmode; sa.A = A;
      for (int k = 0; k < 3; k++) sa.shift[k] = sh[k];
      sa.partials = c->ws[WS_PART].as<double>();
    }
    rc = run_search(c, model, sa, pmode == 1 ? 1 : 0, false, s, true);
    if (rc) return rc;
    if (fused) {
      const bool tm = kernel_timing();
      if (tm) HIPCHK(hipEventRecord(c->e2, s));
      const bool poll = !tm && poll_sums();
      if (poll) arm_sums(c->h_pin);
      HIPCHK(launch_final(sa.partials, rows, c->h_pin, s, (fmode == 3 || fmode == 4) ? (int)ACC_DD : (int)ACC_TOTAL));   // base block only
      if (tm) { HIPCHK(hipEventRecord(c->e3, s)); c->ev2_pending = true; }
      if ((rc = cut_slabs())) return rc;
      if (poll) HIPCHK(await_sums(c->h_pin, s));
      else HIPCHK(hipStreamSynchronize(s));
      std::memcpy(acc_out, c->h_pin, ACC_TOTAL * sizeof(double));
      return TDTK_OK;
    }
    if ((rc = cut_slabs())) return rc;      // (before k_accum: the cut needs the cost bytes only)
  }
  AccumArgs aa{};
  aa.T = model->dev;
  aa.x = data->x; aa.y = data->y; aa.z = data->z;
  aa.nx = data->nx; aa.ny = data->ny; aa.nz = data->nz;
  aa.kpos = c->ws[WS_KPOS].as<int>();
  aa.n = N;
  aa.A = A; aa.inv = inv;
  for (int k = 0; k < 3; k++) aa.shift[k] = sh[k];
  aa.has_D = lum_D ? 1 : 0;
  if (lum_D) std::memcpy(aa.D, lum_D, sizeof aa.D);
  const uint32_t grid = accum_grid(N);
  rc = c->ws[WS_PART].ensure((size_t)grid * ACC_TOTAL * sizeof(double));
  if (rc) return rc;
  aa.partials = c->ws[WS_PART].as<double>();
  // k_final stores the 74 sums straight into pinned host memory (device-visible): no copy-engine hop
  // between the last kernel and the host solve, which matters when an iteration is ~100 us
  const bool tm = kernel_timing();
  if (tm) HIPCHK(hipEventRecord(c->e2, s));
  const bool poll = !tm && poll_sums();
  if (poll) arm_sums(c->h_pin);
  HIPCHK(launch_accum(aa, grid, want, pmode, c->h_pin, s));
  if (tm) { HIPCHK(hipEventRecord(c->e3, s)); c->ev2_pending = true; }
  if (poll) HIPCHK(await_sums(c->h_pin, s));
  else HIPCHK(hipStreamSynchronize(s));
  std::memcpy(acc_out, c->h_pin, ACC_TOTAL * sizeof(double));
  return TDTK_OK;
}

static int scan_idx_to_host(Ctx* c, const tdtk_tree* model, tdtk_scan* data, int32_t* idx_out)
{
  const size_t N = data->N;
  int rc = c->ws[WS_IDX].ensure(N * sizeof(int32_t));
  if (rc) return rc;
  HIPCHK(launch_scatter_idx(c->ws[WS_KPOS].as<int>(), nullptr, data->d_order, model->dev.pts, N,
                            c->ws[WS_IDX].as<int32_t>(), nullptr, c->stream));
  HIPCHK(hipMemcpyAsync(idx_out, c->ws[WS_IDX].p, N * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return TDTK_OK;
}

extern "C" {

// ---- find closest ----------------------------------------------------------------------
int tdtk_find_closest_dev(const tdtk_tree* t, const double* d_q, size_t K, double maxdist2,
                          int32_t* d_idx, double* d_d2, int presorted, void* stream)
{
  if (!t || (!d_q && K)) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(t->device, &c);
  if (rc) return rc;
  if (K == 0) return TDTK_OK;
  hipStream_t s = stream ? (hipStream_t)stream : c->stream;
  int ids[] = {WS_QX, WS_QY, WS_QZ};
  for (int id : ids)
    if ((rc = c->ws[id].ensure(K * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_KPOS].ensure(K * sizeof(int)))) return rc;
  if ((rc = c->ws[WS_D2].ensure(K * sizeof(double)))) return rc;
  const int32_t* order = nullptr;
  if (!presorted) {
    if ((rc = c->ws[WS_ORDER].ensure(K * sizeof(int32_t)))) return rc;
    if ((rc = c->ws[WS_CELL].ensure(K * sizeof(uint32_t)))) return rc;
    if ((rc = c->ws[WS_HIST].ensure(32768 * sizeof(uint32_t)))) return rc;
    BinArgs b{};
    b.q = d_q; b.dir = nullptr; b.n = K;
    for (int a = 0; a < 3; a++) {
      b.lo[a] = t->bbmin[a];
      const double ext = t->bbmax[a] - t->bbmin[a];
      b.scale[a] = (ext > 0) ? 32.0 / ext : 0.0;
    }
    b.hist = c->ws[WS_HIST].as<uint32_t>();
    b.cell = c->ws[WS_CELL].as<uint32_t>();
    b.sx = c->ws[WS_QX].as<double>(); b.sy = c->ws[WS_QY].as<double>(); b.sz = c->ws[WS_QZ].as<double>();
    b.order = c->ws[WS_ORDER].as<int32_t>();
    HIPCHK(launch_bin(b, s));
    order = b.order;
  } else {
    HIPCHK(launch_split_soa(d_q, K, c->ws[WS_QX].as<double>(), c->ws[WS_QY].as<double>(),
                            c->ws[WS_QZ].as<double>(), s));
  }
  SearchArgs sa{};
  sa.x = c->ws[WS_QX].as<double>(); sa.y = c->ws[WS_QY].as<double>(); sa.z = c->ws[WS_QZ].as<double>();
  sa.n = K; sa.maxd2 = maxdist2;
  sa.kpos = c->ws[WS_KPOS].as<int>();
  sa.d2 = c->ws[WS_D2].as<double>();
  rc = run_search(c, t, sa, 0, false, s, true);
  if (rc) return rc;
  HIPCHK(launch_scatter_idx(sa.kpos, sa.d2, order, t->dev.pts, K, d_idx, d_d2, s));
  if (!stream) HIPCHK(hipStreamSynchronize(s));
  else {
    // the kernels queued on the caller's stream use this thread's workspaces: whatever this thread's own stream
    // does next (any later tdtk call) waits for them, so the buffers are never overwritten under a running kernel
    HIPCHK(hipEventRecord(c->e_user, s));
    HIPCHK(hipStreamWaitEvent(c->stream, c->e_user, 0));
  }
  return TDTK_OK;
}

int tdtk_find_closest(const tdtk_tree* t, const double* q, size_t K, double maxdist2, int32_t* idx,
                      double* d2)
{
  if (!t || (!q && K) || (!idx && K)) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(t->device, &c);
  if (rc) return rc;
  if (K == 0) return TDTK_OK;
  if ((rc = c->ws[WS_TMPA].ensure(3 * K * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_IDX].ensure(K * sizeof(int32_t)))) return rc;
  if ((rc = c->ws[WS_TMPB].ensure(K * sizeof(double)))) return rc;
  HIPCHK(hipMemcpyAsync(c->ws[WS_TMPA].p, q, 3 * K * sizeof(double), hipMemcpyHostToDevice, c->stream));
  rc = tdtk_find_closest_dev(t, c->ws[WS_TMPA].as<double>(), K, maxdist2, c->ws[WS_IDX].as<int32_t>(),
                             c->ws[WS_TMPB].as<double>(), 0, c->stream);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(idx, c->ws[WS_IDX].p, K * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  if (d2) HIPCHK(hipMemcpyAsync(d2, c->ws[WS_TMPB].p, K * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return TDTK_OK;
}

int tdtk_find_closest_along_dir(const tdtk_tree* t, const double* q, const double* dir, size_t K,
                                double maxdist2, int32_t* idx, double* d2)
{
  if (!t || ((!q || !dir || !idx) && K)) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(t->device, &c);
  if (rc) return rc;
  if (K == 0) return TDTK_OK;
  hipStream_t s = c->stream;
  int dbl[] = {WS_QX, WS_QY, WS_QZ, WS_DX, WS_DY, WS_DZ, WS_D2, WS_TMPB};
  for (int id : dbl)
    if ((rc = c->ws[id].ensure(K * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_TMPA].ensure(6 * K * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_KPOS].ensure(K * sizeof(int)))) return rc;
  if ((rc = c->ws[WS_IDX].ensure(K * sizeof(int32_t)))) return rc;
  if ((rc = c->ws[WS_ORDER].ensure(K * sizeof(int32_t)))) return rc;
  if ((rc = c->ws[WS_CELL].ensure(K * sizeof(uint32_t)))) return rc;
  if ((rc = c->ws[WS_HIST].ensure(32768 * sizeof(uint32_t)))) return rc;
  double* dq = c->ws[WS_TMPA].as<double>();
  double* dd = dq + 3 * K;
  HIPCHK(hipMemcpyAsync(dq, q, 3 * K * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemcpyAsync(dd, dir, 3 * K * sizeof(double), hipMemcpyHostToDevice, s));
  BinArgs b{};
  b.q = dq; b.dir = dd; b.n = K;
  for (int a = 0; a < 3; a++) {
    b.lo[a] = t->bbmin[a];
    const double ext = t->bbmax[a] - t->bbmin[a];
    b.scale[a] = (ext > 0) ? 32.0 / ext : 0.0;
  }
  b.hist = c->ws[WS_HIST].as<uint32_t>();
  b.cell = c->ws[WS_CELL].as<uint32_t>();
  b.sx = c->ws[WS_QX].as<double>(); b.sy = c->ws[WS_QY].as<double>(); b.sz = c->ws[WS_QZ].as<double>();
  b.sdx = c->ws[WS_DX].as<double>(); b.sdy = c->ws[WS_DY].as<double>(); b.sdz = c->ws[WS_DZ].as<double>();
  b.order = c->ws[WS_ORDER].as<int32_t>();
  HIPCHK(launch_bin(b, s));
  SearchArgs sa{};
  sa.x = b.sx; sa.y = b.sy; sa.z = b.sz;
  sa.nx = b.sdx; sa.ny = b.sdy; sa.nz = b.sdz;
  sa.n = K; sa.maxd2 = maxdist2;
  sa.kpos = c->ws[WS_KPOS].as<int>();
  sa.d2 = c->ws[WS_D2].as<double>();
  rc = run_search(c, t, sa, 2, false, s, true);
  if (rc) return rc;
  HIPCHK(launch_scatter_idx(sa.kpos, sa.d2, b.order, t->dev.pts, K, c->ws[WS_IDX].as<int32_t>(),
                            c->ws[WS_TMPB].as<double>(), s));
  HIPCHK(hipMemcpyAsync(idx, c->ws[WS_IDX].p, K * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  if (d2) HIPCHK(hipMemcpyAsync(d2, c->ws[WS_TMPB].p, K * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return TDTK_OK;
}

int tdtk_count_visits(const tdtk_tree* t, const double* q, size_t K, double maxdist2, uint64_t counters[3])
{
  if (!t || !q || !counters) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(t->device, &c);
  if (rc) return rc;
  hipStream_t s = c->stream;
  int ids[] = {WS_QX, WS_QY, WS_QZ};
  for (int id : ids)
    if ((rc = c->ws[id].ensure(K * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_TMPA].ensure(3 * K * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_KPOS].ensure(K * sizeof(int)))) return rc;
  if ((rc = c->ws[WS_CNT].ensure(3 * sizeof(unsigned long long)))) return rc;
  HIPCHK(hipMemcpyAsync(c->ws[WS_TMPA].p, q, 3 * K * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(hipMemsetAsync(c->ws[WS_CNT].p, 0, 3 * sizeof(unsigned long long), s));
  HIPCHK(launch_split_soa(c->ws[WS_TMPA].as<double>(), K, c->ws[WS_QX].as<double>(),
                          c->ws[WS_QY].as<double>(), c->ws[WS_QZ].as<double>(), s));
  SearchArgs sa{};
  sa.x = c->ws[WS_QX].as<double>(); sa.y = c->ws[WS_QY].as<double>(); sa.z = c->ws[WS_QZ].as<double>();
  sa.n = K; sa.maxd2 = maxdist2;
  sa.kpos = c->ws[WS_KPOS].as<int>();
  sa.counters = c->ws[WS_CNT].as<unsigned long long>();
  rc = run_search(c, t, sa, 0, true, s, false);
  if (rc) return rc;
  unsigned long long h[3];
  HIPCHK(hipMemcpyAsync(h, c->ws[WS_CNT].p, sizeof h, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  for (int k = 0; k < 3; k++) counters[k] = h[k];
  return TDTK_OK;
}

int tdtk_last_kernel_ms(double* nn_ms)
{
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("no device"); return TDTK_EDEVICE; }
  Ctx* c;
  int rc = get_ctx(dev, &c, false);
  if (rc) return rc;
  return collect_ms(c, nn_ms);
}

int tdtk_kernel_timing(int on)
{
  const int was = kernel_timing() ? 1 : 0;
  g_kernel_timing.store(on ? 1 : 0, std::memory_order_relaxed);
  return was;
}

int tdtk_last_timings(double out[4])
{
  if (!out) { set_error("NULL argument"); return TDTK_EINVAL; }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) { set_error("no device"); return TDTK_EDEVICE; }
  Ctx* c;
  int rc = get_ctx(dev, &c, false);
  if (rc) return rc;
  rc = collect_ms(c, &out[0], &out[1]);
  out[2] = c->last_normals_ms;
  out[3] = c->last_build_ms;
  return rc;
}

// every FindClosest pass this thread runs on `device` from now on uses the instrumented instantiation of the
// kernel it would have used (same traversal, same warm radius, same results) and adds to the counters; on == 2: the
// searches also start cold (no warm start, no deferred quick check): what the counters then hold is the walk of the
// reference's _FindClosest (kdTreeImpl.h:345-383) -- SURVEY 8(d)'s n_int / n_pts -- on the same queries, same results
int tdtk_visit_counting(int device, int on)
{
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  if (on) {
    if ((rc = c->d_counters.ensure(16 * sizeof(unsigned long long)))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemset(c->d_counters.p, 0, 16 * sizeof(unsigned long long)));
    c->counted_queries = 0;
    c->counted_ann_queries = 0;
  }
  c->counting = on != 0;
  c->count_cold = on == 2;
  return TDTK_OK;
}

int tdtk_visit_counters(int device, uint64_t out[8])
{
  if (!out) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  for (int k = 0; k < 8; k++) out[k] = 0;
  out[3] = c->counted_queries;
  out[6] = c->counted_ann_queries;
  if (!c->d_counters.p) return TDTK_OK;
  HIPCHK(hipDeviceSynchronize());   // link passes run on auxiliary streams
  unsigned long long h[8];
  HIPCHK(hipMemcpy(h, c->d_counters.p, sizeof h, hipMemcpyDeviceToHost));
  for (int k = 0; k < 3; k++) out[k] = h[k];
  out[4] = h[4]; out[5] = h[5];
  out[7] = h[7];
  return TDTK_OK;
}

#ifdef TDTK_LAB
// lab: trips of the waves of the instrumented persistent-lane launches since tdtk_visit_counting(device, 1) through the node
// walk (out[0]) and the bucket scan (out[1]): with the visit counters, lane-visits / (64 x trips) = how full a phase's trips are
extern "C" int tdtk_lab_trip_counters(int device, uint64_t out[2])
{
  if (!out) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  out[0] = out[1] = 0;
  if (!c->d_counters.p) return TDTK_OK;
  HIPCHK(hipDeviceSynchronize());
  unsigned long long h[2];
  HIPCHK(hipMemcpy(h, c->d_counters.as<unsigned long long>() + 8, sizeof h, hipMemcpyDeviceToHost));
  out[0] = h[0]; out[1] = h[1];
  return TDTK_OK;
}
#endif

// measured roofline denominators: kind 0 = HBM stream copy (16 B per lane, `bytes` read + `bytes`
// written per pass, the buffers far beyond the 256 MB Infinity Cache when bytes >= 1 GB); kind 1 = repeated reads of
// a buffer that fits the L2s (bytes <= 16 MB: every workgroup sweeps its own XCD-local slice); GB/s of the best pass.
int tdtk_measure_bandwidth(int device, int kind, size_t bytes, int reps, double* gbs)
{
  if (!gbs || bytes < 4096 || reps < 1 || kind < 0 || kind > 1) { set_error("bad argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  bytes &= ~(size_t)4095;
  DevBuf a, b;
  if ((rc = a.ensure(bytes))) return rc;
  if (kind == 0 && (rc = b.ensure(bytes))) return rc;
  HIPCHK(hipMemsetAsync(a.p, 1, bytes, c->stream));
  double best = 0.0;
  const int variants[3] = {0, 2, 3};          // stream copy: plain, non-temporal, larger grid -- the best one counts
  for (int v = 0; v < (kind == 0 ? 3 : 1); v++)
    for (int r = 0; r < reps + 1; r++) {
      HIPCHK(hipEventRecord(c->e0, c->stream));
      double moved = 0.0;
      HIPCHK(launch_bandwidth(kind == 0 ? variants[v] : kind, a.p, b.p, bytes, &moved, c->stream));
      HIPCHK(hipEventRecord(c->e1, c->stream));
      HIPCHK(hipEventSynchronize(c->e1));
      float ms = 0;
      HIPCHK(hipEventElapsedTime(&ms, c->e0, c->e1));
      if (r > 0 && ms > 0) best = std::max(best, moved / (ms * 1e-3) / 1e9);
    }
  *gbs = best;
  return TDTK_OK;
}

// ---- octree reduction ("-r <voxelSize>", centre mode) -----------------------------------
// Scan::calcReducedPoints (scan.cc:577-603) with reduction_nrpts == 0: BOctTree(xyz, n, voxelSize)
// then GetOctTreeCenter.  See reduce.hip.
int tdtk_reduce_octree(const double* xyz, size_t n, double voxel_size, int device, double* out_xyz, size_t* n_out)
{
  if (!n_out) { set_error("n_out is NULL"); return TDTK_EINVAL; }
  *n_out = 0;
  if (n == 0) return TDTK_OK;   // an empty root has no occupied child (Boctree.h:1268-1300)
  if (!xyz || !out_xyz) { set_error("NULL points"); return TDTK_EINVAL; }
  if (!(voxel_size > 0)) { set_error("voxel size must be > 0"); return TDTK_EINVAL; }
  if (n >= (1ull << 32) - 1) { set_error("scan too large"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  hipStream_t s = c->stream;
  const size_t sort_tmp = oct_sort_temp_bytes(n), scan_tmp = scan_u32_temp_bytes(n + 1);
  const size_t tmpb = sort_tmp > scan_tmp ? sort_tmp : scan_tmp;
  if ((rc = c->ws[WS_TMPA].ensure(6 * n * sizeof(double)))) return rc;          // points in | centres out
  if ((rc = c->ws[WS_QX].ensure(2 * n * sizeof(uint64_t)))) return rc;          // keys a|b
  if ((rc = c->ws[WS_CELL].ensure(2 * (n + 1) * sizeof(uint32_t)))) return rc;  // flags | slots
  if ((rc = c->ws[WS_TMPB].ensure(tmpb + 256))) return rc;
  if ((rc = c->ws[WS_BOX].ensure(bbox_temp_bytes() + 8 * sizeof(double)))) return rc;
  double* d_in = c->ws[WS_TMPA].as<double>();
  double* d_out = d_in + 3 * n;
  double* d_box = c->ws[WS_BOX].as<double>();
  uint64_t* keys_a = c->ws[WS_QX].as<uint64_t>();
  uint64_t* keys_b = keys_a + n;
  uint32_t* flags = c->ws[WS_CELL].as<uint32_t>();
  uint32_t* slot = flags + (n + 1);
  HIPCHK(hipMemcpyAsync(d_in, xyz, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(launch_bbox(d_in, n, d_box + 8, d_box, s));
  HIPCHK(hipMemcpyAsync(c->h_pin, d_box, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const double* b = c->h_pin;
  // root cube, Boctree.h:248-255
  OctRoot R;
  for (int a = 0; a < 3; a++) R.center[a] = 0.5 * (b[a] + b[3 + a]);
  R.size = std::max(std::max(0.5 * (b[3] - b[0]), 0.5 * (b[4] - b[1])), 0.5 * (b[5] - b[2]));
  R.size += 1.0;
  if (!std::isfinite(R.size)) { set_error("non-finite coordinates"); return TDTK_EINVAL; }
  // the root's children exist unconditionally (:257-268); a child is a leaf once its size <= voxelSize (:1166)
  R.depth = 1;
  for (double sz = R.size / 2.0; sz > voxel_size; sz /= 2.0) R.depth++;
  if (3 * R.depth > 63) { set_error("voxel size too small for this extent (more than 21 octree levels)"); return TDTK_EINVAL; }
  HIPCHK(launch_oct_keys_sorted(d_in, n, R, keys_a, keys_b, c->ws[WS_TMPB].p, sort_tmp, s));
  HIPCHK(launch_oct_heads(keys_b, n, flags, s));
  HIPCHK(launch_scan_u32(flags, slot, n + 1, c->ws[WS_TMPB].p, scan_tmp, s));
  HIPCHK(launch_oct_centres(keys_b, flags, slot, n, R, d_out, s));
  uint32_t cells = 0;
  HIPCHK(hipMemcpyAsync(&cells, slot + n, sizeof cells, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  HIPCHK(hipMemcpyAsync(out_xyz, d_out, 3 * (size_t)cells * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *n_out = cells;
  return TDTK_OK;
}

// Octree reduction with `-O <nrpts>` (Scan::calcReducedPoints, scan.cc:586-596): nrpts == 0 the leaf centres (above),
// nrpts == 1 one random point per occupied leaf (GetOctTreeRandom, Boctree.h:985-1018), nrpts > 1 up to nrpts random points
// per leaf (Boctree.h:1020-1062 with rm_scatter == false).  The leaves, their depth-first order and the order of the
// points inside a leaf are computed on the device (reduce.hip); the draws are std::rand() on the host, leaf by leaf in
// that order, as the reference makes them -- like `rnd`, reproducible against a serial build of the reference only
// (SURVEY N-d).  Not offered: nrpts == -1 (GetOctTreeAvg adds into memory it never initialises, Boctree.h:961-964) and
// rm_scatter (its loop forgets to advance the child pointer for a leaf it skips, Boctree.h:1032-1040).
int tdtk_reduce_octree_nrpts(const double* xyz, size_t n, double voxel_size, int nrpts, int rm_scatter, int device, double* out_xyz,
                             size_t* n_out)
{
  if (nrpts == 0 && !rm_scatter) return tdtk_reduce_octree(xyz, n, voxel_size, device, out_xyz, n_out);
  if (!n_out) { set_error("n_out is NULL"); return TDTK_EINVAL; }
  *n_out = 0;
  if (nrpts < 0) { set_error("reduction_nrpts == -1 (GetOctTreeAvg) reads uninitialised memory in the reference: not reproducible"); return TDTK_EUNSUP; }
  if (rm_scatter) { set_error("rm_scatter walks a stale child pointer in the reference (Boctree.h:1032-1040): not reproducible"); return TDTK_EUNSUP; }
  if (n == 0) return TDTK_OK;
  if (!xyz || !out_xyz) { set_error("NULL points"); return TDTK_EINVAL; }
  if (!(voxel_size > 0)) { set_error("voxel size must be > 0"); return TDTK_EINVAL; }
  if (n >= (1ull << 31)) { set_error("scan too large"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  hipStream_t s = c->stream;
  const size_t sort_tmp = oct_sort_temp_bytes(n), scan_tmp = scan_u32_temp_bytes(n + 1), scan64_tmp = scan_u64_temp_bytes(n + 1);
  const size_t tmpb = std::max(sort_tmp, std::max(scan_tmp, scan64_tmp));
  if ((rc = c->ws[WS_TMPA].ensure(6 * n * sizeof(double)))) return rc;           // points in | points out
  if ((rc = c->ws[WS_QX].ensure(2 * n * sizeof(uint64_t)))) return rc;           // keys by point | sorted
  if ((rc = c->ws[WS_CELL].ensure(3 * (n + 1) * sizeof(uint32_t)))) return rc;   // leaf flags | slots | starts
  if ((rc = c->ws[WS_TMPB].ensure(tmpb + 256))) return rc;
  if ((rc = c->ws[WS_BOX].ensure(bbox_temp_bytes() + 8 * sizeof(double)))) return rc;
  if ((rc = c->ws[WS_QY].ensure(7 * n * sizeof(uint32_t)))) return rc;           // perm | segS segE mid posL posR | sel
  if ((rc = c->ws[WS_QZ].ensure(2 * (n + 1) * sizeof(uint64_t)))) return rc;     // flags | their scan
  double* d_in = c->ws[WS_TMPA].as<double>();
  double* d_out = d_in + 3 * n;
  double* d_box = c->ws[WS_BOX].as<double>();
  uint64_t* keys_a = c->ws[WS_QX].as<uint64_t>();
  uint64_t* keys_b = keys_a + n;
  uint32_t* flags = c->ws[WS_CELL].as<uint32_t>();
  uint32_t* slot = flags + (n + 1);
  uint32_t* starts = slot + (n + 1);
  uint32_t* perm = c->ws[WS_QY].as<uint32_t>();
  uint32_t* work = perm + n;
  uint32_t* d_sel = perm + 6 * n;
  uint64_t* f64 = c->ws[WS_QZ].as<uint64_t>();
  uint64_t* P64 = f64 + (n + 1);
  HIPCHK(hipMemcpyAsync(d_in, xyz, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(launch_bbox(d_in, n, d_box + 8, d_box, s));
  HIPCHK(hipMemcpyAsync(c->h_pin, d_box, 6 * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  const double* b = c->h_pin;
  OctRoot R;                          // root cube, Boctree.h:248-255
  for (int a = 0; a < 3; a++) R.center[a] = 0.5 * (b[a] + b[3 + a]);
  R.size = std::max(std::max(0.5 * (b[3] - b[0]), 0.5 * (b[4] - b[1])), 0.5 * (b[5] - b[2]));
  R.size += 1.0;
  if (!std::isfinite(R.size)) { set_error("non-finite coordinates"); return TDTK_EINVAL; }
  R.depth = 1;
  for (double sz = R.size / 2.0; sz > voxel_size; sz /= 2.0) R.depth++;
  if (3 * R.depth > 63) { set_error("voxel size too small for this extent (more than 21 octree levels)"); return TDTK_EINVAL; }
  HIPCHK(launch_oct_keys_sorted(d_in, n, R, keys_a, keys_b, c->ws[WS_TMPB].p, sort_tmp, s));
  HIPCHK(launch_oct_heads(keys_b, n, flags, s));
  HIPCHK(launch_scan_u32(flags, slot, n + 1, c->ws[WS_TMPB].p, scan_tmp, s));
  HIPCHK(launch_oct_leaf_starts(flags, slot, n, starts, s));
  HIPCHK(launch_oct_leaf_order(keys_a, keys_b, n, R.depth, perm, work, f64, P64, c->ws[WS_TMPB].p, scan64_tmp, s));
  uint32_t cells = 0;
  HIPCHK(hipMemcpyAsync(&cells, slot + n, sizeof cells, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  std::vector<uint32_t> st((size_t)cells + 1);
  HIPCHK(hipMemcpy(st.data(), starts, st.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
  // the draws, leaf by leaf in depth-first order (globals.icc:607-610: rand(rnd) = (int)(rnd * std::rand() / (RAND_MAX + 1.0)))
  auto draw = [](int rnd) { return (int)((double)rnd * (double)std::rand() / (RAND_MAX + 1.0)); };
  std::vector<uint32_t> sel;
  sel.reserve(nrpts == 1 ? cells : n);
  std::vector<int> pick;
  for (uint32_t l = 0; l < cells; l++) {
    const uint32_t start = st[l], len = st[l + 1] - st[l];
    if (nrpts == 1) { sel.push_back(start + (uint32_t)draw((int)len)); continue; }
    if ((uint32_t)nrpts >= len) { for (uint32_t j = 0; j < len; j++) sel.push_back(start + j); continue; }
    pick.clear();
    while ((int)pick.size() < nrpts) {           // std::set<int>::insert of rand(length - 1)
      const int t = draw((int)len - 1);
      if (std::find(pick.begin(), pick.end(), t) == pick.end()) pick.push_back(t);
    }
    std::sort(pick.begin(), pick.end());
    for (int t : pick) sel.push_back(start + (uint32_t)t);
  }
  const size_t m = sel.size();
  if (m > n) { set_error("internal: more points kept than given"); return TDTK_EDEVICE; }
  if (m) {
    HIPCHK(hipMemcpyAsync(d_sel, sel.data(), m * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    HIPCHK(launch_oct_gather(d_in, perm, d_sel, m, d_out, s));
    HIPCHK(hipMemcpyAsync(out_xyz, d_out, 3 * m * sizeof(double), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
  }
  *n_out = m;
  return TDTK_OK;
}

// ---- normals (Scan::calcNormals -> calculateNormalsApxKNN, scan.cc:398-427, normals.cc:35-111) ----------------
// d_xyz: [n][3] in the caller's order (the ANN tree starts from the identity permutation, kd_tree.cpp:259-262);
// d_normals [n][3] and d_knn (nullable, [n][k]) come back in the caller's order.
static int normals_on_device(Ctx* c, const double* d_xyz, size_t n, int k, const double rPos[3], double eps,
                             double* d_normals, int32_t* d_knn)
{
  int rc;
  hipStream_t s = c->stream;
  if ((rc = c->ws[WS_ARENA].ensure(ann_build_arena_bytes(n)))) return rc;
  if ((rc = c->ws[WS_QX].ensure((n ? n : 1) * sizeof(AnnNode)))) return rc;
  if ((rc = c->ws[WS_QY].ensure(n * sizeof(KdPoint)))) return rc;
  if ((rc = c->ws[WS_BOX].ensure(bbox_temp_bytes() + 8 * sizeof(double)))) return rc;
  AnnNode* nodes = c->ws[WS_QX].as<AnnNode>();
  KdPoint* pts = c->ws[WS_QY].as<KdPoint>();
  double* d_bb = c->ws[WS_BOX].as<double>();
  AnnBuildResult r = ann_build_tree(d_xyz, n, c->ws[WS_ARENA].p, nodes, pts, d_bb, s);
  if (r.err != hipSuccess) {
    if (r.degenerate) { set_error("non-finite coordinates"); return TDTK_EINVAL; }
    set_error(std::string("ann_build_tree: ") + hipGetErrorString(r.err));
    return TDTK_EDEVICE;
  }
  const size_t spill = ann_spill_entries(n, r.max_depth);
  if ((rc = c->ws[WS_OVF_REF].ensure((spill + 1) * sizeof(uint32_t)))) return rc;
  if ((rc = c->ws[WS_OVF_M2].ensure((spill + 1) * sizeof(double)))) return rc;
  unsigned long long* d_cnt = nullptr;
  if (c->counting) { d_cnt = c->d_counters.as<unsigned long long>() + 4; c->counted_ann_queries += n; }
  HIPCHK(hipEventRecord(c->e4, s));
  HIPCHK(launch_ann_normals(nodes, r.root_ref, pts, n, k, eps, d_bb, rPos, c->ws[WS_OVF_REF].as<uint32_t>(),
                            c->ws[WS_OVF_M2].as<double>(), r.max_depth, d_normals, d_knn, d_cnt, s));
  HIPCHK(hipEventRecord(c->e5, s));
  c->ev4_pending = true;
  return TDTK_OK;
}

static int normals_check_args(size_t n, int k, const double* rPos, double eps)
{
  if (!rPos) { set_error("rPos is NULL"); return TDTK_EINVAL; }
  if (n == 0) { set_error("Could not calculate normals, XYZ data is empty"); return TDTK_EINVAL; }   // scan.cc:408-409
  if (k < 1 || k > 32) { set_error("k must be in 1..32"); return TDTK_EINVAL; }
  if ((size_t)k > n) { set_error("Requesting more near neighbors than data points"); return TDTK_EINVAL; }   // kd_search.cpp:103-105
  if (!(eps >= 0) || !std::isfinite(eps)) { set_error("eps must be >= 0"); return TDTK_EINVAL; }
  if (n >= (1ull << 29)) { set_error("scan too large (29-bit point positions)"); return TDTK_EINVAL; }
  return TDTK_OK;
}

int tdtk_normals_apx_knn(const double* xyz, size_t n, int k, const double rPos[3], double eps, int device,
                         double* normals_out, int32_t* knn_out)
{
  int rc;
  if ((rc = normals_check_args(n, k, rPos, eps))) return rc;
  if (!xyz || !normals_out) { set_error("NULL points"); return TDTK_EINVAL; }
  Ctx* c;
  if ((rc = get_ctx(device, &c))) return rc;
  hipStream_t s = c->stream;
  if ((rc = c->ws[WS_TMPA].ensure(6 * n * sizeof(double)))) return rc;   // points in | normals out
  if (knn_out && (rc = c->ws[WS_IDX].ensure(n * (size_t)k * sizeof(int32_t)))) return rc;
  double* d_in = c->ws[WS_TMPA].as<double>();
  double* d_nrm = d_in + 3 * n;
  int32_t* d_knn = knn_out ? c->ws[WS_IDX].as<int32_t>() : nullptr;
  HIPCHK(hipMemcpyAsync(d_in, xyz, 3 * n * sizeof(double), hipMemcpyHostToDevice, s));
  if ((rc = normals_on_device(c, d_in, n, k, rPos, eps, d_nrm, d_knn))) return rc;
  HIPCHK(hipMemcpyAsync(normals_out, d_nrm, 3 * n * sizeof(double), hipMemcpyDeviceToHost, s));
  if (knn_out) HIPCHK(hipMemcpyAsync(knn_out, d_knn, n * (size_t)k * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return TDTK_OK;
}

int tdtk_scan_calc_normals(tdtk_scan* sc, int k, const double rPos[3], double eps)
{
  if (!sc) { set_error("NULL argument"); return TDTK_EINVAL; }
  int rc;
  if ((rc = normals_check_args(sc->N, k, rPos, eps))) return rc;
  Ctx* c;
  if ((rc = get_ctx(sc->device, &c))) return rc;
  hipStream_t s = c->stream;
  const size_t n = sc->N;
  if ((rc = c->ws[WS_TMPA].ensure(6 * n * sizeof(double)))) return rc;
  double* d_in = c->ws[WS_TMPA].as<double>();
  double* d_nrm = d_in + 3 * n;
  if ((rc = scan_settle(c, sc))) return rc;
  // the resident points back in the caller's order, the normals back into the resident order
  HIPCHK(launch_unsort_aos(sc->x, sc->y, sc->z, sc->d_order, n, d_in, s));
  if ((rc = normals_on_device(c, d_in, n, k, rPos, eps, d_nrm, nullptr))) return rc;
  if (!sc->nx) {
    HIPCHK(handle_malloc((void**)&sc->nx, n * sizeof(double)));
    HIPCHK(handle_malloc((void**)&sc->ny, n * sizeof(double)));
    HIPCHK(handle_malloc((void**)&sc->nz, n * sizeof(double)));
  }
  HIPCHK(launch_gather_soa(d_nrm, reinterpret_cast<const uint32_t*>(sc->d_order), n, sc->nx, sc->ny, sc->nz, s));
  HIPCHK(hipStreamSynchronize(s));
  return TDTK_OK;
}

// ---- resident scan ---------------------------------------------------------------------
int tdtk_scan_create(const double* xyz, const double* nrm, size_t N, int device, tdtk_scan** out)
{
  if (!out) { set_error("out is NULL"); return TDTK_EINVAL; }
  *out = nullptr;
  if (!xyz && N) { set_error("NULL points"); return TDTK_EINVAL; }
  if (N >= (1ull << 32)) { set_error("scan too large"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  std::unique_ptr<tdtk_scan> sc(new tdtk_scan);
  sc->device = device; sc->N = N;
  if (N == 0) { *out = sc.release(); return TDTK_OK; }
  hipStream_t s = c->stream;
  const size_t tmp_bytes = morton_sort_temp_bytes(N);
  if ((rc = c->ws[WS_TMPA].ensure(3 * N * sizeof(double)))) return rc;        // AoS staging
  if ((rc = c->ws[WS_CELL].ensure(2 * N * sizeof(uint32_t)))) return rc;      // keys a|b
  if ((rc = c->ws[WS_ORDER].ensure(N * sizeof(uint32_t)))) return rc;         // idx a
  if ((rc = c->ws[WS_TMPB].ensure(tmp_bytes + 256))) return rc;
  if ((rc = c->ws[WS_BOX].ensure(bbox_temp_bytes() + 8 * sizeof(double)))) return rc;
  double* d_box = c->ws[WS_BOX].as<double>();
  double* d_aos = c->ws[WS_TMPA].as<double>();
  uint32_t* keys_a = c->ws[WS_CELL].as<uint32_t>();
  uint32_t* keys_b = keys_a + N;
  uint32_t* idx_a = c->ws[WS_ORDER].as<uint32_t>();
  HIPCHK(handle_malloc((void**)&sc->d_order, N * sizeof(int32_t)));
  HIPCHK(handle_malloc((void**)&sc->x, N * sizeof(double)));
  HIPCHK(handle_malloc((void**)&sc->y, N * sizeof(double)));
  HIPCHK(handle_malloc((void**)&sc->z, N * sizeof(double)));
  HIPCHK(hipMemcpyAsync(d_aos, xyz, 3 * N * sizeof(double), hipMemcpyHostToDevice, s));
  HIPCHK(launch_bbox(d_aos, N, d_box + 8, d_box, s));
  HIPCHK(launch_morton_order(d_aos, N, d_box, keys_a, idx_a, keys_b, reinterpret_cast<uint32_t*>(sc->d_order),
                             c->ws[WS_TMPB].p, tmp_bytes, s));
  HIPCHK(launch_gather_soa(d_aos, reinterpret_cast<const uint32_t*>(sc->d_order), N, sc->x, sc->y, sc->z, s));
  if (nrm) {
    HIPCHK(handle_malloc((void**)&sc->nx, N * sizeof(double)));
    HIPCHK(handle_malloc((void**)&sc->ny, N * sizeof(double)));
    HIPCHK(handle_malloc((void**)&sc->nz, N * sizeof(double)));
    HIPCHK(hipMemcpyAsync(d_aos, nrm, 3 * N * sizeof(double), hipMemcpyHostToDevice, s));
    HIPCHK(launch_gather_soa(d_aos, reinterpret_cast<const uint32_t*>(sc->d_order), N, sc->nx, sc->ny, sc->nz, s));
  }
  HIPCHK(hipStreamSynchronize(s));
  *out = sc.release();
  return TDTK_OK;
}

void tdtk_scan_destroy(tdtk_scan* s)
{
  if (!s) return;
  wait_deferred(s->device);   // a deferred k_transform2_batch may still be writing this scan's arrays
  delete s;   // ~tdtk_scan releases the device arrays
}

size_t tdtk_scan_size(const tdtk_scan* s) { return s ? s->N : 0; }

int tdtk_scan_transform(tdtk_scan* s, const double alignxf[16])
{
  if (!s || !alignxf) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(s->device, &c);
  if (rc) return rc;
  if ((rc = scan_keep_original(c, s))) return rc;
  std::unique_lock<std::recursive_mutex> lk(g_moves_mu);
  if (!s->pending.empty()) {
    // behind moves that are still queued: this one joins the queue (order is what matters)
    scan_queue_move(s, alignxf);
    if (!lazy_moves() || s->pending.size() >= LAZY_CHAIN_MAX) {
      if ((rc = scan_settle(c, s))) return rc;
      HIPCHK(hipStreamSynchronize(c->stream));
    }
    return TDTK_OK;
  }
  lk.unlock();
  Mat4 A;
  std::memcpy(A.m, alignxf, sizeof A.m);
  HIPCHK(launch_transform(s->x, s->y, s->z, s->nx, s->ny, s->nz, s->N, A, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return TDTK_OK;
}

static int scan_to_host(Ctx* c, const tdtk_scan* s, const double* x, const double* y, const double* z, double* out);

int tdtk_scan_mark_original(tdtk_scan* s)
{
  if (!s) { set_error("NULL argument"); return TDTK_EINVAL; }
  (void)hipSetDevice(s->device);
  wait_deferred(s->device);   // the saved original may be in use by a move that was left running
  if (s->npend.load(std::memory_order_acquire) != 0) {  // "original" = the points as they are now, queued moves included
    Ctx* c;
    int rc = get_ctx(s->device, &c);
    if (rc) return rc;
    if ((rc = scan_settle(c, s))) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  double* p[] = {s->ox, s->oy, s->oz};
  for (double* q : p)
    if (q) pool_free(q);
  s->ox = s->oy = s->oz = nullptr;
  s->track_original = true;
  return TDTK_OK;
}

int tdtk_scan_download_original(const tdtk_scan* s, double* xyz_out)
{
  if (!s || !xyz_out) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(s->device, &c);
  if (rc) return rc;
  if (!s->ox && (rc = scan_settle(c, s))) return rc;
  return scan_to_host(c, s, s->ox ? s->ox : s->x, s->ox ? s->oy : s->y, s->ox ? s->oz : s->z, xyz_out);
}

// sorted SoA -> caller-order AoS on the device, then one contiguous copy to the host
static int scan_to_host(Ctx* c, const tdtk_scan* s, const double* x, const double* y, const double* z, double* out)
{
  const size_t N = s->N;
  if (!N) return TDTK_OK;
  int rc = c->ws[WS_TMPA].ensure(3 * N * sizeof(double));
  if (rc) return rc;
  HIPCHK(launch_unsort_aos(x, y, z, s->d_order, N, c->ws[WS_TMPA].as<double>(), c->stream));
  HIPCHK(hipMemcpyAsync(out, c->ws[WS_TMPA].p, 3 * N * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return TDTK_OK;
}

int tdtk_scan_download(const tdtk_scan* s, double* xyz_out, double* nrm_out)
{
  if (!s || !xyz_out) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(s->device, &c);
  if (rc) return rc;
  if ((rc = scan_settle(c, s))) return rc;
  if ((rc = scan_to_host(c, s, s->x, s->y, s->z, xyz_out))) return rc;
  if (nrm_out && s->nx) rc = scan_to_host(c, s, s->nx, s->ny, s->nz, nrm_out);
  return rc;
}

int tdtk_scan_pairs(const tdtk_tree* model, const double A[16], tdtk_scan* data, int pmode, double maxd2,
                    uint32_t want, const double* lum_D, int32_t* idx_out, tdtk_pair_sums* sums)
{
  if (!model || !A || !data || !sums) { set_error("NULL argument"); return TDTK_EINVAL; }
  if (pmode < 0 || pmode > 2) { set_error("bad pairing mode"); return TDTK_EINVAL; }
  if ((pmode != 0 || (want & TDTK_WANT_NAPX)) && data->N && !data->nx) {   // an empty scan pairs with nothing
    set_error("this pairing mode / minimizer needs normals");
    return TDTK_EINVAL;
  }
  if (model->device != data->device) { set_error("tree and scan live on different devices"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(model->device, &c);
  if (rc) return rc;
  std::memset(sums, 0, sizeof *sums);
  sums->n_queries = data->N;
  if (data->N == 0) return TDTK_OK;
  double acc[ACC_TOTAL], shift[3];
  rc = scan_pass(c, model, A, data, pmode, maxd2, want, lum_D, nullptr, true, acc, shift);
  if (rc) return rc;
  finish_sums(acc, shift, data->N, want, sums);
  if (lum_D) sums->lum_sumd2 = acc[ACC_LSS];  // second-pass residual rides here (see tdtk_lum_link)
  if (idx_out) rc = scan_idx_to_host(c, model, data, idx_out);
  collect_ms(c, nullptr);
  return rc;
}

// icp6D::Point_Point_Error (icp6D.cc:293-367)
int tdtk_point_point_error(const tdtk_tree* model, const double A[16], tdtk_scan* data, double max_dist_match,
                           double scale_max, uint64_t* np_out, double* error_out)
{
  if (!model || !A || !data || !error_out) { set_error("NULL argument"); return TDTK_EINVAL; }
  if (model->device != data->device) { set_error("tree and scan live on different devices"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(model->device, &c);
  if (rc) return rc;
  if (np_out) *np_out = 0;
  *error_out = 0.0;
  const size_t N = data->N;
  if (N == 0) { *error_out = std::nan(""); return TDTK_OK; }   // 0 / 0 in the reference
  hipStream_t s = c->stream;
  if ((rc = c->ws[WS_KPOS].ensure(N * sizeof(int)))) return rc;
  if ((rc = scan_settle(c, data))) return rc;
  Mat4 Am, inv;
  std::memcpy(Am.m, A, sizeof Am.m);
  m4inv(A, inv.m);
  SearchArgs sa{};
  sa.x = data->x; sa.y = data->y; sa.z = data->z;
  sa.n = N; sa.inv = inv; sa.has_inv = 1;
  sa.maxd2 = max_dist_match * max_dist_match;
  sa.kpos = c->ws[WS_KPOS].as<int>();
  if ((rc = run_search(c, model, sa, 0, false, s, true))) return rc;
  AccumArgs aa{};
  aa.T = model->dev;
  aa.x = data->x; aa.y = data->y; aa.z = data->z;
  aa.kpos = sa.kpos; aa.n = N; aa.A = Am; aa.inv = inv;
  const uint32_t grid = accum_grid(N);
  if ((rc = c->ws[WS_PART].ensure((size_t)grid * ACC_TOTAL * sizeof(double)))) return rc;
  const double scale = std::log(scale_max) / (max_dist_match * max_dist_match);   // icp6D.cc:299
  HIPCHK(launch_pp_error(aa, grid, scale, c->ws[WS_PART].as<double>(), c->h_pin, s));
  HIPCHK(hipStreamSynchronize(s));
  collect_ms(c, nullptr);
  const double se = c->h_pin[0], cn = c->h_pin[1];
  if (np_out) *np_out = (uint64_t)(cn + 0.5);
  *error_out = (-0.39894228 * se) / cn;   // error -= 0.39894228 * exp(dist * scale) per pair; error / nr_ppairs
  return TDTK_OK;
}

int tdtk_get_pt_pairs(const tdtk_tree* t, const double A[16], const double* xyz_r, const double* normal_r,
                      size_t start, size_t end, int rnd, int pmode, double maxd2, uint32_t want,
                      const double* lum_D, int32_t* idx_out, double* p1_out, double* p2_out,
                      double* pn_out, tdtk_pair_sums* sums)
{
  if (!t || !A || !xyz_r || !sums || end < start) { set_error("bad argument"); return TDTK_EINVAL; }
  if ((pmode == 1 || pmode == 2 || (want & TDTK_WANT_NAPX)) && !normal_r) {
    set_error("this pairing mode / minimizer needs normals");
    return TDTK_EINVAL;
  }
  // rnd > 1: "take about 1/rnd-th of the numbers only" (searchTree.cc:118, globals.icc:607-610):
  // one std::rand() per candidate, consumed in index order like a serial (non-OpenMP) reference
  // build does.  The keep-mask is drawn on the host; only the kept queries go to the GPU.
  std::vector<double> kept_xyz, kept_nrm;
  std::vector<size_t> kept_pos;
  const double* q_xyz = xyz_r + 3 * start;
  const double* q_nrm = normal_r ? normal_r + 3 * start : nullptr;
  size_t n = end - start;
  const size_t n_all = n;
  if (rnd > 1) {
    for (size_t i = 0; i < n_all; i++) {
      const int r = (int)((double)rnd * (double)std::rand() / (RAND_MAX + 1.0));
      if (r != 0) continue;
      kept_pos.push_back(i);
      for (int k = 0; k < 3; k++) kept_xyz.push_back(q_xyz[3 * i + k]);
      if (q_nrm) for (int k = 0; k < 3; k++) kept_nrm.push_back(q_nrm[3 * i + k]);
    }
    n = kept_pos.size();
    q_xyz = kept_xyz.data();
    if (q_nrm) q_nrm = kept_nrm.data();
    if (idx_out) for (size_t i = 0; i < n_all; i++) idx_out[i] = -1;
  }
  tdtk_scan* sc = nullptr;
  int rc = tdtk_scan_create(q_xyz, q_nrm, n, t->device, &sc);
  if (rc) return rc;
  std::vector<int32_t> idx_local;
  int32_t* idx = nullptr;
  const bool want_pairs = p1_out || p2_out || pn_out;
  if (rnd > 1 || (!idx_out && want_pairs)) { idx_local.resize(n ? n : 1); idx = idx_local.data(); }
  else idx = idx_out;
  rc = tdtk_scan_pairs(t, A, sc, pmode, maxd2, want, lum_D, idx, sums);
  if (rc) { tdtk_scan_destroy(sc); return rc; }
  sums->n_queries = n;
  if (want_pairs && n && sums->n) {
    // the PtPair(s, t, normal) list of searchTree.cc:147-180 in query order, built on the device:
    // found flags (caller order) -> prefix sum -> one kernel writes the compact lists
    Ctx* c;
    if ((rc = get_ctx(t->device, &c))) { tdtk_scan_destroy(sc); return rc; }
    hipStream_t s = c->stream;
    const size_t np = sums->n;
    const size_t tmpb = scan_u32_temp_bytes(n + 1);
    if ((rc = c->ws[WS_CELL].ensure(2 * (n + 1) * sizeof(uint32_t))) || (rc = c->ws[WS_TMPB].ensure(tmpb + 256)) ||
        (rc = c->ws[WS_TMPA].ensure(9 * np * sizeof(double)))) { tdtk_scan_destroy(sc); return rc; }
    uint32_t* flags = c->ws[WS_CELL].as<uint32_t>();
    uint32_t* slot = flags + (n + 1);
    double* d_p1 = c->ws[WS_TMPA].as<double>();
    double* d_p2 = d_p1 + 3 * np;
    double* d_pn = d_p2 + 3 * np;
    auto bail = [&](int code) { tdtk_scan_destroy(sc); return code; };
    if (hipMemsetAsync(flags + n, 0, sizeof(uint32_t), s) != hipSuccess) return bail(TDTK_EDEVICE);
    if (launch_found_flags(c->ws[WS_KPOS].as<int>(), sc->d_order, n, flags, s) != hipSuccess) return bail(TDTK_EDEVICE);
    if (launch_scan_u32(flags, slot, n + 1, c->ws[WS_TMPB].p, tmpb, s) != hipSuccess) return bail(TDTK_EDEVICE);
    PairListArgs pa{};
    pa.T = t->dev;
    pa.x = sc->x; pa.y = sc->y; pa.z = sc->z;
    const bool use_n = sc->nx && (pmode != 0 || (want & TDTK_WANT_NAPX));
    pa.nx = use_n ? sc->nx : nullptr; pa.ny = use_n ? sc->ny : nullptr; pa.nz = use_n ? sc->nz : nullptr;
    pa.kpos = c->ws[WS_KPOS].as<int>();
    pa.order = sc->d_order; pa.slot = slot; pa.n = n;
    std::memcpy(pa.A.m, A, sizeof pa.A.m);
    m4inv(A, pa.inv.m);
    pa.p1 = p1_out ? d_p1 : nullptr; pa.p2 = p2_out ? d_p2 : nullptr; pa.pn = pn_out ? d_pn : nullptr;
    if (pn_out && !use_n && hipMemsetAsync(d_pn, 0, 3 * np * sizeof(double), s) != hipSuccess) return bail(TDTK_EDEVICE);
    if (use_n || !pn_out) { /* kernel writes pn when normals take part */ }
    else pa.pn = nullptr;
    if (launch_pair_list(pa, pmode, s) != hipSuccess) return bail(TDTK_EDEVICE);
    bool ok = true;
    if (p1_out) ok &= hipMemcpyAsync(p1_out, d_p1, 3 * np * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess;
    if (p2_out) ok &= hipMemcpyAsync(p2_out, d_p2, 3 * np * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess;
    if (pn_out) ok &= hipMemcpyAsync(pn_out, d_pn, 3 * np * sizeof(double), hipMemcpyDeviceToHost, s) == hipSuccess;
    ok &= hipStreamSynchronize(s) == hipSuccess;
    if (!ok) { set_error("pair list copy failed"); return bail(TDTK_EDEVICE); }
  }
  tdtk_scan_destroy(sc);
  if (rnd > 1 && idx_out)
    for (size_t i = 0; i < n; i++) idx_out[kept_pos[i]] = idx[i];
  return TDTK_OK;
}

// ---- diagnostics: is the resident tree bit-identical to the host builder's? ----------------------
int tdtk_tree_verify(const tdtk_tree* t, uint64_t mismatches[4])
{
  if (!t || !mismatches) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(t->device, &c);
  if (rc) return rc;
  // The resident arrays, with the padding of the buckets to whole groups (tree_pad_buckets) undone: the leaves in the
  // order of their (padded) starts give back the packed array and the references the builder emitted.
  std::vector<KdPoint> padded(t->Mp);
  HIPCHK(hipMemcpy(padded.data(), t->d_pts, padded.size() * sizeof(KdPoint), hipMemcpyDeviceToHost));
  std::vector<KdNode> dn(t->info.n_internal);
  std::vector<LeafEntry> dl;
  if (!dn.empty()) HIPCHK(hipMemcpy(dn.data(), t->d_nodes, dn.size() * sizeof(KdNode), hipMemcpyDeviceToHost));
  if (t->d_leaf) { dl.resize(t->info.n_leaves); HIPCHK(hipMemcpy(dl.data(), t->d_leaf, dl.size() * sizeof(LeafEntry), hipMemcpyDeviceToHost)); }
  std::vector<KdPoint> pts;
  uint64_t group_errors = 0;
  // the search-side copies of a node's split half (the hot record's, and the 16-byte one the deferred quick check reads) say
  // what the node says
  if (t->d_hot && !dn.empty()) {
    std::vector<KdHot> hot(dn.size());
    HIPCHK(hipMemcpy(hot.data(), t->d_hot, hot.size() * sizeof(KdHot), hipMemcpyDeviceToHost));
    struct Half { double splitval; uint32_t c1, c2; };
    std::vector<Half> half;
    if (t->d_split) { half.resize(dn.size()); HIPCHK(hipMemcpy(half.data(), t->d_split, half.size() * sizeof(Half), hipMemcpyDeviceToHost)); }
    for (size_t i = 0; i < dn.size(); i++) {
      if (std::memcmp(&hot[i].splitval, &dn[i].splitval, 8) != 0 || hot[i].c1 != dn[i].c1 || hot[i].c2 != dn[i].c2) group_errors++;
      if (!half.empty() && (std::memcmp(&half[i].splitval, &dn[i].splitval, 8) != 0 || half[i].c1 != dn[i].c1 || half[i].c2 != dn[i].c2)) group_errors++;
    }
  }
  if (t->d_grp) {
    const uint32_t cbv = t->dev.cb, cm = (cbv >= 32) ? 0xFFFFFFFFu : ((1u << cbv) - 1u);
    struct Run { uint32_t start, count; uint32_t* ref; LeafEntry* le; };
    std::vector<Run> runs;
    for (KdNode& nd : dn)
      for (uint32_t* r : {&nd.c1, &nd.c2})
        if (*r & REF_LEAF) {
          const uint32_t v = *r & REF_VAL;
          if (t->d_leaf) runs.push_back({(uint32_t)dl[v].start, (uint32_t)dl[v].count, nullptr, &dl[v]});
          else runs.push_back({v >> cbv, v & cm, r, nullptr});
        }
    std::sort(runs.begin(), runs.end(), [](const Run& a, const Run& b) { return a.start < b.start; });
    std::vector<float> shadow((size_t)t->Mp * 3);
    HIPCHK(hipMemcpy(shadow.data(), t->d_grp, shadow.size() * sizeof(float), hipMemcpyDeviceToHost));
    // the 16-bit shadow (TreeDev::q16): every slot's grid indices recomputed here with the grid the tree carries
    std::vector<uint16_t> q16;
    if (t->d_q16) {
      q16.resize((size_t)t->Mp * 3);
      HIPCHK(hipMemcpy(q16.data(), t->d_q16, q16.size() * sizeof(uint16_t), hipMemcpyDeviceToHost));
    }
    auto grid_index = [&](double v, int a) -> uint16_t {       // kernels.hip: q16_index
      const double u = (v - t->q_lo[a]) * t->q_scale + 0.5;
      int i = (!(u >= 0.0)) ? -32768 : ((!(u < 65536.0)) ? 32767 : (int)u - 32768);
      return (uint16_t)(i & 0xFFFF);
    };
    pts.reserve(t->M);
    uint32_t expect = 0;
    for (const Run& r : runs) {
      if (r.start != expect || (r.start & 3u) || r.count == 0 || (size_t)r.start + r.count > t->Mp) { group_errors++; break; }
      const uint32_t packed = (uint32_t)pts.size(), ng = (r.count + 3u) >> 2;
      for (uint32_t j = 0; j < 4 * ng; j++) {
        const KdPoint& P = padded[r.start + j];
        const KdPoint& L = padded[r.start + std::min(j, r.count - 1)];
        if (j < r.count) pts.push_back(P);
        else if (std::memcmp(&P, &L, sizeof P) != 0) group_errors++;     // a pad slot repeats the bucket's last point
        const size_t g = (r.start + j) >> 2, k = j & 3u;
        if (shadow[g * 12 + k] != (float)L.x || shadow[g * 12 + 4 + k] != (float)L.y || shadow[g * 12 + 8 + k] != (float)L.z) group_errors++;
        if (!q16.empty()) {
          // two slots per 12 bytes: { (x0, y0), (x1, y1), (z0, z1) } as six uint16
          const size_t pr = (size_t)(r.start + j) >> 1, hi = (r.start + j) & 1u;
          const uint16_t* w = &q16[pr * 6];
          if (w[2 * hi] != grid_index(L.x, 0) || w[2 * hi + 1] != grid_index(L.y, 1) || w[4 + hi] != grid_index(L.z, 2)) group_errors++;
        }
      }
      expect = r.start + 4 * ng;
      if (r.le) r.le->start = (int32_t)packed;
      else *r.ref = (*r.ref & ~REF_VAL) | (packed << cbv) | r.count;
    }
    if (expect != t->Mp || pts.size() != t->M) group_errors++;
    if (group_errors) { mismatches[0] = mismatches[1] = mismatches[2] = 0; mismatches[3] = group_errors; return TDTK_OK; }
  } else {
    pts = padded;
    if (group_errors) { mismatches[0] = mismatches[1] = mismatches[2] = 0; mismatches[3] = group_errors; return TDTK_OK; }
  }
  // recover the caller's array from the resident points (each carries its caller index)
  std::vector<double> xyz(3 * t->M);
  for (size_t k = 0; k < t->M; k++) {
    const size_t o = (size_t)pts[k].orig;
    if (o >= t->M) { mismatches[0] = mismatches[1] = mismatches[2] = 0; mismatches[3] = 1; return TDTK_OK; }
    xyz[3 * o] = pts[k].x; xyz[3 * o + 1] = pts[k].y; xyz[3 * o + 2] = pts[k].z;
  }
  HostTree H;
  std::string err;
  if (!build_tree(xyz.data(), t->M, t->bucket, H, err)) { set_error(err); return TDTK_EINVAL; }
  std::vector<KdNode> nodes(H.nodes.size());
  std::vector<double> rr(H.nodes.size());
  mismatches[0] = mismatches[1] = mismatches[2] = mismatches[3] = 0;
  if (H.n_internal != t->info.n_internal || H.n_leaves != t->info.n_leaves || H.max_depth != t->info.max_depth ||
      H.max_leaf_points != t->info.max_leaf_points || H.root_ref != t->dev.root_ref || (uint32_t)H.cb != t->dev.cb ||
      H.table_mode != (t->d_leaf != nullptr))
    mismatches[3] = 1;
  if (mismatches[3] == 0) {
    if (!nodes.empty()) {
      nodes = dn;    // with the references pointing into the packed array again
      HIPCHK(hipMemcpy(rr.data(), t->d_r, rr.size() * sizeof(double), hipMemcpyDeviceToHost));
    }
    for (size_t i = 0; i < nodes.size(); i++) {
      const KdNode &a = nodes[i], &b = H.nodes[i];
      // == on doubles: +0 and -0 compare equal (the sign of a zero box centre never decides anything)
      if (!(a.cx == b.cx && a.cy == b.cy && a.cz == b.cz && a.hx == b.hx && a.hy == b.hy && a.hz == b.hz &&
            a.splitval == b.splitval && a.c1 == b.c1 && a.c2 == b.c2)) {
        mismatches[0]++;
        if (kLab && lab_env("TDTK_VERIFY_DUMP"))
          fprintf(stderr, "VERIFY node %zu: dev c(%.17g %.17g %.17g) h(%.17g %.17g %.17g) split %.17g c1 %08x c2 %08x\n"
                          "            host c(%.17g %.17g %.17g) h(%.17g %.17g %.17g) split %.17g c1 %08x c2 %08x\n",
                  i, a.cx, a.cy, a.cz, a.hx, a.hy, a.hz, a.splitval, a.c1, a.c2, b.cx, b.cy, b.cz, b.hx, b.hy, b.hz, b.splitval, b.c1, b.c2);
      }
      if (rr[i] != H.node_r[i]) mismatches[1]++;
    }
    for (size_t i = 0; i < pts.size(); i++)
      if (!(pts[i].x == H.pts[i].x && pts[i].y == H.pts[i].y && pts[i].z == H.pts[i].z && pts[i].orig == H.pts[i].orig))
        mismatches[2]++;
    if (H.table_mode) {
      std::vector<LeafEntry> lt = dl;
      // leaf ids may be numbered differently; compare through the references instead
      auto leaf_of = [&](const std::vector<LeafEntry>& tab, uint32_t ref) { return tab[ref & REF_VAL]; };
      for (size_t i = 0; i < nodes.size(); i++)
        for (int k = 0; k < 2; k++) {
          const uint32_t ra = k ? nodes[i].c2 : nodes[i].c1, rb = k ? H.nodes[i].c2 : H.nodes[i].c1;
          if ((ra & REF_LEAF) != (rb & REF_LEAF)) continue;
          if (ra & REF_LEAF) {
            const LeafEntry x = leaf_of(lt, ra), y = leaf_of(H.leaf_tab, rb);
            if (x.start != y.start || x.count != y.count) mismatches[3]++;
          }
        }
    }
  }
  return TDTK_OK;
}

// ---- host-only diagnostics -----------------------------------------------------------------
int tdtk_host_tree_layout(const double* xyz, size_t M, int bucket_size, int32_t* perm_out, uint64_t stats[4])
{
  HostTree H;
  std::string err;
  if (!build_tree(xyz, M, bucket_size, H, err)) { set_error(err); return TDTK_EINVAL; }
  if (perm_out)
    for (size_t k = 0; k < M; k++) perm_out[k] = H.pts[k].orig;
  if (stats) { stats[0] = H.n_internal; stats[1] = H.n_leaves; stats[2] = H.max_depth; stats[3] = H.max_leaf_points; }
  return TDTK_OK;
}
int tdtk_host_m4inv(const double in[16], double out[16]) { return m4inv(in, out); }
void tdtk_host_mmult(const double a[16], const double b[16], double out[16]) { mmult(a, b, out); }
void tdtk_host_euler_to_matrix4(const double rPos[3], const double rPosTheta[3], double out[16]) { euler_to_matrix4(rPos, rPosTheta, out); }
void tdtk_host_matrix4_to_euler(const double in[16], double rPosTheta[3], double rPos[3]) { matrix4_to_euler(in, rPosTheta, rPos); }
void tdtk_host_quat_to_matrix4(const double quat[4], const double t[3], double out[16]) { quat_to_matrix4(quat, t, out); }
void tdtk_host_matrix4_to_quat(const double in[16], double quat[4], double t[3]) { matrix4_to_quat_t(in, quat, t); }

// ---- minimizers / solves ------------------------------------------------------------------
int tdtk_align(int algo, const tdtk_pair_sums* sums, double alignxf[16], double* rms)
{
  if (!sums || !alignxf) { set_error("NULL argument"); return TDTK_EINVAL; }
  std::string err;
  int rc = align_from_sums(algo, *sums, alignxf, rms, err);
  if (rc) set_error(err);
  return rc;
}

int tdtk_solve_spd(const double* G, const double* B, int n, double* x)
{
  if (!G || !B || !x || n <= 0) { set_error("bad argument"); return TDTK_EINVAL; }
  std::vector<double> xs(n);
  if (!solve_spd_dense(n, G, B, xs.data(), 0.00001)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
  std::memcpy(x, xs.data(), sizeof(double) * n);
  return TDTK_OK;
}

// ---- icp6D::match -----------------------------------------------------------------------------
// Diagnostics (off by default): while on, every pass of tdtk_icp_match also hashes its correspondences on the device (k_idx_hash:
// the K5 hash of SURVEY 8(c) over the caller-order index array) -- the loop's indices can then be compared with a CPU run of the reference's search
// iteration by iteration at sizes where downloading a million indices per iteration is not an option.
static std::atomic<int> g_icp_hashes{0};
constexpr int ICP_HASH_CAP = 1024;
// the words of the calling thread's last tdtk_icp_match, whatever device it ran on (no context is looked up, none created)
static thread_local std::vector<uint64_t> t_last_hashes;
int tdtk_icp_index_hashes(int on)
{
  if (on < 0) return g_icp_hashes.load(std::memory_order_relaxed);     // a question, not a switch
  return g_icp_hashes.exchange(on ? 1 : 0);
}
int tdtk_icp_last_hashes(uint64_t* out, int cap, int* n_out)
{
  if (!n_out || (cap > 0 && !out) || cap < 0) { set_error("bad argument"); return TDTK_EINVAL; }
  const int n = (int)std::min<size_t>(t_last_hashes.size(), (size_t)cap);
  for (int i = 0; i < n; i++) out[i] = t_last_hashes[(size_t)i];
  *n_out = (int)t_last_hashes.size();
  return TDTK_OK;
}

// One -R pass's keep-mask (searchTree.cc:116-118: `if (rnd > 1 && rand(rnd) != 0) continue;` for i = 0 .. N-1): one std::rand()
// per point in the caller's index order, drawn here, now -- never ahead of the pass that uses it, so the process's random
// stream is consumed exactly as a serial build of the reference consumes it (SURVEY N-d) -- and sent as N / 8 bytes; a
// small kernel turns it into one byte per query at its sorted position, which the search kernels read (SearchArgs::skip).
static int draw_keep_mask(Ctx* c, const tdtk_scan* data, int rnd, const unsigned char** skip_out)
{
  const size_t N = data->N, nbytes = (N + 7) / 8;
  int rc;
  if ((rc = c->d_mask.ensure(nbytes + 16)) || (rc = c->d_skip.ensure(N + 16))) return rc;
  c->h_mask.assign(nbytes, 0);
  for (size_t i = 0; i < N; i++) {
    const int r = (int)((double)rnd * (double)std::rand() / (RAND_MAX + 1.0));     // globals.icc:607-610
    if (r == 0) c->h_mask[i >> 3] |= (unsigned char)(1u << (i & 7u));
  }
  void* staged = nullptr;
  // (the staging buffer is reused by the next pass's mask: the copy below has been consumed by then -- every pass ends with a wait)
  if ((rc = stage_pinned(c, c->h_mask.data(), nbytes, &staged))) return rc;
  HIPCHK(hipMemcpyAsync(c->d_mask.p, staged, nbytes, hipMemcpyHostToDevice, c->stream));
  HIPCHK(launch_skip_from_mask(c->d_mask.as<unsigned char>(), data->d_order, N, c->d_skip.as<unsigned char>(), c->stream));
  *skip_out = c->d_skip.as<unsigned char>();
  return TDTK_OK;
}

#ifdef TDTK_LAB
// ---- lab: the ICP loop without the host in it (loop_dev.h; round 6, a measured negative: NEGATIVES.md) -----------------------
// For the batches of the four-lanes-per-query kernel (a real scan after -r reduction: tens of thousands of points) an
// iteration is two short kernels and a host round trip that costs as much as either.  Here the host only FEEDS the stream:
// launch k searches for iteration k after its every workgroup has made iteration k-1's solve for itself (kernels.hip:
// loop_prologue); `ahead` launches are enqueued beyond the one whose row the host is waiting for, and the host follows the
// rows of the record in pinned memory, topping the queue up.  Launches still queued when the loop ends see the flag and return
// at once.  -a 1 (QUAT), closest-point pairing, no -R, no diagnostics, at most 512 rows of pair sums (~32K points); everything
// else keeps the stepped loop below.  TDTK_ICP_DEVICE_LOOP=1 switches it on (lab library only).
constexpr int ICP_LOOP_RING = 64;
static int icp_loop_ahead()          // (read per match: a getenv is nothing beside a match, and tests flip it)
{
  const char* on = getenv("TDTK_ICP_DEVICE_LOOP");
  if (!(on && on[0] == '1')) return 0;
  const char* e = lab_env("TDTK_LOOP_AHEAD");
  const int a = e ? atoi(e) : 3;
  return std::max(1, std::min(ICP_LOOP_RING / 2, a));
}
struct IcpLoopOut {
  int iter = 0;                    // rows consumed = iterations whose solve the host has seen
  bool finished = false;           // the loop ended on the device (converged / last iteration / too few pairs)
  bool few_pairs = false;
  bool need_host = false;          // iteration `iter`'s sums are in acc / shift: the host solves it and carries on stepping
  int converged = 0;
  double ret = 0.0, prev_ret = 0.0;
  bool have_pending = false;
  double pend[16];
  double acc[ACC_TOTAL], shift[3];
};
static hipError_t await_row(const double* row, hipStream_t s)
{
  const volatile uint64_t* w = reinterpret_cast<const volatile uint64_t*>(row + ICP_ROW_READY);
  const auto t0 = std::chrono::steady_clock::now();
  bool yielding = false;
  for (uint32_t spins = 1;; spins++) {
    if (*w != SUMS_ARMED) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
    if (yielding) {
      sched_yield();
      if ((spins & 0x3Fu) != 0) continue;
    } else {
      __builtin_ia32_pause();
      if ((spins & 0x3Fu) == 0 &&
          std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 200.0) yielding = true;
      if ((spins & 0x3FFu) != 0) continue;
    }
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) return (*w != SUMS_ARMED) ? hipSuccess : hipErrorUnknown;   // (drained and the row never came)
    if (q != hipErrorNotReady) return q;
  }
}
// the loop's device block at its start: history and flags zero, the record's ring named and armed
static int icp_loop_reset(Ctx* c)
{
  int rc;
  if ((rc = c->d_loop.ensure(sizeof(IcpLoopDev)))) return rc;
  if (!c->h_loop) HIPCHK(hipHostMalloc((void**)&c->h_loop, sizeof(double) * ICP_ROW * ICP_LOOP_RING + sizeof(IcpLoopDev), hipHostMallocCoherent));
  for (int r = 0; r < ICP_LOOP_RING; r++)
    reinterpret_cast<volatile uint64_t*>(c->h_loop + (size_t)r * ICP_ROW)[ICP_ROW_READY] = SUMS_ARMED;
  IcpLoopDev* init = reinterpret_cast<IcpLoopDev*>(c->h_loop + (size_t)ICP_ROW * ICP_LOOP_RING);    // (pinned: behind the ring)
  std::memset(init, 0, sizeof *init);
  init->rows_host = c->h_loop;
  init->row_cap = ICP_LOOP_RING;
  HIPCHK(hipMemcpyAsync(c->d_loop.p, init, sizeof *init, hipMemcpyHostToDevice, c->stream));
  return TDTK_OK;
}
static int icp_device_loop(Ctx* c, const tdtk_tree* model, const double* A16, tdtk_scan* data, const tdtk_icp_params* prm,
                           bool warm_ok, double* data_transMat, double* data_dalignxf, tdtk_icp_result* res, double* trace,
                           int trace_cap, IcpLoopOut& o)
{
  hipStream_t s = c->stream;
  const size_t N = data->N;
  const int ahead = icp_loop_ahead(), max_iter = prm->max_num_iterations;
  int rc;
  if ((rc = c->ws[WS_KPOS].ensure(N * sizeof(int)))) return rc;
  const uint32_t rows = search_fused_rows(N);
  const size_t pstride = (size_t)rows * ICP_LOOP_COLS;    // two buffers of rows ([column][row]): launch k writes buffer k & 1, reads the other
  if ((rc = c->ws[WS_PART].ensure(2 * pstride * sizeof(double)))) return rc;
  if ((rc = icp_loop_reset(c))) return rc;
  Mat4 A, inv;
  std::memcpy(A.m, A16, sizeof A.m);
  m4inv(A16, inv.m);  // searchTree.cc:110
  double sh[3];
  sh[0] = model->centre[0] * A16[0] + model->centre[1] * A16[4] + model->centre[2] * A16[8] + A16[12];
  sh[1] = model->centre[0] * A16[1] + model->centre[1] * A16[5] + model->centre[2] * A16[9] + A16[13];
  sh[2] = model->centre[0] * A16[2] + model->centre[1] * A16[6] + model->centre[2] * A16[10] + A16[14];
  SearchArgs sa0{};
  sa0.x = data->x; sa0.y = data->y; sa0.z = data->z;
  sa0.nx = data->nx; sa0.ny = data->ny; sa0.nz = data->nz;
  sa0.n = N;
  sa0.inv = inv; sa0.has_inv = 1;
  sa0.maxd2 = prm->max_dist_match2;
  sa0.kpos = c->ws[WS_KPOS].as<int>();
  sa0.fuse = 4; sa0.A = A;
  for (int k = 0; k < 3; k++) sa0.shift[k] = sh[k];
  sa0.loop = c->d_loop.as<IcpLoopDev>();
  sa0.loop_rows = (int)rows; sa0.loop_max_iter = max_iter; sa0.loop_eps = prm->epsilon_icp;
  const double margin = search_margin(model, prm->max_dist_match2);
  double* const P = c->ws[WS_PART].as<double>();
  // Launch k searches for iteration k; its prologue makes iteration k-1's solve and writes that iteration's row.  Launch
  // max_iter exists for its prologue alone (the last iteration's solve ends the loop before anything is searched).
  int enq = 0;
  for (;;) {
    while (enq <= max_iter && enq < o.iter + 1 + ahead) {
      SearchArgs sa = sa0;
      sa.loop_iter = enq;
      sa.partials = P + (size_t)(enq & 1) * pstride;
      sa.loop_prev = P + (size_t)((enq + 1) & 1) * pstride;
      sa.warm = (enq > 0 && warm_ok) ? 1 : 0;   // WS_KPOS holds the previous pass's hits by the time this launch runs
      sa.margin = sa.warm ? margin : 0.0;
      if ((rc = run_search(c, model, sa, 0, false, s, false))) return rc;
      enq++;
    }
    double* row = c->h_loop + (size_t)(o.iter % ICP_LOOP_RING) * ICP_ROW;
    HIPCHK(await_row(row, s));
    const int status = (int)row[ICP_ROW_STATUS];
    const uint64_t n = (uint64_t)(row[ICP_ROW_N] + 0.5);
    res->last_pairs = n;
    // (either way launch o.iter has searched, i.e. it has applied the transform of the row before: nothing is pending)
    if (status == ICP_ROW_FEW_PAIRS) { o.finished = true; o.few_pairs = true; o.have_pending = false; break; }
    if (status == ICP_ROW_NEED_HOST) {
      o.have_pending = false;
      std::memset(o.acc, 0, sizeof o.acc);
      std::memcpy(o.acc, row + ICP_ROW_ACC, ICP_LOOP_COLS * sizeof(double));
      for (int k = 0; k < 3; k++) o.shift[k] = sh[k];
      o.need_host = true;
      break;
    }
    const double ret = row[ICP_ROW_RMS];
    const double* alignxf = row + ICP_ROW_XF;
    if (!prm->quiet) std::printf("QUAT RMS point-to-point error = %10.7f  using %6llu points\n", ret, (unsigned long long)n);
    if (trace && o.iter < trace_cap) {
      double* tr = trace + (size_t)o.iter * 18;
      tr[0] = (double)n; tr[1] = ret;
      std::memcpy(tr + 2, alignxf, 16 * sizeof(double));
    }
    res->last_rms = ret;
    o.prev_ret = o.ret; o.ret = ret;
    std::memcpy(o.pend, alignxf, sizeof o.pend);
    o.have_pending = true;
    if (data_transMat) mmult(alignxf, data_transMat, data_transMat);  // scan.cc:878-898
    if (data_dalignxf) mmult(alignxf, data_dalignxf, data_dalignxf);
    reinterpret_cast<volatile uint64_t*>(row)[ICP_ROW_READY] = SUMS_ARMED;     // the ring: this slot's next turn
    if (status == ICP_ROW_CONVERGED || status == ICP_ROW_LAST) {
      o.converged = status == ICP_ROW_CONVERGED ? 1 : 0;
      o.finished = true;
      break;                       // (o.iter stays the index of this iteration: what icp6D::match returns)
    }
    o.iter++;
  }
  return TDTK_OK;
}

extern "C" int tdtk_lab_icp_device_solve(int device, const double acc17[17], const double shift[3], double alignxf[16], double* rms, int* status)
{
  if (!acc17 || !shift || !alignxf || !rms || !status) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(device, &c);
  if (rc) return rc;
  hipStream_t s = c->stream;
  if ((rc = c->ws[WS_PART].ensure(ACC_TOTAL * sizeof(double)))) return rc;
  double rowv[ACC_TOTAL] = {0.0};
  std::memcpy(rowv, acc17, ICP_LOOP_COLS * sizeof(double));
  HIPCHK(hipMemcpyAsync(c->ws[WS_PART].p, rowv, sizeof rowv, hipMemcpyHostToDevice, s));
  HIPCHK(hipStreamSynchronize(s));                      // (rowv is pageable: the copy has read it)
  if ((rc = icp_loop_reset(c))) return rc;
  HIPCHK(launch_solve_once(c->ws[WS_PART].as<double>(), 1, shift, c->d_loop.as<IcpLoopDev>(), s));
  HIPCHK(hipStreamSynchronize(s));
  const double* row = c->h_loop;
  std::memcpy(alignxf, row + ICP_ROW_XF, 16 * sizeof(double));
  *rms = row[ICP_ROW_RMS];
  *status = (int)row[ICP_ROW_STATUS];
  return TDTK_OK;
}
#endif

static int icp_match_impl(const tdtk_tree* model, const double model_dalignxf[16], tdtk_scan* data,
                          double data_transMat[16], double data_dalignxf[16], const tdtk_icp_params* prm, int rnd,
                          tdtk_icp_result* res, double* trace, int trace_cap);
int tdtk_icp_match(const tdtk_tree* model, const double model_dalignxf[16], tdtk_scan* data,
                   double data_transMat[16], double data_dalignxf[16], const tdtk_icp_params* prm,
                   tdtk_icp_result* res, double* trace, int trace_cap)
{
  return icp_match_impl(model, model_dalignxf, data, data_transMat, data_dalignxf, prm, 1, res, trace, trace_cap);
}
int tdtk_icp_match_rnd(const tdtk_tree* model, const double model_dalignxf[16], tdtk_scan* data,
                       double data_transMat[16], double data_dalignxf[16], const tdtk_icp_params* prm, int rnd,
                       tdtk_icp_result* res, double* trace, int trace_cap)
{
  return icp_match_impl(model, model_dalignxf, data, data_transMat, data_dalignxf, prm, rnd, res, trace, trace_cap);
}
static int icp_match_impl(const tdtk_tree* model, const double model_dalignxf[16], tdtk_scan* data,
                          double data_transMat[16], double data_dalignxf[16], const tdtk_icp_params* prm, int rnd,
                          tdtk_icp_result* res, double* trace, int trace_cap)
{
  if (!model || !model_dalignxf || !data || !prm || !res) { set_error("NULL argument"); return TDTK_EINVAL; }
  if (prm->max_dist_match2 < 0.0 || prm->max_num_iterations < 0) {
    set_error("ERROR [ICP6D]: max_dist_match and max_num_iterations have to be >= 0");  // icp6D.cc:67-78
    return TDTK_EINVAL;
  }
  const int algo = prm->algo;
  if (algo < TDTK_ALGO_QUAT || algo > TDTK_ALGO_NAPX) {
    set_error("This minimization algorithm is not implemented");
    return TDTK_EINVAL;
  }
  const bool serial_only = algo == TDTK_ALGO_ORTHO || algo == TDTK_ALGO_DUAL || algo == TDTK_ALGO_HELIX ||
                           algo == TDTK_ALGO_LUMEULER || algo == TDTK_ALGO_LUMQUAT || algo == TDTK_ALGO_QUAT_SCALE;
  if ((algo == TDTK_ALGO_LUMEULER || algo == TDTK_ALGO_LUMQUAT) && !data_transMat) {
    set_error("LUMEULER / LUMQUAT need the current scan's transMat");
    return TDTK_EINVAL;
  }
  const int pmode = prm->pairing_mode;
  if ((pmode != 0 || algo == TDTK_ALGO_NAPX) && !data->nx) { set_error("normals required"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(model->device, &c);
  if (rc) return rc;
  std::memset(res, 0, sizeof *res);
  if (prm->max_num_iterations == 0 || data->N == 0) return TDTK_OK;  // icp6D.cc:112-114

  if ((rc = scan_settle(c, data))) return rc;          // queued moves first: "original" and the loop both start from them
  if ((rc = scan_keep_original(c, data))) return rc;   // the loop moves the scan in place
  const unsigned want = (algo == TDTK_ALGO_APX) ? TDTK_WANT_APX
                        : (algo == TDTK_ALGO_NAPX ? TDTK_WANT_NAPX : (serial_only ? TDTK_WANT_MOM2 : 0u));
  double ret = 0.0, prev_ret = 0.0, prev_prev_ret = 0.0;
  double alignxf[16], pend[16];
  bool have_pending = false;
  double nn_total = 0.0, sums_total = 0.0;
  const double t0 = now_ms();
  int iter = 0;
  int converged = 0;
  const char* warm_env = lab_env("TDTK_WARM_START");
  const bool warm_ok = !(warm_env && warm_env[0] == '0');
  const bool hashing = g_icp_hashes.load(std::memory_order_relaxed) != 0;
  c->last_hashes.clear();
  t_last_hashes.clear();
  if (hashing) {
    if ((rc = c->d_hash.ensure(ICP_HASH_CAP * sizeof(unsigned long long)))) return rc;
    HIPCHK(hipMemsetAsync(c->d_hash.p, 0, ICP_HASH_CAP * sizeof(unsigned long long), c->stream));
  }
  bool loop_done = false, resume_with_acc = false;
  double acc[ACC_TOTAL], shift[3];
#ifdef TDTK_LAB
  // lab: the small-scan loop that runs without the host (see icp_device_loop): -a 1, closest points, every point a candidate
  if (icp_loop_ahead() > 0 && algo == TDTK_ALGO_QUAT && pmode == 0 && want == 0u && rnd <= 1 && !hashing && !c->counting &&
      !kernel_timing() && fuse_mode(data->N) == 4 && search_multi_class(data->N) == 10 && search_fused_rows(data->N) <= (uint32_t)ICP_LOOP_MAX_ROWS) {
    IcpLoopOut o;
    if ((rc = icp_device_loop(c, model, model_dalignxf, data, prm, warm_ok, data_transMat, data_dalignxf, res, trace, trace_cap, o))) return rc;
    iter = o.iter;
    ret = o.ret; prev_ret = o.prev_ret;
    have_pending = o.have_pending;
    if (have_pending) std::memcpy(pend, o.pend, sizeof pend);
    converged = o.converged;
    loop_done = o.finished;
    if (o.need_host) {             // a solve that did not come out finite on the device: this iteration's sums, the host's solver
      std::memcpy(acc, o.acc, sizeof acc);
      for (int k = 0; k < 3; k++) shift[k] = o.shift[k];
      resume_with_acc = true;
      // (the launches still queued behind the failed solve return at once: the flag is up; the stepped loop below takes over)
    }
  }
#endif
  for (; !loop_done && iter < prm->max_num_iterations; iter++) {
    prev_prev_ret = prev_ret;
    prev_ret = ret;
    // -R: this pass's candidates (drawn now, in the reference's order)
    const unsigned char* skip = nullptr;
    if (rnd > 1 && (rc = draw_keep_mask(c, data, rnd, &skip))) return rc;
    // the previous iteration's alignxf is applied to the points inside the search kernel
    // (the warm start stays exact under -R: WS_KPOS holds -1 for whoever was not drawn last time -- a cold start --, and for
    // the others a point of THIS tree, whose distance bounds the nearest neighbour's however the scan has moved since)
    if (!resume_with_acc) {
      rc = scan_pass(c, model, model_dalignxf, data, pmode, prm->max_dist_match2, want, nullptr,
                     have_pending ? pend : nullptr, true, acc, shift, iter > 0 && warm_ok, skip);
      if (rc) return rc;
      have_pending = false;
    }
    resume_with_acc = false;
    if (hashing && iter < ICP_HASH_CAP)    // this pass's correspondences are still in the workspace (sorted positions)
      HIPCHK(launch_idx_hash(c->ws[WS_KPOS].as<int>(), data->d_order, model->dev.pts, data->N,
                             c->d_hash.as<unsigned long long>() + iter, c->stream));
    double ms = 0, sms = 0;
    collect_ms(c, &ms, &sms);
    nn_total += ms;
    sums_total += sms;
    tdtk_pair_sums sums;
    finish_sums(acc, shift, data->N, want, &sums);
    res->last_pairs = sums.n;
    if (sums.n > 3) {  // icp6D.cc:235-243 (serial branch semantics)
      std::string err;
      double rms = 0;
      // getAlgorithmID() 3 / 8: alignxf enters as CurrentScan->get_transMat() (icp6D.cc:237-241)
      if (data_transMat) std::memcpy(alignxf, data_transMat, sizeof alignxf);
      rc = align_from_sums(algo, sums, alignxf, &rms, err);
      if (rc == TDTK_ESOLVE) {
        // the reference prints and keeps going with ret = -1 and the unchanged alignxf buffer
        ret = -1.0;
        m4identity(alignxf);
      } else if (rc) { set_error(err); return rc; }
      else ret = rms;
    } else {
      break;
    }
    if (!prm->quiet) {
      static const char* tag[] = {"", "QUAT", "SVD", "ORTHO", "DUALQUAT", "HELIX", "APX", "LUMEULER", "LUMQUAT",
                                  "QUAT SCALE", "APX"};
      std::printf("%s RMS point-to-%s error = %10.7f  using %6llu points\n", tag[algo],
                  algo == TDTK_ALGO_NAPX ? "plane" : "point", ret, (unsigned long long)sums.n);
    }
    if (trace && iter < trace_cap) {
      double* row = trace + (size_t)iter * 18;
      row[0] = (double)sums.n; row[1] = ret;
      std::memcpy(row + 2, alignxf, sizeof alignxf);
    }
    res->last_rms = ret;
    // CurrentScan->transform(alignxf, ...): points lazily (next search or epilogue), matrices now
    std::memcpy(pend, alignxf, sizeof pend);
    have_pending = true;
    if (data_transMat) mmult(alignxf, data_transMat, data_transMat);  // scan.cc:878-898
    if (data_dalignxf) mmult(alignxf, data_dalignxf, data_dalignxf);
    if ((std::fabs(ret - prev_ret) < prm->epsilon_icp && std::fabs(ret - prev_prev_ret) < prm->epsilon_icp) ||
        iter == prm->max_num_iterations - 1) {
      converged = (iter != prm->max_num_iterations - 1) ? 1 : 0;
      break;
    }
  }
  if (have_pending) {
    Mat4 P;
    std::memcpy(P.m, pend, sizeof P.m);
    HIPCHK(launch_transform(data->x, data->y, data->z, data->nx, data->ny, data->nz, data->N, P, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (hashing) {
    const int passes = std::min(ICP_HASH_CAP, std::min(iter + 1, prm->max_num_iterations));
    c->last_hashes.assign((size_t)passes, 0);
    HIPCHK(hipMemcpyAsync(c->last_hashes.data(), c->d_hash.p, (size_t)passes * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    t_last_hashes = c->last_hashes;
  }
  res->iterations = iter;
  res->converged = converged;
  res->total_ms = now_ms() - t0;
  res->nn_ms = nn_total;
  res->sums_ms = sums_total;
  if (!prm->quiet) std::printf("TIME  %ld   ITER %d\n", (long)res->total_ms, iter);
  return TDTK_OK;
}

// ---- lum6DEuler::covarianceEuler -----------------------------------------------------------
int tdtk_lum_link(const tdtk_tree* first, const double first_dalignxf[16], tdtk_scan* second, double maxd2,
                  double C[36], double CD[6], uint64_t* m_out, double* ss_out)
{
  if (!first || !first_dalignxf || !second || !C || !CD) { set_error("NULL argument"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(first->device, &c);
  if (rc) return rc;
  std::memset(C, 0, 36 * sizeof(double));
  std::memset(CD, 0, 6 * sizeof(double));
  if (m_out) *m_out = 0;
  if (ss_out) *ss_out = 0;
  if (second->N == 0) return TDTK_OK;
  double acc[ACC_TOTAL], shift[3];
  rc = scan_pass(c, first, first_dalignxf, second, 0, maxd2, TDTK_WANT_LUM, nullptr, nullptr, true, acc, shift);
  if (rc) return rc;
  collect_ms(c, nullptr);
  const uint64_t m = (uint64_t)(acc[ACC_N] + 0.5);
  if (m_out) *m_out = m;
  if (m <= 2) return TDTK_OK;  // "This case should not occur": zeros (lum6Deuler.cc:234-249)
  const double* L = acc + ACC_L;
  const double sx = L[0], sy = L[1], sz = L[2], xpy = L[3], xpz = L[4], ypz = L[5], xy = L[6], xz = L[7], yz = L[8];
  double MM[36] = {0};
  auto M = [&](int r, int col) -> double& { return MM[(r - 1) * 6 + (col - 1)]; };  // 1-based like newmat
  M(1, 1) = M(2, 2) = M(3, 3) = (double)m;
  M(4, 4) = ypz; M(5, 5) = xpy; M(6, 6) = xpz;
  M(1, 5) = M(5, 1) = -sy; M(1, 6) = M(6, 1) = sz;
  M(2, 4) = M(4, 2) = -sz; M(2, 5) = M(5, 2) = sx;
  M(3, 4) = M(4, 3) = sy;  M(3, 6) = M(6, 3) = -sx;
  M(4, 5) = M(5, 4) = -xz; M(4, 6) = M(6, 4) = -xy; M(5, 6) = M(6, 5) = -yz;
  const double* MZ = L + 9;
  double MMi[36], D[6];
  if (!invert_dense(6, MM, MMi)) { set_error("singular link matrix"); return TDTK_ESOLVE; }
  for (int r = 0; r < 6; r++) {
    double v = 0;
    for (int k = 0; k < 6; k++) v += MMi[r * 6 + k] * MZ[k];
    D[r] = v;
  }
  // second pass over the same correspondences (kpos is still in the workspace)
  double acc2[ACC_TOTAL];
  rc = scan_pass(c, first, first_dalignxf, second, 0, maxd2, TDTK_WANT_LUM, D, nullptr, false, acc2, shift);
  if (rc) return rc;
  double ss = acc2[ACC_LSS] / (2.0 * (double)m - 3.0);
  if (ss_out) *ss_out = ss;
  if (ss < 0.0000000000001) return TDTK_OK;  // lum6Deuler.cc:215-226: zeros
  ss = 1.0 / ss;
  for (int k = 0; k < 36; k++) C[k] = MM[k] * ss;
  for (int k = 0; k < 6; k++) CD[k] = MZ[k] * ss;
  return TDTK_OK;
}

// ---- batched whole-scan passes over a list of links --------------------------------------------
// search + sums of every link, kernels enqueued back to back, one host sync.  Links over small scans are dealt to
// up to 8 auxiliary streams (each pass alone leaves most of the machine idle); big scans keep the one stream.
// acc: [nlinks][ACC_TOTAL] raw columns; shifts: [nlinks][3].
static bool fuse_lum_enabled()
{
  const char* e = lab_env("TDTK_FUSE_LUM");
  return e && e[0] == '1';
}

static int link_batch_max()
{
  const char* e = lab_env("TDTK_LINK_BATCH");   // 0 / 1: the lanes below
  int v = e ? atoi(e) : 128;   // (84 links of 1M points: one launch 11.20 ms per LUM round, 64 + 20: 11.22-11.27, 42 + 42: 11.22)
  if (v > 128) v = 128;
  return v;
}

// The 17 sums of a lum6DEuler link added up by the search waves themselves, each over its own slab after its last query
// (k_search_refill_multi<.., 5>): the pass over (x, y, z, hit, pts[hit]) that k_accum_multi makes disappears into the
// search launch -- 84 links of 1M points: search 8.9 -> 9.5 ms, pair sums 1.04 -> 0, LUM round 10.95 -> 10.55 ms.  The rows
// of partial sums are then the slabs, so every such launch uses one slab length and a lone link takes this path too: a
// link's sums do not depend on how many links (or ranks) share the work.  TDTK_LINK_FUSE=0: k_accum_multi.
static bool link_sums_in_search(const Ctx* c, unsigned want, size_t maxN)
{
  const char* e = lab_env("TDTK_LINK_FUSE");
  if (e && e[0] == '0') return false;
  // (experiments with the slab layout change the rows; TDTK_LINK_FUSE=2 keeps the sums inside for such sweeps)
  if ((lab_env("TDTK_REFILL_QPW") || lab_env("TDTK_LINK_PHASES")) && !(e && e[0] == '2')) return false;
  const int th = search_multi_thresh(maxN);
  return !c->counting && want == (TDTK_WANT_LUM | ACC_WANT_NO_CROSS) && search_multi_class(maxN) == 20 && (th == 16 || th == 32);
}

// All links of the call in launches of up to `gb` links each: ONE search launch, one pair-sum launch and one final
// reduction per group (k_search_refill_multi and friends: workgroups of link k+1 fill the tail of link k), everything on
// the context's stream -- what the lanes below get from three streams, without depending on how the runtime maps streams
// to hardware queues, and with a handful of launches instead of three per link.
static int links_device_pass_batched(Ctx* c, int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                                     tdtk_scan* const* second, double maxd2, unsigned want, double* d_out,
                                     const std::vector<double>& shifts, int gb)
{
  hipStream_t s = c->stream;
  int rc;
  size_t maxN = 0;
  int max_need = 0;
  for (int i = 0; i < nlinks; i++) {
    maxN = std::max(maxN, second[i]->N);
    max_need = std::max(max_need, (int)first[i]->info.max_depth - 1 - search_lds_depth());
  }
  const int G = std::min(gb, nlinks);
  // A wave hands out each piece of its slab with the queries that were expensive in the previous pass of the same link
  // first, as the single pass does (k_search_refill, ORDER): 84 links of 1M points 9.18 -> 8.91 ms per LUM round.  The
  // order changes which lanes share a trip, never a result.  TDTK_LINK_ORDERED=0: in slab order.
  const bool ordered_env = [] { const char* e = lab_env("TDTK_LINK_ORDERED"); return !(e && e[0] == '0'); }();   // (per call: tests flip it)
  const bool ordered = ordered_env && !c->counting && search_multi_class(maxN) == 20;
  const bool fuse_links = link_sums_in_search(c, want, maxN);
  if (ordered) {
    while ((int)c->link_costs.size() < nlinks) c->link_costs.emplace_back(new LinkCost);
    for (int p = 0; p < nlinks; p++) {
      LinkCost* lc = c->link_costs[p].get();
      const void* before = lc->cost.p;
      if ((rc = lc->cost.ensure(maxN))) return rc;
      if (lc->cost.p != before) lc->tree = nullptr;
    }
  }
  while ((int)c->slots.size() < G) c->slots.emplace_back(new Lane);
  for (int g = 0; g < G; g++) {          // every buffer at its final size before anything is enqueued
    Lane* sl = c->slots[g].get();
    {
      const void* before = sl->kpos.p;
      if ((rc = sl->kpos.ensure(maxN * sizeof(int)))) return rc;
      if (sl->kpos.p != before) sl->k_tree = sl->k_scan = 0;
    }
    {
      size_t rows = accum_grid(maxN);
      if (fuse_links) {
        SearchArgs probe{};
        probe.n = maxN;
        rows = std::max<size_t>(rows, search_multi_prepare(probe, G, true));
      }
      if ((rc = sl->part.ensure(rows * ACC_TOTAL * sizeof(double)))) return rc;
    }
    if (max_need > 0) {
      // the stack overflow area of a batch: one column per lane of ITS grid (a tenth of what the stand-alone kernels'
      // common area needs, and there are up to 128 of these)
      SearchArgs probe{};
      probe.n = maxN;
      const size_t lanes = (size_t)std::max(search_multi_prepare(probe, 1, fuse_links), search_multi_prepare(probe, G, fuse_links)) * 256;   // the shortest slab any
                                                                                                        // group gets; 256: the widest workgroup
      if ((rc = sl->ovf_m2.ensure(lanes * max_need * sizeof(double)))) return rc;
      if ((rc = sl->ovf_ref.ensure(lanes * max_need * sizeof(uint32_t)))) return rc;
    }
  }
  const int ngroups = (nlinks + G - 1) / G;
  // Launch order: links that search the SAME tree next to each other (a chain link i -> i+1 and the loop closures out
  // of scan i), so that the tree one of them has pulled into the L2s / the Infinity Cache is still there for the next;
  // otherwise the caller's order.  The workgroups of a launch are dispatched in table order, so position p of the table
  // is what runs p-th; every link still writes its own row of d_out (TDTK_LINK_ORDER=0: the caller's order).
  std::vector<int> ord(nlinks);
  for (int i = 0; i < nlinks; i++) ord[i] = i;
  {
    static const bool keep = [] { const char* e = lab_env("TDTK_LINK_ORDER"); return e && e[0] == '0'; }();
    if (!keep) {
      std::map<const tdtk_tree*, int> seen;
      std::vector<int> key(nlinks);
      for (int i = 0; i < nlinks; i++) key[i] = seen.emplace(first[i], i).first->second;   // first link that uses this tree
      std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return key[a] < key[b]; });
    }
  }
  // Lazy scan moves (tdtk_scan::pending): the persistent-lane launch carries them out itself; the small-batch kernels do
  // not, their scans are moved first.  Everything that can fail is done before the first scan's state changes.
  const bool lazy = search_multi_class(maxN) == 20;
  const bool link_warm = [] { const char* e = lab_env("TDTK_LINK_WARM"); return !(e && e[0] == '0'); }();
  size_t nmat_max = 0;
  if (!lazy) {
    if ((rc = scans_settle(c, second, nlinks))) return rc;
  } else {
    {   // a chain longer than the launch applies itself (a scan no link of this rank has read for rounds): one pass now
      std::vector<const tdtk_scan*> longc;
      for (int i = 0; i < nlinks; i++)
        if (second[i]->pending.size() > (size_t)SEARCH_LAZY_MAX) longc.push_back(second[i]);
      if (!longc.empty() && (rc = scans_settle(c, longc.data(), (int)longc.size()))) return rc;
    }
    {   // a copy of its own for every reader of a moving scan but the first of its launch (which gets the spare arrays)
      std::map<const tdtk_scan*, int> first_in;      // scan -> the launch that moves it
      for (int p = 0; p < nlinks; p++) {
        const tdtk_scan* sc = second[ord[p]];
        if (sc->pending.empty()) continue;
        const int gi = p / G;
        const auto it = first_in.find(sc);
        if (it == first_in.end()) first_in[sc] = gi;
        else if (it->second == gi && (rc = c->slots[p - gi * G]->moved.ensure(3 * maxN * sizeof(double)))) return rc;
      }
    }
    for (int i = 0; i < nlinks; i++) {
      tdtk_scan* sc = second[i];
      if (sc->pending.empty()) continue;
      nmat_max += sc->pending.size();
      if ((rc = scan_ensure_spare(sc))) return rc;
    }
  }
  // one table for the whole call: per link its search and pair-sum arguments and final descriptor, per group the two
  // workgroup-offset lists, then the chains of queued moves
  const size_t o_sa = 0, o_aa = o_sa + sizeof(SearchArgs) * nlinks, o_fd = o_aa + sizeof(AccumArgs) * nlinks,
               o_sb = o_fd + sizeof(FinalDesc) * nlinks, o_ab = o_sb + sizeof(uint32_t) * (size_t)(nlinks + ngroups),
               o_mv = ((o_ab + sizeof(uint32_t) * (size_t)(nlinks + ngroups) + 127) / 128) * 128,
               total = o_mv + nmat_max * sizeof(Mat4);
  if ((rc = c->multi_args.ensure(total))) return rc;
  if ((rc = stage_reserve(c, total))) return rc;
  std::vector<char> tab(total, 0);
  Mat4* hmv = reinterpret_cast<Mat4*>(tab.data() + o_mv);
  const Mat4* dmv = reinterpret_cast<const Mat4*>(static_cast<char*>(c->multi_args.p) + o_mv);
  size_t nmat = 0;
  SearchArgs* hsa = reinterpret_cast<SearchArgs*>(tab.data() + o_sa);
  AccumArgs* haa = reinterpret_cast<AccumArgs*>(tab.data() + o_aa);
  FinalDesc* hfd = reinterpret_cast<FinalDesc*>(tab.data() + o_fd);
  uint32_t* hsb = reinterpret_cast<uint32_t*>(tab.data() + o_sb);
  uint32_t* hab = reinterpret_cast<uint32_t*>(tab.data() + o_ab);
  std::vector<uint32_t> s_total(ngroups), a_total(ngroups);
  struct Held { Lane* sl; uint64_t tree, scan; size_t n; };
  std::vector<Held> held;
  if (c->counting) { for (int i = 0; i < nlinks; i++) c->counted_queries += second[i]->N; }
  for (int gi = 0; gi < ngroups; gi++) {
    const int l0 = gi * G, l1 = std::min(nlinks, l0 + G);
    uint32_t sb = 0, ab = 0;
    std::map<tdtk_scan*, std::pair<size_t, int>> moving;   // scans this launch moves: where their chain sits, its length
    for (int p = l0; p < l1; p++) {
      const int i = p, li = ord[p];      // i: position in the tables; li: the caller's link
      const tdtk_tree* t = first[li];
      tdtk_scan* data = second[li];
      bool owner = false;
      if (lazy && !data->pending.empty() && moving.find(data) == moving.end()) {
        // the first link of the launch that reads the scan owns its update
        moving[data] = std::make_pair(nmat, (int)data->pending.size());
        for (const Mat4& m : data->pending) hmv[nmat++] = m;
        owner = true;
      }
      const auto mv = moving.find(data);
      Lane* sl = c->slots[i - l0].get();
      const double* A16 = first_dalignxf + 16 * (size_t)li;
      Mat4 A, inv;
      std::memcpy(A.m, A16, sizeof A.m);
      m4inv(A16, inv.m);
      SearchArgs sa{};
      sa.T = t->dev;
      sa.x = data->x; sa.y = data->y; sa.z = data->z;
      sa.n = data->N; sa.inv = inv; sa.has_inv = 1; sa.maxd2 = maxd2;
      sa.kpos = sl->kpos.as<int>();
      // the previous pass of this very link left its hits here (graph-SLAM rounds repeat their links): each search starts from
      // its previous hit, a point of this tree whatever the scans have done since (k_search's warm start; same index, same d2)
      sa.warm = (link_warm && !c->count_cold && sl->k_tree == t->uid && sl->k_scan == data->uid && sl->k_n == data->N) ? 1 : 0;
      sa.margin = sa.warm ? search_margin(t, maxd2) : 0.0;
      // (what the slot will hold is written down once every launch of the call is enqueued: a call that fails on the way must
      //  not leave a slot named after hits that were never written)
      sl->k_tree = sl->k_scan = 0;
      held.push_back({sl, t->uid, data->uid, data->N});
      const int need = (int)t->info.max_depth - 1 - search_lds_depth();
      if (need > 0) { sa.ovf_m2 = sl->ovf_m2.as<double>(); sa.ovf_ref = sl->ovf_ref.as<uint32_t>(); }
      if (c->counting) sa.counters = c->d_counters.as<unsigned long long>();
      const uint32_t nb = search_multi_prepare(sa, l1 - l0, fuse_links);
      if (ordered) {
        LinkCost* lc = c->link_costs[p].get();
        sa.cost = lc->cost.as<unsigned char>();
        sa.use_cost = (lc->tree == (const void*)t && lc->scan == (const void*)data && lc->n == data->N) ? 1 : 0;
        lc->tree = t; lc->scan = data; lc->n = data->N;
      }
      if (fuse_links) { sa.fuse = 5; sa.A = A; sa.partials = sl->part.as<double>(); }
      if (mv != moving.end()) {
        // the link's waves move their slabs first: out of the scan's arrays into the link's own copy (SearchArgs::moves)
        sa.moves = dmv + mv->second.first; sa.nmoves = mv->second.second;
        sa.sx = data->x; sa.sy = data->y; sa.sz = data->z;
        if (owner) { sa.x = data->ax; sa.y = data->ay; sa.z = data->az; sa.nx = data->nx; sa.ny = data->ny; sa.nz = data->nz; }
        else { double* m = sl->moved.as<double>(); sa.x = m; sa.y = m + maxN; sa.z = m + 2 * maxN; }
      }
      hsb[i + gi] = sb; sb += nb;
      hsa[i] = sa;
      AccumArgs aa{};
      aa.T = t->dev;
      aa.x = data->x; aa.y = data->y; aa.z = data->z;
      if (mv != moving.end()) { aa.x = sa.x; aa.y = sa.y; aa.z = sa.z; }   // (k_accum_multi runs behind the search launch)
      aa.kpos = sa.kpos; aa.n = data->N; aa.A = A; aa.inv = inv;
      for (int k = 0; k < 3; k++) aa.shift[k] = shifts[3 * li + k];
      aa.partials = sl->part.as<double>();
      const uint32_t ag = accum_grid(data->N);
      hab[i + gi] = ab; ab += ag;
      haa[i] = aa;
      hfd[i].partials = aa.partials; hfd[i].out = d_out + (size_t)li * ACC_TOTAL; hfd[i].rows = fuse_links ? (int)nb : (int)ag;
      hfd[i].pad = (want == (TDTK_WANT_LUM | ACC_WANT_NO_CROSS)) ? 1 : 0;   // k_final_multi: which columns the rows fill
    }
    hsb[l1 + gi] = sb; hab[l1 + gi] = ab;
    s_total[gi] = sb; a_total[gi] = ab;
    // behind this launch the spare arrays hold the scan (the launches of the call run in this order on one stream)
    for (auto& kv : moving) {
      tdtk_scan* sc = kv.first;
      std::swap(sc->x, sc->ax); std::swap(sc->y, sc->ay); std::swap(sc->z, sc->az);
      sc->pending.clear();
      sc->npend.store(0, std::memory_order_release);
    }
  }
  void* staged = nullptr;
  if ((rc = stage_pinned(c, tab.data(), total, &staged))) return rc;
  HIPCHK(hipMemcpyAsync(c->multi_args.p, staged, total, hipMemcpyHostToDevice, s));
  char* dbase = static_cast<char*>(c->multi_args.p);
  const int cls = search_multi_class(maxN);
  // idle lanes a wave collects before it hands out new queries: 32 once a launch is many generations of waves (84 links of
  // 1M points, ordered: 8.93 -> 8.85 ms), as for single passes of 4M queries and more (refill_thresh)
  // (22 links: 2.627 ms with 16 against 2.652 with 32; 11 links 1.416 against 1.422)
  const int thresh = (G > 32 && cls == 20 && !lab_env("TDTK_REFILL_THRESH")) ? 32 : search_multi_thresh(maxN);
  for (int gi = 0; gi < ngroups; gi++) {
    const int l0 = gi * G, l1 = std::min(nlinks, l0 + G), nb = l1 - l0;
    const bool timed = (gi == ngroups - 1) && kernel_timing();   // tdtk_last_kernel_ms: the last group's search launch
    if (timed) HIPCHK(hipEventRecord(c->e0, s));
    HIPCHK(launch_search_multi(reinterpret_cast<const SearchArgs*>(dbase + o_sa) + l0,
                               reinterpret_cast<const uint32_t*>(dbase + o_sb) + l0 + gi, nb, s_total[gi], cls, thresh, c->counting, s, ordered, fuse_links));
    if (timed) { HIPCHK(hipEventRecord(c->e1, s)); c->ev_pending = true; }
    HIPCHK(launch_accum_multi(reinterpret_cast<const AccumArgs*>(dbase + o_aa) + l0,
                              reinterpret_cast<const uint32_t*>(dbase + o_ab) + l0 + gi, nb, a_total[gi], want,
                              reinterpret_cast<const FinalDesc*>(dbase + o_fd) + l0, s, fuse_links));
  }
  for (const Held& h : held) { h.sl->k_tree = h.tree; h.sl->k_scan = h.scan; h.sl->k_n = h.n; }   // (a slot used by several groups: the last one's)
  return TDTK_OK;
}

static int links_device_pass(Ctx* c, int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                             tdtk_scan* const* second, double maxd2, unsigned want, std::vector<double>& acc,
                             std::vector<double>& shifts)
{
  hipStream_t s = c->stream;
  int rc;
  if ((rc = c->ws[WS_TMPB].ensure((size_t)nlinks * ACC_TOTAL * sizeof(double)))) return rc;
  double* d_out = c->ws[WS_TMPB].as<double>();
  shifts.assign(3 * (size_t)nlinks, 0.0);
  size_t maxN = 0;
  for (int i = 0; i < nlinks; i++) {
    if (!first[i] || !second[i]) { set_error("NULL link member"); return TDTK_EINVAL; }
    if (first[i]->device != c->device || second[i]->device != c->device) { set_error("link members on another device"); return TDTK_EINVAL; }
    maxN = std::max(maxN, second[i]->N);
  }
  // With the sums added up inside the search launch a link's sums depend on the path it takes, so the path must depend on
  // the link alone and not on its company (else a rank's share of a graph with scans of several size classes would give
  // other bits than the whole): links of big scans go through the batched launch, by refill-threshold group, whatever
  // else the call holds.
  if (nlinks > 1 && link_batch_max() > 1) {
    std::vector<int> key(nlinks);
    bool mixed = false, any = false;
    for (int i = 0; i < nlinks; i++) {
      const size_t N = second[i]->N;
      key[i] = (N > 0 && link_sums_in_search(c, want, N)) ? 100 + search_multi_thresh(N) : 0;
      mixed = mixed || key[i] != key[0];
      any = any || key[i] != 0;
    }
    if (mixed && any) {
      acc.assign((size_t)nlinks * ACC_TOTAL, 0.0);
      std::vector<char> done(nlinks, 0);
      for (int i0 = 0; i0 < nlinks; i0++) {
        if (done[i0]) continue;
        std::vector<int> idx;
        for (int i = i0; i < nlinks; i++) if (!done[i] && key[i] == key[i0]) { idx.push_back(i); done[i] = 1; }
        const int n2 = (int)idx.size();
        std::vector<const tdtk_tree*> f2(n2);
        std::vector<tdtk_scan*> s2(n2);
        std::vector<double> d2(16 * (size_t)n2), a2, sh2;
        for (int k = 0; k < n2; k++) {
          f2[k] = first[idx[k]]; s2[k] = second[idx[k]];
          std::memcpy(&d2[16 * (size_t)k], first_dalignxf + 16 * (size_t)idx[k], 16 * sizeof(double));
        }
        if ((rc = links_device_pass(c, n2, f2.data(), d2.data(), s2.data(), maxd2, want, a2, sh2))) return rc;
        for (int k = 0; k < n2; k++) {
          std::memcpy(&acc[(size_t)idx[k] * ACC_TOTAL], &a2[(size_t)k * ACC_TOTAL], ACC_TOTAL * sizeof(double));
          for (int q = 0; q < 3; q++) shifts[3 * (size_t)idx[k] + q] = sh2[3 * (size_t)k + q];
        }
      }
      return TDTK_OK;
    }
  }
  {
    const int gb = link_batch_max();
    bool ok = gb > 1 && (nlinks > 1 || link_sums_in_search(c, want, maxN)) &&
              (want == (TDTK_WANT_LUM | ACC_WANT_NO_CROSS) || (want & 7u) == TDTK_WANT_LUM || (want & 7u) == 0u);
    const int cls = search_multi_class(maxN);
    ok = ok && cls != 0;
    for (int i = 0; i < nlinks && ok; i++)      // one kernel family (and refill threshold) for the whole call
      ok = second[i]->N > 0 && search_multi_class(second[i]->N) == cls &&
           (cls != 20 || search_multi_thresh(second[i]->N) == search_multi_thresh(maxN));
    if (ok) {
      for (int i = 0; i < nlinks; i++) {
        const tdtk_tree* t = first[i];
        const double* A16 = first_dalignxf + 16 * (size_t)i;
        for (int k = 0; k < 3; k++)
          shifts[3 * i + k] = t->centre[0] * A16[k] + t->centre[1] * A16[4 + k] + t->centre[2] * A16[8 + k] + A16[12 + k];
      }
      HIPCHK(hipMemsetAsync(d_out, 0, (size_t)nlinks * ACC_TOTAL * sizeof(double), s));
      // the launch may carry out queued scan moves into the spare arrays and swap them in: until it has run no other
      // thread may find "nothing queued" on those scans and read them on its own stream -- the lock is held to the sync
      // (threads with nothing to settle never take it)
      // (with more than one context alive another host thread may queue a move on one of these scans between an unlocked
      //  test and the launch's own reading of the queues: then the lock is taken first and the test made under it)
      std::unique_lock<std::recursive_mutex> lk(g_moves_mu, std::defer_lock);
      if (g_ctx_live.load() > 1) lk.lock();
      bool moving_any = false;
      for (int i = 0; i < nlinks && !moving_any; i++) moving_any = second[i]->npend.load(std::memory_order_acquire) != 0;
      if (moving_any && !lk.owns_lock()) lk.lock();
      if ((rc = links_device_pass_batched(c, nlinks, first, first_dalignxf, second, maxd2, want, d_out, shifts, gb))) return rc;
      acc.assign((size_t)nlinks * ACC_TOTAL, 0.0);
      HIPCHK(hipMemcpyAsync(acc.data(), d_out, acc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
      HIPCHK(hipStreamSynchronize(s));
      if (lk.owns_lock()) lk.unlock();
      collect_ms(c, nullptr);
      return TDTK_OK;
    }
  }
  if ((rc = scans_settle(c, second, nlinks))) return rc;     // (the passes below read the scans as they are)
  int max_lanes = 8, forced = 0;
  if (const char* e = lab_env("TDTK_LINK_LANES")) forced = std::max(1, std::min(16, atoi(e)));
  // small scans: a pass is latency-bound, 4 side by side (32 x 60K points, 41 links: 1 lane 2.35 ms, 2: 1.52, 3: 1.30,
  // 4: 1.11, 6: 1.26, 8: 1.19); big scans: 3, so the thin tail of one search and the small sum kernels overlap with
  // the next search (84 links of 1M: 1 lane 16.3 ms, 2: 13.0, 3: 12.2; more lanes than hardware queues lose, see
  // below; tools/gs_lanes_probe.py, tools/small_graph_probe.py)
  const int L = (nlinks > 1) ? std::min(nlinks, forced ? forced : std::min(max_lanes, maxN <= (size_t)262144 ? 4 : 3)) : 1;
  HIPCHK(hipMemsetAsync(d_out, 0, (size_t)nlinks * ACC_TOTAL * sizeof(double), s));
  if (L > 1) {
    // The runtime multiplexes streams onto 4 hardware queues (GPU_MAX_HW_QUEUES): streams beyond that share a queue
    // and their kernels no longer overlap -- with the null stream in use and three lanes BESIDE the context's stream,
    // 84 link passes took 15.9 ms instead of 13.2 (and four lanes lost to three in every probe).  Lane 0 is therefore
    // the context's own stream: three lanes + the null stream = four queues.
    while ((int)c->lanes.size() < L) {
      std::unique_ptr<Lane> ln(new Lane);
      if (c->lanes.empty()) { ln->s = s; ln->owns = false; }
      else HIPCHK(hipStreamCreateWithFlags(&ln->s, hipStreamNonBlocking));
      c->lanes.push_back(std::move(ln));
    }
    HIPCHK(hipStreamSynchronize(s));   // d_out is zeroed before any lane writes into it
    for (int l = 0; l < L; l++) {
      if ((rc = c->lanes[l]->kpos.ensure(maxN * sizeof(int)))) return rc;
      const size_t rows = std::max<size_t>(accum_grid(maxN), search_fused_rows(maxN, L));
      if ((rc = c->lanes[l]->part.ensure(rows * ACC_TOTAL * sizeof(double)))) return rc;
    }
  } else {
    if ((rc = c->ws[WS_KPOS].ensure(maxN * sizeof(int)))) return rc;
    if ((rc = c->ws[WS_PART].ensure((size_t)accum_grid(maxN) * ACC_TOTAL * sizeof(double)))) return rc;
  }
  for (int i = 0; i < nlinks; i++) {
    const tdtk_tree* t = first[i];
    tdtk_scan* data = second[i];
    const double* A16 = first_dalignxf + 16 * (size_t)i;
    for (int k = 0; k < 3; k++)
      shifts[3 * i + k] = t->centre[0] * A16[k] + t->centre[1] * A16[4 + k] + t->centre[2] * A16[8 + k] + A16[12 + k];
    if (data->N == 0) continue;
    Mat4 A, inv;
    std::memcpy(A.m, A16, sizeof A.m);
    m4inv(A16, inv.m);
    Lane* ln = (L > 1) ? c->lanes[i % L].get() : nullptr;
    hipStream_t ls = ln ? ln->s : s;
    SearchArgs sa{};
    sa.side_by_side = L;
    sa.x = data->x; sa.y = data->y; sa.z = data->z;
    sa.n = data->N; sa.inv = inv; sa.has_inv = 1; sa.maxd2 = maxd2;
    sa.kpos = ln ? ln->kpos.as<int>() : c->ws[WS_KPOS].as<int>();
    // a lum6DEuler link on a lane: its 17 sums come out of the search itself (accumulated when a query retires), so
    // the pass over (x, y, z, hit, pts[hit]) that k_accum would make disappears.  Side by side the passes run ~3 waves
    // per SIMD by choice, so the 36 accumulator registers cost no occupancy here (in the single-stream ICP loop they do)
    uint32_t fused_rows = 0;
    if (ln && want == (TDTK_WANT_LUM | ACC_WANT_NO_CROSS) && search_can_fuse(sa.n) && fuse_lum_enabled()) {
      fused_rows = search_fused_rows(sa.n, L);
      sa.fuse = 2; sa.A = A;
      for (int k = 0; k < 3; k++) sa.shift[k] = shifts[3 * i + k];
      sa.partials = ln->part.as<double>();
    }
    if (ln) {
      sa.T = t->dev;
      const uint32_t grid = search_grid(sa.n);
      if ((rc = prepare_overflow_in(ln->ovf_m2, ln->ovf_ref, t, sa.n, sa))) return rc;
      if (search_uses_queue(sa.n) && (rc = ln->qc.attach(sa))) return rc;
      if (c->counting) { sa.counters = c->d_counters.as<unsigned long long>(); c->counted_queries += sa.n; }
      const bool timed = (i == nlinks - 1) && kernel_timing();   // tdtk_last_kernel_ms: this search, running beside the other lanes'
      if (timed) HIPCHK(hipEventRecord(c->e0, ls));
      HIPCHK(launch_search(sa, grid, 0, c->counting, ls));
      if (timed) { HIPCHK(hipEventRecord(c->e1, ls)); c->ev_pending = true; }
    } else {
      if ((rc = run_search(c, t, sa, 0, false, s, i == nlinks - 1))) return rc;
    }
    if (fused_rows) {
      HIPCHK(launch_final(sa.partials, fused_rows, d_out + (size_t)i * ACC_TOTAL, ls));
      continue;
    }
    AccumArgs aa{};
    aa.T = t->dev;
    aa.x = data->x; aa.y = data->y; aa.z = data->z;
    aa.kpos = sa.kpos; aa.n = data->N; aa.A = A; aa.inv = inv;
    for (int k = 0; k < 3; k++) aa.shift[k] = shifts[3 * i + k];
    aa.partials = ln ? ln->part.as<double>() : c->ws[WS_PART].as<double>();
    HIPCHK(launch_accum(aa, accum_grid(data->N), want, 0, d_out + (size_t)i * ACC_TOTAL, ls));
  }
  for (int l = 0; l < L && L > 1; l++) HIPCHK(hipStreamSynchronize(c->lanes[l]->s));
  acc.assign((size_t)nlinks * ACC_TOTAL, 0.0);
  HIPCHK(hipMemcpyAsync(acc.data(), d_out, acc.size() * sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  collect_ms(c, nullptr);
  return TDTK_OK;
}

int tdtk_links_pair_sums(int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                         tdtk_scan* const* second, double maxd2, uint32_t want, tdtk_pair_sums* sums)
{
  if (nlinks < 0 || (nlinks && (!first || !first_dalignxf || !second || !sums))) { set_error("bad argument"); return TDTK_EINVAL; }
  if (nlinks == 0) return TDTK_OK;
  if (want & TDTK_WANT_NAPX) { set_error("NAPX needs normals: use tdtk_scan_pairs"); return TDTK_EUNSUP; }
  if (!first[0]) { set_error("NULL link member"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(first[0]->device, &c);
  if (rc) return rc;
  std::vector<double> acc, shifts;
  if ((rc = links_device_pass(c, nlinks, first, first_dalignxf, second, maxd2, want, acc, shifts))) return rc;
  for (int i = 0; i < nlinks; i++)
    finish_sums(acc.data() + (size_t)i * ACC_TOTAL, shifts.data() + 3 * i, second[i]->N, want, sums + i);
  return TDTK_OK;
}

int tdtk_solve_chol_upper(const double* G, const double* B, int n, double* x)
{
  if (!G || !B || !x || n <= 0) { set_error("bad argument"); return TDTK_EINVAL; }
  std::vector<double> S((size_t)n * n);
  for (int i = 0; i < n; i++)
    for (int j = i; j < n; j++) S[(size_t)i * n + j] = S[(size_t)j * n + i] = G[(size_t)i * n + j];
  std::vector<double> xs(n);
  if (!solve_spd_dense(n, S.data(), B, xs.data(), 0.00001)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
  std::memcpy(x, xs.data(), sizeof(double) * n);
  return TDTK_OK;
}

int tdtk_invert(const double* A, int n, double* Ainv)
{
  if (!A || !Ainv || n <= 0) { set_error("bad argument"); return TDTK_EINVAL; }
  if (!invert_dense(n, A, Ainv)) { set_error("singular matrix"); return TDTK_ESOLVE; }
  return TDTK_OK;
}

// ---- batched links + native pose update (graph-SLAM inner loop without per-link round trips) ----
int tdtk_lum_links(int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                   tdtk_scan* const* second, double maxd2, double* C, double* CD, uint64_t* m_out, double* ss_out)
{
  if (nlinks < 0 || (nlinks && (!first || !first_dalignxf || !second || !C || !CD))) { set_error("bad argument"); return TDTK_EINVAL; }
  if (nlinks == 0) return TDTK_OK;
  if (!first[0]) { set_error("NULL link member"); return TDTK_EINVAL; }
  Ctx* c;
  int rc = get_ctx(first[0]->device, &c);
  if (rc) return rc;
  std::vector<double> acc, shifts;
  // n, sum |delta|^2 and the 15 LUM sums are all a link's block needs
  if ((rc = links_device_pass(c, nlinks, first, first_dalignxf, second, maxd2, TDTK_WANT_LUM | ACC_WANT_NO_CROSS, acc, shifts))) return rc;
  for (int i = 0; i < nlinks; i++) {
    double* Ci = C + 36 * (size_t)i;
    double* CDi = CD + 6 * (size_t)i;
    std::memset(Ci, 0, 36 * sizeof(double));
    std::memset(CDi, 0, 6 * sizeof(double));
    const double* a = acc.data() + (size_t)i * ACC_TOTAL;
    const uint64_t m = (uint64_t)(a[ACC_N] + 0.5);
    if (m_out) m_out[i] = m;
    if (ss_out) ss_out[i] = 0.0;
    if (m <= 2) continue;
    const double* L = a + ACC_L;
    double MM[36] = {0};
    auto M = [&](int r, int col) -> double& { return MM[(r - 1) * 6 + (col - 1)]; };
    M(1, 1) = M(2, 2) = M(3, 3) = (double)m;
    M(4, 4) = L[5]; M(5, 5) = L[3]; M(6, 6) = L[4];
    M(1, 5) = M(5, 1) = -L[1]; M(1, 6) = M(6, 1) = L[2];
    M(2, 4) = M(4, 2) = -L[2]; M(2, 5) = M(5, 2) = L[0];
    M(3, 4) = M(4, 3) = L[1];  M(3, 6) = M(6, 3) = -L[0];
    M(4, 5) = M(5, 4) = -L[7]; M(4, 6) = M(6, 4) = -L[6]; M(5, 6) = M(6, 5) = -L[8];
    const double* MZ = L + 9;
    double MMi[36], D[6];
    if (!invert_dense(6, MM, MMi)) { set_error("singular link matrix"); return TDTK_ESOLVE; }
    double dmz = 0.0;
    for (int r = 0; r < 6; r++) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += MMi[r * 6 + k] * MZ[k];
      D[r] = v;
      dmz += v * MZ[r];
    }
    // sum |delta - A(u) D|^2 = sum|delta|^2 - 2 D.MZ + D.MM.D = sum|delta|^2 - D.MZ for MM D = MZ
    double ss = (a[ACC_SUM] - dmz) / (2.0 * (double)m - 3.0);
    if (ss_out) ss_out[i] = ss;
    if (ss < 0.0000000000001) continue;
    ss = 1.0 / ss;
    for (int k = 0; k < 36; k++) Ci[k] = MM[k] * ss;
    for (int k = 0; k < 6; k++) CDi[k] = MZ[k] * ss;
  }
  return TDTK_OK;
}

// FillGB3D's scatter (lum6Deuler.cc:285-300) over ALL links in link order + solveSparseCholesky
// (graphSlam6D.cc:345-379).  Links whose blocks are all zero (m <= 2) contribute nothing, as there.
int tdtk_lum_assemble_solve(int nlinks, const int32_t* from, const int32_t* to, const double* C, const double* CD,
                            int nscans, double* X, double* G_out, double* B_out)
{
  if (nlinks < 0 || nscans < 2 || !X || (nlinks && (!from || !to || !C || !CD))) { set_error("bad argument"); return TDTK_EINVAL; }
  const int n = nscans - 1, N = 6 * n;
  thread_local std::vector<double> G, B;   // kept per host thread: no fresh pages every LUM round
  if (!G_out && !B_out) {
    // Nobody wants the dense G: the blocks go straight into the skyline the solve factors (round 5: clearing and re-reading
    // 1.1 MB of dense G per round were a third of the 0.19 ms the solve cost a 64-scan graph).  A block row reaches left to
    // the smallest scan it shares a link with; inside that stretch the arithmetic is solve_spd_dense's on the dense copy --
    // the same sums in link order, the same |v| > 1e-5 filter, the same first surviving column per row -- so X is the same
    // bit for bit.
    thread_local std::vector<int> bmin, first;
    thread_local std::vector<size_t> off, base;
    thread_local std::vector<double> sky, y;
    bmin.resize(n);
    for (int r = 0; r < n; r++) bmin[r] = r;
    for (int l = 0; l < nlinks; l++) {
      const int a = from[l] - 1, b = to[l] - 1;
      if (a >= n || b >= n || a < -1 || b < -1) { set_error("link endpoint out of range"); return TDTK_EINVAL; }
      if (a >= 0 && b >= 0) { const int hi = a > b ? a : b, lo = a > b ? b : a; if (lo < bmin[hi]) bmin[hi] = lo; }
    }
    base.resize((size_t)N); first.resize((size_t)N); off.resize((size_t)N + 1);
    size_t total = 0;
    for (int i = 0; i < N; i++) { base[i] = total; total += (size_t)(i - 6 * bmin[i / 6] + 1); }
    sky.assign(total, 0.0); B.assign((size_t)N, 0.0);
    auto at = [&](int i, int k) -> double& { return sky[base[i] + (size_t)(k - 6 * bmin[i / 6])]; };   // k <= i, inside the row's stretch
    for (int l = 0; l < nlinks; l++) {
      const int a = from[l] - 1, b = to[l] - 1;
      const double* Cab = C + 36 * (size_t)l;
      const double* CDab = CD + 6 * (size_t)l;
      auto add = [&](int r, int c, double sgn) {       // the lower triangle only (the solve reads nothing else)
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++)
            if (c * 6 + j <= r * 6 + i) at(r * 6 + i, c * 6 + j) += sgn * Cab[i * 6 + j];
      };
      if (a >= 0) { for (int i = 0; i < 6; i++) B[a * 6 + i] += CDab[i]; add(a, a, 1.0); }
      if (b >= 0) { for (int i = 0; i < 6; i++) B[b * 6 + i] -= CDab[i]; add(b, b, 1.0); }
      // (a self link, from == to: the dense fill subtracts the block twice from the diagonal it has just added it to twice --
      //  net zero, in that order; the same here)
      if (a >= 0 && b >= 0) { if (a > b) add(a, b, -1.0); else if (a < b) add(b, a, -1.0); else { add(a, a, -1.0); add(a, a, -1.0); } }
    }
    for (int i = 0; i < N; i++) {
      const int f0 = 6 * bmin[i / 6];
      double* row = sky.data() + base[i];
      int f = f0;
      while (f < i && !(std::fabs(row[f - f0]) > 0.00001)) ++f;
      first[i] = f;
      off[i] = base[i] + (size_t)(f - f0);
      for (int k = f; k <= i; k++) if (!(std::fabs(row[k - f0]) > 0.00001)) row[k - f0] = 0.0;
    }
    off[N] = total;
    y.resize((size_t)N);
    if (!skyline_solve(N, first.data(), off.data(), sky.data(), B.data(), y.data(), X)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
    return TDTK_OK;
  }
  G.assign((size_t)N * N, 0.0); B.assign((size_t)N, 0.0);
  for (int l = 0; l < nlinks; l++) {
    const int a = from[l] - 1, b = to[l] - 1;
    if (a >= n || b >= n || a < -1 || b < -1) { set_error("link endpoint out of range"); return TDTK_EINVAL; }
    const double* Cab = C + 36 * (size_t)l;
    const double* CDab = CD + 6 * (size_t)l;
    auto add = [&](int r, int c, double sgn) {
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) G[(size_t)(r * 6 + i) * N + (c * 6 + j)] += sgn * Cab[i * 6 + j];
    };
    if (a >= 0) { for (int i = 0; i < 6; i++) B[a * 6 + i] += CDab[i]; add(a, a, 1.0); }
    if (b >= 0) { for (int i = 0; i < 6; i++) B[b * 6 + i] -= CDab[i]; add(b, b, 1.0); }
    if (a >= 0 && b >= 0) { add(a, b, -1.0); add(b, a, -1.0); }
  }
  if (G_out) std::memcpy(G_out, G.data(), sizeof(double) * G.size());
  if (B_out) std::memcpy(B_out, B.data(), sizeof(double) * B.size());
  if (!solve_spd_dense(N, G.data(), B.data(), X, 0.00001)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
  return TDTK_OK;
}

// many resident scans moved in place by A1 and then (optionally) A2.  The moves are queued on the scans (tdtk_scan::pending)
// and carried out by whoever reads a scan next; TDTK_LAZY_MOVES=0 (or a chain that has grown long): one launch now, left
// running behind the process-wide fence.
static int queue_scan_moves(Ctx* c, const std::vector<tdtk_scan*>& moved)
{
  if (moved.empty()) return TDTK_OK;
  std::vector<const tdtk_scan*> now;
  for (tdtk_scan* sc : moved)
    if (!lazy_moves() || sc->npend.load(std::memory_order_acquire) >= LAZY_CHAIN_MAX) now.push_back(sc);
  if (now.empty()) return TDTK_OK;
  int rc = scans_settle(c, now.data(), (int)now.size());
  if (rc) return rc;
  return defer_fence(c);
}

int tdtk_scans_transform2(int count, tdtk_scan* const* scans, const double* A1, const double* A2)
{
  if (count < 0 || (count && (!scans || !A1))) { set_error("bad argument"); return TDTK_EINVAL; }
  Ctx* c = nullptr;
  std::vector<tdtk_scan*> moved;
  for (int i = 0; i < count; i++) {
    tdtk_scan* sc = scans[i];
    if (!sc || !sc->N) continue;
    if (!c) { int rc = get_ctx(sc->device, &c); if (rc) return rc; }
    if (sc->device != c->device) { set_error("resident scans of one call must live on one device"); return TDTK_EINVAL; }
    { int rk = scan_keep_original(c, sc); if (rk) return rk; }
    scan_queue_move(sc, A1 + 16 * (size_t)i);
    if (A2) scan_queue_move(sc, A2 + 16 * (size_t)i);
    moved.push_back(sc);
  }
  if (!c) return TDTK_OK;
  return queue_scan_moves(c, moved);
}

int tdtk_lum_update_poses(int nscans, const double* X, double* transMat, double* dalignxf, double* rPos,
                          double* rPosTheta, tdtk_scan* const* scans, double* xf_out, double* ret)
{
  if (nscans <= 0 || !X || !transMat || !dalignxf || !rPos || !rPosTheta) { set_error("bad argument"); return TDTK_EINVAL; }
  double sum_position_diff = 0.0;
  Ctx* c = nullptr;
  std::vector<tdtk_scan*> moved;
  for (int i = 1; i < nscans; i++) {
    double* tm = transMat + 16 * (size_t)i;
    double* da = dalignxf + 16 * (size_t)i;
    double* rp = rPos + 3 * (size_t)i;
    double* rt = rPosTheta + 3 * (size_t)i;
    const double xa = rp[0], ya = rp[1], za = rp[2];
    const double ctx = std::cos(rt[0]), stx = std::sin(rt[0]), cty = std::cos(rt[1]), sty = std::sin(rt[1]);
    double Ha[36] = {0}, Hi[36];
    for (int k = 0; k < 6; k++) Ha[k * 6 + k] = 1.0;
    Ha[0 * 6 + 4] = -za * ctx + ya * stx;
    Ha[0 * 6 + 5] = ya * cty * ctx + za * stx * cty;
    Ha[1 * 6 + 3] = za;
    Ha[1 * 6 + 4] = -xa * stx;
    Ha[1 * 6 + 5] = -xa * ctx * cty + za * sty;
    Ha[2 * 6 + 3] = -ya;
    Ha[2 * 6 + 4] = xa * ctx;
    Ha[2 * 6 + 5] = -xa * cty * stx - ya * sty;
    Ha[3 * 6 + 5] = sty;
    Ha[4 * 6 + 4] = stx;
    Ha[4 * 6 + 5] = ctx * cty;
    Ha[5 * 6 + 4] = ctx;
    Ha[5 * 6 + 5] = -stx * cty;
    if (!invert_dense(6, Ha, Hi)) { set_error("singular pose Jacobian"); return TDTK_ESOLVE; }
    const double* Xi = X + 6 * (size_t)(i - 1);
    double result[6];
    for (int r = 0; r < 6; r++) {
      double v = 0;
      for (int k = 0; k < 6; k++) v += Hi[r * 6 + k] * Xi[k];
      result[r] = v;
    }
    double nP[3], nT[3];
    for (int k = 0; k < 3; k++) { nP[k] = rp[k] - result[k]; nT[k] = rt[k] - result[k + 3]; }
    // Scan::transformToEuler (scan.cc:1061-1083)
    double tinv[16], axf[16];
    m4inv(tm, tinv);
    mmult(tinv, tm, tm); matrix4_to_euler(tm, rt, rp); mmult(tinv, da, da);   // transform(tinv, INVALID)
    euler_to_matrix4(nP, nT, axf);
    mmult(axf, tm, tm); matrix4_to_euler(tm, rt, rp); mmult(axf, da, da);     // transform(alignxf, LUM, ..)
    if (xf_out) { std::memcpy(xf_out + 32 * (size_t)i, tinv, sizeof tinv); std::memcpy(xf_out + 32 * (size_t)i + 16, axf, sizeof axf); }
    if (scans && scans[i] && scans[i]->N) {
      if (!c) { int rc = get_ctx(scans[i]->device, &c); if (rc) return rc; }
      tdtk_scan* sc = scans[i];
      if (sc->device != c->device) { set_error("resident scans of one call must live on one device"); return TDTK_EINVAL; }
      { int rk = scan_keep_original(c, sc); if (rk) return rk; }
      scan_queue_move(sc, tinv);     // Scan::transformToEuler: out of the old pose ...
      scan_queue_move(sc, axf);      // ... into the new one
      moved.push_back(sc);
    }
    sum_position_diff += std::sqrt(result[0] * result[0] + result[1] * result[1] + result[2] * result[2]);
  }
  if (c) {
    int rc = queue_scan_moves(c, moved);
    if (rc) return rc;
  }
  if (ret) *ret = sum_position_diff / (double)nscans;
  return TDTK_OK;
}

// ---- graph-SLAM back-ends (-G 1..4): per-link blocks, scatter + solve + pose update ------------------
// One pair of entry points for lum6DEuler (lum6Deuler.cc), lum6DQuat (lum6Dquat.cc), ghelix6DQ2
// (ghelix6DQ2.cc) and gapx6D (gapx6D.cc).  A rank computes the blocks of its links (device passes batched, one
// sync), the blocks of all links are exchanged (one all-reduce: every link has one owner, the others hold zeros)
// and every rank runs tdtk_graph_solve_update on the complete set: scatter in link order, solve, pose update.
static const int kGraphBlock[5] = {0, 42, 56, 43, 49};

int tdtk_graph_block_doubles(int backend)
{
  if (backend < 1 || backend > 4) return 0;
  return kGraphBlock[backend];
}

int tdtk_graph_link_blocks(int backend, int nlinks, const tdtk_tree* const* first, const double* first_dalignxf,
                           tdtk_scan* const* second, double maxd2, double* blocks)
{
  if (backend < 1 || backend > 4) { set_error("unknown graph back-end"); return TDTK_EINVAL; }
  if (nlinks < 0 || (nlinks && (!first || !first_dalignxf || !second || !blocks))) { set_error("bad argument"); return TDTK_EINVAL; }
  if (nlinks == 0) return TDTK_OK;
  const int Bn = kGraphBlock[backend];
  std::memset(blocks, 0, sizeof(double) * (size_t)nlinks * Bn);
  if (backend == TDTK_GRAPH_LUMEULER) {
    std::vector<double> C((size_t)nlinks * 36), CD((size_t)nlinks * 6);
    int rc = tdtk_lum_links(nlinks, first, first_dalignxf, second, maxd2, C.data(), CD.data(), nullptr, nullptr);
    if (rc) return rc;
    for (int i = 0; i < nlinks; i++) {
      std::memcpy(blocks + (size_t)i * Bn, C.data() + (size_t)i * 36, 36 * sizeof(double));
      std::memcpy(blocks + (size_t)i * Bn + 36, CD.data() + (size_t)i * 6, 6 * sizeof(double));
    }
    return TDTK_OK;
  }
  const uint32_t want = (backend == TDTK_GRAPH_LUMQUAT) ? TDTK_WANT_LUM
                        : (backend == TDTK_GRAPH_GHELIX ? TDTK_WANT_MOM2 : TDTK_WANT_GAPX);
  std::vector<tdtk_pair_sums> sums((size_t)nlinks);
  int rc = tdtk_links_pair_sums(nlinks, first, first_dalignxf, second, maxd2, want, sums.data());
  if (rc) return rc;
  for (int i = 0; i < nlinks; i++) {
    const tdtk_pair_sums& s = sums[i];
    double* b = blocks + (size_t)i * Bn;
    const double m = (double)s.n;
    if (backend == TDTK_GRAPH_LUMQUAT) {
      // lum6DQuat::covarianceQuat (lum6Dquat.cc:81-248); ss from the normal equations (MM D = MZ)
      if (s.n <= 2) continue;
      const double* L = s.lum;
      const double sx = L[0], sy = L[1], sz = L[2], xpy = L[3], xpz = L[4], ypz = L[5], xy = L[6], xz = L[7], yz = L[8];
      const double MZ[7] = {L[9], L[10], L[11], s.lum_udot, -L[12], -L[14], -L[13]};
      double MM[49] = {0};
      auto set = [&](int r, int c, double v) { MM[r * 7 + c] = MM[c * 7 + r] = v; };
      set(0, 0, m); set(1, 1, m); set(2, 2, m);
      set(3, 3, (xpy + xpz + ypz) / 2.0); set(4, 4, ypz); set(5, 5, xpz); set(6, 6, xpy);
      set(0, 3, sx); set(0, 5, -sz); set(0, 6, sy);
      set(1, 3, sy); set(1, 4, sz);  set(1, 6, -sx);
      set(2, 3, sz); set(2, 4, -sy); set(2, 5, sx);
      set(4, 5, -xy); set(4, 6, -xz); set(5, 6, -yz);
      double MMi[49], D[7], dmz = 0.0;
      if (!invert_dense(7, MM, MMi)) { set_error("singular link matrix"); return TDTK_ESOLVE; }
      for (int r = 0; r < 7; r++) {
        double v = 0;
        for (int k = 0; k < 7; k++) v += MMi[r * 7 + k] * MZ[k];
        D[r] = v;
        dmz += v * MZ[r];
      }
      // the reference sums squared residuals (lum6Dquat.cc:199-214), which cannot go negative; the normal-equation
      // form can by rounding when the link is (nearly) perfectly aligned: clamp, and leave a zero block where the
      // division would blow up (the guard lum6DEuler has at lum6Deuler.cc:215-226)
      double ss = (s.sum - dmz) / (2.0 * m - 3.0);
      if (ss < 0.0) ss = 0.0;
      if (ss < 0.0000000000001) continue;
      for (int k = 0; k < 49; k++) b[k] = MM[k] / ss;
      for (int k = 0; k < 7; k++) b[49 + k] = MZ[k] / ss;
    } else if (backend == TDTK_GRAPH_GHELIX) {
      // ghelix6DQ2::genBBdForLinkedPair (ghelix6DQ2.cc:88-150): raw moments of p2, antisymmetric part of sum p1 p2^T
      if (s.n <= 1) continue;
      const double* cm = s.centroid_m;
      const double* cd = s.centroid_d;
      double DD[3][3], X[3][3];
      const double* dd = s.mom_dd;
      const double ddm[3][3] = {{dd[0], dd[1], dd[2]}, {dd[1], dd[3], dd[4]}, {dd[2], dd[4], dd[5]}};
      for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) {
          DD[r][c] = ddm[r][c] + m * cd[r] * cd[c];
          X[r][c] = s.Si[r * 3 + c] + m * cm[r] * cd[c];
        }
      const double s2[3] = {m * cd[0], m * cd[1], m * cd[2]};
      double* Blk = b + 1;
      auto set = [&](int r, int c, double v) { Blk[r * 6 + c] = Blk[c * 6 + r] = v; };
      set(3, 3, m); set(4, 4, m); set(5, 5, m);
      set(0, 4, -s2[2]); set(1, 3, s2[2]);
      set(0, 5, s2[1]);  set(2, 3, -s2[1]);
      set(2, 4, s2[0]);  set(1, 5, -s2[0]);
      set(0, 1, -DD[0][1]); set(0, 2, -DD[0][2]); set(1, 2, -DD[1][2]);
      set(0, 0, DD[2][2] + DD[1][1]); set(1, 1, DD[2][2] + DD[0][0]); set(2, 2, DD[0][0] + DD[1][1]);
      double* bd1 = b + 37;
      bd1[0] = X[2][1] - X[1][2]; bd1[1] = X[0][2] - X[2][0]; bd1[2] = X[1][0] - X[0][1];
      for (int k = 0; k < 3; k++) bd1[3 + k] = m * (cm[k] - cd[k]);
      b[0] = 1.0;
    } else {
      // gapx6D::genBArotForLinkedPair (gapx6D.cc:153-310); the centroids feed the translation step for
      // every link, also the "empty" ones (gapx6D.cc:424-431)
      for (int k = 0; k < 3; k++) { b[1 + k] = s.centroid_m[k]; b[4 + k] = s.centroid_d[k]; }
      if (s.n <= 1) continue;
      b[0] = 1.0;
      std::memcpy(b + 7, s.gapx_MkMkt, 9 * sizeof(double));
      std::memcpy(b + 16, s.gapx_DkDkt, 9 * sizeof(double));
      std::memcpy(b + 25, s.gapx_MkDkt, 9 * sizeof(double));
      std::memcpy(b + 34, s.gapx_DkMkt, 9 * sizeof(double));
      std::memcpy(b + 43, s.gapx_Ak1, 3 * sizeof(double));
      std::memcpy(b + 46, s.gapx_Ak2, 3 * sizeof(double));
    }
  }
  return TDTK_OK;
}

// Scan::transform(A1) [then Scan::transform(A2)] for scans 1..nscans-1: matrices as Scan::transformMatrix
// (scan.cc:878-898), resident point sets in one launch.  A1 / A2 are [nscans][16]; A2 nullable.
static int apply_scan_moves(int nscans, const double* A1, const double* A2, double* transMat, double* dalignxf,
                            double* rPos, double* rPosTheta, tdtk_scan* const* scans, double* xf_out)
{
  for (int i = 1; i < nscans; i++) {
    double* tm = transMat + 16 * (size_t)i;
    double* da = dalignxf + 16 * (size_t)i;
    const double* a1 = A1 + 16 * (size_t)i;
    mmult(a1, tm, tm); matrix4_to_euler(tm, rPosTheta + 3 * (size_t)i, rPos + 3 * (size_t)i); mmult(a1, da, da);
    if (A2) {
      const double* a2 = A2 + 16 * (size_t)i;
      mmult(a2, tm, tm); matrix4_to_euler(tm, rPosTheta + 3 * (size_t)i, rPos + 3 * (size_t)i); mmult(a2, da, da);
    }
    if (xf_out) {
      std::memcpy(xf_out + 32 * (size_t)i, a1, 16 * sizeof(double));
      if (A2) std::memcpy(xf_out + 32 * (size_t)i + 16, A2 + 16 * (size_t)i, 16 * sizeof(double));
      else m4identity(xf_out + 32 * (size_t)i + 16);
    }
  }
  if (!scans) return TDTK_OK;
  return tdtk_scans_transform2(nscans - 1, scans + 1, A1 + 16, A2 ? A2 + 16 : nullptr);
}

int tdtk_graph_solve_update(int backend, int nlinks, const int32_t* from, const int32_t* to, const double* blocks,
                            int nscans, double* transMat, double* dalignxf, double* rPos, double* rPosTheta,
                            tdtk_scan* const* scans, double* state, double* xf_out, double* ret)
{
  if (backend < 1 || backend > 4) { set_error("unknown graph back-end"); return TDTK_EINVAL; }
  if (nlinks < 0 || nscans < 2 || !transMat || !dalignxf || !rPos || !rPosTheta || (nlinks && (!from || !to || !blocks))) {
    set_error("bad argument");
    return TDTK_EINVAL;
  }
  const int n = nscans - 1, Bn = kGraphBlock[backend];
  for (int l = 0; l < nlinks; l++)
    if (from[l] < 0 || to[l] < 0 || from[l] >= nscans || to[l] >= nscans) { set_error("link endpoint out of range"); return TDTK_EINVAL; }

  if (backend == TDTK_GRAPH_LUMEULER) {
    std::vector<double> C((size_t)nlinks * 36), CD((size_t)nlinks * 6), X((size_t)6 * n);
    for (int i = 0; i < nlinks; i++) {
      std::memcpy(C.data() + (size_t)i * 36, blocks + (size_t)i * Bn, 36 * sizeof(double));
      std::memcpy(CD.data() + (size_t)i * 6, blocks + (size_t)i * Bn + 36, 6 * sizeof(double));
    }
    int rc = tdtk_lum_assemble_solve(nlinks, from, to, C.data(), CD.data(), nscans, X.data(), nullptr, nullptr);
    if (rc) return rc;
    return tdtk_lum_update_poses(nscans, X.data(), transMat, dalignxf, rPos, rPosTheta, scans, xf_out, ret);
  }

  std::vector<double> A1((size_t)nscans * 16, 0.0), A2;
  double sum_position_diff = 0.0;

  if (backend == TDTK_GRAPH_LUMQUAT) {
    // FillGB3D (lum6Dquat.cc:248-276): diagonal blocks accumulate, off-diagonal blocks are ASSIGNED
    const int N = 7 * n;
    std::vector<double> G((size_t)N * N, 0.0), B((size_t)N, 0.0), X((size_t)N);
    for (int l = 0; l < nlinks; l++) {
      const int a = from[l] - 1, b = to[l] - 1;
      const double* Cab = blocks + (size_t)l * Bn;
      const double* CDab = Cab + 49;
      auto blk = [&](int r, int c, double sgn, bool assign) {
        for (int i = 0; i < 7; i++)
          for (int j = 0; j < 7; j++) {
            double& g = G[(size_t)(r * 7 + i) * N + (c * 7 + j)];
            g = assign ? sgn * Cab[i * 7 + j] : g + sgn * Cab[i * 7 + j];
          }
      };
      if (a >= 0) { for (int i = 0; i < 7; i++) B[a * 7 + i] += CDab[i]; blk(a, a, 1.0, false); }
      if (b >= 0) { for (int i = 0; i < 7; i++) B[b * 7 + i] -= CDab[i]; blk(b, b, 1.0, false); }
      if (a >= 0 && b >= 0) { blk(a, b, -1.0, true); blk(b, a, -1.0, true); }
    }
    if (!solve_spd_dense(N, G.data(), B.data(), X.data(), 0.00001)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
    A2.assign((size_t)nscans * 16, 0.0);
    for (int i = 1; i < nscans; i++) {
      // lum6Dquat.cc:347-441
      const double* tm = transMat + 16 * (size_t)i;
      double quat[4], tq[3];
      matrix4_to_quat_t(tm, quat, tq);                    // rQuat follows transMat (scan.cc:886)
      const double xa = rPos[3 * i], ya = rPos[3 * i + 1], za = rPos[3 * i + 2];
      const double p = quat[0], q = quat[1], r = quat[2], s = quat[3];
      const double px = p * xa, py = p * ya, pz = p * za, qx = q * xa, qy = q * ya, qz = q * za;
      const double rx = r * xa, ry = r * ya, rz = r * za, sx = s * xa, sy = s * ya, sz = s * za;
      double Ha[49] = {0}, Hi[49];
      for (int k = 0; k < 7; k++) Ha[k * 7 + k] = 1.0;
      auto H = [&](int rr, int cc) -> double& { return Ha[rr * 7 + cc]; };
      H(3, 3) = 2 * p; H(4, 3) = 2 * q; H(5, 3) = 2 * r; H(6, 3) = 2 * s;
      H(3, 4) = 2 * q; H(4, 4) = -2 * p; H(5, 4) = -2 * s; H(6, 4) = 2 * r;
      H(3, 5) = 2 * r; H(4, 5) = 2 * s; H(5, 5) = -2 * p; H(6, 5) = -2 * q;
      H(3, 6) = 2 * s; H(4, 6) = -2 * r; H(5, 6) = 2 * q; H(6, 6) = -2 * p;
      H(0, 3) = -2 * (px + sy - rz); H(1, 3) = -2 * (-sx + py + qz); H(2, 3) = -2 * (rx - qy + pz);
      H(0, 4) = -2 * (qx + ry + sz); H(1, 4) = -2 * (-rx + qy - pz); H(2, 4) = -2 * (-sx + py + qz);
      H(0, 5) = -2 * (rx - qy + pz); H(1, 5) = -2 * (qx + ry + sz);  H(2, 5) = -2 * (-px - sy + rz);
      H(0, 6) = -2 * (sx - py - qz); H(1, 6) = -2 * (px + sy - rz);  H(2, 6) = -2 * (qx + ry + sz);
      if (!invert_dense(7, Ha, Hi)) { set_error("singular pose Jacobian"); return TDTK_ESOLVE; }
      double result[7];
      for (int rr = 0; rr < 7; rr++) {
        double v = 0;
        for (int k = 0; k < 7; k++) v += Hi[rr * 7 + k] * X[(size_t)(i - 1) * 7 + k];
        result[rr] = v;
      }
      const double nP[3] = {xa - result[0], ya - result[1], za - result[2]};
      double nQ[4] = {p - result[3], q - result[4], r - result[5], s - result[6]};
      const double ql = std::sqrt(nQ[0] * nQ[0] + nQ[1] * nQ[1] + nQ[2] * nQ[2] + nQ[3] * nQ[3]);   // Normalize4
      for (double& v : nQ) v /= ql;
      m4inv(tm, A1.data() + 16 * (size_t)i);                // Scan::transformToQuat (scan.cc:1093-1104)
      quat_to_matrix4(nQ, nP, A2.data() + 16 * (size_t)i);
      sum_position_diff += std::sqrt(result[0] * result[0] + result[1] * result[1] + result[2] * result[2]);
    }
  } else if (backend == TDTK_GRAPH_GHELIX) {
    // state = B (N x N) | bd (N), zeroed by the caller once per doGraphSlam6D call (ghelix6DQ2.cc:329-330)
    if (!state) { set_error("ghelix6DQ2 needs its state vector"); return TDTK_EINVAL; }
    const int N = 6 * n;
    double* B = state;
    double* bd = state + (size_t)N * N;
    for (int l = 0; l < nlinks; l++) {
      const double* b = blocks + (size_t)l * Bn;
      if (b[0] == 0.0) continue;
      const int fa = from[l], fb = to[l];
      if (fb == 0) { set_error("ghelix6DQ2: a link must not end at the fixed scan 0"); return TDTK_EINVAL; }
      const double* Blk = b + 1;
      const double* bd1 = b + 37;
      const int a = (fa - 1) * 6, c = (fb - 1) * 6;
      auto add = [&](int r0, int c0, double sgn) {
        for (int i = 0; i < 6; i++)
          for (int j = 0; j < 6; j++) B[(size_t)(r0 + i) * N + (c0 + j)] += sgn * Blk[i * 6 + j];
      };
      if (fa != 0) { add(a, a, 1.0); for (int i = 0; i < 6; i++) bd[a + i] += bd1[i]; }
      add(c, c, 1.0);
      for (int i = 0; i < 6; i++) bd[c + i] -= bd1[i];        // bd2 == -bd1 term by term (:126-138)
      if (fa != 0) { add(a, c, -1.0); add(c, a, -1.0); }
    }
    std::vector<double> ccs((size_t)N);
    if (!solve_spd_dense(N, B, bd, ccs.data(), 0.00001)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
    for (int i = 1; i < nscans; i++) {
      double* a1 = A1.data() + 16 * (size_t)i;
      helix_compute_rt(ccs.data() + (size_t)(i - 1) * 6, a1);
      sum_position_diff += std::sqrt(a1[12] * a1[12] + a1[13] * a1[13] + a1[14] * a1[14]);
    }
  } else {
    // gapx6D::doGraphSlam6D (gapx6D.cc:323-542).  state = T (3n), kept over the iterations of one call
    if (!state) { set_error("gapx6D needs its translation state vector"); return TDTK_EINVAL; }
    const int N = 3 * n;
    std::vector<double> B((size_t)N * N, 0.0), A((size_t)N, 0.0), X((size_t)N);
    for (int l = 0; l < nlinks; l++) {
      const double* b = blocks + (size_t)l * Bn;
      if (b[0] == 0.0) continue;
      sum_position_diff += 1.0;                            // genBArotForLinkedPair returns 1.0 per link (sic)
      const int f = from[l], sx = to[l];
      if (sx == 0) { set_error("gapx6D: a link must not end at the fixed scan 0"); return TDTK_EINVAL; }
      const int a = (f - 1) * 3, c = (sx - 1) * 3;
      auto add = [&](int r0, int c0, const double* M9) {
        for (int i = 0; i < 3; i++)
          for (int j = 0; j < 3; j++) B[(size_t)(r0 + i) * N + (c0 + j)] += M9[i * 3 + j];
      };
      if (f != 0) {
        for (int i = 0; i < 3; i++) A[a + i] += b[43 + i];
        add(a, a, b + 7); add(a, c, b + 34); add(c, a, b + 25);
      }
      for (int i = 0; i < 3; i++) A[c + i] += b[46 + i];
      add(c, c, b + 16);
    }
    {   // cs_cholsol reads the upper triangle only (see tdtk_solve_chol_upper)
      std::vector<double> S((size_t)N * N);
      for (int i = 0; i < N; i++)
        for (int j = i; j < N; j++) S[(size_t)i * N + j] = S[(size_t)j * N + i] = B[(size_t)i * N + j];
      if (!solve_spd_dense(N, S.data(), A.data(), X.data(), 0.00001)) { set_error("matrix is not positive definite"); return TDTK_ESOLVE; }
    }
    // translation system (genBAtransForLinkedPair, gapx6D.cc:76-137)
    std::vector<double> Bt((size_t)n * n, 0.0), At((size_t)N, 0.0), Bti((size_t)n * n);
    auto rot_apply = [&](const double* x, const double* p, double* out) {
      double a16[16];
      const double zero[3] = {0, 0, 0};
      apx_compute_rt(x, zero, a16);
      out[0] = (p[0] * a16[0] + p[1] * a16[4] + p[2] * a16[8]) + a16[12];    // Point::transform
      out[1] = (p[0] * a16[1] + p[1] * a16[5] + p[2] * a16[9]) + a16[13];
      out[2] = (p[0] * a16[2] + p[1] * a16[6] + p[2] * a16[10]) + a16[14];
    };
    for (int l = 0; l < nlinks; l++) {
      const double* b = blocks + (size_t)l * Bn;
      const int f = from[l], sx = to[l];
      if (sx == 0) { set_error("gapx6D: a link must not end at the fixed scan 0"); return TDTK_EINVAL; }
      const double zero[3] = {0, 0, 0};
      double r1[3], r2[3];
      rot_apply(f != 0 ? X.data() + (size_t)(f - 1) * 3 : zero, b + 1, r1);
      rot_apply(X.data() + (size_t)(sx - 1) * 3, b + 4, r2);
      const double Ak1[3] = {r1[0] - r2[0], r1[1] - r2[1], r1[2] - r2[2]};
      if (f != 0) {
        for (int i = 0; i < 3; i++) At[(size_t)(f - 1) * 3 + i] -= Ak1[i];
        Bt[(size_t)(f - 1) * n + (f - 1)] += 1.0;
        Bt[(size_t)(f - 1) * n + (sx - 1)] -= 1.0;
        Bt[(size_t)(sx - 1) * n + (f - 1)] -= 1.0;
      }
      for (int i = 0; i < 3; i++) At[(size_t)(sx - 1) * 3 + i] += Ak1[i];
      Bt[(size_t)(sx - 1) * n + (sx - 1)] += 1.0;
    }
    if (!invert_dense(n, Bt.data(), Bti.data())) { set_error("singular translation system"); return TDTK_ESOLVE; }
    for (int i = 0; i < n; i++)
      for (int k = 0; k < 3; k++) {
        double v = 0;
        for (int j = 0; j < n; j++) v += Bti[(size_t)i * n + j] * At[(size_t)j * 3 + k];
        state[(size_t)i * 3 + k] += v;
      }
    for (int i = 1; i < nscans; i++) {
      const double* dx = state + (size_t)(i - 1) * 3;
      apx_compute_rt(X.data() + (size_t)(i - 1) * 3, dx, A1.data() + 16 * (size_t)i);
      sum_position_diff += std::sqrt(dx[0] * dx[0] + dx[1] * dx[1] + dx[2] * dx[2]);
    }
  }
  int rc = apply_scan_moves(nscans, A1.data(), A2.empty() ? nullptr : A2.data(), transMat, dalignxf, rPos, rPosTheta,
                            scans, xf_out);
  if (rc) return rc;
  if (ret) *ret = sum_position_diff / (double)nscans;
  return TDTK_OK;
}

}  // extern "C"
