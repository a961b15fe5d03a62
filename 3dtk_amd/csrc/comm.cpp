// RCCL exchange of the graph-SLAM link blocks, inside the library (include/tdtk_hip.h: tdtk_comm_*,
// tdtk_graph_exchange, tdtk_graph_iteration).  One process per GPU; the only collective of the path is ONE
// all-reduce (sum, fp64) of the per-link blocks per global iteration (lum6Deuler.cc:265-303 shards over links).
//
// RCCL is bound at run time (dlopen): a process that already carries an RCCL -- PyTorch bundles its own next to its
// HIP runtime -- must not get a second one, and a single-GPU user of the library needs none at all.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "tdtk_internal.h"

using namespace tdtk;

namespace {
struct Rccl {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) getUniqueId = nullptr;
  decltype(&ncclCommInitRank) commInitRank = nullptr;
  decltype(&ncclCommDestroy) commDestroy = nullptr;
  decltype(&ncclCommAbort) commAbort = nullptr;
  decltype(&ncclCommCount) commCount = nullptr;
  decltype(&ncclAllReduce) allReduce = nullptr;
  decltype(&ncclGetErrorString) getErrorString = nullptr;
  std::string where;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl()
{
  // a copy that is already in the process first (RTLD_NOLOAD), then the ROCm installation
  const char* resident[] = {"librccl.so", "librccl.so.1"};
  for (const char* n : resident)
    if (!g_rccl.h && (g_rccl.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) g_rccl.where = std::string(n) + " (already loaded)";
  const char* fresh[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
  for (const char* n : fresh)
    if (!g_rccl.h && (g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) g_rccl.where = n;
  if (!g_rccl.h) return;
  g_rccl.getUniqueId = (decltype(g_rccl.getUniqueId))dlsym(g_rccl.h, "ncclGetUniqueId");
  g_rccl.commInitRank = (decltype(g_rccl.commInitRank))dlsym(g_rccl.h, "ncclCommInitRank");
  g_rccl.commDestroy = (decltype(g_rccl.commDestroy))dlsym(g_rccl.h, "ncclCommDestroy");
  g_rccl.allReduce = (decltype(g_rccl.allReduce))dlsym(g_rccl.h, "ncclAllReduce");
  g_rccl.commAbort = (decltype(g_rccl.commAbort))dlsym(g_rccl.h, "ncclCommAbort");      // optional: only the failure path uses it
  g_rccl.commCount = (decltype(g_rccl.commCount))dlsym(g_rccl.h, "ncclCommCount");
  g_rccl.getErrorString = (decltype(g_rccl.getErrorString))dlsym(g_rccl.h, "ncclGetErrorString");
  if (!g_rccl.getUniqueId || !g_rccl.commInitRank || !g_rccl.commDestroy || !g_rccl.allReduce) g_rccl.h = nullptr;
}

bool have_rccl()
{
  std::call_once(g_rccl_once, load_rccl);
  if (!g_rccl.h) set_error("RCCL (librccl.so) could not be loaded");
  return g_rccl.h != nullptr;
}

int nccl_fail(const char* what, ncclResult_t r)
{
  set_error(std::string(what) + ": " + (g_rccl.getErrorString ? g_rccl.getErrorString(r) : "RCCL error"));
  return TDTK_EDEVICE;
}
}  // namespace

struct tdtk_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t comm = nullptr;
  // no stream of its own: the collective runs on the calling thread's context stream (the runtime has four hardware
  // queues for all streams of the process, and the link passes want three of them -- see links_device_pass)
  double* d_buf = nullptr;
  size_t cap = 0;
  double* h_pin = nullptr;   // pinned staging: the copies around the collective are asynchronous on `stream`
  size_t h_cap = 0;
  uint64_t n_allreduce = 0;  // collectives issued (tests / bench: proves the RCCL path ran)
  int rccl_world = 0;        // what RCCL itself reports for the communicator (ncclCommCount); must equal `world`
  bool dead = false;         // aborted after a failure of this rank that could not be carried through the collective
};

namespace {
// staging for n doubles on both sides; called at creation (default size) so that the exchange of an ordinary graph never
// allocates between a rank's link passes and the collective
int reserve(tdtk_comm* c, size_t n)
{
  if (n > c->cap) {
    if (c->d_buf) (void)hipFree(c->d_buf);
    c->d_buf = nullptr; c->cap = 0;
    const size_t want = n + n / 4 + 64;
    if (hipMalloc((void**)&c->d_buf, want * sizeof(double)) != hipSuccess) {
      // device memory the handle pool keeps on its shelf is invisible to this allocation: give it back, try once more --
      // failing here ends in ncclCommAbort on every rank
      (void)hipGetLastError();
      pool_trim();
      if (hipMalloc((void**)&c->d_buf, want * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); set_error("hipMalloc failed"); return TDTK_ENOMEM; }
    }
    c->cap = want;
  }
  if (n > c->h_cap) {
    if (c->h_pin) (void)hipHostFree(c->h_pin);
    c->h_pin = nullptr; c->h_cap = 0;
    const size_t want = n + n / 4 + 64;
    if (hipHostMalloc((void**)&c->h_pin, want * sizeof(double), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      pool_trim();
      if (hipHostMalloc((void**)&c->h_pin, want * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); set_error("hipHostMalloc failed"); return TDTK_ENOMEM; }
    }
    c->h_cap = want;
  }
  return TDTK_OK;
}
// A failure of THIS rank that cannot be reported through the collective (no staging memory, a failed copy or launch): the
// other ranks are, or soon will be, inside ncclAllReduce waiting for us.  Abort the communicator so that they come back
// with an error instead of waiting forever; the communicator is unusable afterwards on every rank.
int abort_comm(tdtk_comm* c, int rc)
{
  const std::string why = tdtk_last_error();
  if (c->comm && g_rccl.commAbort) { g_rccl.commAbort(c->comm); c->comm = nullptr; }
  c->dead = true;
  set_error("communicator aborted after a local failure (" + why + "): the peers' collective returns an error");
  return rc;
}
constexpr size_t DEFAULT_EXCHANGE_DOUBLES = 1u << 17;   // 1 MB: 2 600 links of lum6DEuler blocks
}  // namespace

extern "C" {

int tdtk_comm_unique_id(char id[TDTK_COMM_ID_BYTES])
{
  if (!id) { set_error("NULL argument"); return TDTK_EINVAL; }
  if (!have_rccl()) return TDTK_EDEVICE;
  static_assert(sizeof(ncclUniqueId) <= TDTK_COMM_ID_BYTES, "ncclUniqueId does not fit TDTK_COMM_ID_BYTES");
  ncclUniqueId u;
  ncclResult_t r = g_rccl.getUniqueId(&u);
  if (r != ncclSuccess) return nccl_fail("ncclGetUniqueId", r);
  std::memset(id, 0, TDTK_COMM_ID_BYTES);
  std::memcpy(id, &u, sizeof u);
  return TDTK_OK;
}

int tdtk_comm_create(const char id[TDTK_COMM_ID_BYTES], int rank, int world, int device, tdtk_comm** out)
{
  if (!out) { set_error("out is NULL"); return TDTK_EINVAL; }
  *out = nullptr;
  if (!id || world < 1 || rank < 0 || rank >= world) { set_error("bad rank / world"); return TDTK_EINVAL; }
  if (!have_rccl()) return TDTK_EDEVICE;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_error("no HIP device available (lib3dtk_hip has no CPU fallback)"); return TDTK_EDEVICE; }
  if (device < 0 || device >= ndev) { set_error("bad device ordinal"); return TDTK_EINVAL; }
  if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice failed"); return TDTK_EDEVICE; }
  tdtk_comm* c = new tdtk_comm;
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof u);
  ncclResult_t r = g_rccl.commInitRank(&c->comm, world, u, rank);
  if (r != ncclSuccess) { delete c; return nccl_fail("ncclCommInitRank", r); }
  // what RCCL itself says about the communicator must be what the caller asked for (a wrong id / a stale rendezvous
  // would otherwise show up as a sum over fewer ranks than the deal of the links assumed)
  c->rccl_world = world;
  if (g_rccl.commCount) {
    int cnt = 0;
    if (g_rccl.commCount(c->comm, &cnt) == ncclSuccess) c->rccl_world = cnt;
  }
  if (c->rccl_world != world) {
    set_error("RCCL reports a communicator of " + std::to_string(c->rccl_world) + " ranks, the caller asked for " + std::to_string(world));
    tdtk_comm_destroy(c);
    return TDTK_EDEVICE;
  }
  if (int rc = reserve(c, DEFAULT_EXCHANGE_DOUBLES)) { tdtk_comm_destroy(c); return rc; }
  *out = c;
  return TDTK_OK;
}

void tdtk_comm_destroy(tdtk_comm* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipDeviceSynchronize();
  if (c->comm && g_rccl.commDestroy) g_rccl.commDestroy(c->comm);
  if (c->d_buf) (void)hipFree(c->d_buf);
  if (c->h_pin) (void)hipHostFree(c->h_pin);
  delete c;
}

int tdtk_comm_info(const tdtk_comm* c, int* rank, int* world, uint64_t* n_allreduce)
{
  if (!c) { set_error("NULL argument"); return TDTK_EINVAL; }
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  if (n_allreduce) *n_allreduce = c->n_allreduce;
  return TDTK_OK;
}

int tdtk_comm_rccl_world(const tdtk_comm* c)
{
  if (!c) { set_error("NULL argument"); return TDTK_EINVAL; }
  return c->rccl_world;
}

// blocks (host, n doubles) <- sum over the ranks, in place.  Every link has exactly one owner and everybody else
// holds zeros there, so the sum is exact and the result is the same on every rank and for every world size.
// `local_rc` != 0: this rank failed before the exchange (its link passes, say).  It still takes part -- with zeros for
// its blocks and a raised status slot behind them -- so that nobody waits for it, and EVERY rank returns an error
// afterwards (TDTK_EPEER where the failure was somebody else's).
static int exchange_with_status(tdtk_comm* c, double* blocks, size_t n, int local_rc)
{
  if (!c || (!blocks && n)) { set_error("NULL argument"); return TDTK_EINVAL; }
  if (c->dead || !c->comm) { set_error("communicator was aborted after an earlier failure"); return TDTK_EDEVICE; }
  const std::string local_why = local_rc ? tdtk_last_error() : "";
  if (hipSetDevice(c->device) != hipSuccess) { set_error("hipSetDevice failed"); return abort_comm(c, TDTK_EDEVICE); }
  if (kLab) {   // lab (tests): this rank cannot get its staging memory -- the failure the collective cannot carry
    const char* f = lab_env("TDTK_COMM_FAIL");
    if (f && std::strcmp(f, "reserve") == 0) { set_error("hipMalloc failed (injected: TDTK_COMM_FAIL=reserve)"); return abort_comm(c, TDTK_ENOMEM); }
  }
  if (int rc = reserve(c, n + 1)) return abort_comm(c, rc);
  hipStream_t stream = nullptr;
  {
    void* sp = nullptr;
    int rc = ctx_stream(c->device, &sp);
    if (rc) return abort_comm(c, rc);
    stream = static_cast<hipStream_t>(sp);
  }
  if (local_rc) std::memset(c->h_pin, 0, n * sizeof(double));
  else std::memcpy(c->h_pin, blocks, n * sizeof(double));
  c->h_pin[n] = local_rc ? 1.0 : 0.0;
  if (hipMemcpyAsync(c->d_buf, c->h_pin, (n + 1) * sizeof(double), hipMemcpyHostToDevice, stream) != hipSuccess) { set_error("H2D failed"); return abort_comm(c, TDTK_EDEVICE); }
  ncclResult_t r = g_rccl.allReduce(c->d_buf, c->d_buf, n + 1, ncclDouble, ncclSum, c->comm, stream);
  if (r != ncclSuccess) { nccl_fail("ncclAllReduce", r); return abort_comm(c, TDTK_EDEVICE); }
  c->n_allreduce++;
  if (hipMemcpyAsync(c->h_pin, c->d_buf, (n + 1) * sizeof(double), hipMemcpyDeviceToHost, stream) != hipSuccess) { set_error("D2H failed"); return abort_comm(c, TDTK_EDEVICE); }
  if (hipStreamSynchronize(stream) != hipSuccess) { set_error("stream sync failed after the all-reduce"); return abort_comm(c, TDTK_EDEVICE); }
  if (local_rc) { set_error(local_why + " (reported to the other ranks through the exchange)"); return local_rc; }
  if (c->h_pin[n] != 0.0) {
    set_error(std::to_string((int)c->h_pin[n]) + " other rank(s) failed before the exchange; this iteration is void on every rank");
    return TDTK_EPEER;
  }
  std::memcpy(blocks, c->h_pin, n * sizeof(double));
  return TDTK_OK;
}

int tdtk_graph_exchange(tdtk_comm* c, double* blocks, size_t n)
{
  if (!c || (!blocks && n)) { set_error("NULL argument"); return TDTK_EINVAL; }
  return exchange_with_status(c, blocks, n, 0);
}

// Who evaluates which link (FillGB3D's `omp parallel for` over links becomes one process per GPU).  A link costs one
// whole-scan pass over its SECOND scan.  Scans of equal size (or scan_points == NULL): chain links i -> i+1
// round-robin by their index, loop closures by (from + to) % world -- so a closure that appears or disappears between
// rounds (the graph is rebuilt from the poses every round, slam6D.cc:501-532) moves no other link and with it no
// resident tree or scan.  Scans of different size: longest processing time first -- links by descending point count of
// the second scan (ties: lower link index), each to the rank with the least points so far (ties: lower rank).
int tdtk_graph_deal_links(int nlinks, const int32_t* from, const int32_t* to, const uint64_t* scan_points, int nscans,
                          int world, int32_t* owner)
{
  if (nlinks < 0 || world < 1 || (nlinks && (!from || !to || !owner))) { set_error("bad argument"); return TDTK_EINVAL; }
  bool equal = true;
  if (scan_points)
    for (int i = 1; i < nscans; i++)
      if (scan_points[i] != scan_points[0]) { equal = false; break; }
  for (int l = 0; l < nlinks; l++)
    if (from[l] < 0 || to[l] < 0 || (scan_points && (from[l] >= nscans || to[l] >= nscans))) { set_error("link endpoint out of range"); return TDTK_EINVAL; }
  if (equal) {
    const int chain = nscans > 0 ? nscans - 1 : nlinks;
    for (int l = 0; l < nlinks; l++)
      owner[l] = (l < chain && to[l] == from[l] + 1) ? (l % world) : ((from[l] + to[l]) % world);
    return TDTK_OK;
  }
  std::vector<int> order(nlinks);
  for (int l = 0; l < nlinks; l++) order[l] = l;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scan_points[to[a]] > scan_points[to[b]]; });
  std::vector<uint64_t> load(world, 0);
  for (int l : order) {
    int best = 0;
    for (int r = 1; r < world; r++)
      if (load[r] < load[best]) best = r;
    owner[l] = best;
    load[best] += scan_points[to[l]];
  }
  return TDTK_OK;
}

// One global iteration of doGraphSlam6D of back-end `backend` with nothing but C++ between the link passes and the
// pose update: this rank's links (mine[0..n_mine), indices into the link list) in one batched device call, their
// blocks scattered into the zeroed block list of ALL links, one ncclAllReduce (skipped when comm is NULL or the world
// has one rank and TDTK_FORCE_ALLREDUCE is not set), then scatter + solve + pose update on every rank.
int tdtk_graph_iteration(int backend, tdtk_comm* comm, int nlinks, const int32_t* from, const int32_t* to, int n_mine,
                         const int32_t* mine, const tdtk_tree* const* first, const double* first_dalignxf,
                         tdtk_scan* const* second, double max_dist_match2, int nscans, double* transMat, double* dalignxf,
                         double* rPos, double* rPosTheta, tdtk_scan* const* scans, double* state, double* xf_out, double* ret)
{
  const int Bn = tdtk_graph_block_doubles(backend);
  if (Bn == 0) { set_error("unknown graph back-end"); return TDTK_EINVAL; }
  if (nlinks < 0 || n_mine < 0 || n_mine > nlinks || (n_mine && (!mine || !first || !first_dalignxf || !second))) {
    set_error("bad argument");
    return TDTK_EINVAL;
  }
  // everything that can be checked is checked BEFORE the link passes: an argument error is the same on every rank (or the
  // caller's bug on this one) and must not surface between "my links are done" and the collective
  for (int k = 0; k < n_mine; k++)
    if (mine[k] < 0 || mine[k] >= nlinks) { set_error("link index out of range"); return TDTK_EINVAL; }
  if (nlinks && (!from || !to)) { set_error("bad argument"); return TDTK_EINVAL; }
  const char* force = getenv("TDTK_FORCE_ALLREDUCE");
  const bool collective = comm && (comm->world > 1 || (force && force[0] == '1'));
  std::vector<double> blocks((size_t)nlinks * Bn, 0.0), mb((size_t)n_mine * Bn);
  int rc = tdtk_graph_link_blocks(backend, n_mine, first, first_dalignxf, second, max_dist_match2, mb.data());
  if (kLab && !rc) {   // lab (tests): this rank's link passes fail -- the failure the collective's status slot carries
    const char* f = lab_env("TDTK_COMM_FAIL");
    if (f && std::strcmp(f, "links") == 0) { set_error("link passes failed (injected: TDTK_COMM_FAIL=links)"); rc = TDTK_EDEVICE; }
  }
  if (rc && !collective) return rc;
  if (!rc)
    for (int k = 0; k < n_mine; k++) std::memcpy(&blocks[(size_t)mine[k] * Bn], &mb[(size_t)k * Bn], Bn * sizeof(double));
  if (collective) {
    // a rank whose link passes failed still enters the one collective of the iteration (status slot raised), so the
    // others are not left waiting in ncclAllReduce; all ranks return an error
    if ((rc = exchange_with_status(comm, blocks.data(), blocks.size(), rc))) return rc;
  }
  return tdtk_graph_solve_update(backend, nlinks, from, to, blocks.data(), nscans, transMat, dalignxf, rPos, rPosTheta,
                                 scans, state, xf_out, ret);
}

}  // extern "C"
