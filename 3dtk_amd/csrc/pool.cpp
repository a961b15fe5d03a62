// Device memory of the handles (trees, resident scans), handed out again instead of returned to the driver.
//
// A search tree is six allocations (points, padded points, shadow groups, node records, radii, hot records) and one free
// of 1-35 MB each, a resident scan four to ten; hipMalloc / hipFree of such blocks cost tens of microseconds apiece and
// hipFree synchronises the whole device.  slam6D builds and drops trees all the time (a MetaScan's tree per match,
// 64 trees per graph-SLAM set-up), so freed blocks are kept -- up to TDTK_POOL_MB megabytes PER DEVICE, default 1024
// (a third of a percent of an MI355X's HBM; 0 turns the pool off) -- and the next request of about that size (no more
// than a quarter larger) takes one of them.  Memory on the shelf is invisible to every other allocator of the process
// (RCCL, rocPRIM, a host framework): tdtk_pool_trim() hands it back, every allocation of this library that fails is
// retried after a trim (pool_malloc here, the exchange's staging in comm.cpp), and a host that shares the GPU with
// another framework should set TDTK_POOL_MB to what it can spare.
// pool_free keeps hipFree's contract: when it returns, nothing on the device uses the block any more (it synchronises
// the device before it shelves a block, which is what hipFree itself does before unmapping one).
#include <cstdlib>
#include <map>
#include <mutex>

#include <hip/hip_runtime.h>

#include "tdtk_internal.h"

namespace tdtk {
namespace {
struct Shelf {
  std::mutex mu;
  std::multimap<size_t, void*> free_blocks[16];   // per device: capacity -> block
  std::map<void*, std::pair<size_t, int>> live;   // block -> (capacity, device)
  size_t shelved[16] = {0};                       // per device
};
Shelf& shelf() { static Shelf* s = new Shelf; return *s; }     // never destroyed: handles may outlive static destructors
size_t limit_bytes()
{
  static const size_t lim = [] {
    const char* e = getenv("TDTK_POOL_MB");
    const long mb = e ? atol(e) : 1024;
    return (size_t)(mb > 0 ? mb : 0) << 20;
  }();
  return lim;
}
}  // namespace
size_t pool_trim();
namespace {
size_t round_up(size_t b) { const size_t q = (b < ((size_t)1 << 20)) ? 4096 : ((size_t)1 << 16); return (b + q - 1) / q * q; }
}  // namespace

static hipError_t pool_malloc(void** out, size_t bytes)
{
  *out = nullptr;
  if (bytes == 0) bytes = 1;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const size_t want = round_up(bytes);
  Shelf& S = shelf();
  if (limit_bytes() && dev >= 0 && dev < 16) {
    std::lock_guard<std::mutex> lk(S.mu);
    auto& fb = S.free_blocks[dev];
    auto it = fb.lower_bound(want);
    if (it != fb.end() && it->first <= want + want / 4) {
      *out = it->second;
      S.live[*out] = {it->first, dev};
      S.shelved[dev] -= it->first;
      fb.erase(it);
      return hipSuccess;
    }
  }
  hipError_t e = hipMalloc(out, want);
  if (e != hipSuccess) {           // out of memory with blocks on the shelf: give them back and try once more
    (void)hipGetLastError();
    pool_trim();
    e = hipMalloc(out, want);
  }
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lk(S.mu);
    S.live[*out] = {want, dev};
  }
  return e;
}

void pool_free(void* p)
{
  if (!p) return;
  Shelf& S = shelf();
  size_t cap = 0;
  int dev = -1;
  {
    std::lock_guard<std::mutex> lk(S.mu);
    auto it = S.live.find(p);
    if (it != S.live.end()) { cap = it->second.first; dev = it->second.second; S.live.erase(it); }
  }
  if (dev < 0 || dev >= 16 || !limit_bytes() || cap > limit_bytes()) { (void)hipFree(p); return; }
  int cur = 0;
  (void)hipGetDevice(&cur);
  if (cur != dev) (void)hipSetDevice(dev);
  (void)hipDeviceSynchronize();      // hipFree's contract: no kernel touches the block after this call
  if (cur != dev) (void)hipSetDevice(cur);
  std::lock_guard<std::mutex> lk(S.mu);
  // make room on THIS device's shelf (the budget is per device): its smallest blocks go first
  auto& fb = S.free_blocks[dev];
  if (cur != dev) (void)hipSetDevice(dev);
  while (S.shelved[dev] + cap > limit_bytes() && !fb.empty()) {
    auto it = fb.begin();
    (void)hipFree(it->second);
    S.shelved[dev] -= it->first;
    fb.erase(it);
  }
  if (cur != dev) (void)hipSetDevice(cur);
  fb.insert({cap, p});
  S.shelved[dev] += cap;
}

int pool_malloc_raw(void** out, size_t bytes) { return (int)pool_malloc(out, bytes); }

size_t pool_trim()
{
  Shelf& S = shelf();
  std::lock_guard<std::mutex> lk(S.mu);
  size_t released = 0;
  for (int d = 0; d < 16; d++) {
    auto& fb = S.free_blocks[d];
    for (auto& kv : fb) { (void)hipFree(kv.second); released += kv.first; }   // (hipFree takes a pointer of any device)
    fb.clear();
    S.shelved[d] = 0;
  }
  return released;
}

}  // namespace tdtk
