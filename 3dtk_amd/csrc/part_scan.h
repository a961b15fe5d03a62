// The partition of a level in two passes: what the two tree builders (build.hip, ann.hip) share.  See k_part_scan.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
namespace tdtk {
#ifndef WAVE
#define WAVE 64
#endif
#define PS_ROWS 8u
#define PS_THREADS 512u
#define PS_TILE (PS_THREADS * PS_ROWS)
__device__ __forceinline__ unsigned long long ps_pack(uint32_t cnt, uint32_t seen, uint32_t flag, uint32_t epoch)
{
  return ((unsigned long long)flag << 62) | ((unsigned long long)epoch << 54) | ((unsigned long long)seen << 53) | (unsigned long long)(cnt & 0x7FFFFFFu);
}
// a workgroup's tile number: tiles are numbered in the order their workgroups start, so a tile only ever waits for tiles
// that are running (the one that draws the last number puts the counter back for the next launch)
__device__ __forceinline__ uint32_t ps_draw_tile(uint32_t* __restrict__ counter, uint32_t ntiles)
{
  __shared__ uint32_t s_tile;
  if (threadIdx.x == 0) {
    const uint32_t t = atomicAdd(counter, 1u);
    if (t == ntiles - 1u) atomicExch(counter, 0u);
    s_tile = t;
  }
  __syncthreads();
  return s_tile;
}
// The segmented count: bit r of `gebits` / `headbits` = this lane's position of row r (tile base + wave * 64 PS_ROWS + 64 r + lane)
// counts / is the first position of its run.  geb[r] = counting positions of the run in front of that position, as far as
// this wave knows; where bit r of `ext` is set the run started in front of the wave and `win` has to be added.
__device__ __forceinline__ void ps_scan_core(const uint32_t gebits, const uint32_t headbits, const uint32_t tile,
                                             unsigned long long* __restrict__ status, const uint32_t epoch, uint32_t* __restrict__ err,
                                             uint32_t (&geb)[PS_ROWS], uint32_t& ext_out, uint32_t& win_out)
{
  __shared__ uint32_t s_wcnt[PS_THREADS / WAVE], s_wseen[PS_THREADS / WAVE], s_in;
  const uint32_t lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
  // the wave's rows in order: ballots, and a count carried from row to row (wave-uniform)
  const unsigned long long lt_mask = (1ull << lane) - 1ull, le_mask = lt_mask | (1ull << lane);
  uint32_t carry = 0u, seen = 0u, ext = 0u;
#pragma unroll
  for (uint32_t r = 0; r < PS_ROWS; r++) {
    const unsigned long long gm = __ballot((gebits >> r) & 1u), hm = __ballot((headbits >> r) & 1u);
    const unsigned long long hb = hm & le_mask;
    if (hb) {
      const int hl = 63 - __clzll((long long)hb);
      geb[r] = (uint32_t)__popcll(gm & lt_mask & ~((1ull << hl) - 1ull));
    } else {
      geb[r] = carry + (uint32_t)__popcll(gm & lt_mask);
      if (!seen) ext |= 1u << r;                          // the node started in front of this wave: + what comes in
    }
    if (hm) { const int hl = 63 - __clzll((long long)hm); carry = (uint32_t)__popcll(gm >> hl); seen = 1u; }
    else carry += (uint32_t)__popcll(gm);
  }
  if (lane == 0) { s_wcnt[wv] = carry; s_wseen[wv] = seen; }
  __syncthreads();
  if (wv == 0) {
    // the tile's aggregate: the count since the last node start in it (or all of it), and whether there was one
    uint32_t tcnt = 0u, tseen = 0u;
#pragma unroll
    for (uint32_t w = 0; w < PS_THREADS / WAVE; w++) { if (s_wseen[w]) { tcnt = s_wcnt[w]; tseen = 1u; } else tcnt += s_wcnt[w]; }
    uint32_t incoming = 0u;
    if (tile == 0) {
      if (lane == 0) __hip_atomic_store(&status[0], ps_pack(tcnt, 1u, 2u, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&status[tile], ps_pack(tcnt, tseen, tseen ? 2u : 1u, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int pos = (int)tile - 1;
      uint32_t spins = 0;
      for (;;) {
        const int idx = pos - (int)lane;
        const unsigned long long w = (idx >= 0) ? __hip_atomic_load(&status[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                                : ps_pack(0u, 1u, 2u, epoch);      // in front of the first tile: nothing
        const bool valid = ((uint32_t)(w >> 54) & 0xFFu) == epoch && (w >> 62) != 0ull;
        // the nearest predecessor that is final (its count since a node start is known, or a node starts in it) ends
        // the look-back; everything nearer must have its aggregate out
        const unsigned long long pm = __ballot(valid && (w >> 62) == 2ull);
        const unsigned long long vm = __ballot(valid);
        const int p = pm ? (__ffsll((long long)pm) - 1) : 64;
        const unsigned long long need = (p >= 64) ? ~0ull : ((2ull << p) - 1ull);
        if ((vm & need) != need) {
          if (++spins > (1u << 22)) { if (lane == 0) atomicOr(err, 0x10000u); break; }
          __builtin_amdgcn_s_sleep(1);
          continue;
        }
        uint32_t part = ((int)lane <= p) ? (uint32_t)(w & 0x7FFFFFFull) : 0u;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) part += (uint32_t)__shfl_xor((int)part, off, WAVE);
        incoming += part;
        if (p < 64) break;
        pos -= WAVE;
      }
      // a tile in which no node starts is final only now: the count at its end since the node start in front of it
      if (lane == 0 && !tseen) __hip_atomic_store(&status[tile], ps_pack(incoming + tcnt, 0u, 2u, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (lane == 0) s_in = incoming;
  }
  __syncthreads();
  // what comes into this wave: the waves in front of it back to a node start, the tile's incoming count behind them
  uint32_t win = 0u;
  {
    bool open = true;
    for (int w = (int)wv - 1; w >= 0 && open; w--) { win += s_wcnt[w]; if (s_wseen[w]) open = false; }
    if (open) win += s_in;
  }
  ext_out = ext; win_out = win;
}
size_t part_state_bytes(size_t n);      // bytes of the look-back state (zeroed once per build; epochs 1 .. 254)
}  // namespace tdtk
