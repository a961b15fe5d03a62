// Offline model, second generation: lane-level replay of k_search_refill's scheduling policies PLUS a SIMD timing model
// (W resident waves share one vector ALU; every trip is `issue` cycles on it followed by a memory wait that other waves
// may fill).  Used in round 3 to choose between (A) the kernel as it is, (B) capped inner loops with carried-over lanes
// and (C) a per-wave pool of query states in LDS from which 64 steps of ONE kind are drawn per trip.
//   gcc -O2 -o /tmp/wave_pool tools/sim/wave_pool.c -lm && /tmp/wave_pool [points] [offset] [noise] [radius] [warm]
// Same stand-alone tree / trace generator as wave_sched.c (same shape family as the reference's tree, not bit-identical).
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double c[3], h[3], split; int axis, c1, c2, start, count; } Node;
static Node* nodes; static int nn;
static double* P; static int* idx;
static uint64_t rs = 88172645463325252ull;
static double urand(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (rs >> 11) * (1.0 / 9007199254740992.0); }
static double nrand(void) { double u = urand(), v = urand(); return sqrt(-2 * log(u + 1e-300)) * cos(6.283185307179586 * v); }
static int build(int lo, int n)
{
  int me = nn++;
  Node* nd = &nodes[me];
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300}, mean[3] = {0, 0, 0};
  for (int i = lo; i < lo + n; i++)
    for (int k = 0; k < 3; k++) { double v = P[3 * idx[i] + k]; if (v < mn[k]) mn[k] = v; if (v > mx[k]) mx[k] = v; mean[k] += v; }
  for (int k = 0; k < 3; k++) { nd->c[k] = 0.5 * (mn[k] + mx[k]); nd->h[k] = 0.5 * (mx[k] - mn[k]); mean[k] /= n; }
  if (n <= 20) { nd->start = lo; nd->count = n; return me; }
  int ax = 0; if (nd->h[1] > nd->h[ax]) ax = 1; if (nd->h[2] > nd->h[ax]) ax = 2;
  nd->axis = ax; nd->split = mean[ax]; nd->count = 0;
  int i = lo, j = lo + n - 1;
  while (i <= j) { while (i <= j && P[3 * idx[i] + ax] < nd->split) i++; while (i <= j && P[3 * idx[j] + ax] >= nd->split) j--; if (i < j) { int t = idx[i]; idx[i] = idx[j]; idx[j] = t; } }
  int nl = i - lo;
  if (nl == 0 || nl == n) { nd->start = lo; nd->count = n; return me; }
  int a = build(lo, nl), b = build(lo + nl, n - nl);
  nodes[me].c1 = a; nodes[me].c2 = b;
  return me;
}
typedef struct { int nseg; short seg[64]; } Trace;   // nodes, bucket points, nodes, bucket points, ..., trailing nodes
static double q[3], best; static int bk; static Trace* tr; static int cur_nodes; static int FREE_PRUNED = 0; static long n_pruned = 0, n_visits = 0;
static void emit_bucket(int k) { if (tr->nseg < 62) { tr->seg[tr->nseg++] = (short)cur_nodes; tr->seg[tr->nseg++] = (short)k; } cur_nodes = 0; }
static void visit(int ni)
{
  const Node* nd = &nodes[ni];
  if (nd->count > 0) {
    emit_bucket(nd->count);
    for (int i = nd->start; i < nd->start + nd->count; i++) {
      const double* p = P + 3 * idx[i];
      double d = (p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]);
      if (d < best) { best = d; bk = idx[i]; }
    }
    return;
  }
  double a = fmax(fmax(fabs(q[0] - nd->c[0]) - nd->h[0], fabs(q[1] - nd->c[1]) - nd->h[1]), fabs(q[2] - nd->c[2]) - nd->h[2]);
  n_visits++;
  if (a >= 0 && a * a >= best) { n_pruned++; if (!FREE_PRUNED) cur_nodes++; return; }
  cur_nodes++;
  double myd = q[nd->axis] - nd->split;
  if (myd < 0) { visit(nd->c1); if (myd * myd < best) visit(nd->c2); }
  else { visit(nd->c2); if (myd * myd < best) visit(nd->c1); }
}
static uint64_t morton(const double* p)
{
  uint64_t m = 0;
  for (int k = 0; k < 3; k++) {
    uint32_t v = (uint32_t)((p[k] + 1100.0) / 2200.0 * 1023.0);
    for (int b = 0; b < 10; b++) m |= (uint64_t)((v >> b) & 1) << (3 * b + k);
  }
  return m;
}
typedef struct { uint64_t key; int i; } KI;
static int cmpki(const void* a, const void* b) { uint64_t x = ((const KI*)a)->key, y = ((const KI*)b)->key; return x < y ? -1 : x > y; }

// ---- per-wave trip sequences --------------------------------------------------------------------------------------
// a trip = (issue instructions, memory round trip yes/no); a wave's run is a list of trips
typedef struct { float issue; unsigned char mem; } Trip;
typedef struct { Trip* t; int n, cap; double lanes_node, lanes_pt, node_trips, pt_trips, valu; } Run;
static void push_trip(Run* r, double issue, int mem)
{
  if (r->n == r->cap) { r->cap = r->cap ? 2 * r->cap : 256; r->t = realloc(r->t, sizeof(Trip) * r->cap); }
  r->t[r->n].issue = (float)issue; r->t[r->n].mem = (unsigned char)mem; r->n++; r->valu += issue;
}
static double C_NODE = 32, C_PT4 = 56, C_POP = 8, C_REFILL = 60, C_LOOP = 6;
static double C_SEL = 40;     // pool: choosing 64 states + reading / writing them in LDS, per trip
static int cost_key(const Trace* t) { int c = 0; for (int s = 0; s < t->nseg; s++) c += (s & 1) ? 4 : t->seg[s]; return c; }
static int cmp_cost_desc(const void* a, const void* b) { return cost_key((const Trace*)b) - cost_key((const Trace*)a); }

// lane / slot state
typedef struct { int q, seg, rem; } Slot;    // rem: nodes left (even seg) or 4-point trips left (odd seg)
static void slot_start(Slot* s, const Trace* T, int qi) { s->q = qi; s->seg = 0; s->rem = T[qi].seg[0]; }
// after finishing the current segment: move to the next one; returns 0 when the query is finished
static int slot_advance(Slot* s, const Trace* T)
{
  const Trace* t = &T[s->q];
  for (;;) {
    s->seg++;
    if (s->seg >= t->nseg) { s->q = -1; return 0; }
    s->rem = (s->seg & 1) ? (t->seg[s->seg] + 3) / 4 : t->seg[s->seg];
    if (s->rem > 0) return 1;       // a zero-node walk (the popped child is a bucket) costs nothing
  }
}

// (A) the kernel as it is (K1 = K2 = 1: inner loops run until no lane is left in them), or (B) capped loops
static void run_nested(const Trace* T, int nq, int thresh, int K1, int K2, Run* r)
{
  Slot L[64];
  for (int l = 0; l < 64; l++) L[l].q = -1;
  int next = 0;
  for (;;) {
    int idle = 0;
    for (int l = 0; l < 64; l++) if (L[l].q < 0) idle++;
    if (next < nq && (idle == 64 || idle >= thresh)) {
      for (int l = 0; l < 64 && next < nq; l++) if (L[l].q < 0) { slot_start(&L[l], T, next++); if (L[l].rem == 0) slot_advance(&L[l], T); }
      push_trip(r, C_REFILL, 1);
    }
    int nw = 0, ns = 0;
    for (int l = 0; l < 64; l++) if (L[l].q >= 0) { if (L[l].seg & 1) ns++; else nw++; }
    if (nw + ns == 0) { if (next >= nq) break; continue; }
    int did = 0;
    int run1 = nw > 0 && (nw >= K1 || ns == 0);
    const int run2 = ns > 0 && (ns >= K2 || nw == 0);
    if (!run1 && !run2 && nw >= ns) run1 = 1;
    // phase 1
    if (run1) do {
      push_trip(r, C_NODE, 1); r->node_trips++; r->lanes_node += nw; did = 1;
      for (int l = 0; l < 64; l++) if (L[l].q >= 0 && !(L[l].seg & 1)) {
        if (--L[l].rem == 0) { if (slot_advance(&L[l], T)) { if (L[l].seg & 1) { ns++; nw--; } } else nw--; }
      }
    } while (nw > 0 && (nw >= K1 || ns == 0));
    push_trip(r, C_LOOP, 0);
    int did2 = 0;
    if (ns > 0 && (ns >= K2 || nw == 0 || !did)) do {
      push_trip(r, C_PT4, 1); r->pt_trips++; r->lanes_pt += ns; did2 = 1;
      int still = 0;
      for (int l = 0; l < 64; l++) if (L[l].q >= 0 && (L[l].seg & 1) && L[l].rem > 0) {
        if (--L[l].rem == 0) L[l].rem = -1; else still++;
      }
      ns = still;
    } while (ns > 0 && (ns >= K2 || nw == 0));
    if (did2) push_trip(r, 2 * C_POP, 2);
    for (int l = 0; l < 64; l++) if (L[l].q >= 0 && (L[l].seg & 1) && L[l].rem == -1) slot_advance(&L[l], T);
  }
}

// (C) the pool: NS slots per wave; per trip the kind with more candidates is chosen (nodes preferred when >= 64 of them)
static void run_pool(const Trace* T, int nq, int NS, int refill_free, Run* r)
{
  Slot* S = malloc(sizeof(Slot) * NS);
  for (int i = 0; i < NS; i++) S[i].q = -1;
  int next = 0;
  for (;;) {
    int fr = 0;
    for (int i = 0; i < NS; i++) if (S[i].q < 0) fr++;
    if (next < nq && (fr == NS || fr >= refill_free)) {
      int filled = 0;
      for (int i = 0; i < NS && next < nq && filled < 64; i++) if (S[i].q < 0) { slot_start(&S[i], T, next++); if (S[i].rem == 0) slot_advance(&S[i], T); filled++; }
      push_trip(r, C_REFILL, 1);
      continue;
    }
    int nw = 0, ns = 0;
    for (int i = 0; i < NS; i++) if (S[i].q >= 0) { if (S[i].seg & 1) ns++; else nw++; }
    if (nw + ns == 0) { if (next >= nq) break; continue; }
    const int kind = (nw >= 64) ? 0 : (ns >= 64) ? 1 : (nw >= ns ? 0 : 1);
    int taken = 0;
    for (int i = 0; i < NS && taken < 64; i++) if (S[i].q >= 0 && ((S[i].seg & 1) == kind)) {
      taken++;
      if (--S[i].rem == 0) slot_advance(&S[i], T);
    }
    if (kind == 0) { push_trip(r, C_NODE + C_SEL, 1); r->node_trips++; r->lanes_node += taken; }
    else { push_trip(r, C_PT4 + C_SEL + C_POP, 1); r->pt_trips++; r->lanes_pt += taken; }
  }
  free(S);
}

// (D) J query contexts per lane (one in registers, J-1 parked in LDS), wave-majority vote on the kind of step, a lane whose
// active context is of the other kind swaps to a parked one of the chosen kind (if it has one), capped inner loops
static double C_SWAP = 26, C_VOTE = 8;
static void run_ctx(const Trace* T, int nq, int J, int thresh, int K1, int K2, double bias, Run* r)
{
  Slot S[64][4]; int act[64];
  for (int l = 0; l < 64; l++) { act[l] = 0; for (int j = 0; j < J; j++) S[l][j].q = -1; }
  int next = 0;
  for (;;) {
    int fr = 0, total = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < J; j++) { total++; if (S[l][j].q < 0) fr++; }
    if (next < nq && (fr == total || fr >= thresh)) {
      // one refill event hands out at most 64 queries: one per lane (into the lane's first free context)
      for (int l = 0; l < 64 && next < nq; l++) for (int j = 0; j < J; j++) if (S[l][j].q < 0) { slot_start(&S[l][j], T, next++); if (S[l][j].rem == 0) slot_advance(&S[l][j], T); break; }
      push_trip(r, C_REFILL, 1);
    }
    int cn = 0, cs = 0;
    for (int l = 0; l < 64; l++) { int hn = 0, hs = 0; for (int j = 0; j < J; j++) if (S[l][j].q >= 0) { if (S[l][j].seg & 1) hs = 1; else hn = 1; } cn += hn; cs += hs; }
    if (cn + cs == 0) { if (next >= nq) break; continue; }
    const int kind = (cn * bias >= cs) ? 0 : 1;
    // select
    int sel[64], swaps = 0, n = 0;
    for (int l = 0; l < 64; l++) {
      sel[l] = -1;
      if (S[l][act[l]].q >= 0 && (S[l][act[l]].seg & 1) == kind) sel[l] = act[l];
      else for (int j = 0; j < J; j++) if (S[l][j].q >= 0 && (S[l][j].seg & 1) == kind) { sel[l] = j; swaps++; act[l] = j; break; }
      if (sel[l] >= 0) n++;
    }
    push_trip(r, C_VOTE + (swaps ? C_SWAP : 0), swaps ? 1 : 0);
    const int K = kind == 0 ? K1 : K2;
    int first = 1;
    while (n > 0 && (first || n >= K)) {
      first = 0;
      if (kind == 0) { push_trip(r, C_NODE, 1); r->node_trips++; r->lanes_node += n; }
      else { push_trip(r, C_PT4, 1); r->pt_trips++; r->lanes_pt += n; }
      for (int l = 0; l < 64; l++) if (sel[l] >= 0) {
        Slot* s = &S[l][sel[l]];
        if (--s->rem == 0) {
          if (kind == 1) { s->rem = -1; sel[l] = -1; n--; }       // bucket through: pops at the end of the loop
          else { if (!slot_advance(s, T) || (s->seg & 1)) { sel[l] = -1; n--; } }
        }
      }
    }
    if (kind == 1) {
      push_trip(r, 2 * C_POP, 1);
      for (int l = 0; l < 64; l++) for (int j = 0; j < J; j++) if (S[l][j].q >= 0 && (S[l][j].seg & 1) && S[l][j].rem == -1) slot_advance(&S[l][j], T);
    }
  }
}

// (E) J queries per lane side by side in registers: the nested loops of the kernel, but every trip takes one step for each of
// the lane's J contexts that needs one (the loads of all J are in flight together: ONE memory wait per trip)
static void run_multi(const Trace* T, int nq, int J, int thresh, Run* r)
{
  Slot S[64][4];
  for (int l = 0; l < 64; l++) for (int j = 0; j < J; j++) S[l][j].q = -1;
  int next = 0;
  for (;;) {
    int fr = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < J; j++) if (S[l][j].q < 0) fr++;
    if (next < nq && (fr == 64 * J || fr >= thresh)) {
      for (int j = 0; j < J; j++) for (int l = 0; l < 64 && next < nq; l++) if (S[l][j].q < 0) { slot_start(&S[l][j], T, next++); if (S[l][j].rem == 0) slot_advance(&S[l][j], T); }
      push_trip(r, C_REFILL, 1);
    }
    int any = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < J; j++) if (S[l][j].q >= 0) any = 1;
    if (!any) { if (next >= nq) break; continue; }
    // phase 1: node trips until no context is walking
    for (;;) {
      double issue = 0; int lanes = 0, sets = 0;
      for (int j = 0; j < J; j++) {
        int n = 0;
        for (int l = 0; l < 64; l++) if (S[l][j].q >= 0 && !(S[l][j].seg & 1)) n++;
        if (n) { issue += C_NODE; lanes += n; sets++; }
      }
      if (!sets) break;
      push_trip(r, issue, 1); r->node_trips += sets; r->lanes_node += lanes;
      for (int j = 0; j < J; j++) for (int l = 0; l < 64; l++) if (S[l][j].q >= 0 && !(S[l][j].seg & 1)) { if (--S[l][j].rem == 0) slot_advance(&S[l][j], T); }
    }
    push_trip(r, C_LOOP, 0);
    int didb = 0;
    for (;;) {
      double issue = 0; int lanes = 0, sets = 0;
      for (int j = 0; j < J; j++) {
        int n = 0;
        for (int l = 0; l < 64; l++) if (S[l][j].q >= 0 && (S[l][j].seg & 1) && S[l][j].rem > 0) n++;
        if (n) { issue += C_PT4; lanes += n; sets++; }
      }
      if (!sets) break;
      didb = 1;
      push_trip(r, issue, 1); r->pt_trips += sets; r->lanes_pt += lanes;
      for (int j = 0; j < J; j++) for (int l = 0; l < 64; l++) if (S[l][j].q >= 0 && (S[l][j].seg & 1) && S[l][j].rem > 0) { if (--S[l][j].rem == 0) S[l][j].rem = -1; }
    }
    if (didb) push_trip(r, 2 * C_POP * J, 2);
    for (int j = 0; j < J; j++) for (int l = 0; l < 64; l++) if (S[l][j].q >= 0 && (S[l][j].seg & 1) && S[l][j].rem == -1) slot_advance(&S[l][j], T);
  }
}

// ---- SIMD timing: W waves share one vector ALU ------------------------------------------------------------------------
// each trip occupies the ALU for 4 * issue cycles (wave64 on a 16-lane SIMD), then the wave waits `lat` cycles if the
// trip ends in a memory round trip.  Returns cycles until all W waves are through; *busy = ALU busy fraction.
static double simd_time(Run* runs, int W, double lat, double* busy)
{
  int pos[16]; double ready[16];
  for (int w = 0; w < W; w++) { pos[w] = 0; ready[w] = 0; }
  double t = 0, work = 0;
  for (;;) {
    int pick = -1; double br = 1e300;
    for (int w = 0; w < W; w++) if (pos[w] < runs[w].n && ready[w] < br) { br = ready[w]; pick = w; }
    if (pick < 0) break;
    if (t < br) t = br;
    const Trip* tp = &runs[pick].t[pos[pick]++];
    const double d = 4.0 * tp->issue;
    t += d; work += d;
    ready[pick] = t + (tp->mem == 1 ? lat : tp->mem == 2 ? 130.0 : 0);
  }
  double end = t;
  for (int w = 0; w < W; w++) if (ready[w] > end) end = ready[w];
  *busy = work / end;
  return end;
}

typedef struct { const char* name; int type; double W; int qpw; int p1, p2, p3; double bias; int ordered; } Pol;
// type 0: nested (thresh p1, K1 p2, K2 p3); 1: pool (slots p1, refill_free p2); 2: contexts (J p1, K1 p2, K2 p3)
static const Pol POL[] = {
  {"A  kernel: nested loops, 224/wave, W=4.4", 0, 4.4, 224, 16, 1, 1, 1, 1},
  {"A' same, slab order", 0, 4.4, 224, 16, 1, 1, 1, 0},
  {"A  W=6 160/wave", 0, 6, 160, 16, 1, 1, 1, 1},
  {"A  W=6 192/wave (5.2 in flight)", 0, 5.2, 192, 16, 1, 1, 1, 1},
  {"A  W=4 256/wave", 0, 4, 256, 16, 1, 1, 1, 1},
  {"A  W=2.6 384/wave", 0, 2.6, 384, 16, 1, 1, 1, 1},
  {"A  W=2 512/wave", 0, 2, 512, 16, 1, 1, 1, 1},
  {"B  capped loops K 16/16", 0, 4.4, 224, 16, 16, 16, 1, 1},
  {"C  pool 128 slots W=4", 1, 4, 256, 128, 32, 0, 1, 1},
  {"C  pool 96 slots W=5", 1, 5, 205, 96, 32, 0, 1, 1},
  {"E  2 queries/lane W=4 256/wave", 3, 4, 256, 2, 16, 0, 1, 1},
  {"E  2 queries/lane W=4 thresh 32", 3, 4, 256, 2, 32, 0, 1, 1},
  {"E  2 queries/lane W=4.4 224/wave", 3, 4.4, 224, 2, 16, 0, 1, 1},
  {"E  2 queries/lane W=3 341/wave", 3, 3, 341, 2, 32, 0, 1, 1},
  {"E  3 queries/lane W=3 341/wave", 3, 3, 341, 3, 32, 0, 1, 1},
  {"E  2 queries/lane W=4 slab order", 3, 4, 256, 2, 16, 0, 1, 0},
  {"D  J=2 W=4 K 1/1", 2, 4, 256, 2, 1, 1, 1, 1},
  {"D  J=2 W=4 K 16/16", 2, 4, 256, 2, 16, 16, 1, 1},
  {"D  J=2 W=4 K 32/32", 2, 4, 256, 2, 32, 32, 1, 1},
  {"D  J=2 W=4 K 40/32", 2, 4, 256, 2, 40, 32, 1, 1},
  {"D  J=2 W=4 K 48/40", 2, 4, 256, 2, 48, 40, 1, 1},
  {"D  J=2 W=4 K 32/32 slab order", 2, 4, 256, 2, 32, 32, 1, 0},
  {"D  J=2 W=4 K 32/32 bias 0.7", 2, 4, 256, 2, 32, 32, 0.7, 1},
  {"D  J=2 W=4 K 32/32 bias 1.4", 2, 4, 256, 2, 32, 32, 1.4, 1},
  {"D  J=2 W=4 K 32/32 320/wave", 2, 4, 320, 2, 32, 32, 1, 1},
  {"D  J=2 W=5 K 32/32", 2, 5, 205, 2, 32, 32, 1, 1},
  {"D  J=3 W=3 K 32/32", 2, 3, 341, 3, 32, 32, 1, 1},
  {"D  J=3 W=3 K 48/48", 2, 3, 341, 3, 48, 48, 1, 1},
  {"D  J=3 W=2 K 48/48", 2, 2, 512, 3, 48, 48, 1, 1},
  {"D  J=4 W=2 K 48/48", 2, 2, 512, 4, 48, 48, 1, 1},
};
int main(int argc, char** argv)
{
  int M = argc > 1 ? atoi(argv[1]) : 1000000;
  double offset = argc > 2 ? atof(argv[2]) : 0.0, noise = argc > 3 ? atof(argv[3]) : 1.0, radius = argc > 4 ? atof(argv[4]) : 25.0;
  int warm = argc > 5 ? atoi(argv[5]) : 1;
  double LAT = argc > 6 ? atof(argv[6]) : 1200;
  FREE_PRUNED = argc > 7 ? atoi(argv[7]) : 0;
  P = malloc(sizeof(double) * 3 * M); idx = malloc(sizeof(int) * M); nodes = malloc(sizeof(Node) * (M / 4 + 16));
  for (int i = 0; i < 3 * M; i++) P[i] = urand() * 2000 - 1000;
  for (int i = 0; i < M; i++) idx[i] = i;
  build(0, M);
  int NQ = M; double* Q = malloc(sizeof(double) * 3 * NQ); KI* ki = malloc(sizeof(KI) * NQ);
  for (int i = 0; i < NQ; i++) { for (int k = 0; k < 3; k++) Q[3 * i + k] = P[3 * i + k] + noise * nrand() + offset * (k == 0 ? 1.0 : k == 1 ? -0.5 : 0.3); ki[i].key = morton(Q + 3 * i); ki[i].i = i; }
  qsort(ki, NQ, sizeof(KI), cmpki);
  const int span = 2600, nsimd = 40;      // a SIMD's neighbourhood of the sorted scan
  const int NPOL = (int)(sizeof(POL) / sizeof(POL[0]));
  double tsum[64] = {0}, bsum[64] = {0}, vsum[64] = {0}, ntr[64] = {0}, ptr_[64] = {0}, ln[64] = {0}, lp[64] = {0};
  Trace* T = malloc(sizeof(Trace) * span);
  static Trace T2[4096];
  double tot_nodes = 0, tot_leaves = 0, tot_pts = 0; long nqs = 0;
  for (int s = 0; s < nsimd; s++) {
    size_t base = (size_t)((double)s / nsimd * (NQ - span));
    for (int j = 0; j < span; j++) {
      int qi = ki[base + j].i;
      for (int k = 0; k < 3; k++) q[k] = Q[3 * qi + k];
      best = radius * radius;
      if (warm) { double d = 0; for (int k = 0; k < 3; k++) d += (P[3 * qi + k] - q[k]) * (P[3 * qi + k] - q[k]); if (d < best) best = d * (1 + 1e-15) + 1e-300; }
      bk = -1; tr = &T[j]; tr->nseg = 0; cur_nodes = 0;
      visit(0);
      if (tr->nseg < 63) tr->seg[tr->nseg++] = (short)cur_nodes;
      for (int u = 0; u < tr->nseg; u++) { if (u & 1) { tot_leaves++; tot_pts += tr->seg[u]; } else tot_nodes += tr->seg[u]; }
      nqs++;
    }
    for (int p = 0; p < NPOL; p++) {
      const Pol* pl = &POL[p];
      const int Wlo = (int)floor(pl->W), Whi = (int)ceil(pl->W);
      for (int Wsim = Wlo; Wsim <= Whi; Wsim++) {
        const double wgt = (Wlo == Whi) ? 1.0 : (Wsim == Wlo ? (Whi - pl->W) : (pl->W - Wlo));
        Run runs[16]; memset(runs, 0, sizeof runs);
        for (int w = 0; w < Wsim; w++) {
          const int lo = (w * pl->qpw) % (span - pl->qpw + 1);
          memcpy(T2, T + lo, sizeof(Trace) * pl->qpw);
          if (pl->ordered) qsort(T2, pl->qpw, sizeof(Trace), cmp_cost_desc);
          if (pl->type == 0) run_nested(T2, pl->qpw, pl->p1, pl->p2, pl->p3, &runs[w]);
          else if (pl->type == 1) run_pool(T2, pl->qpw, pl->p1, pl->p2, &runs[w]);
          else if (pl->type == 2) run_ctx(T2, pl->qpw, pl->p1, 16, pl->p2, pl->p3, pl->bias, &runs[w]);
          else run_multi(T2, pl->qpw, pl->p1, pl->p2, &runs[w]);
        }
        double busy; const double t = simd_time(runs, Wsim, LAT, &busy);
        const double nqd = (double)Wsim * pl->qpw;
        tsum[p] += wgt * t / nqd / nsimd; bsum[p] += wgt * busy / nsimd;
        for (int w = 0; w < Wsim; w++) {
          vsum[p] += wgt * runs[w].valu / nqd / nsimd; ntr[p] += wgt * runs[w].node_trips / nqd / nsimd; ptr_[p] += wgt * runs[w].pt_trips / nqd / nsimd;
          ln[p] += wgt * runs[w].lanes_node / nsimd / nqd; lp[p] += wgt * runs[w].lanes_pt / nsimd / nqd;
          free(runs[w].t);
        }
      }
    }
  }
  printf("internal-node visits %.2f per query, pruned by the box test %.2f (free: %d)\n", (double)n_visits / nqs, (double)n_pruned / nqs, FREE_PRUNED);
  printf("%d points, offset %.1f noise %.1f radius %.1f warm %d, latency %.0f cycles: per query %.2f nodes, %.2f leaves, %.2f points\n", M, offset, noise, radius, warm, LAT,
         tot_nodes / nqs, tot_leaves / nqs, tot_pts / nqs);
  for (int p = 0; p < NPOL; p++)
    printf("%-34s cyc/query/SIMD %6.1f (x%.2f) ALU busy %4.1f%% VALU/query %5.1f node trips %5.2f (eff %4.1f%%) pt trips %5.2f (eff %4.1f%%) lane eff %4.1f%%\n", POL[p].name, tsum[p],
           tsum[p] / tsum[0], 100 * bsum[p], vsum[p], ntr[p], 100 * ln[p] / (64 * ntr[p] + 1e-9), ptr_[p], 100 * lp[p] / (64 * ptr_[p] + 1e-9),
           100 * (ln[p] * C_NODE + lp[p] * C_PT4) / (64 * vsum[p]));
  return 0;
}
