// Offline model of how a wave64 of k_search_refill spends its VALU issue slots, for trying scheduling policies without
// a GPU.  Stand-alone: builds its own bucketed kd-tree (mean split on the longest box axis, bucket 20 -- the same
// shape family as the reference's, not bit-identical, which a scheduling study does not need), runs the reference's
// traversal per query (near child first, far child if myd^2 < best, box test on entry) and records each query's
// sequence of segments (internal nodes walked, then a bucket of k points, ...).  Queries are Morton-sorted and dealt
// to waves in slabs like the kernel does.
//   gcc -O2 -o /tmp/wave_sched tools/sim/wave_sched.c -lm && /tmp/wave_sched [points] [offset] [noise] [radius]
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double c[3], h[3], split; int axis, c1, c2, start, count; } Node;   // count > 0: leaf
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

// per-query trace: segs[] = nodes walked, bucket points, nodes walked, bucket points, ..., trailing nodes (no bucket: -1)
typedef struct { int nseg; short seg[64]; } Trace;
static double q[3], best; static int bk; static Trace* tr; static int cur_nodes;
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
  cur_nodes++;
  double a = fmax(fmax(fabs(q[0] - nd->c[0]) - nd->h[0], fabs(q[1] - nd->c[1]) - nd->h[1]), fabs(q[2] - nd->c[2]) - nd->h[2]);
  if (a >= 0 && a * a >= best) return;
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

// ---- wave model -------------------------------------------------------------------------
static double C_NODE = 30, C_PT4 = 56, C_POP = 8, C_REFILL = 60, C_LOOP = 6;
typedef struct { double valu, node_trips, pt_trips, refills; double lane_node, lane_pt; } Cost;

// policy: thresh = idle lanes needed for a refill; helpers = idle lanes help scanning buckets (0/1)
static void run_wave(const Trace* T, int nq, int thresh, int helpers, Cost* c)
{
  int lane_q[64], lane_seg[64], lane_left_nodes[64];
  for (int l = 0; l < 64; l++) lane_q[l] = -1;
  int next = 0;
  for (;;) {
    int idle = 0, active = 0;
    for (int l = 0; l < 64; l++) if (lane_q[l] < 0) idle++; else active++;
    if (next < nq && (active == 0 || idle >= thresh)) {
      for (int l = 0; l < 64 && next < nq; l++) if (lane_q[l] < 0) { lane_q[l] = next++; lane_seg[l] = 0; }
      c->valu += C_REFILL; c->refills += 1;
    }
    active = 0;
    for (int l = 0; l < 64; l++) if (lane_q[l] >= 0) active++;
    if (!active) break;
    // phase 1: every active lane walks the nodes of its current segment
    int trips = 0, lanes_sum = 0;
    for (int l = 0; l < 64; l++) if (lane_q[l] >= 0) { int n = T[lane_q[l]].seg[lane_seg[l]]; if (n > trips) trips = n; lanes_sum += n; }
    c->valu += trips * C_NODE + C_LOOP; c->node_trips += trips; c->lane_node += lanes_sum;
    // phase 2: buckets
    int ptrips = 0, psum = 0, nb = 0;
    for (int l = 0; l < 64; l++) if (lane_q[l] >= 0) {
      const Trace* t = &T[lane_q[l]];
      int k = (lane_seg[l] + 1 < t->nseg) ? t->seg[lane_seg[l] + 1] : -1;
      if (k > 0) { int tt = (k + 3) / 4; if (tt > ptrips) ptrips = tt; psum += tt; nb++; }
    }
    if (helpers && nb > 0) {
      int share = 64 / nb; if (share > 4) share = 4;     // lanes per bucket
      if (share >= 2) { ptrips = (ptrips + share - 1) / share; c->valu += 24; }   // broadcast + combine
    }
    if (nb) { c->valu += ptrips * C_PT4 + C_POP * 2; c->pt_trips += ptrips; c->lane_pt += psum; }
    for (int l = 0; l < 64; l++) if (lane_q[l] >= 0) {
      const Trace* t = &T[lane_q[l]];
      lane_seg[l] += 2;
      if (lane_seg[l] >= t->nseg) lane_q[l] = -1;
    }
  }
}

// "expensive queries first" (the previous ICP iteration's bucket count as the key): does the drain at the end of a
// slab get shorter, and the spread of the waves' lifetimes narrower?  The slab is handed out in `pieces` pieces.
static int cost_mode = 0;   // 0: buckets visited (what the kernel records), 1: node trips + point trips
static int cost_of(const Trace* t)
{
  int c = 0;
  if (cost_mode == 0) { for (int s = 1; s < t->nseg; s += 2) c++; return c; }
  for (int s = 0; s < t->nseg; s++) c += (s & 1) ? (t->seg[s] + 3) / 4 : t->seg[s];
  return c;
}
static int cmp_cost_desc(const void* a, const void* b) { return cost_of((const Trace*)b) - cost_of((const Trace*)a); }
static void order_slab(Trace* T, int nq, int pieces, int classes)
{
  const int per = nq / pieces;
  for (int p = 0; p < pieces; p++) {
    Trace* t = T + p * per;
    const int n = (p == pieces - 1) ? nq - p * per : per;
    if (classes == 0) qsort(t, n, sizeof(Trace), cmp_cost_desc);          // full order (qsort is not stable: fine for a model)
    else {                                                                 // two classes: >= `classes` buckets first, stable
      Trace tmp[512]; int k = 0;
      for (int i = 0; i < n; i++) if (cost_of(&t[i]) >= classes) tmp[k++] = t[i];
      for (int i = 0; i < n; i++) if (cost_of(&t[i]) < classes) tmp[k++] = t[i];
      memcpy(t, tmp, sizeof(Trace) * n);
    }
  }
}

int main(int argc, char** argv)
{
  int M = argc > 1 ? atoi(argv[1]) : 1000000;
  double offset = argc > 2 ? atof(argv[2]) : 0.0, noise = argc > 3 ? atof(argv[3]) : 1.0, radius = argc > 4 ? atof(argv[4]) : 25.0;
  int warm = argc > 5 ? atoi(argv[5]) : 1;
  P = malloc(sizeof(double) * 3 * M); idx = malloc(sizeof(int) * M); nodes = malloc(sizeof(Node) * (M / 4 + 16));
  for (int i = 0; i < 3 * M; i++) P[i] = urand() * 2000 - 1000;
  for (int i = 0; i < M; i++) idx[i] = i;
  build(0, M);
  // queries: model + noise + offset, Morton-sorted; a sample of slabs
  int NQ = M; double* Q = malloc(sizeof(double) * 3 * NQ); KI* ki = malloc(sizeof(KI) * NQ);
  for (int i = 0; i < NQ; i++) { for (int k = 0; k < 3; k++) Q[3 * i + k] = P[3 * i + k] + noise * nrand() + offset * (k == 0 ? 1.0 : k == 1 ? -0.5 : 0.3); ki[i].key = morton(Q + 3 * i); ki[i].i = i; }
  qsort(ki, NQ, sizeof(KI), cmpki);
  const int qpw = 224, nwaves = 400;
  Trace* T = malloc(sizeof(Trace) * qpw);
  double tot_nodes = 0, tot_leaves = 0, tot_pts = 0; long nqs = 0;
  Cost pol[8]; memset(pol, 0, sizeof pol);
  double lpt_sum[4] = {0, 0, 0, 0}, lpt_sq[4] = {0, 0, 0, 0};
  const char* names[8] = {"thresh 16 (kernel)", "thresh 8", "thresh 32", "thresh 64 (drain)", "thresh 16 + bucket helpers", "thresh 32 + bucket helpers", "thresh 1", "thresh 48"};
  for (int w = 0; w < nwaves; w++) {
    size_t base = (size_t)((double)w / nwaves * (NQ - qpw));
    for (int j = 0; j < qpw; j++) {
      int qi = ki[base + j].i;
      for (int k = 0; k < 3; k++) q[k] = Q[3 * qi + k];
      best = radius * radius;
      if (warm) {   // radius just above the distance to the point the previous pass found (here: the query's own origin)
        double d = 0; for (int k = 0; k < 3; k++) d += (P[3 * qi + k] - q[k]) * (P[3 * qi + k] - q[k]);
        if (d < best) best = d * (1 + 1e-15) + 1e-300;
      }
      bk = -1; tr = &T[j]; tr->nseg = 0; cur_nodes = 0;
      visit(0);
      if (tr->nseg < 63) tr->seg[tr->nseg++] = (short)cur_nodes;   // trailing walk without a bucket
      for (int s = 0; s < tr->nseg; s++) { if (s & 1) { tot_leaves++; tot_pts += tr->seg[s]; } else tot_nodes += tr->seg[s]; }
      nqs++;
    }
    run_wave(T, qpw, 16, 0, &pol[0]); run_wave(T, qpw, 8, 0, &pol[1]); run_wave(T, qpw, 32, 0, &pol[2]); run_wave(T, qpw, 64, 0, &pol[3]);
    run_wave(T, qpw, 16, 1, &pol[4]); run_wave(T, qpw, 32, 1, &pol[5]); run_wave(T, qpw, 1, 0, &pol[6]); run_wave(T, qpw, 48, 0, &pol[7]);
    {   // per-wave totals of the kernel's policy in slab order, in "expensive first" order (full / two classes)
      static Trace T2[512];
      Cost a0; memset(&a0, 0, sizeof a0); run_wave(T, qpw, 16, 0, &a0);
      memcpy(T2, T, sizeof(Trace) * qpw); order_slab(T2, qpw, 2, 0);
      Cost a1; memset(&a1, 0, sizeof a1); run_wave(T2, qpw, 16, 0, &a1);
      memcpy(T2, T, sizeof(Trace) * qpw); order_slab(T2, qpw, 1, 0);
      Cost a2; memset(&a2, 0, sizeof a2); run_wave(T2, qpw, 16, 0, &a2);
      cost_mode = 1;
      memcpy(T2, T, sizeof(Trace) * qpw); order_slab(T2, qpw, 1, 0);
      Cost a3; memset(&a3, 0, sizeof a3); run_wave(T2, qpw, 16, 0, &a3);
      cost_mode = 0;
      lpt_sum[3] += a3.valu; lpt_sq[3] += a3.valu * a3.valu;
      lpt_sum[0] += a0.valu; lpt_sq[0] += a0.valu * a0.valu;
      lpt_sum[1] += a1.valu; lpt_sq[1] += a1.valu * a1.valu;
      lpt_sum[2] += a2.valu; lpt_sq[2] += a2.valu * a2.valu;
    }
  }
  printf("%d points, offset %.1f noise %.1f radius %.1f warm %d: per query %.2f nodes, %.2f leaves, %.2f points\n", M, offset, noise, radius, warm,
         tot_nodes / nqs, tot_leaves / nqs, tot_pts / nqs);
  for (int p = 0; p < 8; p++)
    printf("%-30s VALU wave-instr/query %6.1f   node trips/query %5.2f (lane eff %4.1f%%)   point trips/query %5.2f (lane eff %4.1f%%)  refills/query %.3f\n", names[p],
           pol[p].valu / nqs, pol[p].node_trips / nqs, 100.0 * pol[p].lane_node / (64.0 * pol[p].node_trips), pol[p].pt_trips / nqs,
           100.0 * pol[p].lane_pt / (64.0 * pol[p].pt_trips), pol[p].refills / nqs);
  const char* ln[4] = {"slab order", "expensive first (buckets), 2 pieces", "expensive first (buckets), 1 piece", "expensive first (all trips), 1 piece"};
  for (int k = 0; k < 4; k++) {
    const double m = lpt_sum[k] / nwaves, sd = sqrt(lpt_sq[k] / nwaves - m * m);
    printf("%-44s VALU instr / wave: mean %8.0f  sd %6.0f (%.1f %%)  mean + 3.5 sd %8.0f\n", ln[k], m, sd, 100 * sd / m, m + 3.5 * sd);
  }
  return 0;
}
