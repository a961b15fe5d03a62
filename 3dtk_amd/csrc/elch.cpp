// ELCH loop closing (-L 1), host side: the graph balancer of elch6D::graph_balancer (src/slam6d/elch6D.cc:186-279)
// without Boost.Graph, and the merge of several whole-scan pair-sum blocks into one (a MetaScan as the DATA scan of
// icp6D::match: Scan::getPtPairsParallel walks its member scans, scan.cc:1305-1327, and Align_Parallel merges the
// per-call partial sums, icp6Dquat.cc:533-588).  The data-parallel parts of elch6Deuler::close_loop
// (elch6Deuler.cc:44-138) -- one covarianceEuler per graph edge, one ICP match, one transform of every scan -- are
// the library's batched link passes / resident match / batched transform; this file is the control flow between them.
//
// Parity: unpinned (elch6D.cc needs Boost.Graph; no reference test pins its output).  Boost's
// dijkstra_shortest_paths relaxes with a strict '<' and leaves predecessor[v] == v for unreached vertices; both are
// kept.  Among EXACTLY equal path lengths Boost's 4-ary heap order decides, which is not reproduced -- edge weights
// here are |diag(C^-1)| of real links, where exact ties do not occur.  Round 3: the balancer is written on flat arrays
// (edge switches, position-indexed open ends) instead of following the reference's list-iterator formulation.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

#include "tdtk_internal.h"

using namespace tdtk;

namespace {

// The loop graph as flat arrays: edge e joins ea[e] and eb[e] with weight ew[e] and can be switched off; every vertex
// lists its incident edges in the order they were given (inc[first[v] .. first[v+1])), which is the order the
// relaxations of the shortest-path search run in.  Parallel edges are allowed; a self loop is listed once.
struct LoopGraph {
  int nv = 0;
  std::vector<int> ea, eb, first, inc, live_deg;
  std::vector<double> ew;
  std::vector<char> on;

  void build(int nvertices, int nedges, const int32_t* from, const int32_t* to, const double* w)
  {
    nv = nvertices;
    ea.assign(from, from + nedges); eb.assign(to, to + nedges); ew.assign(w, w + nedges);
    on.assign((size_t)nedges, 1);
    live_deg.assign((size_t)nv, 0);
    first.assign((size_t)nv + 1, 0);
    for (int e = 0; e < nedges; e++) { first[ea[e] + 1]++; if (eb[e] != ea[e]) first[eb[e] + 1]++; }
    for (int v = 0; v < nv; v++) { live_deg[v] = first[v + 1]; first[v + 1] += first[v]; }
    inc.resize((size_t)first[nv]);
    std::vector<int> fill(first.begin(), first.end() - 1);
    for (int e = 0; e < nedges; e++) { inc[fill[ea[e]]++] = e; if (eb[e] != ea[e]) inc[fill[eb[e]]++] = e; }
  }
  int other(int e, int v) const { return ea[e] == v ? eb[e] : ea[e]; }
  void switch_off(int e)
  {
    if (!on[e]) return;
    on[e] = 0;
    live_deg[ea[e]]--;
    if (eb[e] != ea[e]) live_deg[eb[e]]--;
  }
  // every edge between u and v goes (what boost::remove_edge(u, v, g) does on a multigraph)
  void cut_between(int u, int v)
  {
    for (int k = first[u]; k < first[u + 1]; k++)
      if (on[inc[k]] && other(inc[k], u) == v) switch_off(inc[k]);
  }
  void isolate(int v)
  {
    for (int k = first[v]; k < first[v + 1]; k++) switch_off(inc[k]);
  }
};

// Shortest paths from `src` over the edges still switched on.  The loop graphs of ELCH have tens to hundreds of vertices,
// so the next vertex is found by a plain scan (smallest tentative length, lowest index among equals -- the order a
// (length, vertex) min-heap pops in); a neighbour is re-parented only by a strictly shorter path.  parent[v] == v and
// length == DBL_MAX mark a vertex that cannot be reached: the balancer tests exactly that.
void shortest_paths(const LoopGraph& g, int src, int* parent, double* length, std::vector<char>& settled)
{
  const double unreached = std::numeric_limits<double>::max();
  for (int v = 0; v < g.nv; v++) { parent[v] = v; length[v] = unreached; }
  settled.assign((size_t)g.nv, 0);
  length[src] = 0.0;
  for (;;) {
    int u = -1;
    for (int v = 0; v < g.nv; v++)
      if (!settled[v] && length[v] != unreached && (u < 0 || length[v] < length[u])) u = v;
    if (u < 0) return;
    settled[u] = 1;
    for (int k = g.first[u]; k < g.first[u + 1]; k++) {
      const int e = g.inc[k];
      if (!g.on[e]) continue;
      const int t = g.other(e, u);
      const double via = length[u] + g.ew[e];
      if (via < length[t]) { length[t] = via; parent[t] = u; }
    }
  }
}
}  // namespace

extern "C" {

// elch6D::graph_balancer (elch6D.cc:186-279): distribute the loop-closing error over the vertices of the loop graph.
// weights[f] = 0, weights[l] = 1; repeatedly the two closest "open ends" (vertices whose weight is known and that still
// have edges) are joined by their shortest path, the vertices on it get weights interpolated by path length, the path's
// edges leave the graph and its inner vertices become open ends themselves; an open end that reaches no other one
// starts a dangling part, which inherits its weight.  The result depends on the order in which pairs are joined, so the
// scan order of the reference is kept: open ends are tried in the order they appeared, candidates after them in the same
// order, the first strictly shortest pair wins.
int tdtk_elch_graph_balancer(int nvertices, int nedges, const int32_t* from, const int32_t* to, const double* w, int f,
                             int l, double* weights)
{
  if (nvertices <= 0 || nedges < 0 || !weights || (nedges && (!from || !to || !w)) || f < 0 || l < 0 || f >= nvertices ||
      l >= nvertices) {
    set_error("bad argument");
    return TDTK_EINVAL;
  }
  for (int e = 0; e < nedges; e++)
    if (from[e] < 0 || to[e] < 0 || from[e] >= nvertices || to[e] >= nvertices) { set_error("edge endpoint out of range"); return TDTK_EINVAL; }
  LoopGraph g;
  g.build(nvertices, nedges, from, to, w);

  // open ends in order of appearance; a retired entry keeps its place (gone[k]) until the round is over
  std::vector<int> open_end{f, l};
  std::vector<char> gone{0, 0};
  std::vector<int> dangling;                      // roots of the parts that inherit a weight, in the order found
  weights[f] = 0;
  weights[l] = 1;
  std::vector<int> par((size_t)nvertices), best_par((size_t)nvertices);
  std::vector<double> len((size_t)nvertices), best_len((size_t)nvertices);
  std::vector<char> settled;

  for (;;) {
    // compact the list (order preserved)
    size_t keep = 0;
    for (size_t k = 0; k < open_end.size(); k++)
      if (!gone[k]) open_end[keep++] = open_end[k];
    open_end.resize(keep);
    gone.assign(keep, 0);
    if (open_end.empty()) break;

    double shortest = -1.0;                       // length of the best pair of this round, < 0: none yet
    size_t a_pos = 0, b_pos = 0;                  // its two ends, as positions in open_end
    const size_t count = open_end.size();         // ends appended below belong to the next round's scan only at its tail
    for (size_t i = 0; i < count; i++) {
      shortest_paths(g, open_end[i], par.data(), len.data(), settled);
      bool improved = false;
      for (size_t j = i + 1; j < count; j++) {
        if (gone[j]) continue;
        const int t = open_end[j];
        if (par[t] != t && (shortest < 0 || len[t] < shortest)) { shortest = len[t]; a_pos = i; b_pos = j; improved = true; }
      }
      if (improved) { par.swap(best_par); len.swap(best_len); }
      if (shortest < 0) {                         // reaches no later end and nothing has been found yet: a dangling part
        dangling.push_back(open_end[i]);
        gone[i] = 1;
      }
    }
    if (shortest < 0) continue;                   // everything left was dangling
    const int a = open_end[a_pos], b = open_end[b_pos];
    // walk the path from b back to a: interpolate, take the path's edges out, inner vertices that keep edges open up
    g.cut_between(b, best_par[b]);
    for (int v = best_par[b]; v != a; v = best_par[v]) {
      weights[v] = weights[a] + (weights[b] - weights[a]) * best_len[v] / best_len[b];
      g.cut_between(v, best_par[v]);
      if (g.live_deg[v] > 0) { open_end.push_back(v); gone.push_back(0); }
    }
    if (g.live_deg[a] == 0) gone[a_pos] = 1;
    if (g.live_deg[b] == 0) gone[b_pos] = 1;
  }

  // dangling parts: breadth first from each root, every neighbour takes the weight of the vertex it hangs on
  for (size_t head = 0; head < dangling.size(); head++) {
    const int s = dangling[head];
    for (int k = g.first[s]; k < g.first[s + 1]; k++) {
      const int e = g.inc[k];
      if (!g.on[e]) continue;
      const int t = g.other(e, s);
      weights[t] = weights[s];
      if (g.live_deg[t] > 1) dangling.push_back(t);
    }
    g.isolate(s);
  }
  return TDTK_OK;
}

// (n, sum, centroids, Si) of several passes -> the block of their union: what Align_Parallel computes from
// per-thread partial sums before it solves (icp6Dquat.cc:533-578, with the serial-Align normalisation the rest of
// the library uses): cm = sum n_i cm_i / N, Si = sum Si_i + n_i (cm_i - cm)(cd_i - cd)^T.  Base block only.
int tdtk_pair_sums_merge(int count, const tdtk_pair_sums* parts, tdtk_pair_sums* out)
{
  if (count < 0 || !out || (count && !parts)) { set_error("bad argument"); return TDTK_EINVAL; }
  std::memset(out, 0, sizeof *out);
  double N = 0.0;
  for (int i = 0; i < count; i++) {
    out->n_queries += parts[i].n_queries;
    out->n += parts[i].n;
    out->sum += parts[i].sum;
    N += (double)parts[i].n;
    for (int k = 0; k < 3; k++) {
      out->centroid_m[k] += (double)parts[i].n * parts[i].centroid_m[k];
      out->centroid_d[k] += (double)parts[i].n * parts[i].centroid_d[k];
    }
  }
  out->lum_sumd2 = out->sum;
  if (out->n == 0) return TDTK_OK;
  for (int k = 0; k < 3; k++) { out->centroid_m[k] /= N; out->centroid_d[k] /= N; }
  for (int i = 0; i < count; i++) {
    const double n = (double)parts[i].n;
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++)
        out->Si[a * 3 + b] += parts[i].Si[a * 3 + b] +
                              n * (parts[i].centroid_m[a] - out->centroid_m[a]) * (parts[i].centroid_d[b] - out->centroid_d[b]);
  }
  return TDTK_OK;
}

// Graph::Graph(int nodes, double cldist2, int loopsize) (src/slam6d/graph.cc:107-130): the chain i -> i + 1, then every pair
// (j, k), k - j > loopsize, whose scanner positions are closer than cldist (Dist2, globals.icc:238-245), j-major.
int tdtk_graph_links(int nscans, const double* rPos, double cldist2, int loopsize, int32_t* from, int32_t* to, int cap, int* nlinks)
{
  if (nscans < 0 || !nlinks || (nscans && !rPos) || cap < 0 || (cap && (!from || !to))) { tdtk::set_error("bad argument"); return TDTK_EINVAL; }
  int n = 0;
  auto put = [&](int a, int b) { if (n < cap) { from[n] = a; to[n] = b; } ++n; };
  for (int i = 0; i + 1 < nscans; i++) put(i, i + 1);
  for (int j = 0; j < nscans; j++)
    for (int k = j + 1; k < nscans; k++) {
      if (!(k - j > loopsize)) continue;
      const double dx = rPos[3 * k] - rPos[3 * j], dy = rPos[3 * k + 1] - rPos[3 * j + 1], dz = rPos[3 * k + 2] - rPos[3 * j + 2];
      if (dx * dx + dy * dy + dz * dz < cldist2) put(j, k);
    }
  *nlinks = n;       // (more than cap: nothing beyond cap was written -- call again with room)
  return TDTK_OK;
}

}  // extern "C"
