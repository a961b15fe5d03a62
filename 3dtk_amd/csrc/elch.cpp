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
// here are |diag(C^-1)| of real links, where exact ties do not occur.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <list>
#include <queue>
#include <vector>

#include "tdtk_internal.h"

using namespace tdtk;

namespace {
struct UGraph {   // adjacency_list<listS, vecS, undirectedS> with an edge weight: parallel edges allowed
  int n;
  std::vector<std::list<std::pair<int, double>>> adj;
  explicit UGraph(int n_) : n(n_), adj(n_) {}
  void add_edge(int a, int b, double w) { adj[a].push_back({b, w}); if (a != b) adj[b].push_back({a, w}); }
  void remove_edge(int a, int b)   // removes ALL edges between a and b, like boost::remove_edge(u, v, g)
  {
    adj[a].remove_if([b](const std::pair<int, double>& e) { return e.first == b; });
    if (a != b) adj[b].remove_if([a](const std::pair<int, double>& e) { return e.first == a; });
  }
  int degree(int v) const { return (int)adj[v].size(); }
  void clear_vertex(int v)
  {
    for (auto& e : adj[v])
      if (e.first != v) adj[e.first].remove_if([v](const std::pair<int, double>& x) { return x.first == v; });
    adj[v].clear();
  }
};

void dijkstra(const UGraph& g, int s, std::vector<int>& p, std::vector<double>& d)
{
  const double inf = std::numeric_limits<double>::max();
  for (int v = 0; v < g.n; v++) { p[v] = v; d[v] = inf; }
  d[s] = 0.0;
  typedef std::pair<double, int> QE;
  std::priority_queue<QE, std::vector<QE>, std::greater<QE>> q;
  q.push({0.0, s});
  std::vector<char> done(g.n, 0);
  while (!q.empty()) {
    const QE top = q.top(); q.pop();
    const int u = top.second;
    if (done[u] || top.first > d[u]) continue;
    done[u] = 1;
    for (const auto& e : g.adj[u]) {
      const double nd = d[u] + e.second;
      if (nd < d[e.first]) { d[e.first] = nd; p[e.first] = u; q.push({nd, e.first}); }
    }
  }
}
}  // namespace

extern "C" {

int tdtk_elch_graph_balancer(int nvertices, int nedges, const int32_t* from, const int32_t* to, const double* w, int f,
                             int l, double* weights)
{
  if (nvertices <= 0 || nedges < 0 || !weights || (nedges && (!from || !to || !w)) || f < 0 || l < 0 || f >= nvertices ||
      l >= nvertices) {
    set_error("bad argument");
    return TDTK_EINVAL;
  }
  UGraph g(nvertices);
  for (int e = 0; e < nedges; e++) {
    if (from[e] < 0 || to[e] < 0 || from[e] >= nvertices || to[e] >= nvertices) { set_error("edge endpoint out of range"); return TDTK_EINVAL; }
    g.add_edge(from[e], to[e], w[e]);
  }
  std::list<int> crossings, branches;
  crossings.push_back(f);
  crossings.push_back(l);
  weights[f] = 0;
  weights[l] = 1;
  std::vector<int> p(nvertices), p_min(nvertices);
  std::vector<double> d(nvertices), d_min(nvertices);
  double dist;
  bool do_swap = false;
  std::list<int>::iterator si, ei, s_min, e_min;
  // process all junctions (elch6D.cc:203-252)
  while (!crossings.empty()) {
    dist = -1;
    for (si = crossings.begin(); si != crossings.end();) {
      dijkstra(g, *si, p, d);
      ei = si;
      ei++;
      for (; ei != crossings.end(); ei++) {
        if (*ei != p[*ei] && (dist < 0 || d[*ei] < dist)) {
          dist = d[*ei];
          s_min = si;
          e_min = ei;
          do_swap = true;
        }
      }
      if (do_swap) {
        std::swap(p, p_min);
        std::swap(d, d_min);
        do_swap = false;
      }
      if (dist < 0) {          // vertex starts a branch
        branches.push_back(*si);
        si = crossings.erase(si);
      } else {
        si++;
      }
    }
    if (dist > -1) {
      g.remove_edge(*e_min, p_min[*e_min]);
      for (int i = p_min[*e_min]; i != *s_min; i = p_min[i]) {
        weights[i] = weights[*s_min] + (weights[*e_min] - weights[*s_min]) * d_min[i] / d_min[*e_min];
        g.remove_edge(i, p_min[i]);
        if (g.degree(i) > 0) crossings.push_back(i);
      }
      if (g.degree(*s_min) == 0) crossings.erase(s_min);
      if (g.degree(*e_min) == 0) crossings.erase(e_min);
    }
  }
  // error propagation (elch6D.cc:262-278)
  while (!branches.empty()) {
    const int s = branches.front();
    branches.pop_front();
    for (const auto& e : g.adj[s]) {
      weights[e.first] = weights[s];
      if (g.degree(e.first) > 1) branches.push_back(e.first);
    }
    g.clear_vertex(s);
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

}  // extern "C"
