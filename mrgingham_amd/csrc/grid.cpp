// Host-side grid finder: orders the detector's corner candidates into a gridn x gridn board.
//
// This is the caller side of the hot path: the reference keeps it
// on the host (find_grid.cc, ~100-300 points per frame, microseconds) and so does this library.
// It restates mrgingham::find_grid_from_points (find_grid.cc:1216-1445) and the helpers it uses
// (:88-140 neighbour iteration, :204-312 sequence growing, :314-346 sequence search,
// :502-569 sequence candidates, :780-823 crossing test, :825-951 outer 4-cycles,
// :953-1003 equal-and-opposite cycles, :1025-1190 clockwise cycle and top edge,
// :1192-1214 sequence lookup), written from the algorithm, not copied.
//
// PARITY UNPINNED.  The reference reads its neighbour structure off boost::polygon's Voronoi
// diagram (find_grid.cc:7, :1226-1227); boost is absent here and the reference has no test for this
// file, so nothing executable pins this restatement.  The neighbour structure below is the Delaunay
// triangulation of the same integer points (the dual of that Voronoi diagram) with exact integer
// predicates; neighbours of a site are visited counter-clockwise (in the x,y plane as numbers) like
// boost's edge->next(), sites in boost's cell order (sorted by x, then y).  Where the reference takes
// "the first neighbour that matches" (find_grid.cc:216-222) the starting neighbour of the rotation is
// an implementation detail of boost that is not reproduced; on clean boards exactly one neighbour
// matches, and the final grid is fixed by geometry (unique 4-cycles, clockwise test, top edge), not by
// visiting order.  Exactly cocircular quadruples (a degenerate Voronoi vertex) get one of their two
// diagonals here and none in boost.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <chrono>
#if defined(__x86_64__)
#include <x86intrin.h>
#endif
#include <vector>

#include <sys/stat.h>

#include "grid.h"

namespace mrg {

// Test hook (mrgingham_amd_find_grid_from_points_perturbed): the two things about the reference's visiting
// order that cannot be checked against boost here.  ring_seed != 0 starts every site's neighbour ring at a
// pseudo-random position (boost's incident_edge() start is an implementation detail); last_match takes the
// LAST neighbour that continues a sequence instead of the first (find_grid.cc:216-222).  The tests assert
// that neither changes any result.
thread_local GridDebugSequence g_grid_debug_sequence = {false, 0, 0};
thread_local bool g_grid_debug = false;
thread_local GridPerturbation g_grid_perturbation{0u, false};
thread_local GridPhaseClock g_grid_clock = {0, 0, 0, 0, 0, 0};

namespace {

using i64 = long long;
using i128 = __int128;

// ---------------------------------------------------------------------------------------------
// Delaunay triangulation (sweep insertion + Lawson flips, exact predicates)
// ---------------------------------------------------------------------------------------------
struct Tri {
    int v[3];  // counter-clockwise
    int n[3];  // n[k] = triangle across the edge opposite v[k], or -1
};

inline i64 orient(const PointI& a, const PointI& b, const PointI& c) {
    return (i64)(b.x - a.x) * (i64)(c.y - a.y) - (i64)(b.y - a.y) * (i64)(c.x - a.x);
}

// > 0 when d lies strictly inside the circumcircle of the counter-clockwise triangle a,b,c
inline int incircle(const PointI& a, const PointI& b, const PointI& c, const PointI& d) {
    const i64 ax = (i64)a.x - d.x, ay = (i64)a.y - d.y;
    const i64 bx = (i64)b.x - d.x, by = (i64)b.y - d.y;
    const i64 cx = (i64)c.x - d.x, cy = (i64)c.y - d.y;
    {
        // floating-point filter: with differences below 2^26 (every coordinate this library produces: 32767 px * 1000
        // < 2^25) the squared lengths and the 2x2 minors are exact in double, each of the three products is rounded
        // once and the two additions once each: a determinant further from 0 than 5 half-ulps of the terms has its sign
        const double fa2 = (double)(ax * ax + ay * ay), fb2 = (double)(bx * bx + by * by), fc2 = (double)(cx * cx + cy * cy);
        const double t1 = fa2 * (double)(bx * cy - by * cx), t2 = fb2 * (double)(ax * cy - ay * cx),
                     t3 = fc2 * (double)(ax * by - ay * bx);
        const double det = t1 - t2 + t3;
        const double bound = 1e-15 * (std::fabs(t1) + std::fabs(t2) + std::fabs(t3));
        auto in26 = [](i64 v) { return v > -(1ll << 26) && v < (1ll << 26); };
        const bool small = in26(ax) && in26(ay) && in26(bx) && in26(by) && in26(cx) && in26(cy);
        if (small && (det > bound || det < -bound)) return det > 0 ? 1 : -1;
    }
    const i128 a2 = (i128)ax * ax + (i128)ay * ay;
    const i128 b2 = (i128)bx * bx + (i128)by * by;
    const i128 c2 = (i128)cx * cx + (i128)cy * cy;
    const i128 det = a2 * ((i128)bx * cy - (i128)by * cx) - b2 * ((i128)ax * cy - (i128)ay * cx) +
                     c2 * ((i128)ax * by - (i128)ay * bx);
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

class Delaunay {
public:
    explicit Delaunay(const std::vector<PointI>& pts) : p(pts) {}

    // false when there is no triangle at all (fewer than 3 distinct points, or all collinear)
    bool build() {
        const int n = (int)p.size();
        tris.reserve(2 * (size_t)n + 8);
        order.resize(n);
        for (int i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int a, int b) {
            return p[a].x != p[b].x ? p[a].x < p[b].x : (p[a].y != p[b].y ? p[a].y < p[b].y : a < b);
        });
        // distinct points, in sweep order
        std::vector<int> s;
        for (int i : order)
            if (s.empty() || p[s.back()].x != p[i].x || p[s.back()].y != p[i].y) s.push_back(i);
        if (s.size() < 3) return false;
        // the first point that is not collinear with the first two; the collinear prefix s[0..k-1]
        size_t k = 2;
        while (k < s.size() && orient(p[s[0]], p[s[1]], p[s[k]]) == 0) ++k;
        if (k == s.size()) return false;
        // fan from s[k] over the collinear prefix
        const bool left = orient(p[s[0]], p[s[1]], p[s[k]]) > 0;
        for (size_t i = 0; i + 1 < k; ++i) {
            Tri t;
            if (left) { t.v[0] = s[i]; t.v[1] = s[i + 1]; t.v[2] = s[k]; }
            else { t.v[0] = s[i + 1]; t.v[1] = s[i]; t.v[2] = s[k]; }
            t.n[0] = t.n[1] = t.n[2] = -1;
            tris.push_back(t);
        }
        link_all();
        // convex hull as a counter-clockwise cycle
        hull_next.assign(n, -1);
        hull_prev.assign(n, -1);
        {
            std::vector<int> h;
            if (left) { for (size_t i = 0; i < k; ++i) h.push_back(s[i]); h.push_back(s[k]); }
            else { for (size_t i = k; i-- > 0;) h.push_back(s[i]); h.push_back(s[k]); }
            for (size_t i = 0; i < h.size(); ++i) {
                hull_next[h[i]] = h[(i + 1) % h.size()];
                hull_prev[h[(i + 1) % h.size()]] = h[i];
            }
        }
        int last = s[k];
        for (size_t i = k + 1; i < s.size(); ++i) {
            insert_outside(s[i], last);
            last = s[i];
        }
        return true;
    }

    const std::vector<PointI>& p;
    std::vector<int> order;  // all point indices sorted by (x, y): boost's cell order
    std::vector<Tri> tris;

private:
    std::vector<int> hull_next, hull_prev;
    // hull edge a -> hull_next[a] belongs to triangle hull_tri[a], opposite its vertex slot hull_slot[a]
    std::vector<int> hull_tri, hull_slot;
    std::vector<int> fresh;                      // scratch of insert_outside
    std::vector<std::pair<int, int>> flip_stack;  // scratch of legalize

    void note_hull_edges(int t) {
        for (int k = 0; k < 3; ++k)
            if (tris[t].n[k] < 0) {
                const int a = tris[t].v[(k + 1) % 3];
                hull_tri[a] = t;
                hull_slot[a] = k;
            }
    }

    // the initial fan: consecutive triangles share the edge from the apex to a prefix point
    void link_all() {
        hull_tri.assign(p.size(), -1);
        hull_slot.assign(p.size(), -1);
        for (int t = 0; t + 1 < (int)tris.size(); ++t)
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j)
                    if (tris[t].v[(k + 1) % 3] == tris[t + 1].v[(j + 2) % 3] &&
                        tris[t].v[(k + 2) % 3] == tris[t + 1].v[(j + 1) % 3]) {
                        tris[t].n[k] = t + 1;
                        tris[t + 1].n[j] = t;
                    }
        for (int t = 0; t < (int)tris.size(); ++t) note_hull_edges(t);
    }

    // new triangle (b, a, q) on the visible hull edge a -> b; `prev` = the triangle added just before
    // on the neighbouring hull edge (it shares the edge a-q), or -1
    int add_tri_on_hull(int a, int b, int q, int prev) {
        Tri t;
        t.v[0] = b; t.v[1] = a; t.v[2] = q;
        t.n[0] = prev;          // edge (a, q)
        t.n[1] = -1;            // edge (q, b): shared with the next new triangle, or hull
        t.n[2] = hull_tri[a];   // edge (b, a): the old hull edge
        const int id = (int)tris.size();
        tris.push_back(t);
        tris[hull_tri[a]].n[hull_slot[a]] = id;
        if (prev >= 0) tris[prev].n[1] = id;  // prev = (a, a_prev, q): its edge (q, a) is opposite slot 1
        return id;
    }

    // q lies outside the current hull (sweep order guarantees it); `start` is a hull vertex it sees
    void insert_outside(int q, int start) {
        // visible hull edges (a -> hull_next[a]) are those with q strictly to their right
        int lo = start, hi = start;
        while (orient(p[hull_prev[lo]], p[lo], p[q]) < 0) lo = hull_prev[lo];
        while (orient(p[hi], p[hull_next[hi]], p[q]) < 0) hi = hull_next[hi];
        if (lo == hi) {
            // q is collinear with the hull edges at `start` on both sides: cannot happen for a point
            // strictly beyond the sweep line unless the hull is degenerate; nothing visible.
            return;
        }
        fresh.clear();
        int prev = -1;
        for (int a = lo; a != hi;) {
            const int b = hull_next[a];
            prev = add_tri_on_hull(a, b, q, prev);  // (a,b) is a ccw hull edge seen from outside: b,a,q is ccw
            fresh.push_back(prev);
            a = b;
        }
        hull_next[lo] = q; hull_prev[q] = lo;
        hull_next[q] = hi; hull_prev[hi] = q;
        // vertices strictly between lo and hi left the hull; the two new hull edges are lo -> q, q -> hi
        for (int t : fresh) note_hull_edges(t);
        for (size_t i = 0; i < fresh.size(); ++i) legalize(fresh[i], 2);
    }

    // Lawson: the edge of triangle t opposite its vertex slot k
    void legalize(int t0, int k0) {
        std::vector<std::pair<int, int>>& stack = flip_stack;
        stack.clear();
        stack.push_back({t0, k0});
        while (!stack.empty()) {
            auto [t, k] = stack.back();
            stack.pop_back();
            const int u = tris[t].n[k];
            if (u < 0) continue;
            const int a = tris[t].v[k], b = tris[t].v[(k + 1) % 3], c = tris[t].v[(k + 2) % 3];
            int ku = -1;
            for (int j = 0; j < 3; ++j)
                if (tris[u].n[j] == t && tris[u].v[(j + 1) % 3] == c && tris[u].v[(j + 2) % 3] == b) ku = j;
            if (ku < 0) continue;
            const int d = tris[u].v[ku];
            if (incircle(p[a], p[b], p[c], p[d]) <= 0) continue;
            // flip edge (b,c) -> (a,d): triangles (a,b,d) and (a,d,c)
            const int n_ab = tris[t].n[(k + 2) % 3], n_ca = tris[t].n[(k + 1) % 3];
            const int n_bd = tris[u].n[(ku + 1) % 3], n_dc = tris[u].n[(ku + 2) % 3];
            // note: in u, vertex order is d, c, b (ccw): edge opposite c is (b,d), opposite b is (d,c)
            tris[t].v[0] = a; tris[t].v[1] = b; tris[t].v[2] = d;
            tris[t].n[0] = n_bd; tris[t].n[1] = u; tris[t].n[2] = n_ab;
            tris[u].v[0] = a; tris[u].v[1] = d; tris[u].v[2] = c;
            tris[u].n[0] = n_dc; tris[u].n[1] = n_ca; tris[u].n[2] = t;
            auto relink = [&](int nb, int from, int to) {
                if (nb < 0) return;
                for (int j = 0; j < 3; ++j)
                    if (tris[nb].n[j] == from) { tris[nb].n[j] = to; return; }
            };
            relink(n_bd, u, t);
            relink(n_ca, t, u);
            note_hull_edges(t);
            note_hull_edges(u);
            stack.push_back({t, 0});
            stack.push_back({u, 0});
        }
    }
};

// ---------------------------------------------------------------------------------------------
// What find_grid.cc reads off the Voronoi diagram, per site: the counter-clockwise ring of
// neighbouring sites, whether consecutive neighbours close a triangle with the site, and the
// site on the far side of that triangle's outer edge (find_grid.cc:41-87).
// ---------------------------------------------------------------------------------------------
// (flat: ring of site v = entries [off[v], off[v + 1]) of nbr / tri / far)
struct SiteGraph {
    std::vector<int> order;  // sites in cell order
    std::vector<int> off;    // n + 1
    std::vector<int> nbr;    // neighbour sites, counter-clockwise
    std::vector<char> tri;   // tri[k]: (site, nbr[k], nbr[k+1]) is a triangle
    std::vector<int> far;    // far[k]: the site opposite `site` across edge (nbr[k], nbr[k+1]), or -1
};

bool build_site_graph(const std::vector<PointI>& pts, SiteGraph& g) {
    Delaunay dt(pts);
    if (!dt.build()) return false;
    const int n = (int)pts.size();
    g.order = dt.order;
    g.off.assign((size_t)n + 1, 0);
    const size_t guess = 3 * dt.tris.size() + (size_t)n;
    g.nbr.clear(); g.tri.clear(); g.far.clear();
    g.nbr.reserve(guess); g.tri.reserve(guess); g.far.reserve(guess);
    // one incident triangle per vertex, preferring one whose clockwise side is open (hull start)
    std::vector<int> inc(n, -1);
    for (int t = 0; t < (int)dt.tris.size(); ++t)
        for (int k = 0; k < 3; ++k) {
            const int v = dt.tris[t].v[k];
            // the edge (v, v_next) is opposite slot (k+2)%3; the edge (v_prev, v) opposite (k+1)%3.
            // rotating clockwise around v crosses edge (v, v[k+1]) -> neighbour n[(k+2)%3]
            if (inc[v] < 0 || dt.tris[t].n[(k + 2) % 3] < 0) inc[v] = t;
        }
    for (int v = 0; v < n; ++v) {
        g.off[v] = (int)g.nbr.size();
        if (inc[v] < 0) continue;  // duplicate of another point: no cell of its own
        const size_t first = g.nbr.size();
        // walk counter-clockwise around v starting at inc[v]
        int t = inc[v];
        const int t_first = t;
        bool closed = false;
        while (true) {
            int k = 0;
            while (dt.tris[t].v[k] != v) ++k;
            const int b = dt.tris[t].v[(k + 1) % 3], c = dt.tris[t].v[(k + 2) % 3];
            if (g.nbr.size() == first) g.nbr.push_back(b);
            g.tri.push_back(1);
            // far vertex across (b,c): the triangle opposite v
            const int u = dt.tris[t].n[k];
            int d = -1;
            if (u >= 0)
                for (int j = 0; j < 3; ++j)
                    if (dt.tris[u].v[j] != b && dt.tris[u].v[j] != c) d = dt.tris[u].v[j];
            g.far.push_back(d);
            // next triangle counter-clockwise: across edge (v, c), which is opposite slot (k+1)%3
            const int nx = dt.tris[t].n[(k + 1) % 3];
            if (nx == t_first) { closed = true; break; }
            g.nbr.push_back(c);
            if (nx < 0) break;
            t = nx;
        }
        if (!closed) {
            // hull site: the step from the last neighbour back to the first has no triangle
            g.tri.push_back(0);
            g.far.push_back(-1);
        }
    }
    g.off[n] = (int)g.nbr.size();
    return true;
}

// The neighbours the reference considers from a site, in its order: for every ring position the
// direct neighbour, then the "in-between" site across the triangle's far edge when it lies
// angularly between the two (find_grid.cc:88-140).
template <typename F>
bool for_each_adjacent(const SiteGraph& g, const std::vector<PointI>& pts, int c, F&& visit) {
    struct { const int* nbr; const char* tri; const int* far; } r{g.nbr.data() + g.off[c], g.tri.data() + g.off[c],
                                                                   g.far.data() + g.off[c]};
    const int deg = g.off[c + 1] - g.off[c];
    const PointI& pt = pts[c];
    int start = 0;
    if (g_grid_perturbation.ring_seed && deg > 0) {
        uint32_t hsh = (uint32_t)c * 2654435761u ^ g_grid_perturbation.ring_seed;
        hsh ^= hsh >> 15; hsh *= 0x2c1b3c6du; hsh ^= hsh >> 12;
        start = (int)(hsh % (uint32_t)deg);
    }
    for (int kk = 0; kk < deg; ++kk) {
        const int k = (kk + start) % deg;
        const int b = r.nbr[k];
        if (visit(b, PointI{pts[b].x - pt.x, pts[b].y - pt.y})) return true;
        if (deg < 2) continue;
        const int cn = r.nbr[(k + 1) % deg];
        const i64 v0x = pts[b].x - pt.x, v0y = pts[b].y - pt.y;
        const i64 v1x = pts[cn].x - pt.x, v1y = pts[cn].y - pt.y;
        if (v1x * v0y > v0x * v1y) continue;  // not an acute turn: graph boundary (:116-117)
        if (!r.tri[k]) continue;              // the two edges do not close a triangle (:121-122)
        const int d = r.far[k];
        if (d < 0) continue;
        const i64 vmx = pts[d].x - pt.x, vmy = pts[d].y - pt.y;
        if (v1x * vmy > vmx * v1y) continue;  // must lie angularly between its neighbours (:128-131)
        if (vmx * v0y > v0x * vmy) continue;
        if (visit(d, PointI{(int)vmx, (int)vmy})) return true;
    }
    return false;
}

// thresholds, find_grid.cc:204-207
constexpr int kScale = 1000;  // FIND_GRID_SCALE, mrgingham-internal.h:3
constexpr double kSpacingCos = 0.984;
constexpr double kLenRatioMin = 0.7, kLenRatioMax = 1.4, kLenRatioDev = 0.35;

// The adjacency of every site in the reference's visiting order, built once: the sequence search
// walks it tens of thousands of times per frame.  Flat: site c's entries are [off[c], off[c + 1]).
struct Adj {
    int site;
    PointI delta;
    double dx, dy;  // delta again, as the doubles the angle test multiplies
    double len;
};
struct AdjLists {
    std::vector<int> off;
    std::vector<Adj> a;
    struct Range {
        const Adj *b, *e;
        const Adj* begin() const { return b; }
        const Adj* end() const { return e; }
    };
    Range operator[](int c) const { return Range{a.data() + off[c], a.data() + off[c + 1]}; }
};

AdjLists build_adjacency(const SiteGraph& g, const std::vector<PointI>& pts) {
    AdjLists adj;
    adj.off.assign(pts.size() + 1, 0);
    adj.a.reserve(2 * g.nbr.size());
    for (int c = 0; c < (int)pts.size(); ++c) {
        adj.off[c] = (int)adj.a.size();
        for_each_adjacent(g, pts, c, [&](int cand, PointI delta) {
            adj.a.push_back(Adj{cand, delta, (double)delta.x, (double)delta.y, std::hypot((double)delta.x, (double)delta.y)});
            return false;
        });
    }
    adj.off[pts.size()] = (int)adj.a.size();
    return adj;
}

struct SeqStats {  // HypothesisStatistics, :166-172
    PointI delta_last;
    double lx, ly;  // delta_last as doubles
    double last_len;
    double ratio_sum;
    int ratio_n;
};

// get_adjacent_cell_along_sequence, :209-312: the first neighbour continuing the sequence, or -1
// `trace` != nullptr: the candidates, for the --debug-sequence messages of :247-306 (coordinates in whole pixels)
int next_along_sequence(const AdjLists& adj, int c, SeqStats& st, const std::vector<PointI>* trace = nullptr) {
    const Adj* chosen = nullptr;
    double chosen_ratio = 0.0;
    const int S = kScale;
    const double lx = st.lx, ly = st.ly, last_len = st.last_len;
    const bool last_match = g_grid_perturbation.last_match;
    for (const Adj& a : adj[c]) {
        if (trace)
            fprintf(stderr, "Considering connection in sequence from (%d,%d) -> (%d,%d); delta (%d,%d) ..... \n",
                    (*trace)[c].x / S, (*trace)[c].y / S, (*trace)[a.site].x / S, (*trace)[a.site].y / S, a.delta.x / S,
                    a.delta.y / S);
        const double dot = lx * a.dx + ly * a.dy;
        const double den = last_len * a.len;
        // most neighbours point somewhere else entirely: decided without the division wherever the quotient is
        // further from the threshold than any rounding could move it (the quotient itself is the reference's, :257-260)
        if (!trace && dot < (kSpacingCos - 1e-6) * den) continue;
        const double cos_err = dot / den;
        if (cos_err < kSpacingCos) {
            if (trace)
                fprintf(stderr, "..... rejecting. Angle is wrong. I wanted cos_err>=threshold, but saw %f<%f\n", cos_err,
                        kSpacingCos);
            continue;
        }
        const double ratio = a.len / last_len;
        if (ratio < kLenRatioMin || ratio > kLenRatioMax) {
            if (trace)
                fprintf(stderr, "..... rejecting. Lengths are wrong. I wanted abs(length_ratio)<=threshold, but saw %f<%f or %f>%f\n",
                        ratio, kLenRatioMin, ratio, kLenRatioMax);
            continue;
        }
        if (st.ratio_n > 2) {
            const double dev = ratio - st.ratio_sum / (double)st.ratio_n;
            if (dev < -kLenRatioDev || dev > kLenRatioDev) {
                if (trace)
                    fprintf(stderr, "..... rejecting. Lengths are wrong. I wanted abs(length_ratio_deviation)<=threshold, but saw %f>%f\n",
                            std::fabs(dev), kLenRatioDev);
                continue;
            }
        }
        chosen = &a;
        chosen_ratio = ratio;
        if (trace) fprintf(stderr, "..... accepting!\n\n");
        if (!last_match) break;  // the reference: the first match (:216-222)
    }
    if (!chosen) return -1;
    st.ratio_sum += chosen_ratio;
    st.ratio_n++;
    st.delta_last = chosen->delta;
    st.lx = chosen->dx;
    st.ly = chosen->dy;
    st.last_len = chosen->len;
    return chosen->site;
}

struct Sequence {  // CandidateSequence, :148-162
    int c0, c1, clast;
    PointD delta_mean;
};

// walks n_remaining steps from c along delta; fills `path` (if given) with the sites visited
// (`delta_len` = hypot(delta) when the caller has it already -- the adjacency lists do --, < 0 otherwise)
int walk_sequence(const AdjLists& adj, PointI delta, int c, int n_remaining, PointD* delta_mean, std::vector<int>* path,
                  const std::vector<PointI>* trace = nullptr, double delta_len = -1.0) {
    SeqStats st{delta, (double)delta.x, (double)delta.y,
                delta_len >= 0.0 ? delta_len : std::hypot((double)delta.x, (double)delta.y), 0.0, 0};
    double sx = delta.x, sy = delta.y;
    int last = -1;
    for (int i = 0; i < n_remaining; ++i) {
        const int nx = next_along_sequence(adj, c, st, trace);
        if (nx < 0) return -1;
        sx += st.delta_last.x;
        sy += st.delta_last.y;
        if (path) path->push_back(nx);
        last = nx;
        c = nx;
    }
    if (delta_mean) { delta_mean->x = sx / (double)(n_remaining + 1); delta_mean->y = sy / (double)(n_remaining + 1); }
    return last;
}

// The sequence-candidate search (:502-569) walks from EVERY adjacency of every site.  While fewer than three length
// ratios have been seen (:289: the deviation test needs length_ratio_N > 2) the neighbour that continues a walk is a
// function of the edge it arrived by alone: the first neighbour that passes the angle and length-ratio tests against
// that edge.  Found once per edge and remembered; with the deviation test on, the scan starts at that neighbour
// (whatever comes before it fails a test that does not depend on the history).  Same neighbours, same sums as
// walk_sequence, ~3x fewer tests.
struct WalkMemo {
    std::vector<int> nxt;       // per adjacency entry: -2 not looked at yet, -1 none, else the continuing entry
    std::vector<double> ratio;  // its length ratio
};
inline bool continues(const Adj& in, const Adj& a, double* ratio_out) {
    const double dot = in.dx * a.dx + in.dy * a.dy;
    const double den = in.len * a.len;
    if (dot < (kSpacingCos - 1e-6) * den) return false;  // (see next_along_sequence)
    if (dot / den < kSpacingCos) return false;
    const double ratio = a.len / in.len;
    if (ratio < kLenRatioMin || ratio > kLenRatioMax) return false;
    *ratio_out = ratio;
    return true;
}
int walk_sequence_memo(const AdjLists& adj, WalkMemo& m, int e0, int n_remaining, PointD* delta_mean) {
    int e = e0;
    double sx = adj.a[e].delta.x, sy = adj.a[e].delta.y, ratio_sum = 0.0;
    int ratio_n = 0, last = -1;
    for (int i = 0; i < n_remaining; ++i) {
        const Adj& in = adj.a[e];
        int f = m.nxt[e];
        if (f == -2) {
            f = -1;
            for (int k = adj.off[in.site]; k < adj.off[in.site + 1]; ++k)
                if (continues(in, adj.a[k], &m.ratio[e])) { f = k; break; }
            m.nxt[e] = f;
        }
        if (f < 0) return -1;
        double ratio = m.ratio[e];
        if (ratio_n > 2) {
            const double mean = ratio_sum / (double)ratio_n;
            for (;; ) {
                const double dev = ratio - mean;
                if (!(dev < -kLenRatioDev || dev > kLenRatioDev)) break;
                for (++f; f < adj.off[in.site + 1]; ++f)
                    if (continues(in, adj.a[f], &ratio)) break;
                if (f >= adj.off[in.site + 1]) return -1;
            }
        }
        ratio_sum += ratio;
        ratio_n++;
        sx += adj.a[f].delta.x;
        sy += adj.a[f].delta.y;
        last = adj.a[f].site;
        e = f;
    }
    delta_mean->x = sx / (double)(n_remaining + 1);
    delta_mean->y = sy / (double)(n_remaining + 1);
    return last;
}

std::vector<int> sequence_points(const AdjLists& adj, const std::vector<PointI>& pts, const Sequence& s, int gridn) {
    std::vector<int> out{s.c0, s.c1};
    walk_sequence(adj, PointI{pts[s.c1].x - pts[s.c0].x, pts[s.c1].y - pts[s.c0].y}, s.c1, gridn - 2, nullptr, &out);
    return out;
}

// is_crossing, :780-823 (single precision, like the reference)
bool segments_cross(int a0, int a1, int b0, int b1, const std::vector<PointI>& p) {
    const float l0[2] = {(float)(p[a1].x - p[a0].x), (float)(p[a1].y - p[a0].y)};
    const float q0[2] = {(float)(p[b0].x - p[a0].x), (float)(p[b0].y - p[a0].y)};
    const float q1[2] = {(float)(p[b1].x - p[a0].x), (float)(p[b1].y - p[a0].y)};
    const float d2 = l0[0] * l0[0] + l0[1] * l0[1];
    const float r0[2] = {q0[0] * l0[0] + q0[1] * l0[1], -q0[0] * l0[1] + q0[1] * l0[0]};
    const float r1[2] = {q1[0] * l0[0] + q1[1] * l0[1], -q1[0] * l0[1] + q1[1] * l0[0]};
    if (r0[1] * r1[1] > 0) return false;
    if ((r0[0] < 0 && r1[0] < 0) || (r0[0] > d2 && r1[0] > d2)) return false;
    const float k = r0[1] / (r0[1] - r1[1]);
    const float x = r0[0] + k * (r1[0] - r0[0]);
    return x >= 0.0f && x <= d2;
}

struct Cycle { int e[4]; };

struct CycleSearch {
    const std::vector<Sequence>& seq;
    const std::vector<int>& outer;  // indices into seq
    const std::vector<int>& from_off;  // start site -> positions in `outer`: from_pos[from_off[site] .. from_off[site + 1])
    const std::vector<int>& from_pos;
    const std::vector<PointI>& pts;

    // next_outer_edge, :825-951
    bool extend(Cycle& cyc, int count, int start_site) const {
        bool found = false;
        Cycle best{};
        const Sequence& cur = seq[outer[cyc.e[count - 1]]];
        for (int q = from_off[cur.clast]; q < from_off[cur.clast + 1]; ++q) {
            const int pos = from_pos[q];
            const Sequence& nx = seq[outer[pos]];
            if (nx.clast == cur.c0) continue;  // straight back
            if (count != 3) {
                if (nx.clast == start_site) continue;
                if (count == 2 && segments_cross(seq[outer[cyc.e[0]]].c0, seq[outer[cyc.e[0]]].clast, nx.c0, nx.clast, pts))
                    continue;
                cyc.e[count] = pos;
                if (!extend(cyc, count + 1, start_site)) continue;
                if (found) return false;  // must be unique
                found = true;
                best = cyc;
            } else {
                if (nx.clast != start_site) continue;
                if (segments_cross(seq[outer[cyc.e[1]]].c0, seq[outer[cyc.e[1]]].clast, nx.c0, nx.clast, pts)) return false;
                cyc.e[3] = pos;
                return true;
            }
        }
        if (!found) return false;
        cyc = best;
        return true;
    }
};

}  // namespace

namespace {
// ---- the reference's debug dumps (find_grid.cc:385-421, :423-475, :609-779): same file names, header lines,
// columns and stderr messages, so that the plots and scripts written for them keep working
void make_executable(const char* f) { chmod(f, S_IRUSR | S_IRGRP | S_IROTH | S_IWUSR | S_IWGRP | S_IXUSR | S_IXGRP | S_IXOTH); }

void dump_graph(const AdjLists& adj, const std::vector<int>& order, const std::vector<PointI>& p) {
    const char* fn = "/tmp/mrgingham-2-voronoi.vnl";
    FILE* fp = fopen(fn, "w");
    if (!fp) return;
    fprintf(fp, "#!/usr/bin/feedgnuplot --domain --dataid --with 'lines linecolor 0' --square --maxcurves 100000 --set 'yrange [:] rev'\n");
    fprintf(fp, "# x id_edge y\n");
    int e = 0;
    for (int c : order)
        for (const Adj& a : adj[c]) {
            fprintf(fp, "%f %d %f\n", p[c].x / (double)kScale, e, p[c].y / (double)kScale);
            fprintf(fp, "%f %d %f\n", p[a.site].x / (double)kScale, e, p[a.site].y / (double)kScale);
            ++e;
        }
    fclose(fp);
    make_executable(fn);
    fprintf(stderr, "Wrote self-plotting voronoi diagram to %s\n", fn);
}

void dump_interval(FILE* fp, int icand, int ipt, int c0, int c1, const std::vector<PointI>& p) {
    if (c1 < 0) {
        fprintf(fp, "%d %d %f %f - - - - - -\n", icand, ipt, (double)p[c0].x / (double)kScale, (double)p[c0].y / (double)kScale);
        return;
    }
    const double dx = (double)(p[c1].x - p[c0].x) / (double)kScale, dy = (double)(p[c1].y - p[c0].y) / (double)kScale;
    fprintf(fp, "%d %d %f %f %f %f %f %f %f %f\n", icand, ipt, (double)p[c0].x / (double)kScale, (double)p[c0].y / (double)kScale,
            (double)p[c1].x / (double)kScale, (double)p[c1].y / (double)kScale, dx, dy, std::hypot(dx, dy),
            std::atan2(dy, dx) * 180.0 / M_PI);
}

// `which` == nullptr: every sequence candidate; otherwise the candidates with these indices (the outer edges)
void dump_sequences(const char* base, const std::vector<Sequence>& seq, const std::vector<int>* which, const AdjLists& adj,
                    const std::vector<PointI>& p, int gridn) {
    const std::string sparse = std::string(base) + ".vnl", dense = std::string(base) + "-detailed.vnl";
    const int n = which ? (int)which->size() : (int)seq.size();
    auto at = [&](int i) -> const Sequence& { return seq[which ? (*which)[i] : i]; };
    if (FILE* fp = fopen(sparse.c_str(), "w")) {
        fprintf(fp, "#!/usr/bin/feedgnuplot --dom --aut --square --rangesizea 3 --w 'vec size screen 0.01,20 fixed fill' --set 'yr [:] rev'\n");
        fprintf(fp, "# fromx fromy deltax deltay\n");
        for (int i = 0; i < n; ++i)
            fprintf(fp, "%f %f %f %f\n", (double)p[at(i).c0].x / (double)kScale, (double)p[at(i).c0].y / (double)kScale,
                    at(i).delta_mean.x / (double)kScale, at(i).delta_mean.y / (double)kScale);
        fclose(fp);
        make_executable(sparse.c_str());
        fprintf(stderr, "Wrote self-plotting sequence-candidate dump to %s\n", sparse.c_str());
    }
    if (FILE* fp = fopen(dense.c_str(), "w")) {
        fprintf(fp, "# candidateid pointid fromx fromy tox toy deltax deltay len angle\n");
        for (int i = 0; i < n; ++i) {
            const std::vector<int> sp = sequence_points(adj, p, at(i), gridn);
            for (int k = 0; k < (int)sp.size(); ++k)
                dump_interval(fp, i, k, sp[k], k + 1 < (int)sp.size() ? sp[k + 1] : -1, p);
        }
        fclose(fp);
        fprintf(stderr, "Wrote detailed sequence-candidate dump to %s\n", dense.c_str());
    }
}

}  // namespace

static bool find_grid_impl(std::vector<PointD>& out, const std::vector<PointI>& pts, int gridn, double (&lap)[4]);

// The phase clock: the time-stamp counter where there is one (a clock_gettime per mark costs microseconds under some
// sandboxes), the steady clock in nanoseconds elsewhere.  What a tick is worth is worked out when somebody READS the totals:
// from the anchor -- both clocks at the process's first grid-finder call -- to the moment of the question.
#if defined(__x86_64__)
static inline unsigned long long phase_ticks() { return __rdtsc(); }
#else
static inline unsigned long long phase_ticks() {
    return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#endif
namespace {
struct ClockAnchor { std::chrono::steady_clock::time_point c; unsigned long long t; };
const ClockAnchor& clock_anchor() {
    static const ClockAnchor a{std::chrono::steady_clock::now(), phase_ticks()};
    return a;
}
}  // namespace
double grid_clock_tick_us() {
    const ClockAnchor& a = clock_anchor();
    double us;
    unsigned long long t1;
    do {   // (asked within 0.2 ms of the anchor -- only when nothing worth reading has been counted yet: wait that long)
        t1 = phase_ticks();
        us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a.c).count();
    } while (us < 200.0 || t1 == a.t);
    return us / (double)(t1 - a.t);
}

bool find_grid_from_points(std::vector<PointD>& out, const std::vector<PointI>& pts, int gridn) {
    double lap[4] = {0, 0, 0, 0};
    const bool ok = find_grid_impl(out, pts, gridn, lap);
    GridPhaseClock& c = g_grid_clock;
    c.graph_t += lap[0]; c.adjacency_t += lap[1]; c.sequences_t += lap[2]; c.cycles_t += lap[3];
    c.calls++;
    c.found += ok;
    return ok;
}

static bool find_grid_impl(std::vector<PointD>& out, const std::vector<PointI>& pts, int gridn, double (&lap)[4]) {
    struct Lap {   // adds the time since the last mark to lap[i]; the destructor closes the phase that was running
        double (&lap)[4];
        int cur = 0;
        unsigned long long t = (clock_anchor(), phase_ticks());
        void to(int next) {
            const unsigned long long n = phase_ticks();
            lap[cur] += (double)(n - t);
            t = n;
            cur = next;
        }
        ~Lap() { to(cur); }
    } phase{lap};
    const bool debug = g_grid_debug;
    if (gridn < 2 || (int)pts.size() < gridn * gridn) return false;
    SiteGraph g;
    if (!build_site_graph(pts, g)) return false;
    phase.to(1);

    // get_sequence_candidates, :502-569
    const AdjLists adj = build_adjacency(g, pts);
    phase.to(2);
    if (debug) dump_graph(adj, g.order, pts);
    std::vector<Sequence> seq;
    // --debug-sequence (:515-539): the candidate nearest to the given pixel is traced
    int tracing = -1;
    if (g_grid_debug_sequence.on) {
        unsigned long long best = ~0ull;
        for (int c : g.order) {
            const long long dx = (long long)pts[c].x - (long long)kScale * g_grid_debug_sequence.x;
            const long long dy = (long long)pts[c].y - (long long)kScale * g_grid_debug_sequence.y;
            const unsigned long long d2 = (unsigned long long)(dx * dx + dy * dy);
            if (d2 < best) { best = d2; tracing = c; }
        }
        if (tracing >= 0)
            fprintf(stderr, "============== Looking at sequences from (%d,%d)\n", pts[tracing].x / kScale,
                    pts[tracing].y / kScale);
    }
    // (the memoised walk takes the first matching neighbour: the perturbed and the traced search take the plain one)
    const bool plain = g_grid_perturbation.last_match || tracing >= 0;
    WalkMemo memo;
    if (!plain) {
        memo.nxt.assign(adj.a.size(), -2);
        memo.ratio.assign(adj.a.size(), 0.0);
    }
    seq.reserve(adj.a.size() / 4 + 16);
    for (int c : g.order)
        for (int e = adj.off[c]; e < adj.off[c + 1]; ++e) {
            const Adj& a = adj.a[e];
            if (c == tracing)
                fprintf(stderr, "\n\n====== Looking at adjacent point (%d,%d)\n", pts[a.site].x / kScale,
                        pts[a.site].y / kScale);
            PointD mean;
            const int clast = plain ? walk_sequence(adj, a.delta, a.site, gridn - 2, &mean, nullptr,
                                                    c == tracing ? &pts : nullptr, a.len)
                                    : walk_sequence_memo(adj, memo, e, gridn - 2, &mean);
            if (clast >= 0) seq.push_back(Sequence{c, a.site, clast, mean});
        }

    if (debug) {
        dump_sequences("/tmp/mrgingham-3-candidates", seq, nullptr, adj, pts, gridn);
        fprintf(stderr, "got %zd points\n", pts.size());
        fprintf(stderr, "got %zd sequence candidates\n", seq.size());
    }
    phase.to(3);
    // outer-edge candidates: sequences whose start site starts at least two sequences (:1246-1262)
    const int nsites = (int)pts.size();
    std::vector<int> started((size_t)nsites, 0);
    for (const Sequence& s : seq) started[s.c0]++;
    std::vector<int> outer;
    for (int i = 0; i < (int)seq.size(); ++i)
        if (started[seq[i].c0] >= 2) outer.push_back(i);
    if (outer.size() < 8) {
        if (debug) fprintf(stderr, "Too few candidates for an outer edge of the grid. Needed at least 8, got %d\n", (int)outer.size());
        return false;
    }
    if (debug) dump_sequences("/tmp/mrgingham-4-outer-edges", seq, &outer, adj, pts, gridn);
    // positions in `outer` by start site, ascending within a site
    auto by_site = [&](int n, auto&& site_of, std::vector<int>& off, std::vector<int>& pos) {
        off.assign((size_t)nsites + 1, 0);
        for (int i = 0; i < n; ++i) off[site_of(i) + 1]++;
        for (int v = 0; v < nsites; ++v) off[v + 1] += off[v];
        pos.resize((size_t)n);
        std::vector<int> fill(off.begin(), off.end() - 1);
        for (int i = 0; i < n; ++i) pos[fill[site_of(i)]++] = i;
    };
    std::vector<int> from_off, from_pos;
    by_site((int)outer.size(), [&](int i) { return seq[outer[i]].c0; }, from_off, from_pos);

    // 4-cycles of outer edges (:1287-1322)
    std::vector<Cycle> cycles;
    std::vector<char> used(outer.size(), 0);
    const CycleSearch search{seq, outer, from_off, from_pos, pts};
    for (int i = 0; i < (int)outer.size(); ++i) {
        if (used[i]) continue;
        Cycle cyc{};
        cyc.e[0] = i;
        if (!search.extend(cyc, 1, seq[outer[i]].c0)) continue;
        cycles.push_back(cyc);
        for (int k = 0; k < 4; ++k) used[cyc.e[k]] = 1;
    }
    auto dump_cycles = [&](const char* fn, int ncyc, auto&& cyc_of, auto&& label) {
        FILE* fp = fopen(fn, "w");
        if (!fp) return;
        fprintf(fp, "#!/usr/bin/feedgnuplot --datai --dom --aut --square --rangesizea 3 --w 'vec size screen 0.01,20 fixed fill' --set 'yr [:] rev'\n");
        fprintf(fp, "# fromx type fromy deltax deltay\n");
        for (int ic = 0; ic < ncyc; ++ic)
            for (int ie = 0; ie < 4; ++ie) {
                const Sequence& cs = seq[outer[cyc_of(ic).e[ie]]];
                fprintf(fp, "%f %s %f %f %f\n", (double)pts[cs.c0].x / (double)kScale, label(ic, ie).c_str(),
                        (double)pts[cs.c0].y / (double)kScale, cs.delta_mean.x / (double)kScale, cs.delta_mean.y / (double)kScale);
            }
        fclose(fp);
        make_executable(fn);
        fprintf(stderr, "Wrote outer edge cycle dump to %s\n", fn);
    };
    if (debug && !cycles.empty())
        dump_cycles("/tmp/mrgingham-5-outer-edge-cycles", (int)cycles.size(), [&](int ic) -> const Cycle& { return cycles[ic]; },
                    [](int ic, int) { return std::to_string(ic); });
    if (cycles.size() < 2) {
        if (debug) fprintf(stderr, "Found too few 4-cycles. Needed at least 2, got %d\n", (int)cycles.size());
        return false;
    }

    // exactly one equal-and-opposite pair (:953-1003, :1324-1352)
    auto opposite = [&](const Cycle& a, const Cycle& b) {
        int ia = 0, ib = -1;
        const int p0 = seq[outer[a.e[0]]].c0;
        for (int k = 0; k < 4; ++k)
            if (seq[outer[b.e[k]]].clast == p0) { ib = k; break; }
        if (ib < 0) return false;
        for (int i = 0; i < 4; ++i) {
            const Sequence& sa = seq[outer[a.e[ia]]];
            const Sequence& sb = seq[outer[b.e[ib]]];
            if (sa.c0 != sb.clast || sa.clast != sb.c0) return false;
            ia = (ia + 1) % 4;
            ib = (ib + 3) % 4;
        }
        return true;
    };
    int pair[2] = {-1, -1};
    for (int i0 = 0; i0 < (int)cycles.size(); ++i0)
        for (int i1 = i0 + 1; i1 < (int)cycles.size(); ++i1)
            if (opposite(cycles[i0], cycles[i1])) {
                if (pair[0] >= 0) {
                    if (debug) fprintf(stderr, "Found more than one equal-and-opposite pair of outer-edge cycles. Giving up\n");
                    return false;
                }
                pair[0] = i0;
                pair[1] = i1;
            }
    if (pair[0] < 0) {
        if (debug) fprintf(stderr, "Didn't find any equal-and-opposite pairs of outer-edge cycles. Giving up\n");
        return false;
    }

    // clockwise cycle and its top edge (:1025-1190)
    const Cycle* cyc2[2] = {&cycles[pair[0]], &cycles[pair[1]]};
    int v[4][2];
    for (int i = 0; i < 4; ++i) {
        const Sequence& s = seq[outer[cyc2[0]->e[i]]];
        v[i][0] = (pts[s.clast].x - pts[s.c0].x) / 1024;  // FIND_GRID_SCALE_APPROX_POWER2
        v[i][1] = (pts[s.clast].y - pts[s.c0].y) / 1024;
    }
    bool sign[4];
    for (int i0 = 0; i0 < 4; ++i0) {
        const int i1 = (i0 + 1) % 4;
        sign[i0] = v[i1][0] * v[i0][1] < v[i0][0] * v[i1][1];
    }
    int iclockwise;
    if (sign[0] && sign[1] && sign[2] && sign[3]) iclockwise = 0;
    else if (!sign[0] && !sign[1] && !sign[2] && !sign[3]) iclockwise = 1;
    else return false;  // not convex

    int itop[2];
    for (int ic = 0; ic < 2; ++ic) {
        int ymin[2] = {INT32_MAX, INT32_MAX}, emin[2] = {-1, -1}, plo[2] = {0, 0}, phi[2] = {0, 0};
        for (int i = 0; i < 4; ++i) {
            const Sequence& s = seq[outer[cyc2[ic]->e[i]]];
            int y_here, lo, hi;
            if (pts[s.c0].y < pts[s.clast].y) { y_here = pts[s.c0].y; lo = s.c0; hi = s.clast; }
            else { y_here = pts[s.clast].y; lo = s.clast; hi = s.c0; }
            if (y_here < ymin[0]) {
                ymin[1] = ymin[0]; emin[1] = emin[0]; plo[1] = plo[0]; phi[1] = phi[0];
                ymin[0] = y_here; emin[0] = i; plo[0] = lo; phi[0] = hi;
            } else if (y_here < ymin[1]) {
                ymin[1] = y_here; emin[1] = i; plo[1] = lo; phi[1] = hi;
            }
        }
        i64 v0y = (pts[phi[0]].y - pts[plo[0]].y) / 1024, v0x = (pts[phi[0]].x - pts[plo[0]].x) / 1024;
        i64 v1y = (pts[phi[1]].y - pts[plo[1]].y) / 1024, v1x = (pts[phi[1]].x - pts[plo[1]].x) / 1024;
        v0x = v0x > 0 ? v0x : -v0x;
        v1x = v1x > 0 ? v1x : -v1x;
        const i64 cr = (v0x * v1y - v0y * v1x) * (v0x * v1y - v0y * v1x);
        const i64 den = (v0x * v0x + v0y * v0y) * (v1x * v1x + v1y * v1y);
        if ((cr > 0 ? cr : -cr) * 8 < den * 1) return false;  // the two highest edges are too parallel (:1153-1156)
        const i64 l = v0y * v1x, rr = v1y * v0x;
        itop[ic] = ((l > 0 ? l : -l) < (rr > 0 ? rr : -rr)) ? emin[0] : emin[1];
    }

    if (debug)
        dump_cycles("/tmp/mrgingham-6-identified-outer-edge-cycle", 2, [&](int ic) -> const Cycle& { return *cyc2[ic]; },
                    [&](int ic, int ie) {
                        return std::string(ic == iclockwise ? "clockwise" : "counterclockwise") + (itop[ic] == ie ? "-top" : "");
                    });
    // rows between the two vertical outer edges (:1378-1440)
    std::vector<int> sf_off, sf_pos;
    by_site((int)seq.size(), [&](int i) { return seq[i].c0; }, sf_off, sf_pos);
    auto seq_from_to = [&](int from, int to) {
        for (int q = sf_off[from]; q < sf_off[from + 1]; ++q)
            if (seq[sf_pos[q]].clast == to) return sf_pos[q];
        return -1;
    };
    std::vector<int> rows(gridn);
    rows[0] = outer[cyc2[iclockwise]->e[itop[iclockwise]]];
    const int vleft = outer[cyc2[1 - iclockwise]->e[(itop[1 - iclockwise] + 1) % 4]];
    const int vright = outer[cyc2[iclockwise]->e[(itop[iclockwise] + 1) % 4]];
    const std::vector<int> lp = sequence_points(adj, pts, seq[vleft], gridn);
    const std::vector<int> rp = sequence_points(adj, pts, seq[vright], gridn);
    if ((int)lp.size() != gridn || (int)rp.size() != gridn) return false;
    for (int i = 1; i < gridn; ++i) {
        const int s = seq_from_to(lp[i], rp[i]);
        if (s < 0) {
            if (debug) fprintf(stderr, "Couldn't find sequence in row %d\n", i);
            return false;
        }
        rows[i] = s;
        if (seq_from_to(rp[i], lp[i]) < 0) {
            if (debug) fprintf(stderr, "Row %d: left-to-right sequence was found, but right-to-left sequence doesn't exist!\n", i);
            return false;
        }
    }
    for (int i = 0; i < gridn; ++i) {
        const std::vector<int> row = sequence_points(adj, pts, seq[rows[i]], gridn);
        if ((int)row.size() != gridn) return false;
        for (int s : row) out.push_back(PointD{(double)pts[s].x / 1000.0, (double)pts[s].y / 1000.0});  // :353-354
    }
    if (debug) fprintf(stderr, "Success. Found grid\n");
    return true;
}

}  // namespace mrg
