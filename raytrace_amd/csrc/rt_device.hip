// rt_device.hip -- MI355X (gfx950) implementation of the RayTrace per-pixel path behind the
// C ABI of include/rt_hip.h.
//
// What runs on the GPU (reference lines it restates):
//   camera ray + tent filter + 2x2 sub-pixel accumulate   raytrace.cpp:196-205
//   radiance(), made iterative (depth-first, refraction first)   raytrace.cpp:41-95
//   KdTree::check / check_node, made iterative with a per-lane stack   kdtree.cpp:98-142
//   KdTreeNode::check_aabb   kdtree.h:57-76
//   Polygon::intersect       polygon.h:37-60
//   Sphere::intersect / sphere_intersect   sphere.h:13-21, sphere.cpp:3-14
//
// Numerics: all arithmetic is FP64 in the reference's operation order; this file must be
// compiled with -ffp-contract=off so no multiply-add is fused (the CPU path has no FMA), and
// double divide / sqrt are the correctly rounded expansions.  Primary-hit ids and distances
// are therefore bit-identical to the CPU path; colours differ only through sin/cos (last-ulp)
// and through summing emission along the path front-to-back instead of back-to-front.
//
// Layout in HBM (built by rt_upload_scene):
//   NodeRec   64 B, 64-B aligned: box (48), split (8), right child (4), info (4: axis, has-left,
//             own-list length).  Nodes are in pre-order so the left child is node+1.
//   ListInfo  48 B per node: start of the node's own list, and for long lists the root of a
//             culling hierarchy plus the data for its safety margin
//   TriRec    80 B, 16-B aligned, grouped by own list (each triangle is in exactly one list), so a
//             node's list is a contiguous run: p0, E1 = p1-p0, E2 = p2-p0, global id, list rank
//   CullRec   64 B: box + two children, or a run of <= 4 triangles (long own lists only)
//   TriShade  80 B per triangle, same order: nc, col, lig, type (touched only on the chosen hit)
//   SphGeom   32 B: centre, radius;  SphShade 64 B: lig, col, type
//
// Execution: persistent workgroups of 4 wavefronts; each wavefront repeatedly takes one 8x8
// pixel block (a 64-lane packet, one pixel per lane) from an atomic counter, and each lane runs
// its pixel's 4*samps paths to completion, so the per-pixel sums happen in the reference's
// order with no atomics.  The traversal stack lives in LDS ([level][thread], conflict-free);
// levels beyond LDS_STACK spill to a per-thread HBM array sized from the tree depth.
//
// Long own lists.  The reference tests EVERY triangle of a visited node's own list, and its
// median-split tree leaves thousands of straddling triangles on the top nodes.  Testing them all
// is not needed to get the same answer: check_node's result is "smallest t > eps, first in list
// order among equals" plus "does any t < 1e10 exist", and both only depend on the triangles that
// Polygon::intersect accepts.  So each long list gets a small bounding-box hierarchy, built at
// upload, whose boxes are inflated by a margin that provably exceeds how far outside a triangle
// the floating-point Moller-Trumbore test can still accept a ray (see list_margin()); subtrees
// whose inflated box the ray misses are skipped, the surviving triangles go through the unchanged
// exact test, and ties are broken by the stored list rank.  Results are identical by construction.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace {

constexpr double EPS = 1e-5;  // point.h:7
constexpr int BLOCK = 256;    // threads per workgroup (4 wavefronts)
constexpr int LDS_STACK = 32; // traversal-stack levels kept in LDS
constexpr int MAX_SPLIT = 4;  // pending glass branches per path
constexpr unsigned NODE_MASK = 0x0FFFFFFFu;
constexpr int N_STATS = 7;
constexpr int CULL_MIN = 12; // own lists longer than this get a culling hierarchy
constexpr int CULL_LEAF = 4; // triangles per hierarchy leaf

struct NodeRec {
    double mi[3], ma[3];
    double split;
    int right;     // 0 = none (the root is never a child)
    unsigned info; // bits 0-1 axis, bit 2 has-left, bits 3.. own-list length
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be 64 bytes");

struct __attribute__((aligned(16))) TriRec {
    double p0[3], e1[3], e2[3];
    int gid;  // global triangle id
    int rank; // position in the reference's list order (decides between equal distances)
};
static_assert(sizeof(TriRec) == 80, "TriRec must be 80 bytes");

struct __attribute__((aligned(16))) TriShade {
    double nc[3], col[3], lig[3];
    int type, pad;
};
static_assert(sizeof(TriShade) == 80, "TriShade must be 80 bytes");

struct __attribute__((aligned(16))) ListInfo {
    double c[3]; // centre of the list's bounding box
    double k2;   // margin per unit of distance, from the longest triangle edge of the list
    double r2;   // box half-diagonal + longest edge
    int first;   // first TriRec of the list
    int root;    // root CullRec, or -1 when the list is scanned linearly
};
static_assert(sizeof(ListInfo) == 48, "ListInfo must be 48 bytes");

struct __attribute__((aligned(16))) CullRec {
    double lo[3], hi[3];
    int a, b; // b >= 0: children a, b;  b < 0: leaf with -b triangles starting at TriRec a
};
static_assert(sizeof(CullRec) == 64, "CullRec must be 64 bytes");

struct __attribute__((aligned(16))) SphGeom { double px, py, pz, rad; };
struct __attribute__((aligned(16))) SphShade { double lig[3], col[3]; int type, pad0; double pad1; };
static_assert(sizeof(SphShade) == 64, "SphShade must be 64 bytes");

struct DevScene {
    const NodeRec *nodes;
    const ListInfo *lists;
    const CullRec *cull;
    const TriRec *tris;
    const TriShade *tshade;
    const SphGeom *sph;
    const SphShade *sshade;
    unsigned *ovf;      // overflow stack [ovf_depth][n_threads]
    int n_nodes, n_tris, n_sph;
    int ovf_depth;
    int n_threads;
};

struct DevCamera {
    double o[3], d[3], cx[3], cy[3];
    int sizex, sizey;
};

struct DevTile {
    int x0, y0, rows, cols;
    long long out_off; // first pixel of this tile in the packed output
    int blk_first;     // index of this tile's first 8x8 block
    int blk_cols;      // 8x8 blocks per block-row
};

struct d3 { double x, y, z; };
__device__ __forceinline__ d3 mk(double x, double y, double z) { d3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ d3 operator+(d3 a, d3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ d3 operator-(d3 a, d3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ d3 operator*(d3 a, double b) { return mk(a.x * b, a.y * b, a.z * b); }
__device__ __forceinline__ d3 mult(d3 a, d3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ double dot(d3 a, d3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ d3 cross(d3 a, d3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ d3 normv(d3 a) { return a * (1 / sqrt(a.x * a.x + a.y * a.y + a.z * a.z)); } // point.h:49
__device__ __forceinline__ d3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }

struct Stats { unsigned long long rays, visits, tris, sphs, shades, done, culls; };

// drand48 recurrence; every path restarts it from a hashed state
__device__ __forceinline__ double rng_next(unsigned long long &x) {
    x = (x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)x * (1.0 / 281474976710656.0);
}
__host__ __device__ __forceinline__ unsigned long long path_state(unsigned long long seed, unsigned long long path) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ULL * (path + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z >> 16;
}

// kdtree.h:57-76
__device__ __forceinline__ bool aabb_hit(const double (&mi)[3], const double (&ma)[3], const double (&s)[3],
                                         const double (&v)[3]) {
    if (mi[0] < s[0] + EPS && s[0] < ma[0] + EPS && mi[1] < s[1] + EPS && s[1] < ma[1] + EPS && mi[2] < s[2] + EPS &&
        s[2] < ma[2] + EPS)
        return true;
    bool hit = false;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;
        if (fabs(v[i]) > EPS) {
            const double t = ((v[i] > 0 ? mi[i] : ma[i]) - s[i]) / v[i];
            if (t > EPS) {
                const double pj = s[j] + v[j] * t, pk = s[k] + v[k] * t;
                if (mi[j] < pj + EPS && pj < ma[j] + EPS && mi[k] < pk + EPS && pk < ma[k] + EPS) hit = true;
            }
        }
    }
    return hit;
}

// polygon.h:37-60 with E1, E2 precomputed by the same subtractions; returns 1e20 on a miss
__device__ __forceinline__ double tri_hit(const TriRec &T, d3 o, d3 d) {
    const d3 p0 = ld3(T.p0), E1 = ld3(T.e1), E2 = ld3(T.e2);
    const d3 P = cross(d, E2);
    double det = dot(E1, P);
    d3 Tv;
    if (det > 0) Tv = o - p0;
    else { Tv = p0 - o; det = -det; }
    if (det < EPS) return 1e20;
    const double u = dot(Tv, P);
    if (u < 0 || u > det) return 1e20;
    const d3 Q = cross(Tv, E1);
    const double v = dot(d, Q);
    if (v < 0 || u + v > det) return 1e20;
    double t = dot(E2, Q);
    const double fInvDet = 1.0 / det;
    t *= fInvDet;
    return t;
}

// Safety margin for the culling boxes of one own list.  In floating point, Polygon::intersect can
// accept a ray whose exact line passes slightly OUTSIDE the triangle: the rounding error of u, v
// and det is at most ~40 ulp * |T| * L^2 (T = origin - p0, L = longest edge), and the acceptance
// needs det >= eps = 1e-5, so in barycentric terms the slack is <= 40 * 1.1e-16 * |T| * L^2 / 1e-5
// and the accepted point lies within 4.4e-10 * L^2 * (|T| + L) of the triangle.  The margin used is
// (1e-9 * L^2 + 1e-10) * (|o - c| + R + L) with c, R the list box centre / half-diagonal (|T| <=
// |o - c| + R), i.e. more than twice that bound; cull_hit() then inflates by twice the margin again,
// which also swallows the few-ulp error of the slab arithmetic itself.
__device__ __forceinline__ double list_margin(const ListInfo &li, const double (&s)[3]) {
    const double dx = s[0] - li.c[0], dy = s[1] - li.c[1], dz = s[2] - li.c[2];
    return li.k2 * (sqrt(dx * dx + dy * dy + dz * dz) + li.r2);
}

// Conservative ray (t >= 0) versus inflated box: never false for a ray the exact triangle test
// could accept; false positives only cost an extra exact test.
__device__ __forceinline__ bool cull_hit(const CullRec &c, double m2, const double (&s)[3], const double (&inv)[3],
                                         const bool (&par)[3]) {
    double tmin = 0.0, tmax = 1e300;
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double lo = c.lo[i] - m2, hi = c.hi[i] + m2;
        if (par[i]) { if (s[i] < lo || s[i] > hi) ok = false; }
        else {
            const double t1 = (lo - s[i]) * inv[i], t2 = (hi - s[i]) * inv[i];
            const double ta = t1 < t2 ? t1 : t2, tb = t1 < t2 ? t2 : t1;
            tmin = ta > tmin ? ta : tmin;
            tmax = tb < tmax ? tb : tmax;
        }
    }
    return ok && tmin <= tmax;
}

// KdTree::check (kdtree.cpp:111-142) without recursion.  The recursion's return value
// ("this subtree has a hit", dis < 1e10) is the `flag` carried per frame; a stack entry holds
// the interior node whose near child is being walked, which child is the far one, and the
// frame's flag at the time of the descent.  Order per node: near child, own list, far child;
// strict < keeps the first of equal distances, as the nested comparisons of the reference do.
__device__ __forceinline__ void trace_tris(const DevScene &S, d3 o, d3 d, double &best_t, int &best_pos,
                                           unsigned (*stk)[BLOCK], int tid, int gthread, Stats &st) {
    best_t = 1e20;
    best_pos = -1;
    const double s[3] = {o.x, o.y, o.z}, v[3] = {d.x, d.y, d.z};
    // for the culling boxes: reciprocal direction, or "parallel" where a component is (nearly) zero
    const bool par[3] = {fabs(v[0]) < 1e-150, fabs(v[1]) < 1e-150, fabs(v[2]) < 1e-150};
    const double inv[3] = {par[0] ? 0.0 : 1.0 / v[0], par[1] ? 0.0 : 1.0 / v[1], par[2] ? 0.0 : 1.0 / v[2]};
    const double dlen2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    const bool can_cull = dlen2 > 0.25 && dlen2 < 4.0; // the margin analysis assumes a unit-ish direction
    int cur = 0, sp = 0;
    bool flag = false;
    for (;;) {
        int own = -1, farkind = 0;
        bool near_hit = false;
        if (cur >= 0) { // enter node `cur`
            const NodeRec *nd = S.nodes + cur;
            const double mi[3] = {nd->mi[0], nd->mi[1], nd->mi[2]}, ma[3] = {nd->ma[0], nd->ma[1], nd->ma[2]};
            const double split = nd->split;
            const int right = nd->right;
            const unsigned info = nd->info;
            st.visits++;
            if (!aabb_hit(mi, ma, s, v)) cur = -1;
            else {
                const int axis = info & 3;
                const bool hasL = (info >> 2) & 1, hasR = right > 0;
                const double sw = axis == 0 ? s[0] : axis == 1 ? s[1] : s[2];
                const double vw = axis == 0 ? v[0] : axis == 1 ? v[1] : v[2];
                int near = -1;
                if (!hasL && !hasR) { /* leaf, or interior node whose triangles all straddle */ }
                else if (sw < split) { // kdtree.cpp:119-127
                    if (hasL) near = cur + 1;
                    if (hasR && vw > 0) farkind = 2;
                } else if (sw > split) { // kdtree.cpp:128-136
                    if (hasR) near = right;
                    if (hasL && vw < 0) farkind = 1;
                } else { // kdtree.cpp:137-141: on the plane, only the child the ray points into
                    if (vw < 0 && hasL) near = cur + 1;
                    else if (vw > 0 && hasR) near = right;
                }
                if (near >= 0) {
                    const unsigned e = (unsigned)cur | ((unsigned)farkind << 28) | ((unsigned)flag << 30);
                    if (sp < LDS_STACK) stk[sp][tid] = e;
                    else S.ovf[(size_t)(sp - LDS_STACK) * S.n_threads + gthread] = e;
                    sp++;
                    flag = false;
                    cur = near;
                } else {
                    own = cur;
                    cur = -1;
                }
            }
        }
        if (cur < 0 && own < 0) { // a subtree has returned
            if (sp == 0) break;
            sp--;
            const unsigned e = sp < LDS_STACK ? stk[sp][tid] : S.ovf[(size_t)(sp - LDS_STACK) * S.n_threads + gthread];
            own = (int)(e & NODE_MASK);
            farkind = (e >> 28) & 3;
            near_hit = flag;
            flag = ((e >> 30) & 1) | near_hit;
        }
        if (own >= 0) { // check_node (kdtree.cpp:98-109), then the far child
            const NodeRec *nd = S.nodes + own;
            const int right = nd->right;
            const int count = (int)(nd->info >> 3);
            bool any = false;
            if (count > 0) {
                const ListInfo *li = S.lists + own;
                const int first = li->first, root = li->root;
                if (root < 0 || !can_cull) { // short list: the reference's scan, in its order
                    for (int i = 0; i < count; i++) {
                        const double t = tri_hit(S.tris[first + i], o, d);
                        if (t > EPS) {
                            if (t < best_t) { best_t = t; best_pos = first + i; }
                            if (t < 1e10) any = true;
                        }
                    }
                    st.done += (unsigned)count;
                } else { // long list: skip what the ray provably cannot hit, exact test on the rest
                    const double m2 = 2.0 * list_margin(*li, s);
                    double lt = 1e20;
                    int lpos = -1, lrank = 0x7FFFFFFF;
                    const int base = sp;
                    int c = root;
                    for (;;) {
                        const CullRec *cr = S.cull + c;
                        st.culls++;
                        bool descend = false;
                        if (cull_hit(*cr, m2, s, inv, par)) {
                            const int a = cr->a, b = cr->b;
                            if (b < 0) {
                                for (int i = 0; i < -b; i++) {
                                    const TriRec &T = S.tris[a + i];
                                    const double t = tri_hit(T, o, d);
                                    if (t > EPS) {
                                        const int rk = T.rank;
                                        if (t < lt || (t == lt && rk < lrank)) { lt = t; lpos = a + i; lrank = rk; }
                                        if (t < 1e10) any = true;
                                    }
                                }
                                st.done += (unsigned)(-b);
                            } else { // visit a next, b later
                                if (sp < LDS_STACK) stk[sp][tid] = (unsigned)b;
                                else S.ovf[(size_t)(sp - LDS_STACK) * S.n_threads + gthread] = (unsigned)b;
                                sp++;
                                c = a;
                                descend = true;
                            }
                        }
                        if (!descend) {
                            if (sp == base) break;
                            sp--;
                            c = (int)(sp < LDS_STACK ? stk[sp][tid] : S.ovf[(size_t)(sp - LDS_STACK) * S.n_threads + gthread]);
                        }
                    }
                    if (lt < best_t) { best_t = lt; best_pos = lpos; }
                }
            }
            st.tris += (unsigned)count;
            flag = flag | any;
            if (farkind != 0 && !near_hit) cur = farkind == 1 ? own + 1 : right;
            else cur = -1;
        }
    }
}

// sphere.cpp:3-14 over sphere.h:13-21: scan from the last sphere down, strict <
__device__ __forceinline__ void trace_spheres(const DevScene &S, d3 o, d3 d, double &ts, int &ids, Stats &st) {
    ts = 1e20;
    ids = -1;
    for (int i = S.n_sph; i--;) {
        const SphGeom g = S.sph[i];
        const d3 op = mk(g.px, g.py, g.pz) - o;
        const double b = dot(op, d);
        double det = b * b - dot(op, op) + g.rad * g.rad;
        double t;
        if (det < 0) t = 1e20;
        else {
            det = sqrt(det);
            t = b - det;
            if (!(t > EPS)) { t = b + det; if (!(t > EPS)) t = 1e20; }
        }
        if (t < ts) { ts = t; ids = i; }
    }
    st.sphs += (unsigned)S.n_sph;
}

struct Pending { d3 o, d, thr; int depth; int into; };

// radiance() (raytrace.cpp:41-95) as a loop.  Emission is summed front to back with the path
// throughput `thr`; where the reference traces both glass branches (depth <= split_depth) the
// reflection branch is parked and the refraction branch is followed first, which is the order
// the random stream is consumed in.
__device__ __forceinline__ d3 radiance(const DevScene &S, const rt_params &P, d3 ro, d3 rd, unsigned long long &rng,
                                       unsigned (*stk)[BLOCK], int tid, int gthread, Stats &st, int &first_id) {
    d3 acc = mk(0, 0, 0), thr = mk(1, 1, 1);
    int depth = 0;
    bool into = true;
    Pending pend[MAX_SPLIT];
    int np = 0;
    bool primary = true;
    for (;;) {
        double to, ts;
        int tpos, ids;
        st.rays++;
        trace_tris(S, ro, rd, to, tpos, stk, tid, gthread, st);
        trace_spheres(S, ro, rd, ts, ids, st);
        const bool fo = to < 1e10, fs = ts < 1e10;
        bool go_on = false;
        if (fo || fs) {
            d3 x, n, f, lig;
            int type;
            if (ts < to || !fo) { // raytrace.cpp:52-57
                const SphGeom g = S.sph[ids];
                const SphShade sh = S.sshade[ids];
                x = ro + rd * ts;
                n = normv(x - mk(g.px, g.py, g.pz));
                f = ld3(sh.col); type = sh.type; lig = ld3(sh.lig);
                if (primary) first_id = -1 - ids;
            } else { // raytrace.cpp:58-67; getn() returns the face normal (polygon.h:67-69)
                const TriShade *sh = S.tshade + tpos;
                x = ro + rd * to;
                n = ld3(sh->nc);
                f = ld3(sh->col); type = sh->type; lig = ld3(sh->lig);
                if (primary) first_id = S.tris[tpos].gid;
                st.shades++;
            }
            const d3 nl = dot(n, rd) < 0 ? n : n * -1.0;
            acc = acc + mult(thr, lig);
            const double p = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z; // raytrace.cpp:69
            ++depth;
            go_on = true;
            if (depth > P.hard_depth) go_on = false;
            else if (depth > P.rr_depth) { // raytrace.cpp:70-74
                if (P.rr_mode && rng_next(rng) < p) f = f * (1 / p);
                else go_on = false;
            }
            if (go_on) {
                thr = mult(thr, f);
                if (type == RT_DIFF) { // raytrace.cpp:77-82
                    const double r1 = 2 * M_PI * rng_next(rng), r2 = rng_next(rng), r2s = sqrt(r2);
                    const d3 w = nl, u = normv(cross(fabs(w.x) > .1 ? mk(0, 1, 0) : mk(1, 0, 0), w)), vv = cross(w, u);
                    const d3 nd = normv(u * cos(r1) * r2s + vv * sin(r1) * r2s + w * sqrt(1 - r2));
                    ro = x - rd * EPS;
                    rd = nd;
                } else if (type == RT_SPEC) { // raytrace.cpp:83-84
                    const d3 nd = rd - n * 2.0 * dot(n, rd);
                    ro = x - rd * EPS;
                    rd = nd;
                } else { // raytrace.cpp:86-94
                    const d3 reflo = x - rd * EPS, refld = rd - n * 2.0 * dot(n, rd);
                    const double nc = 1, nt = 1.7, nnt = into ? nc / nt : nt / nc, ddn = dot(rd, nl);
                    const double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                    if (cos2t < 0) { ro = reflo; rd = refld; } // total internal reflection
                    else {
                        const d3 tdir = normv(rd * nnt - n * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
                        const double a = nt - nc, b = nt + nc, R0 = a * a / (b * b), c = 1 - (into ? -ddn : dot(tdir, n));
                        const double Re = R0 + (1 - R0) * c * c * c * c * c, Tr = 1 - Re, Pp = .25 + .5 * Re, RP = Re / Pp,
                                     TP = Tr / (1 - Pp);
                        const d3 refro = x + rd * EPS;
                        if (depth > P.split_depth) {
                            if (rng_next(rng) < Pp) { thr = thr * RP; ro = reflo; rd = refld; }
                            else { thr = thr * TP; ro = refro; rd = tdir; into = !into; }
                        } else {
                            Pending &q = pend[np++];
                            q.o = reflo; q.d = refld; q.thr = thr * Re; q.depth = depth; q.into = into;
                            thr = thr * Tr; ro = refro; rd = tdir; into = !into;
                        }
                    }
                }
            }
        } else if (primary) first_id = RT_ID_MISS;
        primary = false;
        if (!go_on) { // this branch of the path is finished: resume a parked branch or stop
            if (np == 0) break;
            const Pending &q = pend[--np];
            ro = q.o; rd = q.d; thr = q.thr; depth = q.depth; into = q.into != 0;
        }
    }
    return acc;
}

__device__ __forceinline__ double clamp01(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; } // raytrace.cpp:28-31

__device__ __forceinline__ void wave_add_stats(const Stats &st, unsigned long long *g) {
    unsigned long long v[N_STATS] = {st.rays, st.visits, st.tris, st.sphs, st.shades, st.done, st.culls};
#pragma unroll
    for (int k = 0; k < N_STATS; k++) {
        unsigned long long x = v[k];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(g + k, x);
    }
}

// One wavefront = one 8x8 pixel block at a time, pulled from `counter`; one pixel per lane.
__global__ __launch_bounds__(BLOCK) void k_render(DevScene S, DevCamera cam, rt_params P, const DevTile *tiles, int n_tiles,
                                                  unsigned n_blocks, unsigned *counter, double *rgb, int *prim,
                                                  unsigned long long *gstats) {
    __shared__ unsigned stk[LDS_STACK][BLOCK];
    const int tid = threadIdx.x, lane = tid & 63;
    const int gthread = blockIdx.x * BLOCK + tid;
    Stats st = {0, 0, 0, 0, 0, 0, 0};
    const d3 camo = ld3(cam.o), camd = ld3(cam.d), cx = ld3(cam.cx), cy = ld3(cam.cy);
    const int sizex = cam.sizex, sizey = cam.sizey, samps = P.samps;
    for (;;) {
        unsigned blk = 0;
        if (lane == 0) blk = atomicAdd(counter, 1u);
        blk = __shfl(blk, 0, 64);
        if (blk >= n_blocks) break;
        int lo = 0, hi = n_tiles - 1; // last tile whose first block is <= blk
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if ((unsigned)tiles[mid].blk_first <= blk) lo = mid; else hi = mid - 1;
        }
        const DevTile T = tiles[lo];
        const int b = (int)blk - T.blk_first;
        const int lrow = (b / T.blk_cols) * 8 + (lane >> 3), lcol = (b % T.blk_cols) * 8 + (lane & 7);
        if (lrow < T.rows && lcol < T.cols) {
            const int x = T.x0 + lrow, y = T.y0 + lcol;
            d3 col = mk(0, 0, 0);
            int first_id = RT_ID_MISS, pix_id = RT_ID_MISS;
            for (int sx = 0; sx < 2; sx++)
                for (int sy = 0; sy < 2; sy++) { // raytrace.cpp:196-206
                    d3 r = mk(0, 0, 0);
                    for (int s = 0; s < samps; s++) {
                        const unsigned long long path =
                            ((((unsigned long long)x * (unsigned long long)sizey + (unsigned long long)y) * 2 + sx) * 2 + sy) *
                                (unsigned long long)samps + s;
                        unsigned long long rng = path_state(P.seed, path);
                        double dx = 0, dy = 0;
                        if (P.jitter) {
                            const double r1 = 2 * rng_next(rng); dx = r1 < 1 ? sqrt(r1) - 1 : 1 - sqrt(2 - r1);
                            const double r2 = 2 * rng_next(rng); dy = r2 < 1 ? sqrt(r2) - 1 : 1 - sqrt(2 - r2);
                        }
                        const d3 dd = cx * (((sx + .5 + dx) / 2 + x) / sizex - .5) + cy * (((sy + .5 + dy) / 2 + y) / sizey - .5) + camd;
                        // raytrace.cpp:203: d is normalised in place before the origin is formed
                        // (g++ evaluates the constructor arguments right to left)
                        const d3 dn = normv(dd);
                        const d3 L = radiance(S, P, camo + dn * 140.0, dn, rng, stk, tid, gthread, st, first_id);
                        if (sx == 0 && sy == 0 && s == 0) pix_id = first_id;
                        r = r + L * (1.0 / samps);
                    }
                    col = col + mk(clamp01(r.x), clamp01(r.y), clamp01(r.z)) * .25;
                }
            const size_t k = (size_t)T.out_off + (size_t)lrow * T.cols + lcol;
            rgb[3 * k] = col.x; rgb[3 * k + 1] = col.y; rgb[3 * k + 2] = col.z;
            if (prim) prim[k] = pix_id;
        }
    }
    wave_add_stats(st, gstats);
}

// raytrace.cpp:46-47 for explicit rays, one ray per lane
__global__ __launch_bounds__(BLOCK) void k_trace(DevScene S, long long n, const double *rays, int *tri_id, double *tri_t,
                                                 int *sph_id, double *sph_t, unsigned long long *gstats) {
    __shared__ unsigned stk[LDS_STACK][BLOCK];
    const int tid = threadIdx.x;
    const int gthread = blockIdx.x * BLOCK + tid;
    Stats st = {0, 0, 0, 0, 0, 0, 0};
    for (long long i = gthread; i < n; i += (long long)gridDim.x * BLOCK) {
        const d3 o = ld3(rays + 6 * i), d = ld3(rays + 6 * i + 3);
        double to, ts;
        int tpos, ids;
        st.rays++;
        trace_tris(S, o, d, to, tpos, stk, tid, gthread, st);
        trace_spheres(S, o, d, ts, ids, st);
        if (to < 1e10 && !(ts < to)) st.shades++;
        tri_id[i] = to < 1e10 ? S.tris[tpos].gid : -1;
        tri_t[i] = to;
        sph_id[i] = ts < 1e10 ? ids : -1;
        sph_t[i] = ts;
    }
    wave_add_stats(st, gstats);
}

__global__ void k_fp64(int op, long long n, const double *a, const double *b, double *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = op == 0 ? a[i] / b[i] : sqrt(a[i]);
}

// ------------------------------------------------------------------------------------ host side

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(RT_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

} // namespace

struct rt_scene {
    int dev = 0;
    DevScene d{};
    void *nodes = nullptr, *first = nullptr, *cull = nullptr, *tris = nullptr, *tshade = nullptr, *sph = nullptr, *sshade = nullptr;
    void *ovf = nullptr;
    int tree_depth = 0;
    int grid = 0;               // persistent workgroups
    unsigned *d_counter = nullptr; // [0] block counter
    unsigned long long *d_stats = nullptr;
    DevTile *d_tiles = nullptr;
    int tiles_cap = 0;
    std::vector<DevTile> h_tiles;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    hipStream_t last_stream = nullptr;
};

namespace {

// Culling hierarchy for one long own list: median splits of the triangle centroids along the
// widest axis down to CULL_LEAF triangles.  `order[first..first+count)` is permuted into leaf
// order; boxes are the exact (uninflated) bounds of the triangles below.
struct CullTri { double lo[3], hi[3], c[3]; int src; };

int cull_rec(std::vector<CullRec> &out, std::vector<CullTri> &bt, int a, int b, int first, int depth, int &max_depth) {
    const int me = (int)out.size();
    out.emplace_back();
    if (depth > max_depth) max_depth = depth;
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (int i = a; i < b; i++)
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], bt[i].lo[k]); hi[k] = std::max(hi[k], bt[i].hi[k]);
            clo[k] = std::min(clo[k], bt[i].c[k]); chi[k] = std::max(chi[k], bt[i].c[k]);
        }
    for (int k = 0; k < 3; k++) { out[me].lo[k] = lo[k]; out[me].hi[k] = hi[k]; }
    if (b - a <= CULL_LEAF) {
        out[me].a = first + a;
        out[me].b = -(b - a);
        return me;
    }
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    const int mid = (a + b) / 2;
    std::nth_element(bt.begin() + a, bt.begin() + mid, bt.begin() + b, [ax](const CullTri &x, const CullTri &y) {
        return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.src < y.src);
    });
    const int l = cull_rec(out, bt, a, mid, first, depth + 1, max_depth);
    const int r = cull_rec(out, bt, mid, b, first, depth + 1, max_depth);
    out[me].a = l;
    out[me].b = r;
    return me;
}

void build_cull(const rt_scene_desc *D, int first, int count, std::vector<int> &order, std::vector<CullRec> &cull, ListInfo &li,
                int &max_depth) {
    std::vector<CullTri> bt((size_t)count);
    double lmax2 = 0;
    for (int j = 0; j < count; j++) {
        const double *v = D->tri_verts + 9 * (size_t)D->list_tri[first + j];
        CullTri &t = bt[j];
        t.src = first + j;
        for (int k = 0; k < 3; k++) {
            t.lo[k] = std::min(v[k], std::min(v[3 + k], v[6 + k]));
            t.hi[k] = std::max(v[k], std::max(v[3 + k], v[6 + k]));
            t.c[k] = (v[k] + v[3 + k] + v[6 + k]) / 3.0;
        }
        for (int e = 0; e < 3; e++) { // squared edge lengths
            const double *p = v + 3 * e, *q = v + 3 * ((e + 1) % 3);
            const double l2 = (p[0] - q[0]) * (p[0] - q[0]) + (p[1] - q[1]) * (p[1] - q[1]) + (p[2] - q[2]) * (p[2] - q[2]);
            lmax2 = std::max(lmax2, l2);
        }
    }
    int depth = 0;
    li.root = cull_rec(cull, bt, 0, count, first, 1, depth);
    max_depth = std::max(max_depth, depth);
    for (int j = 0; j < count; j++) order[first + j] = bt[j].src;
    const CullRec &root = cull[li.root];
    double r2 = 0;
    for (int k = 0; k < 3; k++) {
        li.c[k] = 0.5 * (root.lo[k] + root.hi[k]);
        r2 += 0.25 * (root.hi[k] - root.lo[k]) * (root.hi[k] - root.lo[k]);
    }
    li.k2 = 1e-9 * lmax2 + 1e-10;         // see list_margin()
    li.r2 = std::sqrt(r2) + std::sqrt(lmax2);
}

int validate(const rt_scene_desc *D) {
    if (!D) return fail(RT_E_INVALID, "desc is NULL");
    if (D->n_nodes < 1) return fail(RT_E_INVALID, "n_nodes must be >= 1 (got %d)", D->n_nodes);
    if ((unsigned)D->n_nodes > NODE_MASK) return fail(RT_E_LIMIT, "n_nodes %d exceeds %u", D->n_nodes, NODE_MASK);
    if (D->n_tris < 0 || D->n_spheres < 0) return fail(RT_E_INVALID, "negative primitive count");
    if (D->n_list != D->n_tris) return fail(RT_E_INVALID, "n_list (%d) must equal n_tris (%d)", D->n_list, D->n_tris);
    if (!D->node_box || !D->node_split || !D->node_axis || !D->node_left || !D->node_right || !D->node_first || !D->node_count)
        return fail(RT_E_INVALID, "node arrays missing");
    if (D->n_tris > 0 && (!D->list_tri || !D->tri_verts || !D->tri_nc || !D->tri_col || !D->tri_lig || !D->tri_type))
        return fail(RT_E_INVALID, "triangle arrays missing");
    if (D->n_spheres > 0 && (!D->sph_rad || !D->sph_pos || !D->sph_lig || !D->sph_col || !D->sph_type))
        return fail(RT_E_INVALID, "sphere arrays missing");
    if (D->tree_depth < 1) return fail(RT_E_INVALID, "tree_depth must be >= 1");
    long long run = 0;
    for (int i = 0; i < D->n_nodes; i++) {
        const int l = D->node_left[i], r = D->node_right[i];
        if (l != -1 && l != i + 1) return fail(RT_E_INVALID, "node %d: left child %d is not node+1 (pre-order required)", i, l);
        if (r != -1 && (r <= i || r >= D->n_nodes)) return fail(RT_E_INVALID, "node %d: right child %d out of range", i, r);
        if (l != -1 && l >= D->n_nodes) return fail(RT_E_INVALID, "node %d: left child out of range", i);
        if (D->node_axis[i] < 0 || D->node_axis[i] > 2) return fail(RT_E_INVALID, "node %d: axis %d", i, D->node_axis[i]);
        if (D->node_count[i] < 0 || D->node_count[i] >= (1 << 29)) return fail(RT_E_LIMIT, "node %d: list length %d", i, D->node_count[i]);
        if (D->node_first[i] != run) return fail(RT_E_INVALID, "node %d: own list must start at %lld (pre-order concatenation), got %d", i, run, D->node_first[i]);
        run += D->node_count[i];
    }
    if (run != D->n_list) return fail(RT_E_INVALID, "own lists cover %lld entries, n_list is %d", run, D->n_list);
    std::vector<unsigned char> seen((size_t)D->n_tris, 0);
    for (int i = 0; i < D->n_list; i++) {
        const int g = D->list_tri[i];
        if (g < 0 || g >= D->n_tris) return fail(RT_E_INVALID, "list entry %d: triangle id %d out of range", i, g);
        if (seen[g]) return fail(RT_E_INVALID, "triangle %d appears in more than one own list", g);
        seen[g] = 1;
    }
    for (int i = 0; i < D->n_tris; i++)
        if (D->tri_type[i] < 0 || D->tri_type[i] > 2) return fail(RT_E_INVALID, "triangle %d: type %d", i, D->tri_type[i]);
    for (int i = 0; i < D->n_spheres; i++)
        if (D->sph_type[i] < 0 || D->sph_type[i] > 2) return fail(RT_E_INVALID, "sphere %d: type %d", i, D->sph_type[i]);
    return RT_OK;
}

template <class T>
int upload(void **dst, const std::vector<T> &src) {
    const size_t bytes = (src.empty() ? 1 : src.size()) * sizeof(T);
    HIP_TRY(hipMalloc(dst, bytes));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return RT_OK;
}

int check_params(const rt_camera *cam, const rt_params *p) {
    if (!cam || !p) return fail(RT_E_INVALID, "camera or params is NULL");
    if (cam->sizex < 1 || cam->sizey < 1) return fail(RT_E_INVALID, "image size %d x %d", cam->sizex, cam->sizey);
    if (p->samps < 1) return fail(RT_E_INVALID, "samps must be >= 1");
    if (p->split_depth < 0 || p->split_depth > MAX_SPLIT) return fail(RT_E_LIMIT, "split_depth must be in [0,%d]", MAX_SPLIT);
    if (p->rr_depth < 0) return fail(RT_E_INVALID, "rr_depth must be >= 0");
    if (p->hard_depth < 1 || p->hard_depth > 100000) return fail(RT_E_INVALID, "hard_depth must be in [1,100000]");
    return RT_OK;
}

// raytrace.cpp:155-156
DevCamera make_camera(const rt_camera *c) {
    DevCamera k;
    const double l = 1 / std::sqrt(c->d[0] * c->d[0] + c->d[1] * c->d[1] + c->d[2] * c->d[2]);
    const double d[3] = {c->d[0] * l, c->d[1] * l, c->d[2] * l};
    const double cy[3] = {c->sizey * .5035 / c->sizex, 0, 0};
    double cr[3] = {cy[1] * d[2] - cy[2] * d[1], cy[2] * d[0] - cy[0] * d[2], cy[0] * d[1] - cy[1] * d[0]};
    const double m = 1 / std::sqrt(cr[0] * cr[0] + cr[1] * cr[1] + cr[2] * cr[2]);
    for (int i = 0; i < 3; i++) {
        k.o[i] = c->o[i];
        k.d[i] = d[i];
        k.cy[i] = cy[i];
        k.cx[i] = (cr[i] * m) * -.5035;
    }
    k.sizex = c->sizex;
    k.sizey = c->sizey;
    return k;
}

int launch_render(rt_scene *s, const rt_camera *cam, const rt_tile *tiles, int n_tiles, const rt_params *p, double *d_rgb,
                  int *d_prim, hipStream_t stream) {
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (!tiles || n_tiles < 1) return fail(RT_E_INVALID, "no tiles");
    std::vector<DevTile> ht((size_t)n_tiles);
    long long off = 0, blocks = 0;
    for (int i = 0; i < n_tiles; i++) {
        const rt_tile &t = tiles[i];
        if (t.rows < 1 || t.cols < 1 || t.x0 < 0 || t.y0 < 0 || (long long)t.x0 + t.rows > cam->sizex ||
            (long long)t.y0 + t.cols > cam->sizey)
            return fail(RT_E_INVALID, "tile %d (%d,%d,%d,%d) is empty or outside the %d x %d image", i, t.x0, t.y0, t.rows,
                        t.cols, cam->sizex, cam->sizey);
        DevTile &d = ht[i];
        d.x0 = t.x0; d.y0 = t.y0; d.rows = t.rows; d.cols = t.cols;
        d.out_off = off;
        d.blk_first = (int)blocks;
        d.blk_cols = (t.cols + 7) / 8;
        off += (long long)t.rows * t.cols;
        blocks += (long long)((t.rows + 7) / 8) * d.blk_cols;
        if (blocks > 0x7FFFFFF0LL) return fail(RT_E_LIMIT, "too many pixel blocks in one launch");
    }
    HIP_TRY(hipSetDevice(s->dev));
    if (n_tiles > s->tiles_cap) {
        if (s->d_tiles) HIP_TRY(hipFree(s->d_tiles));
        s->d_tiles = nullptr;
        s->tiles_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->d_tiles, sizeof(DevTile) * (size_t)n_tiles));
        s->tiles_cap = n_tiles;
        s->h_tiles.clear();
    }
    if (s->h_tiles.size() != ht.size() || memcmp(s->h_tiles.data(), ht.data(), sizeof(DevTile) * ht.size()) != 0) {
        s->h_tiles = ht; // the copy below reads s->h_tiles, which stays alive until the next call
        HIP_TRY(hipMemcpyAsync(s->d_tiles, s->h_tiles.data(), sizeof(DevTile) * ht.size(), hipMemcpyHostToDevice, stream));
    }
    HIP_TRY(hipMemsetAsync(s->d_counter, 0, 16, stream));
    const DevCamera dc = make_camera(cam);
    const int grid = (int)std::min<long long>(s->grid, (blocks + 3) / 4);
    HIP_TRY(hipEventRecord(s->ev0, stream));
    hipLaunchKernelGGL(k_render, dim3(grid), dim3(BLOCK), 0, stream, s->d, dc, *p, (const DevTile *)s->d_tiles, n_tiles,
                       (unsigned)blocks, s->d_counter, d_rgb, d_prim, s->d_stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(s->ev1, stream));
    s->timed = true;
    s->last_stream = stream;
    return RT_OK;
}

} // namespace

extern "C" {

const char *rt_last_error(void) { return g_err.c_str(); }

int rt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int rt_device_info(int dev, char *buf, size_t cap) {
    if (!buf || cap == 0) return fail(RT_E_INVALID, "buf is NULL or empty");
    hipDeviceProp_t pr;
    HIP_TRY(hipGetDeviceProperties(&pr, dev));
    snprintf(buf, cap,
             "device %d: %s (%s), %d CUs, wavefront %d, %.1f GiB global memory, %zu KiB LDS per workgroup, "
             "clock %d MHz, memory clock %d MHz, bus %d bit, L2 %d KiB",
             dev, pr.name, pr.gcnArchName, pr.multiProcessorCount, pr.warpSize, pr.totalGlobalMem / 1073741824.0,
             pr.sharedMemPerBlock / 1024, pr.clockRate / 1000, pr.memoryClockRate / 1000, pr.memoryBusWidth, pr.l2CacheSize / 1024);
    return RT_OK;
}

int rt_upload_scene(int dev, const rt_scene_desc *D, rt_scene **out) {
    if (!out) return fail(RT_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate(D);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (dev < 0 || dev >= ndev) return fail(RT_E_INVALID, "device %d not in [0,%d)", dev, ndev);
    HIP_TRY(hipSetDevice(dev));

    std::vector<NodeRec> nodes((size_t)D->n_nodes);
    std::vector<ListInfo> lists((size_t)D->n_nodes);
    std::vector<CullRec> cull;
    std::vector<int> order((size_t)D->n_list); // device position -> entry of list_tri
    for (int i = 0; i < D->n_list; i++) order[i] = i;
    int cull_depth = 0;
    for (int i = 0; i < D->n_nodes; i++) {
        NodeRec &n = nodes[i];
        for (int k = 0; k < 3; k++) { n.mi[k] = D->node_box[6 * i + k]; n.ma[k] = D->node_box[6 * i + 3 + k]; }
        n.split = D->node_split[i];
        n.right = D->node_right[i] > 0 ? D->node_right[i] : 0;
        n.info = (unsigned)D->node_axis[i] | (D->node_left[i] >= 0 ? 4u : 0u) | ((unsigned)D->node_count[i] << 3);
        ListInfo &li = lists[i];
        li.first = D->node_first[i];
        li.root = -1;
        li.c[0] = li.c[1] = li.c[2] = 0; li.k2 = 0; li.r2 = 0;
        if (D->node_count[i] > CULL_MIN) build_cull(D, D->node_first[i], D->node_count[i], order, cull, li, cull_depth);
    }
    if (cull.size() > 0x7FFFFFF0ull) return fail(RT_E_LIMIT, "culling hierarchy too large");
    std::vector<TriRec> tris((size_t)D->n_tris);
    std::vector<TriShade> tsh((size_t)D->n_tris);
    for (int pos = 0; pos < D->n_list; pos++) {
        const int rank = order[pos];
        const int g = D->list_tri[rank];
        const double *v = D->tri_verts + 9 * (size_t)g;
        TriRec &t = tris[pos];
        for (int k = 0; k < 3; k++) {
            t.p0[k] = v[k];
            t.e1[k] = v[3 + k] - v[k]; // polygon.h:39
            t.e2[k] = v[6 + k] - v[k]; // polygon.h:40
        }
        t.gid = g;
        t.rank = rank;
        TriShade &h = tsh[pos];
        for (int k = 0; k < 3; k++) {
            h.nc[k] = D->tri_nc[3 * (size_t)g + k];
            h.col[k] = D->tri_col[3 * (size_t)g + k];
            h.lig[k] = D->tri_lig[3 * (size_t)g + k];
        }
        h.type = D->tri_type[g];
        h.pad = 0;
    }
    std::vector<SphGeom> sg((size_t)D->n_spheres);
    std::vector<SphShade> ss((size_t)D->n_spheres);
    for (int i = 0; i < D->n_spheres; i++) {
        sg[i].px = D->sph_pos[3 * i]; sg[i].py = D->sph_pos[3 * i + 1]; sg[i].pz = D->sph_pos[3 * i + 2];
        sg[i].rad = D->sph_rad[i];
        for (int k = 0; k < 3; k++) { ss[i].lig[k] = D->sph_lig[3 * i + k]; ss[i].col[k] = D->sph_col[3 * i + k]; }
        ss[i].type = D->sph_type[i];
        ss[i].pad0 = 0; ss[i].pad1 = 0;
    }

    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_E_NOMEM, "out of host memory");
    s->dev = dev;
    s->tree_depth = D->tree_depth;
    rc = upload(&s->nodes, nodes);
    if (!rc) rc = upload(&s->first, lists);
    if (!rc) rc = upload(&s->cull, cull);
    if (!rc) rc = upload(&s->tris, tris);
    if (!rc) rc = upload(&s->tshade, tsh);
    if (!rc) rc = upload(&s->sph, sg);
    if (!rc) rc = upload(&s->sshade, ss);
    if (rc) { rt_free_scene(s); return rc; }

    // persistent grid: as many workgroups as the chip holds at this kernel's occupancy
    hipDeviceProp_t pr;
    int per_cu = 0;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render, BLOCK, 0) != hipSuccess || per_cu < 1)
        per_cu = 2;
    s->grid = pr.multiProcessorCount * per_cu;
    const int n_threads = s->grid * BLOCK;
    const int need_depth = D->tree_depth + cull_depth + 2; // main frames plus one list's culling walk on top
    const int ovf_depth = need_depth > LDS_STACK ? need_depth - LDS_STACK : 0;
    hipError_t e = hipSuccess;
    if (ovf_depth > 0) e = hipMalloc(&s->ovf, (size_t)ovf_depth * n_threads * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_counter, 16);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_stats, N_STATS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(s->d_stats, 0, N_STATS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e != hipSuccess) {
        rt_free_scene(s);
        return fail(RT_E_DEVICE, "device allocation failed: %s", hipGetErrorString(e));
    }
    s->d.nodes = (const NodeRec *)s->nodes;
    s->d.lists = (const ListInfo *)s->first;
    s->d.cull = (const CullRec *)s->cull;
    s->d.tris = (const TriRec *)s->tris;
    s->d.tshade = (const TriShade *)s->tshade;
    s->d.sph = (const SphGeom *)s->sph;
    s->d.sshade = (const SphShade *)s->sshade;
    s->d.ovf = (unsigned *)s->ovf;
    s->d.n_nodes = D->n_nodes;
    s->d.n_tris = D->n_tris;
    s->d.n_sph = D->n_spheres;
    s->d.ovf_depth = ovf_depth;
    s->d.n_threads = n_threads;
    *out = s;
    return RT_OK;
}

int rt_free_scene(rt_scene *s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->dev);
    void *ptrs[] = {s->nodes, s->first, s->cull, s->tris, s->tshade, s->sph, s->sshade, s->ovf, s->d_counter, s->d_stats, s->d_tiles};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->ev0) (void)hipEventDestroy(s->ev0);
    if (s->ev1) (void)hipEventDestroy(s->ev1);
    delete s;
    return RT_OK;
}

int rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_tile *tiles, int32_t n_tiles, const rt_params *p,
                           double *d_rgb, int32_t *d_prim_id, void *stream) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (!d_rgb) return fail(RT_E_INVALID, "d_rgb is NULL");
    return launch_render(s, cam, tiles, n_tiles, p, d_rgb, d_prim_id, (hipStream_t)stream);
}

int rt_get_stats(rt_scene *s, rt_stats *stats) {
    if (!s || !stats) return fail(RT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(s->dev));
    HIP_TRY(hipStreamSynchronize(s->last_stream));
    unsigned long long h[N_STATS];
    HIP_TRY(hipMemcpy(h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(s->d_stats, 0, sizeof(h)));
    stats->rays = h[0]; stats->node_visits = h[1]; stats->tri_tests = h[2]; stats->sph_tests = h[3]; stats->tri_shades = h[4];
    stats->tri_tests_done = h[5]; stats->cull_tests = h[6];
    return RT_OK;
}

int rt_last_kernel_ms(rt_scene *s, float *ms) {
    if (!s || !ms) return fail(RT_E_INVALID, "NULL argument");
    if (!s->timed) return fail(RT_E_INVALID, "no render has been launched on this scene");
    HIP_TRY(hipSetDevice(s->dev));
    HIP_TRY(hipEventSynchronize(s->ev1));
    HIP_TRY(hipEventElapsedTime(ms, s->ev0, s->ev1));
    return RT_OK;
}

int rt_render_tile(rt_scene *s, const rt_camera *cam, const rt_tile *tile, const rt_params *p, double *rgb_out,
                   int32_t *prim_id_out, rt_stats *stats) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (!tile || !rgb_out) return fail(RT_E_INVALID, "tile or rgb_out is NULL");
    if (tile->rows < 1 || tile->cols < 1) return fail(RT_E_INVALID, "empty tile");
    HIP_TRY(hipSetDevice(s->dev));
    const size_t npix = (size_t)tile->rows * tile->cols;
    double *d_rgb = nullptr;
    int *d_prim = nullptr;
    HIP_TRY(hipMalloc((void **)&d_rgb, npix * 3 * sizeof(double)));
    hipError_t e = hipMalloc((void **)&d_prim, npix * sizeof(int));
    if (e != hipSuccess) { (void)hipFree(d_rgb); return fail(RT_E_DEVICE, "hipMalloc: %s", hipGetErrorString(e)); }
    int rc = RT_OK;
    if (hipMemset(s->d_stats, 0, N_STATS * sizeof(unsigned long long)) != hipSuccess) rc = fail(RT_E_DEVICE, "hipMemset failed");
    if (!rc) rc = launch_render(s, cam, tile, 1, p, d_rgb, d_prim, nullptr);
    if (!rc) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(rgb_out, d_rgb, npix * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && prim_id_out) e = hipMemcpy(prim_id_out, d_prim, npix * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_E_DEVICE, "render failed: %s", hipGetErrorString(e));
    }
    (void)hipFree(d_rgb);
    (void)hipFree(d_prim);
    if (!rc && stats) rc = rt_get_stats(s, stats);
    return rc;
}

int rt_trace_rays(rt_scene *s, int64_t n, const double *rays, int32_t *tri_id, double *tri_t, int32_t *sph_id, double *sph_t,
                  rt_stats *stats) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (n < 0 || (n > 0 && (!rays || !tri_id || !tri_t || !sph_id || !sph_t))) return fail(RT_E_INVALID, "bad ray buffers");
    HIP_TRY(hipSetDevice(s->dev));
    if (n == 0) {
        if (stats) memset(stats, 0, sizeof(*stats));
        return RT_OK;
    }
    double *d_rays = nullptr, *d_tt = nullptr, *d_st = nullptr;
    int *d_ti = nullptr, *d_si = nullptr;
    hipError_t e = hipMalloc((void **)&d_rays, (size_t)n * 6 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_tt, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_st, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&d_ti, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void **)&d_si, (size_t)n * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_rays, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(s->d_stats, 0, N_STATS * sizeof(unsigned long long));
    if (e == hipSuccess) {
        const long long want = (n + BLOCK - 1) / BLOCK;
        const int grid = (int)std::min<long long>(s->grid, want); // stay within the overflow stack's thread count
        hipLaunchKernelGGL(k_trace, dim3(grid), dim3(BLOCK), 0, nullptr, s->d, (long long)n, (const double *)d_rays, d_ti, d_tt,
                           d_si, d_st, s->d_stats);
        e = hipGetLastError();
        s->last_stream = nullptr;
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(tri_id, d_ti, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tri_t, d_tt, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sph_id, d_si, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sph_t, d_st, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(d_rays); (void)hipFree(d_tt); (void)hipFree(d_st); (void)hipFree(d_ti); (void)hipFree(d_si);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "rt_trace_rays: %s", hipGetErrorString(e));
    if (stats) return rt_get_stats(s, stats);
    return RT_OK;
}

int rt_debug_fp64(int dev, int op, int64_t n, const double *a, const double *b, double *out) {
    if (n < 0 || op < 0 || op > 1 || (n > 0 && (!a || !out || (op == 0 && !b)))) return fail(RT_E_INVALID, "bad arguments");
    if (n == 0) return RT_OK;
    HIP_TRY(hipSetDevice(dev));
    double *da = nullptr, *db = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc((void **)&da, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&db, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&dout, (size_t)n * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(da, a, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(db, op == 0 ? b : a, (size_t)n * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_fp64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, (long long)n, (const double *)da,
                           (const double *)db, dout);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, dout, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(db); (void)hipFree(dout);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "rt_debug_fp64: %s", hipGetErrorString(e));
    return RT_OK;
}

} // extern "C"
