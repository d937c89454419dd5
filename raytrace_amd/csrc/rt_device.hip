// rt_device.hip -- MI355X (gfx950) implementation of the RayTrace per-pixel path behind the
// C ABI of include/rt_hip.h.
//
// What runs on the GPU (reference lines it restates):
//   camera ray + tent filter + 2x2 sub-pixel accumulate   raytrace.cpp:196-205
//   radiance(), made iterative (depth-first, refraction first)   raytrace.cpp:41-95
//   KdTree::check / check_node, made iterative with a per-lane stack   kdtree.cpp:98-142
//   KdTreeNode::check_aabb   kdtree.h:57-76
//   Polygon::intersect       polygon.h:37-60
//   Polygon::getcol with GetImageData (textured triangles)   polygon.h:30-35, 73-89
//   Sphere::intersect / sphere_intersect   sphere.h:13-21, sphere.cpp:3-14
// and, with no counterpart in the reference, explicit light sampling (rt_params.light_mode 1).
//
// Numerics: all arithmetic that decides a result is FP64 in the reference's operation order; this
// file must be compiled with -ffp-contract=off so no multiply-add is fused (the CPU path has no
// FMA), and double divide / sqrt are the correctly rounded expansions.  Primary-hit ids and
// distances are therefore bit-identical to the CPU path; colours differ only through sin/cos
// (last-ulp) and through summing emission along the path front-to-back instead of back-to-front.
// The conservative box tests that only decide what is SKIPPED may fuse and use single-precision
// boxes rounded outward.
//
// Layout in HBM (built by rt_upload_scene): one pool of traversal records addressed in 16-byte
// granules (NodeRec, QuadRec, TriRec below), TriShade 80 B per triangle (nc, col, lig, type, global
// id; touched only on the chosen hit), SphGeom 32 B (centre, radius), SphShade 64 B (lig, col,
// type), and per ray slot a column of 8-byte words (direction, best hit, shading state).
//
// Execution: a persistent grid of one-wavefront workgroups.  Each wavefront owns 152 ray slots (140
// in the build for deep trees, see KernelCfg; one sub-pixel's current path each); slots take sub-pixels from an atomic counter in 8x8-block
// order, a pixel's four always in one wavefront, and the one that finishes last adds the four
// shares in the reference's order.  See k_main for the scheduling.
//
// Own lists.  The reference tests EVERY triangle of a visited node's own list, and its
// median-split tree leaves thousands of straddling triangles on the top nodes.  Testing them all
// is not needed to get the same answer: check_node's result is "smallest t > eps, first in list
// order among equals" plus "does any t < 1e10 exist", and both only depend on the triangles that
// Polygon::intersect accepts.  So each list of two or more triangles gets a bounding-box hierarchy, built at
// upload, whose boxes are inflated by a margin that provably exceeds how far outside a triangle
// the floating-point Moller-Trumbore test can still accept a ray (see ray_margin_factor()); subtrees
// whose inflated box the ray misses are skipped, the surviving triangles go through the unchanged
// exact test, and ties are broken by the stored list rank.  Results are identical by construction.
// The hierarchy's leaves are single triangles or groups of two (three in the build for deep trees)
// neighbours behind one box, which one step tests: a step's fixed cost is what limits the kernel.
// In the build for deep trees, once a wavefront has only a handful of slots left, all its lanes scan ONE
// slot's list together (k_main, "the wavefront scans ..."): the scan is an order-free reduction, and a ray
// through a fan of thousands of slivers would otherwise walk it alone, one step after the other.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "rt_hip.h"

namespace {

#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 2 // wavefronts per SIMD the register allocation is bounded for (LDS decides the real number)
#endif

constexpr double EPS = 1e-5;  // point.h:7
constexpr int WAVE = 64;          // threads per workgroup: one wavefront
// Two builds of the kernel: ray slots per workgroup (64 to 256) and the newest traversal-stack levels kept in LDS (a
// ring, a power of two; older levels live in HBM).  LDS is handed out in pieces of 1280 bytes and a slot takes 68 bytes
// plus 4 per ring level: 152 slots with a 4-level ring are the most that leave room for twelve wavefronts per CU, 140
// with an 8-level ring for eleven.  With equally many wavefronts more slots mean fuller steps; a deep tree moves too
// many levels through HBM with the short ring.  RT_SLOTS / RT_LDS_STACK make both builds the same (test builds).
// (Workgroups of two wavefronts could share 21 pieces, 159 slots each: 0.92 instead of 0.90 of the lanes busy per step,
// but the same kernel runs 2 % slower in workgroups of two at equal slot count -- measured, no gain; RT_WPG keeps the option.)
// group: the most triangles a culling hierarchy keeps behind one box (2 to 4); the triangle step tests a group in one go.
// Every further triangle makes that step longer for the whole wavefront: scenes of few large triangles (shallow trees)
// do best with pairs, meshes of millions of small ones (deep trees, long own lists) with more.
struct KernelCfg { int slots, ring, wpg, group; }; // wpg: wavefronts per workgroup (each with its own slots; they share nothing)
#ifndef RT_GROUP_SHALLOW
#define RT_GROUP_SHALLOW 2
#endif
#ifndef RT_GROUP_DEEP
#define RT_GROUP_DEEP 3
#endif
#if defined(RT_SLOTS) || defined(RT_LDS_STACK)
#ifndef RT_SLOTS
#define RT_SLOTS 140
#endif
#ifndef RT_LDS_STACK
#define RT_LDS_STACK 8
#endif
#ifndef RT_WPG
#define RT_WPG 1
#endif
constexpr KernelCfg CFG_SHALLOW = {RT_SLOTS, RT_LDS_STACK, RT_WPG, RT_GROUP_SHALLOW}, CFG_DEEP = {RT_SLOTS, RT_LDS_STACK, RT_WPG, RT_GROUP_DEEP};
#else
constexpr KernelCfg CFG_SHALLOW = {152, 4, 1, RT_GROUP_SHALLOW}, CFG_DEEP = {140, 8, 1, RT_GROUP_DEEP};
#endif
constexpr int DEEP_TREE = 28; // tree depth from which the 8-level ring is used
// a wavefront's slots in the per-slot arrays in HBM: whole 128-byte lines, so that no two wavefronts write one line
constexpr int slot_stride(int slots) { return (slots + 31) & ~31; }
constexpr int MAX_SPLIT = 4;  // pending glass branches per path
#ifdef RT_PROFILE
constexpr int N_STATS = 28; // 11..20: clock ticks in vote, the four bodies, tail; steps per kind; 21..24: triangle step details; 25..27: ticks from a body's start to the arrival of its loads (node, cull, tri)
#elif defined(RT_EVENTS)
constexpr int N_EVENTS = 22;
constexpr int N_STATS = 11 + 2 * N_EVENTS; // 11..: per event, the steps in which some lane met it, and the lanes that did
#elif defined(RT_DRAIN) // when the wavefronts start, find the work counter dry, and end (100 MHz clock): how a launch ramps up and drains
constexpr int DRAIN_BINS = 300; // wavefronts by the time from dry to end, bins of 10 us
constexpr int N_STATS = 11 + 8 + DRAIN_BINS + 8 + 6; // (+ 6: steps of each kind run for a wavefront's last slot alone; list scans taken over by the wavefront, their batches) // 11 min start, 12 max start, 13 min dry, 14 sum dry, 15 max dry, 16 sum end, 17 max end, 18 wavefronts; after the bins: steps and lanes per kind once the counter is dry
#else
constexpr int N_STATS = 11;
#endif
#ifndef RT_SMALL_LIST
#define RT_SMALL_LIST 1
#endif
// The wavefront-wide list scan (see k_main, "one slot left"): 0 never, 1 in the build for deep trees, 2 in both builds.
#ifndef RT_COOP
#define RT_COOP 1
#endif
constexpr int SMALL_LIST = RT_SMALL_LIST; // own lists up to this length are tested directly, longer ones get a hierarchy
constexpr int SPH_HIER_MIN = 16; // scenes with at least this many spheres get a hierarchy over them

// Every traversal record lives in ONE pool addressed in 16-byte granules, and a work item carries
// the granule of its record:
//   node i        granule 4*i             NodeRec (64 B)
//   its own list  granule g_list + 4*i    QuadRec (64 B): the list's start (+ an 8-byte ListHdr in its own array)
//   its boxes     granule 2*g_list + 3*i  KidRec (48 B): children and own list, for the node's step
//   triangle      5 granules              TriRec (80 B)
//   sphere        2 granules              centre, radius (leaves of the sphere hierarchy)
//   cull record   4 granules              QuadRec (64 B)
struct __attribute__((aligned(16))) NodeRec { // one kd-tree node (KdTree, kdtree.h:96-116)
    double mi[3], ma[3];
    double split;
    unsigned right; // granule of the right child, 0 = none (the root is never a child)
    unsigned info;  // bits 0-1 axis, bit 2 has-left, bits 3.. own-list length
};
static_assert(sizeof(NodeRec) == 64, "NodeRec must be 64 bytes");

// Boxes of up to four children and the children as work items (0 = no child).  A box plane is
// stored as one byte: plane = base[axis] + byte * step[axis], evaluated in single precision with
// one fused multiply-add; the builder picks the bytes so that this very expression is <= the
// exact lower bound and >= the exact upper bound.  As a cull record it is an inner node of a long
// own list's culling hierarchy; as a node's list record it is the hierarchy's root (long list) or
// carries the chained item of the list's first triangle in item[0] (short list).
struct __attribute__((aligned(16))) QuadRec {
    float base[3], step[3];
    unsigned char q[4][3][2]; // [child][axis][lower, upper]
    unsigned item[4];
};
static_assert(sizeof(QuadRec) == 64, "QuadRec must be 64 bytes");
// Boxes of a node's two children and of its own list, quantised like a culling record's (single-precision
// base, step kept as the upper 16 bits of a float, one byte per plane), so that the node's step can skip a
// child whose check_aabb is certain to fail and an own list no triangle of which the ray can meet.
// Box 0 is the left child, 1 the right child, 2 the own list (with its ListHdr::meta beside it).
struct __attribute__((aligned(16))) KidRec {
    float base[3];
    unsigned short step16[3];
    unsigned short flags; // KID_*
    unsigned char q[3][3][2];
    unsigned char pad2[2];
    unsigned meta;        // the own list's margin coefficients (ListHdr::meta)
    unsigned direct;      // the own list as ONE triangle, chain or group item, if it is no longer than a group; else 0
};
static_assert(sizeof(KidRec) == 48, "KidRec is three granules");
constexpr unsigned KID_NOKIDS = 1, KID_LIST = 2, KID_EMPTY = 4; // children's boxes unusable; list box usable; own list empty

struct ListHdr { // per node, beside its list record: what the "near child finished" step needs
    unsigned right; // granule of the node's right child (0 = none), for the far-child step
    unsigned meta;  // LIST_SHORT | length, or a long list's margin coefficients: bits 8-30 the upper bits of the (float)
                    // working coefficient K, bits 0-7 an exponent r; K * 2^r bounds the coefficient of every triangle
};
constexpr unsigned LIST_SHORT = 0x80000000u;

struct __attribute__((aligned(16))) TriRec {
    double p0[3], e1[3], e2[3];
    int pos;  // index of the triangle's TriShade; TRI_LAST set on the last triangle of an own list
    int rank; // position in the reference's list order (decides between equal distances)
};
static_assert(sizeof(TriRec) == 80, "TriRec must be 80 bytes");

struct __attribute__((aligned(16))) TriShade {
    double nc[3], col[3], lig[3];
    int type;
    int gid; // global triangle id
};
static_assert(sizeof(TriShade) == 80, "TriShade must be 80 bytes");

// A textured triangle's texture coordinates and image (Polygon::ts, tex, sizex, sizey; polygon.h:13-15); TriShade::type
// carries its index + 1 above the two type bits.  `off` is the image's first byte in the texture block.
struct __attribute__((aligned(16))) TexTri {
    double ts[6];
    unsigned long long off;
    int sizex, sizey;
};
static_assert(sizeof(TexTri) == 64, "TexTri must be 64 bytes");
// The texture block in HBM: pow(b / 256, 2.2) for the 256 byte values (evaluated by the host's libm, which is what
// the CPU path calls, polygon.h:87), then the TexTri records, then the images.
constexpr size_t TEX_OFF_TRI = 256 * sizeof(double);

struct __attribute__((aligned(16))) SphGeom { double px, py, pz, rad; };
struct __attribute__((aligned(16))) SphShade { double lig[3], col[3]; int type, pad0; double pad1; };
static_assert(sizeof(SphShade) == 64, "SphShade must be 64 bytes");

// What the traversal steps need of the scene: passed to the kernel by value, i.e. held in scalar registers for the
// kernel's whole life.  What only the shade step needs is in ColdArgs.
struct DevScene {
    const char *pool; // nodes + lists, cull records, triangles
    const uint2 *tlist; // per triangle (device position): first list position of its own list, its own list position
    unsigned *ovf;  // overflow stack [ovf_depth][n_slots]
    double *lstate; // per-slot state kept in HBM [LS_WORDS][n_slots]
    unsigned g_list; // granule of node 0's list record
    unsigned g_tri;  // granule of the first triangle record (five granules each, in list order)
    unsigned g_sphc; // granule of the sphere hierarchy's first cull record (the triangles' records lie below it)
    unsigned g_sph;  // granule of sphere 0's copy in the pool (two granules per sphere); tri items from here on are spheres
    unsigned sph_root; // item of the sphere hierarchy's root record, 0 = spheres are scanned in the shade step
    float sph_k, sph_c; // inflation of the sphere boxes: sph_k * (bound on |centre - origin|)^2 + sph_c
    double sph_cen[3], sph_h; // centre and summed half extents of the box around all spheres
    const ListHdr *lhdr; // per node
    float fslack;    // part of the single-precision slab test's safety margin that depends on the scene only
    double padc;   // constant part of the node-box padding: 2.5e-5 + 4e-12 * (largest sum of |box coordinates| of a node)
    float padcf;   // the same in single precision, rounded up
    int n_sph;
    int ovf_depth;
    int n_slots;
    int coop_reserve; // room the wavefront-wide list scan keeps free in its work list (4 x the deepest culling hierarchy + 8); 0: never scans that way
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

// Everything only the shade step reads (camera, integrator controls, shading records, the launch's work list and
// output buffers).  It lives in device memory and the kernel gets a pointer: passed by value these ~90 words would
// sit in scalar registers through the traversal steps, which already spill theirs.
struct ColdArgs {
    DevCamera cam;
    rt_params P;
    const TriShade *tshade;
    const SphGeom *sph;
    const SphShade *sshade;
    const unsigned char *tex; // texture block (see TexTri), NULL without textures
    const int *lights;        // indices of the emissive spheres, ascending (explicit light sampling)
    int n_lights, pad_l;
    double cen[3]; // centre of the scene's vertex box
    double reach;  // its summed half extents + the scene's longest triangle edge
    double absn;   // |centre|_1 + summed half extents
    const DevTile *tiles;
    const int *blk_tile;
    unsigned *counter;
    double *rgb;
    int *prim;
    double *subbuf;
    unsigned *subcnt;
    const double *rays;
    int *tri_id;
    double *tri_t;
    int *sph_id;
    double *sph_t;
    unsigned n_work;
    int n_tiles;
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

// A pointer read from memory has no known address space, and the compiler then uses generic-address loads and stores
// for it (which also count as LDS traffic and order against every LDS access).  All such pointers here point to
// device memory: tell the compiler so.
template <class T>
__device__ __forceinline__ T *as_global(T *p) {
#ifdef __HIP_DEVICE_COMPILE__
    typedef __attribute__((address_space(1))) T *global_ptr;
    // (the value goes through an opaque no-op so that the cast into the global address space is not folded away)
    unsigned long long v = (unsigned long long)p; // (wavefront-uniform wherever this is used: it stays in scalar registers)
    asm("" : "+s"(v));
    return (T *)(global_ptr)v;
#else
    return p;
#endif
}

// Addresses as "wavefront-uniform base + 32-bit lane offset": a load or store takes that form as it is, where a 64-bit
// address per lane costs a shift and two adds of vector arithmetic for every record and every state word (+1.2 % at 4K).
// The pool holds at most 2^28 granules of 16 bytes (rt_prepare_scene refuses more), so a record's byte offset fits 32 bits.
__device__ __forceinline__ const char *pool_at(const char *pool, unsigned granule) { return pool + (unsigned)(granule << 4); }
// A slot's column of the per-slot state in HBM ([word][slot], 8-byte words) for the traversal steps: lw[w * n_slots] is
// word w of the slot, as with a plain pointer to the slot's first word.  All columns together stay below 4 GiB
// (rt_upload_prepared checks).  The shade step keeps the plain pointer: with its two dozen different words the uniform
// bases cost more scalar registers (which the kernel already spills) than the lane arithmetic they save.
struct SlotWords {
    char *base;   // the array (uniform)
    unsigned off; // byte offset of the slot's word in a column
    __device__ __forceinline__ double &operator[](size_t elems) const { return *(double *)((base + elems * 8) + off); }
};

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

// kdtree.h:57-76.  The three axes are tried one after the other, each behind a branch, exactly as
// the reference's loop returns at the first axis that accepts: later divisions are skipped and the
// temporaries of one axis are dead before the next starts.
__device__ __forceinline__ bool aabb_hit(const double (&mi)[3], const double (&ma)[3], const double (&s)[3],
                                         const double (&v)[3]) {
    if (mi[0] < s[0] + EPS && s[0] < ma[0] + EPS && mi[1] < s[1] + EPS && s[1] < ma[1] + EPS && mi[2] < s[2] + EPS &&
        s[2] < ma[2] + EPS)
        return true;
#define RT_AABB_AXIS(i, j, k)                                                                                   \
    if (fabs(v[i]) > EPS) {                                                                                     \
        const double t = ((v[i] > 0 ? mi[i] : ma[i]) - s[i]) / v[i];                                            \
        if (t > EPS) {                                                                                          \
            const double pj = s[j] + v[j] * t, pk = s[k] + v[k] * t;                                            \
            if (mi[j] < pj + EPS && pj < ma[j] + EPS && mi[k] < pk + EPS && pk < ma[k] + EPS) return true;      \
        }                                                                                                       \
    }
    RT_AABB_AXIS(0, 1, 2)
    RT_AABB_AXIS(1, 2, 0)
    RT_AABB_AXIS(2, 0, 1)
#undef RT_AABB_AXIS
    return false;
}

// polygon.h:37-60 with E1, E2 precomputed by the same subtractions; returns 1e20 on a miss
__device__ __forceinline__ double tri_hit(d3 p0, d3 E1, d3 E2, d3 o, d3 d) {
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
// (1e-9 * L^2 + 1e-10) * B with B >= |T| + L for every triangle of the scene: B = |o - C|_1 + H + Lg
// (C, H the centre and the summed half extents of the scene's vertex box, Lg the longest edge of
// the scene), i.e. more than twice that bound; the boxes are then inflated by twice the margin
// again, which also swallows the few-ulp error of the slab arithmetic itself (a term proportional
// to the absolute coordinates covers scenes far from the coordinate origin).  The bound is for a
// unit direction; the rounding errors grow with |d| while the det cut does not, so the margin is
// multiplied by max(1, |d|).  Per list only K = 2 * (1e-9 * L^2 + 1e-10) is stored, per ray the
// factor G = max(1, |d|) * B + 1e-5 * (|o|_1 + |C|_1 + H); a list's inflation is K * G.
__device__ __forceinline__ float ray_margin_factor(const double (&cen)[3], double reach, double absn, double ox, double oy, double oz,
                                                   double dlen) {
    const double b = fabs(ox - cen[0]) + fabs(oy - cen[1]) + fabs(oz - cen[2]) + reach;
    const double g = (dlen > 1.0 ? dlen : 1.0) * b + 1e-5 * (fabs(ox) + fabs(oy) + fabs(oz) + absn);
    return __double2float_ru(g * 1.000001);
}

// Conservative ray (t >= 0) versus inflated box: never false for a ray the exact triangle test
// could accept; a false positive only costs an extra exact test.  `inv` holds 1/direction, or
// 1e300 for a (nearly) zero component: the products then overflow to -inf/+inf on the two sides
// of the origin, which makes that axis accept exactly the origins inside its slab.
__device__ __forceinline__ bool slab_hit(const double (&lo0)[3], const double (&hi0)[3], double m2, const double (&s)[3],
                                         const double (&inv)[3]) {
    double tmin = 0.0, tmax = 1e300;
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const double t1 = ((lo0[i] - m2) - s[i]) * inv[i], t2 = ((hi0[i] + m2) - s[i]) * inv[i];
        tmin = fmax(tmin, fmin(t1, t2));
        tmax = fmin(tmax, fmax(t1, t2));
    }
    return tmin <= tmax;
}

// sphere.cpp:3-14 over sphere.h:13-21: scan from the last sphere down, strict <
__device__ __forceinline__ void trace_spheres(const SphGeom *sph, int n_sph, d3 o, d3 d, double &ts, int &ids) {
    ts = 1e20;
    ids = -1;
    for (int i = n_sph; i--;) {
        const SphGeom g = sph[i];
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
}

// The u, v that Polygon::intersect leaves behind for a ray that hits (polygon.h:37-60; raytrace.cpp:61 calls it a
// second time just for them), with E1, E2 precomputed by the same subtractions.
__device__ __forceinline__ void tri_uv(d3 p0, d3 E1, d3 E2, d3 o, d3 d, double &u, double &v) {
    const d3 P = cross(d, E2);
    double det = dot(E1, P);
    d3 Tv;
    if (det > 0) Tv = o - p0;
    else { Tv = p0 - o; det = -det; }
    u = dot(Tv, P);
    const d3 Q = cross(Tv, E1);
    v = dot(d, Q);
    const double fInvDet = 1.0 / det;
    u *= fInvDet;
    v *= fInvDet;
}

// Polygon::getcol with an image (polygon.h:30-35,73-89).  A coordinate that does not fit an int, or an index outside
// the image (the reference reads out of bounds there), gives the byte 0.
__device__ __forceinline__ d3 tex_color(const unsigned char *blk, const TexTri &x, double u, double v) {
    const double sx = ((x.ts[2] - x.ts[0]) * u + (x.ts[4] - x.ts[0]) * v) + x.ts[0];
    const double sy = ((x.ts[3] - x.ts[1]) * u + (x.ts[5] - x.ts[1]) * v) + x.ts[1];
    const double fx = sx * x.sizex + 0.5, fy = sy * x.sizey + 0.5;
    const double *powtab = (const double *)blk;
    long long idx = -1;
    if (fabs(fx) < 2147483648.0 && fabs(fy) < 2147483648.0)
        idx = (long long)((int)fx % x.sizex) * 3 * x.sizey + 3 * (long long)((int)fy % x.sizey);
    if (idx < 0 || idx + 2 >= 3ll * x.sizex * x.sizey) { const double z = powtab[0]; return mk(z, z, z); }
    const unsigned char *px = blk + x.off + idx;
    return mk(powtab[px[2]], powtab[px[1]], powtab[px[0]]);
}

__device__ __forceinline__ double clamp01(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; } // raytrace.cpp:28-31

// sin and cos of x in [0, 2*pi) (the only argument range radiance() needs, raytrace.cpp:79-81):
// quadrant reduction with a two-part pi/2 and the classic minimax kernels on [-pi/4, pi/4];
// within 1.2e-16 of libm over the whole range, with a fraction of the registers of the general
// routine.
__device__ __forceinline__ void sincos_2pi(double x, double &sn, double &cs) {
    const int k = (int)(x * 0.63661977236758134308 + 0.5);
    const double y = (x - k * 1.57079632673412561417e+00) - k * 6.07710050650619224932e-11;
    const double z = y * y;
    const double rs = 8.33333333332248946124e-03 +
                      z * (-1.98412698298579493134e-04 +
                           z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double ks = y + (z * y) * (-1.66666666666666324348e-01 + z * rs);
    const double rc = z * (4.16666666666666019037e-02 +
                           z * (-1.38888888888741095749e-03 +
                                z * (2.48015872894767294178e-05 +
                                     z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double kc = 1.0 - (0.5 * z - z * rc);
    const int q = k & 3;
    sn = q == 0 ? ks : q == 1 ? kc : q == 2 ? -ks : -kc;
    cs = q == 0 ? kc : q == 1 ? -ks : q == 2 ? -kc : ks;
}

// A traversal work item, 32 bits; the per-ray stack holds these.  The low bits are the granule
// of the item's record in the pool.
//   node item   : granule (29 bits), kind bits 00            enter the node: KdTree::check
//   cull item   : IT_CULL | granule                          one inner record of a list's hierarchy
//   tri item    : IT_TRI | chain << 28 | granule (28 bits)   one exact triangle test; with `chain` the next
//                 triangle of the list follows directly, up to the one marked last (short lists)
//   group item  : IT_GROUP | (n - 2) << 26 | position (26 bits)  n = 2 to 4 triangles that follow each other in the pool
//                 from that list position on (a small range of a list's hierarchy, behind ONE box), tested in one step:
//                 a further test costs a fifth of a step of its own
//   after item  : IT_AFTER | flag << 30 | far << 28 | node granule (28 bits); far: 0 none, 1 left, 2 right child,
//                 3 = a far child whose box the ray is certain to miss (visited in name only)
//                 the node's near child is done: own list, then the far child; `flag` is the
//                 enclosing frame's "has a hit" so far.  Its record is the node's ListRec.
constexpr unsigned IT_CULL = 1u << 29, IT_TRI = 2u << 29, IT_CHAIN = 1u << 28, IT_AFTER = 1u << 31, IDX29 = 0x1FFFFFFFu,
                   IDX28 = 0x0FFFFFFFu;
constexpr unsigned IT_GROUP = 3u << 29, GROUP_POS = 0x03FFFFFFu;
constexpr unsigned TRI_LAST = 0x80000000u; // in TriRec::pos: last triangle of its own list
constexpr unsigned IT_END = 0xFFFFFFFFu; // "the stack is empty": the traversal is complete
// A node-kind item no node can have: "the next stack levels are in HBM".  The step that runs it fetches
// them together with the other lanes' node records instead of making the popping step wait for HBM.
constexpr unsigned IT_REFILL = 0x1FFFFFFFu;
// levels fetched at once by a refill step
#ifdef RT_REFILL
constexpr int refill_levels(int) { return RT_REFILL; }
#else
constexpr int refill_levels(int ring) { return ring >= 8 ? 4 : ring >= 4 ? 3 : 1; }
#endif
enum { K_NODE = 0, K_CULL = 1, K_TRI = 2, K_SHADE = 3, K_DONE = 4 };
enum { ST_RAYS = 0, ST_VISITS, ST_TRIS, ST_SPHS, ST_SHADES, ST_DONE, ST_CULLS, ST_ITERS, ST_LANES, ST_NFETCH, ST_SPHDONE };
// Words of a ray slot's state kept in HBM (8 bytes each, [word][slot] so lanes coalesce): what
// only the shading step touches, plus the ray direction (read together with a record).
//   LS_PST  low: depth (bits 0-19), pending branches (20-22), sub-pixel (23-24), into (25),
//           primary (26), a traversal is waiting to be shaded (27), an emissive sphere at the hit adds its
//           emission (28), the ray is a shadow ray (29), only sphere LS_COL adds its emission, times the
//           number of lights (30); high: sample index
//   LS_OUT  low: index in the packed output (render) / ray index (trace); high: pixel x << 16 | y
//   LS_PEND parked rays (a glass hit's reflection; with light sampling, the bounce ray while the shadow ray is traced),
//           PEND_WORDS each: origin, direction, throughput, depth | into << 32 | no sphere emission << 33
//   LS_COL  the sphere a shadow ray was aimed at / the light a bounce ray stands in for (bit 30 above)
//   LS_V    ray direction;  LS_BT best distance so far
//   LS_HA   low: best triangle (TriShade index)
//   LS_ST   nearest sphere's distance, LS_SI its index (-1 none, -2: not searched through the hierarchy)
enum { LS_ACC = 0, LS_THR = 3, LS_RSUM = 6, LS_COL = 9, LS_RNG = 12, LS_PST = 13, LS_OUT = 14, LS_V = 15, LS_BT = 18, LS_HA = 19,
       LS_ST = 20, LS_SI = 21, LS_PEND = 22 };
constexpr int PEND_WORDS = 10;
constexpr int LS_WORDS = LS_PEND + PEND_WORDS * (MAX_SPLIT + 1); // (+ 1: a bounce ray waiting for its shadow ray)
// per-slot flags word in LDS: stack depth (bits 0-7), the frame's "has a hit" flag (8), the frame's
// number of far children that are only counted (9-11), number of stack levels that live in HBM only (12-19), "the ray
// starts far outside the scene" (20)
constexpr unsigned ST_SP = 0xFFu, ST_FLAG = 0x100u, ST_FAR = 1u << 20; // (ST_FAR: see sG)
// "the ray is tame" (21): every direction component exceeds 2.5e-5 in magnitude and the node-box padding at its origin is
// below 3e-5 -- what the node step's quick acceptance needs of the ray itself, decided once when the ray starts
constexpr unsigned ST_TAME = 1u << 21;
constexpr int ST_LO = 12;
// Far children that are only counted (their box is certainly missed) when the near subtree returns false need no
// stack entry if nothing else waits for that return: the frame keeps their number (up to seven) in three bits of its flags word,
// an after item or marker saves the enclosing frame's number in bits 25-27, and whatever ends the frame (such an
// item coming off the stack, or the end of the traversal) adds the number to the visits if the frame found nothing.
#ifndef RT_NMAX
#define RT_NMAX 7 // (a test build uses 1, so that the marker a frame needs beyond that is used often)
#endif
constexpr unsigned ST_N = 0xE00u, N_MAX = RT_NMAX;
static_assert(N_MAX >= 1 && N_MAX <= 7, "three bits");
constexpr int ST_NSH = 9, IT_NSH = 25;
// (after items and markers carry a node granule / 2 in bits 1-24; bit 0 marks a marker)
__device__ __forceinline__ unsigned after_granule(unsigned x) { return (x & 0x01FFFFFEu) << 1; }
// the kind of step a node, cull, triangle (or pair) item needs
__device__ __forceinline__ unsigned item_kind(unsigned x) {
    const unsigned k = (x >> 29) & 3u;
    return k < 2u ? k : 2u; // (K_NODE, K_CULL, K_TRI)
}

// The whole path -- camera ray, KdTree::check without recursion, check_node, sphere scan, radiance()
// as a loop -- is cut into steps of four kinds:
//     node   enter a kd-tree node: check_aabb, choose near / far child          (kdtree.cpp:111-141)
//     cull   four child boxes of an own list's culling hierarchy; the first such step of a list
//            also does the bookkeeping of "the node's near child has returned"   (kdtree.cpp:122-125)
//     tri    one exact triangle test, or those of a group item                  (kdtree.cpp:101-107)
//     shade  sphere scan, material, next ray / next path / next pixel           (raytrace.cpp:41-95,196-205)
// A wavefront owns SLOTS ray slots (152, or 140 in the build for deep trees).  A slot is a complete path tracer
// state: its ray origin and inverse direction, its current item, flags and the top of its
// traversal stack live in LDS, the rest (direction, best hit, shading state) in HBM, and NOTHING
// about a ray is held in registers between steps.  Each iteration the wavefront picks the kind
// of step most of its slots wait for, the (up to 64) topmost slots on that kind's stack are handed to the
// lanes, every such lane runs the step for its slot (state from LDS, record from the
// pool, both loads issued together) and writes back the few words that changed.  Because the 64
// rays of a packet diverge completely after the first bounce, this keeps most lanes active per
// executed instruction, and the step bodies are straight-line code without loop-carried registers.
//
// TRACE = false: slots pull sub-pixels (8x8 blocks of pixels, a pixel's four in one wavefront) from an atomic
// counter and run each sub-pixel's samps paths in the reference's order.  TRACE = true: slots pull explicit rays
// and report the nearest triangle / sphere (raytrace.cpp:46-47).
// a float kept in 16 bits: sign, exponent and the top 7 mantissa bits, the magnitude rounded up
__device__ __forceinline__ unsigned short half_up_put(float x) { return (unsigned short)((__float_as_uint(x) + 0xFFFFu) >> 16); }
__device__ __forceinline__ float half_up_get(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// a float >= 0 kept in 8 bits: the next power of two (the margins it is used for are far below the boxes' sizes)
__device__ __forceinline__ unsigned char exp_up_put(float x) { return (unsigned char)((__float_as_uint(x) + 0x7FFFFFu) >> 23); }
__device__ __forceinline__ float exp_up_get(unsigned char e) { return __uint_as_float((unsigned)e << 23); }
__device__ __forceinline__ float cull_inv(float x) { return fminf(fmaxf(x, -1e30f), 1e30f); }

#ifdef RT_DRAIN
// per wavefront of the last launch: time from dry to end (100 MHz ticks), steps of the four kinds after dry, steps for its last
// slot alone, list scans taken over by the wavefront and their batches
constexpr int DRAIN_WAVES = 4096;
static __device__ unsigned dr_waves[DRAIN_WAVES][8];
#endif

// SPH = true: the scene has a sphere hierarchy (a second instance, so that scenes without one do not pay for it).
// SLOTS, LDS_STACK, WPG, GROUP: see KernelCfg.
// FEAT selects what the shade step can do beyond the reference's untextured integrator (instances of their own, like
// SPH): 1 = textured triangles, 2 = those and explicit light sampling (rt_params.light_mode 1).
template <bool TRACE, bool SPH, int FEAT, int SLOTS, int LDS_STACK, int WPG, int GROUP>
#ifndef RT_KALIGN
#define RT_KALIGN 256
#endif
__global__ __launch_bounds__(WAVE * WPG, RT_MIN_WAVES) __attribute__((aligned(RT_KALIGN))) void k_main(DevScene S, const ColdArgs *__restrict__ cold, unsigned long long *gstats) {
    constexpr bool TEX = FEAT >= 1, NEE = !TRACE && FEAT >= 2;
    __shared__ double2 sA[WPG * SLOTS]; // origin x, y
    __shared__ double2 sB[WPG * SLOTS]; // origin z, 1/direction x
    __shared__ double2 sC[WPG * SLOTS]; // 1/direction y, z
    // (both kept as the upper half of a float with the magnitude rounded up: a larger margin only culls less)
    __shared__ unsigned char sM[WPG * SLOTS]; // inflation of the current list's culling boxes
    __shared__ unsigned char sG[WPG * SLOTS]; // the ray's margin factor (ST_FAR in the flags word: beyond what the working coefficients cover)
    __shared__ unsigned short sT[WPG * SLOTS]; // the best distance so far (triangles; the same 16-bit form, so never too small)
    __shared__ unsigned stk[LDS_STACK][WPG * SLOTS];
    // the slots that wait for a step, one stack per kind: slot | flags word << 8; item.  Two stacks share an
    // array and grow towards each other (a slot is in at most one of them): node up from 0 and cull down from
    // SLOTS - 1, tri up from SLOTS and shade down from 2 * SLOTS - 1
    __shared__ uint2 sQ[WPG * 2 * SLOTS];
    constexpr int SLOT_STRIDE = slot_stride(SLOTS), REFILL = refill_levels(LDS_STACK);
    static_assert(SLOTS >= WAVE && SLOTS <= 256, "a slot index is kept in 8 bits");
    static_assert((LDS_STACK & (LDS_STACK - 1)) == 0 && REFILL >= 1 && REFILL - 1 <= LDS_STACK, "the refilled levels but the first go to the ring");
    static_assert(SLOT_STRIDE % 32 == 0 && SLOT_STRIDE >= SLOTS, "a wavefront's 4-byte columns of the HBM state start on a cache line");
    // (a workgroup of WPG wavefronts is WPG independent copies of what follows: each has its own part of every array)
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = WPG > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;
    const int wslot0 = wv * SLOTS; // this wavefront's first slot in the LDS arrays
    const unsigned long long lane_lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    auto qindex = [wslot0](int kind, int i) __attribute__((always_inline)) { return 2 * wslot0 + (kind >> 1) * SLOTS + ((kind & 1) ? SLOTS - 1 - i : i); };
    const size_t lss = (size_t)S.n_slots;
    const unsigned gslot0 = (blockIdx.x * WPG + wv) * SLOT_STRIDE;

    // all slots start as "nothing traced yet, needs work"
    for (int i = lane; i < SLOTS; i += WAVE) {
        S.lstate[LS_PST * lss + gslot0 + i] = 0.0;
        sQ[qindex(K_SHADE, i)] = make_uint2((unsigned)i, 0u);
    }
    int q0 = 0, q1 = 0, q2 = 0, q3 = SLOTS; // entries on the four stacks
    // pieces of work taken from the counter and not handed to a slot yet: first piece, count (in LDS because only
    // the lanes of a shade step see an update)
    // (one lane writes what the others read, so the compiler must not keep a lane's old value: relaxed atomic
    // accesses -- a `volatile` array here is accessed through generic-address instructions, and one of those in
    // flight makes the compiler put a full wait before every LDS read that follows)
    __shared__ unsigned sStashAll[WPG * 2];
    unsigned *const sStash = sStashAll + 2 * wv;
    auto stash_get = [&](int i) __attribute__((always_inline)) { return __hip_atomic_load(&sStash[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto stash_put = [&](int i, unsigned v) __attribute__((always_inline)) { __hip_atomic_store(&sStash[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (lane == 0) { stash_put(0, 0u); stash_put(1, 0u); }
    // work counters: wavefront-uniform 32-bit sums of lane counts in scalar registers, flushed to
    // the 64-bit totals in HBM long before they can overflow; only the own-list lengths need a
    // per-lane sum
    unsigned c_rays = 0, c_shades = 0, c_done = 0, c_iters = 0, c_lanes = 0;
    unsigned c_sphdone = 0, c_scans = 0; // spheres tested one by one; rays that scanned all spheres
    unsigned long long l_tris = 0;
    // counted per lane (an add under the lane's own condition is cheaper than a wavefront-wide count and an add to a counter
    // that every path through the step has to carry): node visits, node records fetched, culling records tested
    unsigned l_vis = 0, l_nf = 0, l_culls = 0;
    auto flush_lanes = [&]() __attribute__((always_inline)) {
        unsigned long long vis = l_vis, nf = l_nf, cu = l_culls;
        for (int off = 32; off > 0; off >>= 1) { vis += __shfl_down(vis, off, 64); nf += __shfl_down(nf, off, 64); cu += __shfl_down(cu, off, 64); }
        if (lane == 0) {
            if (vis) atomicAdd(gstats + ST_VISITS, vis);
            if (nf) atomicAdd(gstats + ST_NFETCH, nf);
            if (cu) atomicAdd(gstats + ST_CULLS, cu);
        }
        l_vis = l_nf = l_culls = 0;
    };
    auto flush = [&]() __attribute__((always_inline)) {
        if (lane == 0) {
            if (c_rays) { atomicAdd(gstats + ST_RAYS, (unsigned long long)c_rays); atomicAdd(gstats + ST_SPHS, (unsigned long long)c_rays * (unsigned)S.n_sph); }
            if (c_shades) atomicAdd(gstats + ST_SHADES, (unsigned long long)c_shades);
            if (c_done) atomicAdd(gstats + ST_DONE, (unsigned long long)c_done);
            atomicAdd(gstats + ST_ITERS, (unsigned long long)c_iters);
            atomicAdd(gstats + ST_LANES, (unsigned long long)c_lanes);
            if (c_sphdone | c_scans) atomicAdd(gstats + ST_SPHDONE, (unsigned long long)c_sphdone + (unsigned long long)c_scans * (unsigned)S.n_sph);
        }
        c_rays = c_shades = c_done = c_iters = c_lanes = 0;
        c_sphdone = c_scans = 0;
    };
#define RT_SCOUNT(c, n) (c) += (unsigned)(n)


#ifdef RT_PROFILE
    unsigned long long p_vote = 0, p_body[4] = {0, 0, 0, 0}, p_tail = 0, p_n[4] = {0, 0, 0, 0}, p_tri[4] = {0, 0, 0, 0}, p_load[3] = {0, 0, 0};
#define RT_TICK(x) const unsigned long long x = __builtin_amdgcn_s_memtime()
#else
#define RT_TICK(x)
#endif
#ifdef RT_EVENTS // how often the loop's side paths run: steps with at least one such lane, and lanes
    unsigned long long p_ev[N_EVENTS] = {}, p_evl[N_EVENTS] = {};
#define RT_EV(k, cond) do { const unsigned long long b_ = __ballot(cond); p_ev[k] += b_ != 0; p_evl[k] += __popcll(b_); } while (0)
#else
#define RT_EV(k, cond)
#endif
#ifdef RT_MARKS // comments in the assembly at the borders of the loop's parts (for reading it; no effect on the code)
#define RT_MARK(name) asm volatile("; RTMARK " name)
#else
#define RT_MARK(name)
#endif
#ifdef RT_DRAIN
    const unsigned long long dr_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long dr_dry = 0;
    unsigned dr_steps[4] = {0, 0, 0, 0}, dr_lanes[4] = {0, 0, 0, 0}, dr_lone[6] = {0, 0, 0, 0, 0, 0};
#endif
    constexpr bool COOP = !SPH && (RT_COOP == 2 || (RT_COOP == 1 && LDS_STACK >= 8));
#ifndef RT_COOP_SLOTS
#define RT_COOP_SLOTS 16
#endif
#ifndef RT_COOP_WIDE
#define RT_COOP_WIDE 2
#endif
    constexpr int COOP_WIDE = RT_COOP_WIDE; // stack levels below the slot's item that must be work of the same list for its scan to count as wide
    constexpr int COOP_SLOTS = RT_COOP_SLOTS; // the most slots that may still be active when the wavefront scans one slot's list together
    // items the wavefront-wide scan takes: single triangles (not the chains of short lists) and groups; inner records
    auto coop_leaf = [](unsigned x) __attribute__((always_inline)) { const unsigned k = x >> 29; return k == 3u || (k == 2u && !(x & IT_CHAIN)); };
    auto coop_inner = [](unsigned x) __attribute__((always_inline)) { return (x >> 29) == 1u; };
    for (;;) {
        RT_TICK(tk0);
        RT_MARK("pick");
        // ---- vote: run the kind most slots wait for; its (up to 64) topmost entries go to the lanes
        asm("" : "+s"(q0), "+s"(q1), "+s"(q2), "+s"(q3)); // keep the counts in scalar registers
        int pick = K_NODE, most = q0;
        if (q1 > most) { most = q1; pick = K_CULL; }
        if (q2 > most) { most = q2; pick = K_TRI; }
        if (q3 > most) { most = q3; pick = K_SHADE; }
        if (most == 0) break; // every slot is done
        int nrun = most < WAVE ? most : WAVE;
        // A handful of slots left, and the next one is scanning an own list's hierarchy: a ray through a fan of thousands of
        // slivers takes thousands of such steps one after the other while the other lanes idle (on the Lucy-class frame a
        // fifth of the launch is a handful of such rays, four to a wavefront: a pixel's sub-pixels).  A list's scan is an
        // order-free reduction -- the nearest accepted triangle, the lower list rank among equals, "was any accepted" --
        // so the wavefront then scans that ONE slot's list together: see the step below.
        bool coop = false;
        if (COOP && S.coop_reserve > 0 && (pick == K_CULL || pick == K_TRI) && q0 + q1 + q2 + q3 <= COOP_SLOTS) {
            // (only a scan that has several items to offer: the slot's item and the COOP_WIDE newest levels of its stack are
            // work of the same list; a lone triangle or record is served faster by an ordinary step, which serves all slots
            // at once.  On this build, Mrays/s with one / two / three levels against no such scan: Lucy-class 4K 1 995 / 1 973 /
            // 1 956 against 1 673; a scene of short lists, the Cornell demo, 3 206 / 3 414 / 3 457 against 3 511)
            const uint2 e0 = sQ[qindex(pick, most - 1)];
            const unsigned it0 = (unsigned)__builtin_amdgcn_readfirstlane((int)e0.y), f0 = (unsigned)__builtin_amdgcn_readfirstlane((int)e0.x);
            const int sl0 = (int)(f0 & 0xFFu) + wslot0, sp0 = (int)((f0 >> 8) & ST_SP), slo0 = (int)((f0 >> (8 + ST_LO)) & 0xFFu);
            if ((coop_leaf(it0) || coop_inner(it0)) && sp0 - slo0 >= COOP_WIDE) {
                bool wide = true;
#pragma unroll
                for (int k = 1; k <= COOP_WIDE; k++) {
                    const unsigned x = (unsigned)__builtin_amdgcn_readfirstlane((int)stk[(sp0 - k) & (LDS_STACK - 1)][sl0]);
                    wide = wide && (coop_leaf(x) || coop_inner(x));
                }
                if (wide) { coop = true; nrun = 1; }
            }
        }
        const int rest = most - nrun;
        if (pick == K_NODE) q0 = rest; else if (pick == K_CULL) q1 = rest; else if (pick == K_TRI) q2 = rest; else q3 = rest;
        const bool run = lane < nrun;
        uint2 e = make_uint2(0u, 0u);
        if (run) e = sQ[qindex(pick, rest + lane)];
        const int slot = (int)(e.x & 0xFFu);
        const int sl = slot + wslot0; // its index in the LDS arrays
        const unsigned cur = e.y;
        unsigned st = e.x >> 8;
        double *const ls = S.lstate + gslot0 + slot; // this slot's column of the HBM state
        const SlotWords lw = {(char *)S.lstate, (gslot0 + (unsigned)slot) * 8u}; // (the same for the traversal steps, see SlotWords)
        RT_SCOUNT(c_iters, 1);
        RT_SCOUNT(c_lanes, nrun);
#ifdef RT_DRAIN
        if (dr_dry != 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (pick == k) { dr_steps[k]++; dr_lanes[k] += (unsigned)nrun; }
        }
        if (most == 1 && q0 + q1 + q2 + q3 == 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) if (pick == k) dr_lone[k]++;
        }
#endif

        // the stack: levels [slo, sp) are in the LDS ring (level i at i mod LDS_STACK), levels [0, slo) in HBM
        int sp = (int)(st & ST_SP), slo = (int)((st >> ST_LO) & 0xFFu);
        auto push = [&](unsigned item) __attribute__((always_inline)) {
            if (sp - slo == LDS_STACK) { // ring full: its oldest level moves to HBM
                S.ovf[(unsigned)slo * (unsigned)S.n_slots + gslot0 + (unsigned)slot] = stk[slo & (LDS_STACK - 1)][sl];
                slo++;
            }
            stk[sp & (LDS_STACK - 1)][sl] = item;
            sp++;
        };
        auto pop = [&]() __attribute__((always_inline)) -> unsigned { // (without branches: the read is harmless when not used)
            const unsigned e = stk[(sp - 1) & (LDS_STACK - 1)][sl];
            const bool in_ring = sp > slo;
            // the ring is empty: the level is in HBM and stays there until a refill step
            const unsigned r = sp == 0 ? IT_END : in_ring ? e : IT_REFILL;
            sp -= in_ring ? 1 : 0;
            return r;
        };
        auto is_marker = [](unsigned x) __attribute__((always_inline)) { return (x & (IT_AFTER | 1u)) == (IT_AFTER | 1u) && x != IT_END; };
        unsigned next = 0;      // the item to continue with, unless the tail is to pop one
        unsigned vis_add = 0;   // visits that this lane counts in this step
        // A marker restores the frame's flag; if the near child found nothing, its far child is entered (the marker
        // carries its granule) or, if that child cannot pass its box, counted as a visit.  Returns "pop the next item".
        auto do_marker = [&](unsigned m) __attribute__((always_inline)) -> bool {
            const unsigned fk = (m >> 28) & 3u;
            const bool near_hit = (st & ST_FLAG) != 0; // what the near child's check() returned
            vis_add += near_hit ? 0u : ((st >> ST_NSH) & 7u) + (fk == 3u ? 1u : 0u);
            st = (st & ~ST_N) | (((m >> IT_NSH) & 7u) << ST_NSH);
            if ((m >> 30) & 1u) st |= ST_FLAG;          // back in the enclosing frame
            if (!near_hit && (fk == 1u || fk == 2u)) { next = after_granule(m); return false; } // kdtree.cpp:124,133
            return true;
        };
        // what the step leaves behind for the common tail
        bool want_pop = false;  // take the next item from the stack (or finish the traversal)
        unsigned push_item = 0; // item to push before that (0 = none; the root is never pushed)
        bool done = false;      // the slot found no more work
        RT_TICK(tk1);
        RT_MARK("bodies");

        if (COOP && coop) {
            RT_MARK("coop");
            // ---- the wavefront scans the rest of the slot's current list together.  Two work lists in LDS, leaves and inner
            // records, in the parts of the scheduler's stacks that at most COOP_SLOTS entries per stack leave free.  Leaves
            // first; a batch of inner records only as wide as leaves room for all its children plus S.coop_reserve words,
            // below which the scan goes on one record at a time, depth first, which needs at most three words per level
            // of the hierarchy.
            const int slot_u = __builtin_amdgcn_readfirstlane(slot), sl_u = slot_u + wslot0;
            const unsigned st_u = (unsigned)__builtin_amdgcn_readfirstlane((int)st), cur_u = (unsigned)__builtin_amdgcn_readfirstlane((int)cur);
            int sp_u = __builtin_amdgcn_readfirstlane(sp);
            const int slo_u = __builtin_amdgcn_readfirstlane(slo);
            unsigned *const WT = (unsigned *)(sQ + 2 * wslot0) + 2 * COOP_SLOTS, *const WC = WT + 2 * SLOTS;
            constexpr int CAP = 2 * SLOTS - 4 * COOP_SLOTS; // words of each list
            static_assert(CAP >= 2 * 64 + 16, "the largest reserve (rt_prepare_scene) and a record's children fit");
            int nT = 0, nC = 0;
            int k_steps = 0, k_lanes = 0, k_done = 0; // (work counters of the scan, added to the wavefront's after it)
            auto put = [&](unsigned x) __attribute__((always_inline)) {
                if (coop_leaf(x)) { if (lane == 0) WT[nT] = x; nT++; }
                else { if (lane == 0) WC[nC] = x; nC++; }
            };
            put(cur_u);
            // what the ring holds of the same list on top of anything else comes along (levels in HBM stay: the refill brings them)
            while (sp_u > slo_u) {
                const unsigned x = (unsigned)__builtin_amdgcn_readfirstlane((int)stk[(sp_u - 1) & (LDS_STACK - 1)][sl_u]);
                if (!(coop_leaf(x) || coop_inner(x)) || nT >= 16 || nC >= 16) break;
                put(x);
                sp_u--;
            }
            const SlotWords lu = {(char *)S.lstate, (gslot0 + (unsigned)slot_u) * 8u};
            const double2 a = sA[sl_u], b = sB[sl_u], c = sC[sl_u];
            const d3 o = mk(a.x, a.y, b.x), d = mk(lu[LS_V * lss], lu[(LS_V + 1) * lss], lu[(LS_V + 2) * lss]);
            const float mg = exp_up_get(sM[sl_u]);
            const float tlim = (st_u & ST_FLAG) ? half_up_get(sT[sl_u]) * 1.000001f : INFINITY; // (as the cull step: once the frame has a hit)
            // per lane: the best of the triangles this lane tested -- smallest distance, lower list rank among equals
            double ct = 1e20;
            unsigned cha = 0;
            int crank = 0x7FFFFFFF;
            bool chit = false;
            while (nT + nC > 0) {
                k_steps++;
                if (nT > 0) { // ---- a batch of leaves: the exact tests (kdtree.cpp:105)
                    const int m = nT < WAVE ? nT : WAVE;
                    nT -= m;
                    k_lanes += m;
                    const bool act = lane < m;
                    const unsigned x = act ? WT[nT + lane] : 0u;
                    const bool grp = (x >> 29) == 3u;
                    const unsigned gn = !act ? 0u : grp ? ((x >> 26) & 3u) + 2u : 1u;
                    if (act) {
                        const double2 *p = (const double2 *)pool_at(S.pool, grp ? S.g_tri + 5u * (x & GROUP_POS) : (x & IDX28));
#pragma unroll
                        for (int k = 0; k < GROUP; k++) {
                            if (gn > (unsigned)k) {
                                const double2 *r = p + 5 * k;
                                const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4];
                                const double t = tri_hit(mk(r0.x, r0.y, r1.x), mk(r1.y, r2.x, r2.y), mk(r3.x, r3.y, r4.x), o, d);
                                const long long w = __double_as_longlong(r4.y);
                                if (t > EPS) {
                                    const int rk = (int)(w >> 32);
                                    if (t < ct || (t == ct && t < 1e20 && rk < crank)) { ct = t; cha = (unsigned)w & ~TRI_LAST; crank = rk; }
                                    if (t < 1e10) chit = true;
                                }
                            }
                        }
                    }
                    {
                        int tested = 0;
#pragma unroll
                        for (int k = 0; k < GROUP; k++) tested += (int)__popcll(__ballot(gn > (unsigned)k));
                        k_done += tested;
                    }
                } else { // ---- a batch of inner records: four child boxes each, as in the cull step
                    int m = (CAP - nC - S.coop_reserve) / 4;
                    m = m < 1 ? 1 : m;
                    m = m < nC ? m : nC;
                    m = m < CAP / 4 ? m : CAP / 4; // (no leaf is waiting: all children of the batch fit the leaves' list)
                    m = m < WAVE ? m : WAVE;
                    nC -= m;
                    k_lanes += m;
                    const bool act = lane < m;
                    const unsigned x = act ? WC[nC + lane] : 0u;
                    bool hit[4] = {false, false, false, false};
                    unsigned item[4] = {0u, 0u, 0u, 0u};
                    if (act) {
                        const uint4 *p = (const uint4 *)pool_at(S.pool, x & IDX28);
                        const uint4 r0 = p[0], r1 = p[1], r2 = p[2], it = p[3];
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const float sf[3] = {(float)a.x, (float)a.y, (float)b.x}, ivf[3] = {cull_inv((float)b.y), cull_inv((float)c.x), cull_inv((float)c.y)};
                        const float mt = (mg + S.fslack + 9.6e-7f * (fabsf(sf[0]) + fabsf(sf[1]) + fabsf(sf[2]))) * 1.000001f;
                        const float base[3] = {__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)};
                        const float step[3] = {__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y)};
                        const unsigned qw[6] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w};
                        item[0] = it.x; item[1] = it.y; item[2] = it.z; item[3] = it.w;
                        f2 ka[3], kb[3];
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            const float a1 = step[i] * ivf[i];
                            ka[i] = (f2){a1, a1};
                            kb[i] = (f2){(base[i] - (sf[i] + mt)) * ivf[i], (base[i] - (sf[i] - mt)) * ivf[i]};
                        }
#pragma unroll
                        for (int k = 0; k < 4; k++) {
                            float ta[3], tb[3];
#pragma unroll
                            for (int i = 0; i < 3; i++) {
                                const int bi = (k * 3 + i) * 2;
                                const unsigned wd = qw[bi >> 2];
                                const f2 qf = (bi & 2) ? (f2){(float)((wd >> 16) & 0xFFu), (float)(wd >> 24)} : (f2){(float)(wd & 0xFFu), (float)((wd >> 8) & 0xFFu)};
                                const f2 t = __builtin_elementwise_fma(qf, ka[i], kb[i]);
                                ta[i] = fminf(t.x, t.y);
                                tb[i] = fmaxf(t.x, t.y);
                            }
                            const float tmin = fmaxf(fmaxf(ta[0], ta[1]), fmaxf(ta[2], 0.0f)), tmax = fminf(fminf(tb[0], tb[1]), fminf(tb[2], tlim));
                            hit[k] = (tmin * 0.999999f <= tmax) & (item[k] != 0);
                        }
                        l_culls += 1u;
                    }
                    // the children that are met join the work list, each kind on its side
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const bool lf = hit[k] && coop_leaf(item[k]), in = hit[k] && !coop_leaf(item[k]); // (a hierarchy's children: triangle, group or record)
                        const unsigned long long ml = __ballot(lf), mi = __ballot(in);
                        if (lf) WT[nT + (int)__popcll(ml & lane_lt)] = item[k];
                        if (in) WC[nC + (int)__popcll(mi & lane_lt)] = item[k];
                        nT += (int)__popcll(ml);
                        nC += (int)__popcll(mi);
                    }
                }
            }
#ifdef RT_DRAIN
            dr_lone[4]++; dr_lone[5] += (unsigned)__builtin_amdgcn_readfirstlane(k_steps);
#endif
            RT_SCOUNT(c_iters, __builtin_amdgcn_readfirstlane(k_steps)); RT_SCOUNT(c_lanes, __builtin_amdgcn_readfirstlane(k_lanes)); RT_SCOUNT(c_done, __builtin_amdgcn_readfirstlane(k_done));
            // ---- the wavefront's best: smallest distance, then lower rank; against what the slot holds: strict < keeps the
            // earlier of two equal distances -- inside one list the lower rank, across lists the list visited first
            double bt = ct;
            for (int off = 32; off > 0; off >>= 1) { const double v = __shfl_xor(bt, off, 64); bt = v < bt ? v : bt; }
            int br = ct == bt ? crank : 0x7FFFFFFF;
            for (int off = 32; off > 0; off >>= 1) { const int v = __shfl_xor(br, off, 64); br = v < br ? v : br; }
            const unsigned long long mw = __ballot(ct == bt && crank == br);
            const unsigned bha = (unsigned)__builtin_amdgcn_readlane((int)cha, mw ? __ffsll((long long)mw) - 1 : 0);
            const bool any = __ballot(chit) != 0;
            if (run) {
                const double best_t = ls[LS_BT * lss];
                if (bt < 1e20) {
                    bool better = bt < best_t;
                    if (__builtin_expect(bt == best_t, 0)) {
                        const uint2 lb = S.tlist[((const unsigned *)&ls[LS_HA * lss])[0]], lc = S.tlist[bha];
                        better = lb.x == lc.x && br < (int)lb.y;
                    }
                    if (better) {
                        ls[LS_BT * lss] = bt;
                        sT[sl] = half_up_put((float)bt * 1.000001f);
                        ls[LS_HA * lss] = __longlong_as_double((long long)(unsigned long long)bha);
                    }
                }
                if (any) st |= ST_FLAG;
                sp = sp_u;
                want_pop = true;
            }
        } else
        if (pick == K_NODE) {
            RT_MARK("node");
            const bool refill = run && cur == IT_REFILL;
            const bool marker = run && is_marker(cur); // a second marker in a row (see the tail)
            if (__builtin_expect(marker, 0)) want_pop = do_marker(cur);
            unsigned rf[REFILL];
            if (refill) { // issue the loads of levels sp - 1 (the item to continue with), sp - 2, ...
#pragma unroll
                for (int k = 0; k < REFILL; k++) {
                    const int lv = sp - 1 - k > 0 ? sp - 1 - k : 0;
                    rf[k] = S.ovf[(unsigned)lv * (unsigned)S.n_slots + gslot0 + (unsigned)slot];
                }
            }
            bool skipped = false, skipped_far = false; // a child's visit was decided here (it is still a visit of the reference)
            bool ev_exact = false, ev_fail = false, ev_beyond = false;
            if (run && !refill && !marker) { // ---- enter a node (kdtree.cpp:111-141)
                const double2 *p = (const double2 *)pool_at(S.pool, cur);
                const double2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
                const uint4 *kp = (const uint4 *)pool_at(S.pool, 2u * S.g_list + 3u * (cur >> 2)); // boxes of the children and the own list
                const uint4 k0 = kp[0], k1 = kp[1], k2 = kp[2];
                const float gfac = exp_up_get(sG[sl]);
                const double2 a = sA[sl], b = sB[sl], c = sC[sl];
#ifdef RT_PROFILE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                p_load[0] += __builtin_amdgcn_s_memtime() - tk1;
#endif
                const double s[3] = {a.x, a.y, b.x}, inv[3] = {b.y, c.x, c.y};
                const double mi[3] = {q0.x, q0.y, q1.x}, ma[3] = {q1.y, q2.x, q2.y};
                const double split = q3.x;
                const long long w7 = __double_as_longlong(q3.y);
                const unsigned right = (unsigned)w7, info = (unsigned)(w7 >> 32);
                const unsigned node = cur;
                // check_aabb (kdtree.h:57-76) is decided without its divisions whenever the answer is
                // clear.  With the entry / exit parameters of the three slabs of the exact box:
                //  * it can only accept a ray that meets the box grown by eps plus rounding (`pad`);
                //  * it accepts an origin inside the box grown by eps (its own first test);
                //  * it accepts a ray that enters the exact box at t > 2 eps through a face whose
                //    direction component exceeds 2.5 eps: at the entry point the other two
                //    coordinates are inside the box up to ~1e-9 (coordinates are bounded through
                //    `pad`), far inside the eps tolerance of the reference's face test, so its
                //    loop accepts on that axis.
                // Whatever is left (grazing rays, nearly axis-parallel directions) runs the exact code.
                // (the ray's own part of these conditions -- |direction components| > 2.5 eps, padding below 3e-5 -- is the
                // ray's "tame" flag; the origin-inside test compares the differences the slab parameters need anyway
                // against eps less 1e-9, which implies the reference's own comparison for every origin such a padding
                // allows (|coordinates| below 1.25e6: rounding below 1.5e-10).  A lane that neither test accepts -- a node
                // that is missed, mostly -- goes through the full set of conditions below.)
                double tin = -1e300, tout = 1e300;
                bool in1 = true;
#pragma unroll
                for (int i = 0; i < 3; i++) {
                    const double d1 = mi[i] - s[i], d2 = ma[i] - s[i];
                    const double t1 = d1 * inv[i], t2 = d2 * inv[i];
                    tin = fmax(tin, fmin(t1, t2)); tout = fmin(tout, fmax(t1, t2));
                    in1 &= (d1 < EPS * (1 - 1e-4)) & (d2 > -EPS * (1 - 1e-4));
                }
                const int axis = info & 3;
                const double sw = axis == 0 ? s[0] : axis == 1 ? s[1] : s[2];
                const double iw = axis == 0 ? inv[0] : axis == 1 ? inv[1] : inv[2];
                bool pass = ((st & ST_TAME) != 0) & (in1 | ((tin <= tout) & (tin > 2e-5)));
                bool vpos = iw > 0, vneg = iw < 0; // sign of the direction along the split axis
                if (__builtin_expect(!pass | (fabs(iw) >= 1e299), 0)) { // one lane in a thousand
                    const double pad = S.padc + 4e-12 * (fabs(s[0]) + fabs(s[1]) + fabs(s[2]));
                    const double imax = fmax(fmax(fabs(inv[0]), fabs(inv[1])), fabs(inv[2]));
                    // growing the box by `pad` moves every slab parameter by at most pad * max|1/direction|
                    const double pm = pad * imax;
                    const bool maybe = (tin - pm <= tout + pm) & (tout + pm >= 0.0);
                    const bool inside = (mi[0] < s[0] + EPS) & (s[0] < ma[0] + EPS) & (mi[1] < s[1] + EPS) & (s[1] < ma[1] + EPS) &
                                        (mi[2] < s[2] + EPS) & (s[2] < ma[2] + EPS);
                    const bool clear = (tin <= tout) & (tin > 2e-5) & (imax < 4e4) & (pad < 3e-5);
                    pass = inside | clear;
                    ev_exact = (maybe & !pass) | (fabs(iw) >= 1e299);
                    if ((maybe & !pass) | (fabs(iw) >= 1e299)) { // rare: needs the direction itself
                        const double v[3] = {lw[LS_V * lss], lw[(LS_V + 1) * lss], lw[(LS_V + 2) * lss]};
                        if (!pass) pass = maybe && aabb_hit(mi, ma, s, v);
                        const double vw = axis == 0 ? v[0] : axis == 1 ? v[1] : v[2];
                        vpos = vw > 0; vneg = vw < 0;
                    }
                }
                ev_fail = !pass;
#ifdef RT_EVENTS
                ev_beyond = pass && tin > (double)half_up_get(sT[sl]);
#endif
                if (!pass) want_pop = true;
                else {
                    l_tris += info >> 3; // the own list will be scanned (kdtree.cpp:122,131,137)
                    // Truth values from here on are lane masks (one bit per lane, in scalar registers): combining them costs
                    // no vector instruction, and a select takes the mask as it is.
                    typedef unsigned long long LaneMask;
                    auto lanes = [](LaneMask m) __attribute__((always_inline)) { return __builtin_amdgcn_inverse_ballot_w64(m); };
                    const LaneMask mHasL = __ballot(((info >> 2) & 1u) != 0), mHasR = __ballot(right != 0);
                    const LaneMask mLt = __ballot(sw < split), mGt = __ballot(sw > split), mVpos = __ballot(vpos), mVneg = __ballot(vneg);
                    // kdtree.cpp:119-127 (left of the plane), 128-136 (right of it), 137-141 (on it: only
                    // the child the ray points into)
                    const LaneMask mNearL = mHasL & (mLt | (~mGt & mVneg)), mNearR = mHasR & ~mNearL & (mGt | (~mLt & mVpos));
                    const LaneMask mFarR = mLt & mHasR & mVpos, mFarL = mGt & mHasL & mVneg; // the far child, if the near subtree returns false
                    // A child whose check_aabb is certain to fail is not visited, only counted: check_aabb
                    // accepts nothing that misses the child's box grown by `pad` (see above), and the
                    // quantised boxes contain the exact ones, so the single-precision slab test of the
                    // culling records (same expression, same error bound, inflation `pad`) decides it.
                    // The own list's box (inflated like the boxes of its hierarchy) is tested the same way: a
                    // list the ray cannot meet, or an empty one, needs no step of its own.
                    LaneMask mMet[3], mNokids, mEmpty, mHasList;
                    {
                        typedef float f2 __attribute__((ext_vector_type(2)));
                        const float sf[3] = {(float)s[0], (float)s[1], (float)s[2]};
                        const float ivf[3] = {cull_inv((float)inv[0]), cull_inv((float)inv[1]), cull_inv((float)inv[2])};
                        const float msl = S.fslack + 9.6e-7f * (fabsf(sf[0]) + fabsf(sf[1]) + fabsf(sf[2]));
                        const float kbase[3] = {__uint_as_float(k0.x), __uint_as_float(k0.y), __uint_as_float(k0.z)};
                        const float kstep[3] = {__uint_as_float(k0.w << 16), __uint_as_float(k0.w & 0xFFFF0000u), __uint_as_float(k1.x << 16)};
                        const unsigned kq[5] = {k1.y, k1.z, k1.w, k2.x, k2.y}; // 18 plane bytes: [box][axis][lower, upper]
                        const unsigned kflags = k1.x >> 16;
                        // the list's inflation, as the step that starts the list would compute it
                        const float mgl = ldexpf(__uint_as_float(k2.z), (st & ST_FAR) ? (int)(k2.z & 0xFFu) : 0) * gfac;
                        // one inflation for all three boxes, the larger of the two (a larger one only skips less)
                        // (the node-box padding, rounded up: S.padcf >= padc, 4.1e-12f * |origin|_1 in single precision >= 4e-12 * |origin|_1)
                        const float padf = S.padcf + 4.1e-12f * (fabsf(sf[0]) + fabsf(sf[1]) + fabsf(sf[2]));
                        const float mt = (fmaxf(padf, mgl) + msl) * 1.000001f;
                        f2 ka[3], kb[3];
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            const float a1 = kstep[i] * ivf[i];
                            ka[i] = (f2){a1, a1};
                            kb[i] = (f2){(kbase[i] - (sf[i] + mt)) * ivf[i], (kbase[i] - (sf[i] - mt)) * ivf[i]};
                        }
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            float ta[3], tb[3];
#pragma unroll
                            for (int i = 0; i < 3; i++) {
                                const int bi = (k * 3 + i) * 2;
                                const unsigned wd = kq[bi >> 2];
                                const f2 qf = (bi & 2) ? (f2){(float)((wd >> 16) & 0xFFu), (float)(wd >> 24)} : (f2){(float)(wd & 0xFFu), (float)((wd >> 8) & 0xFFu)};
                                const f2 t = __builtin_elementwise_fma(qf, ka[i], kb[i]);
                                ta[i] = fminf(t.x, t.y);
                                tb[i] = fmaxf(t.x, t.y);
                            }
                            const float tmin = fmaxf(fmaxf(ta[0], ta[1]), fmaxf(ta[2], 0.0f)), tmax = fminf(fminf(tb[0], tb[1]), tb[2]);
                            mMet[k] = __ballot(tmin * 0.999999f <= tmax);
                        }
                        mNokids = __ballot((kflags & KID_NOKIDS) != 0); // boxes marked unusable cull nothing
                        mEmpty = __ballot((kflags & KID_EMPTY) != 0);
                        mHasList = __ballot((kflags & KID_LIST) != 0);
                    }
                    const LaneMask mMetL = mMet[0] | mNokids, mMetR = mMet[1] | mNokids;
                    const LaneMask mListMissed = mEmpty | (mHasList & ~mMet[2]);
                    const LaneMask mHasNear = (mNearL & mMetL) | (mNearR & mMetR); // a near child that cannot pass its box: check() returns false at once
                    const LaneMask mFar0 = mFarL | mFarR, mFar12 = (mFarL & mMetL) | (mFarR & mMetR), mFar3 = mFar0 & ~mFar12; // far3: likewise, if it comes to that
                    const unsigned left = node + 4;
                    const unsigned near = lanes(mNearL) ? left : right, far_item = lanes(mFarL) ? left : right;
                    // the own list and the far child come after the near subtree; the frame's flag
                    // travels in the item and the near subtree starts with a clean one
                    // (far child in the item: 1 left, 2 right, 3 only counted; the frame's flag; its count-only children so far)
                    const unsigned common = IT_AFTER | ((st & ST_FLAG) << 22) | (lanes(mFarL | mFar3) ? 1u << 28 : 0u) | (lanes(mFarR | mFar3) ? 1u << 29 : 0u) |
                                            ((st & ST_N) << (IT_NSH - ST_NSH));
                    const unsigned after = common | (node >> 1);
                    // (three cases)
                    //  * the list has to be scanned: after item under the near subtree, or next if there is none;
                    //  * nothing to scan after the near subtree: a marker (an after item with bit 0 set, which the
                    //    tail of whatever step pops it handles) restores the flag and decides about the far child,
                    //    whose granule it carries in place of the node's -- unless it would have nothing to do
                    //    (no far child, clear flag to restore);
                    //  * no near subtree either: check() goes straight to the far child (kdtree.cpp:124,133), and a
                    //    far child that fails at its box is only counted.
                    const unsigned marker = common | (lanes(mFar12) ? far_item >> 1 : 0u) | 1u;
                    const LaneMask mFlag = __ballot((st & ST_FLAG) != 0), mRoom = __ballot(((st >> ST_NSH) & 7u) < N_MAX);
                    // (a far child that is only counted joins the frame's number instead, while that has room)
                    const LaneMask mCountOnly = mFar3 & ~mFlag & mRoom;
                    const LaneMask mWantMarker = (mFar0 | mFlag) & ~mCountOnly;
                    // A list that is one item, with no near subtree to wait for, needs no step of its own for "the near child
                    // has returned": what that step would do -- count a far child that is certain to fail, put the other kind
                    // on the stack, restore the flag and the count it has just saved -- is done here, and its triangles come next.
                    const unsigned direct = k2.w;
                    const LaneMask mGoDirect = __ballot(direct != 0) & ~mListMissed & ~mHasNear;
                    const LaneMask mOwnFrame = ~mGoDirect;
                    st = lanes((~mListMissed | mHasNear) & mOwnFrame) ? st & ~ST_FLAG : st;
                    st = lanes(mHasNear & mListMissed & mCountOnly) ? st + (1u << ST_NSH) : st;
                    st = lanes((~mListMissed | (mHasNear & mWantMarker)) & mOwnFrame) ? st & ~ST_N : st; // the item starts a frame of its own
                    push_item = lanes(mHasNear & ~mListMissed) ? after : lanes(mHasNear & mWantMarker) ? marker : lanes(mGoDirect & mFar12) ? far_item : 0u;
                    next = lanes(mHasNear) ? near : lanes(mGoDirect) ? direct : lanes(~mListMissed) ? after : lanes(mFar12) ? far_item : 0u;
                    skipped = lanes((mNearL | mNearR) & ~mHasNear);
                    skipped_far = lanes((mListMissed | mGoDirect) & ~mHasNear & mFar3);
                    want_pop = lanes(mListMissed & ~mHasNear & ~mFar12);
                }
            }
#ifndef RT_NO_CONSUME
            if (REFILL == 3) asm volatile("" : : "v"(rf[0]), "v"(rf[1]), "v"(rf[2])); // (delivery taken on every path, see the cull step)
            if (REFILL == 4) asm volatile("" : : "v"(rf[0]), "v"(rf[1]), "v"(rf[2]), "v"(rf[REFILL - 1]));
#endif
            if (refill) { // levels [slo', sp - 1) go to the ring, level sp - 1 continues
                sp--;
                next = rf[0];
#pragma unroll
                for (int k = 1; k < REFILL; k++)
                    if (sp - k >= 0) stk[(sp - k) & (LDS_STACK - 1)][sl] = rf[k];
                slo = sp - (REFILL - 1) > 0 ? sp - (REFILL - 1) : 0;
            }
            RT_EV(0, run && !refill && !marker); RT_EV(1, ev_exact); RT_EV(2, refill); RT_EV(3, marker); RT_EV(4, ev_fail);
            RT_EV(16, run && !refill && !marker && !ev_fail && next != 0 && !want_pop && push_item == 0); // goes on without a push
            RT_EV(20, ev_beyond); // the node's box lies beyond the best hit so far
            const bool fetched = run && !refill && !marker; // lanes that fetched a node's records
            l_nf += fetched ? 1u : 0u;
            vis_add += (fetched ? 1u : 0u) + (skipped ? 1u : 0u) + (skipped_far ? 1u : 0u);
        } else if (pick == K_CULL) {
            RT_MARK("cull");
            bool boxes = false;
            bool ev_one = false, ev_lim = false;
            if (run) { // ---- four child boxes of a list's hierarchy; for an after item, the list's start
                const unsigned g = (cur & IT_AFTER) ? S.g_list + after_granule(cur) : (cur & IDX28);
                const uint4 *p = (const uint4 *)pool_at(S.pool, g);
                const uint4 r0 = p[0], r1 = p[1], r2 = p[2], it = p[3];
                // right child, list header (every lane loads, the others node 0's: a load under a branch of its own
                // makes the compiler wait for all of the loads above before it goes on)
                const uint2 hdr = *(const uint2 *)(S.lhdr + ((cur & IT_AFTER) ? after_granule(cur) >> 2 : 0u));
                const double2 a = sA[sl], b = sB[sl], c = sC[sl];
                float mg = exp_up_get(sM[sl]);
                const float gfac = exp_up_get(sG[sl]); // (read with the rest: only a list's first step uses it, but a read of its own there is a round trip of its own)
                float best_lim = half_up_get(sT[sl]);
                // (for an after item the flag of the enclosing frame is restored below: st or the item's bit)
                // (the sphere hierarchy keeps only the nearest sphere: its boxes are always limited by that distance,
                // a word of the slot's state in HBM that is fetched with the record)
                const bool sph_boxes = SPH && !(cur & IT_AFTER) && (cur & IDX28) >= S.g_sphc;
                double sph_best = 0;
                if (SPH && sph_boxes) sph_best = lw[LS_ST * lss];
#ifndef RT_NO_CONSUME
                // Every lane takes delivery of all it loaded, here: a load whose result some path never reads counts as in
                // flight for the compiler from then on, and it then puts full waits -- which also wait for the stores
                // issued since -- in front of whatever next writes those registers, in this step's tail and in the next step.
                asm volatile("" : : "v"(r0.x), "v"(r0.y), "v"(r0.z), "v"(r0.w), "v"(r1.x), "v"(r1.y), "v"(r1.z), "v"(r1.w), "v"(r2.x), "v"(r2.y),
                             "v"(r2.z), "v"(r2.w), "v"(it.x), "v"(it.y), "v"(it.z), "v"(it.w), "v"(hdr.x), "v"(hdr.y));
                if (SPH) asm volatile("" : : "v"(sph_best));
#endif
#ifdef RT_PROFILE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                p_load[1] += __builtin_amdgcn_s_memtime() - tk1;
#endif
                ev_one = it.y == 0 && it.z == 0 && it.w == 0 && (it.x >> 30) != 0; // (diagnostics)
                const bool lim_on = sph_boxes || ((st & ST_FLAG) != 0) || ((cur & IT_AFTER) && ((cur >> 30) & 1));
                if (sph_boxes) best_lim = (float)sph_best;
                boxes = true;
                if (cur & IT_AFTER) { // near child finished: own list (kdtree.cpp:122,131,137), then far child
                    const unsigned right = hdr.x, meta = hdr.y;
                    const unsigned node = after_granule(cur), farkind = (cur >> 28) & 3;
                    const bool near_hit = (st & ST_FLAG) != 0;         // what the near child's check() returned
                    // the far child's check() fails at its box: a visit, nothing else; likewise those the frame kept count of
                    vis_add += near_hit ? 0u : ((st >> ST_NSH) & 7u) + (farkind == 3 ? 1u : 0u);
                    st = (st & ~ST_N) | (((cur >> IT_NSH) & 7u) << ST_NSH);
                    if ((cur >> 30) & 1) st |= ST_FLAG;                 // back in the enclosing frame
                    if (((farkind - 1u) < 2u) & !near_hit) push(farkind == 1 ? node + 4 : right); // kdtree.cpp:124,133: after the list (one condition, one branch)
                    if (__builtin_expect((meta & LIST_SHORT) != 0, 0)) { // up to SMALL_LIST triangles: a chain of tri steps in list order
                        boxes = false;
                        if ((meta & 0xFF) == 0) want_pop = true;
                        else next = it.x;
                    } else { // inflation of this list's boxes, see ray_margin_factor(); a larger one only culls less
                        // (the exponent in the low mantissa bits only makes the coefficient a little larger)
                        mg = ldexpf(__uint_as_float(meta), (st & ST_FAR) ? (int)(meta & 0xFFu) : 0) * gfac;
                        { const unsigned char h = exp_up_put(mg); sM[sl] = h; mg = exp_up_get(h); }
                    }
                }
                if (SPH && !(cur & IT_AFTER) && (cur & IT_CHAIN)) { // root of the sphere hierarchy: the kd-tree is done
                    // Sphere::intersect (sphere.h:13-21) accepts a unit-length ray only within
                    // r + (5.1e-13 |op|^2 + 3e-16 r^2) / r of the centre (rounding of det and of |d|^2 - 1,
                    // which the ray's start bounds by 1e-12); |op| <= |o - C|_1 + H for every sphere
                    const double gs = fabs(a.x - S.sph_cen[0]) + fabs(a.y - S.sph_cen[1]) + fabs(b.x - S.sph_cen[2]) + S.sph_h;
                    const float gsf = (float)gs * 1.000001f;
                    mg = S.sph_k * gsf * gsf + S.sph_c;
                    { const unsigned char h = exp_up_put(mg); sM[sl] = h; mg = exp_up_get(h); }
                }
                if (boxes) {
                    // Slab test of the four inflated boxes in single precision, two planes per packed
                    // instruction: the ray parameter of the plane base + q * step is evaluated as
                    // q * (step / d) + (base - origin -+ m) / d, one fma per pair of planes.  It may only
                    // err towards "met".  Against the plane the builder verified (the single-precision
                    // fma(q, step, base), within half an ulp of base + q * step) the rounding of the
                    // origin, of the three products, of the two differences and of the fma moves a plane
                    // by less than 2^-24 * (10 C + 5.1 |origin| + 4.1 m), C = largest |coordinate| of the
                    // scene; the inflation m is therefore enlarged by 16 * 2^-24 * (|origin|_1 + C) and by
                    // 1e-6 of itself.  Rounding 1/direction scales all parameters of an axis by
                    // 1 +- 2^-24, for which the interval test is relaxed by 1e-6 relative.  1/direction
                    // is limited to +-1e30 (a direction component of zero has 1e300 there) so that every
                    // product stays finite: on such an axis the parameters are then beyond +-1e18 unless
                    // the origin lies exactly on a moved plane, far outside the parameters of the other
                    // axes (directions are at least 1e-10 long), which is all the test needs of them.
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    const float sf[3] = {(float)a.x, (float)a.y, (float)b.x}, ivf[3] = {cull_inv((float)b.y), cull_inv((float)c.x), cull_inv((float)c.y)};
                    const float mt = (mg + S.fslack + 9.6e-7f * (fabsf(sf[0]) + fabsf(sf[1]) + fabsf(sf[2]))) * 1.000001f;
                    const float base[3] = {__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)};
                    const float step[3] = {__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y)};
                    const unsigned qw[6] = {r1.z, r1.w, r2.x, r2.y, r2.z, r2.w}; // 24 plane bytes: [child][axis][lower, upper]
                    const unsigned item[4] = {it.x, it.y, it.z, it.w};
                    f2 ka[3], kb[3]; // ray parameter of the plane with byte q: q * ka + kb
#pragma unroll
                    for (int i = 0; i < 3; i++) {
                        const float a1 = step[i] * ivf[i];
                        ka[i] = (f2){a1, a1};
                        // lower planes move down by mt, upper planes up
                        kb[i] = (f2){(base[i] - (sf[i] + mt)) * ivf[i], (base[i] - (sf[i] - mt)) * ivf[i]};
                    }
                    // Once the frame has a hit (its check() returns true whatever follows, kdtree.cpp:105-107), a
                    // triangle can only matter by being nearer than the best one: boxes that the ray enters
                    // beyond that distance are dropped.  (Equal distances stay: the parameters err towards "met".)
                    const float tlim = lim_on ? best_lim * 1.000001f : INFINITY;
                    bool hit[4];
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        float ta[3], tb[3];
#pragma unroll
                        for (int i = 0; i < 3; i++) {
                            const int bi = (k * 3 + i) * 2; // byte index of the lower plane
                            const unsigned wd = qw[bi >> 2];
                            const f2 qf = (bi & 2) ? (f2){(float)((wd >> 16) & 0xFFu), (float)(wd >> 24)} : (f2){(float)(wd & 0xFFu), (float)((wd >> 8) & 0xFFu)};
                            const f2 t = __builtin_elementwise_fma(qf, ka[i], kb[i]);
                            ta[i] = fminf(t.x, t.y);
                            tb[i] = fmaxf(t.x, t.y);
                        }
                        const float tmin = fmaxf(fmaxf(ta[0], ta[1]), fmaxf(ta[2], 0.0f)), tmax = fminf(fminf(tb[0], tb[1]), fminf(tb[2], tlim));
                        hit[k] = (tmin * 0.999999f <= tmax) & (item[k] != 0);
#ifdef RT_EVENTS
                        if (hit[k] && !(tmin * 0.999999f <= fminf(tmax, best_lim * 1.000001f))) ev_lim = true;
#endif
                    }
                    // the first child that is met continues, the others go on the stack
                    const int n1 = hit[0], n2 = n1 + hit[1], n3 = n2 + hit[2], n4 = n3 + hit[3];
                    if (n4 == 0) want_pop = true;
                    else {
                        next = hit[0] ? item[0] : hit[1] ? item[1] : hit[2] ? item[2] : item[3];
                        if (LDS_STACK >= 4 || sp - slo + 3 <= LDS_STACK) {
                            if (LDS_STACK >= 4) { // make room first: the oldest levels move to HBM
                                const int over = sp - slo + (n4 - 1) - LDS_STACK;
                                if (__builtin_expect(over > 0, 0)) {
#pragma unroll
                                    for (int k = 0; k < 3; k++)
                                        if (k < over) S.ovf[(unsigned)(slo + k) * (unsigned)S.n_slots + gslot0 + (unsigned)slot] = stk[(slo + k) & (LDS_STACK - 1)][sl];
                                    slo += over;
                                }
                            }
                            if (hit[1] & (n1 > 0)) stk[(sp + n1 - 1) & (LDS_STACK - 1)][sl] = item[1];
                            if (hit[2] & (n2 > 0)) stk[(sp + n2 - 1) & (LDS_STACK - 1)][sl] = item[2];
                            if (hit[3] & (n3 > 0)) stk[(sp + n3 - 1) & (LDS_STACK - 1)][sl] = item[3];
                            sp += n4 - 1;
                        } else {
                            if (hit[1] & (n1 > 0)) push(item[1]);
                            if (hit[2] & (n2 > 0)) push(item[2]);
                            if (hit[3] & (n3 > 0)) push(item[3]);
                        }
                    }
                }
            }
            RT_EV(5, run && (cur & IT_AFTER)); RT_EV(6, boxes); RT_EV(7, boxes && want_pop); RT_EV(8, run && !boxes);
            RT_EV(18, boxes && want_pop && (cur & IT_AFTER)); // a list whose first record already meets nothing
            RT_EV(21, ev_lim); // a box that is met lies beyond the best hit so far (and the frame has no hit)
            RT_EV(19, run && (cur & IT_AFTER) && ev_one); // a list that is one triangle item, entered after a near subtree
            l_culls += boxes ? 1u : 0u;
        } else if (pick == K_TRI) {
            RT_MARK("tri");
            bool is_sph = false;
            if (run) { // ---- one exact triangle test (kdtree.cpp:105), a group's further ones below
                // a group's further triangles follow the first in the pool (a lane with fewer reads its first record
                // again: a load under a branch of its own would be waited for separately)
                const bool grp = (cur >> 29) == 3u;
                const unsigned gn = grp ? ((cur >> 26) & 3u) + 2u : 1u; // triangles of this step
                const double2 *p = (const double2 *)pool_at(S.pool, grp ? S.g_tri + 5u * (cur & GROUP_POS) : (cur & IDX28));
                const double2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
                const long long w9 = __double_as_longlong(q4.y);
                double2 u[GROUP - 1][5];
#pragma unroll
                for (int k = 1; k < GROUP; k++) {
                    const double2 *pk = p + (gn > (unsigned)k ? 5 * k : 0);
#pragma unroll
                    for (int i = 0; i < 5; i++) u[k - 1][i] = pk[i];
                }
                const bool chain = (cur >> 28) == ((IT_TRI | IT_CHAIN) >> 28);
                const d3 d = mk(lw[LS_V * lss], lw[(LS_V + 1) * lss], lw[(LS_V + 2) * lss]);
                const double best_t = lw[LS_BT * lss];
                const double2 a = sA[sl], b = sB[sl];
                is_sph = SPH && !grp && (cur & IDX28) >= S.g_sph;
                double sph_ts = 0; // the nearest sphere so far, fetched with the record
                if (SPH && is_sph) sph_ts = lw[LS_ST * lss];
#ifndef RT_NO_CONSUME
                if (SPH) asm volatile("" : : "v"(sph_ts));
                asm volatile("" : : "v"(q0.x), "v"(q0.y), "v"(q1.x), "v"(q1.y), "v"(q2.x), "v"(q2.y), "v"(q3.x), "v"(q3.y), "v"(q4.x), "v"(q4.y),
                             "v"(d.x), "v"(d.y), "v"(d.z), "v"(best_t)); // (delivery taken on every path, see the cull step)
#pragma unroll
                for (int k = 0; k < GROUP - 1; k++)
                    asm volatile("" : : "v"(u[k][0].x), "v"(u[k][0].y), "v"(u[k][1].x), "v"(u[k][1].y), "v"(u[k][2].x), "v"(u[k][2].y), "v"(u[k][3].x),
                                 "v"(u[k][3].y), "v"(u[k][4].x), "v"(u[k][4].y));
#endif
#ifdef RT_PROFILE
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                p_load[2] += __builtin_amdgcn_s_memtime() - tk1;
#endif
                const d3 o = mk(a.x, a.y, b.x);
                if (is_sph) { // ---- one sphere of the sphere hierarchy (sphere.h:13-21, sphere.cpp:7)
                    const int i = (int)(((cur & IDX28) - S.g_sph) >> 1);
                    const d3 op = mk(q0.x, q0.y, q1.x) - o; // the pool's copy of the sphere: centre, radius
                    const double rad = q1.y;
                    const double bb = dot(op, d);
                    double det = bb * bb - dot(op, op) + rad * rad;
                    double t;
                    if (det < 0) t = 1e20;
                    else {
                        det = sqrt(det);
                        t = bb - det;
                        if (!(t > EPS)) { t = bb + det; if (!(t > EPS)) t = 1e20; }
                    }
                    const double ts = sph_ts;
                    // the reference scans from the last sphere down with strict <: of equal distances the higher index wins
                    if (t < ts || (t == ts && t < 1e20 && i > (int)__double_as_longlong(lw[LS_SI * lss]))) {
                        lw[LS_ST * lss] = t;
                        lw[LS_SI * lss] = __longlong_as_double((long long)i);
                    }
                    want_pop = true;
                }
                // the step's triangles in turn, each against what the ones before left behind; the slot's state is
                // written once, by the last one that won
                double best_cur = best_t;
                unsigned new_ha = 0;
                bool have_new = false;
                auto consider = [&](double t, long long w) __attribute__((always_inline)) {
                    if (t > EPS) {
                        // strict < keeps the earlier of two equal distances; inside one list "earlier"
                        // is the lower rank, across lists it is the list visited first
                        bool better = t < best_cur;
                        if (__builtin_expect(t == best_cur && t < 1e20, 0)) { // rare: the two triangles' lists and ranks decide
                            const unsigned held = have_new ? new_ha : ((const unsigned *)&lw[LS_HA * lss])[0];
                            const uint2 lb = S.tlist[held], lc = S.tlist[(unsigned)w & ~TRI_LAST];
                            better = lb.x == lc.x && (int)(w >> 32) < (int)lb.y;
                        }
                        if (better) { best_cur = t; new_ha = (unsigned)w & ~TRI_LAST; have_new = true; }
                        if (t < 1e10) st |= ST_FLAG;
                    }
                };
                const double t = is_sph ? 0.0 : tri_hit(mk(q0.x, q0.y, q1.x), mk(q1.y, q2.x, q2.y), mk(q3.x, q3.y, q4.x), o, d);
                consider(t, w9);
#pragma unroll
                for (int k = 1; k < GROUP; k++) {
                    const bool has = gn > (unsigned)k;
                    const double2 *r = u[k - 1];
                    if (__ballot(has))
                        consider(has ? tri_hit(mk(r[0].x, r[0].y, r[1].x), mk(r[1].y, r[2].x, r[2].y), mk(r[3].x, r[3].y, r[4].x), o, d) : 0.0, __double_as_longlong(r[4].y));
                }
                if (have_new) {
                    lw[LS_BT * lss] = best_cur;
                    sT[sl] = half_up_put((float)best_cur * 1.000001f); // (t > 0; the factor covers the rounding to single)
                    lw[LS_HA * lss] = __longlong_as_double((long long)(unsigned long long)new_ha);
                }
                if (is_sph) {}
                else if (chain && !((unsigned)w9 & TRI_LAST)) next = cur + 5u; // the list's next triangle
                else want_pop = true;
#ifdef RT_PROFILE
                p_tri[0] += (unsigned)__popcll(__ballot(chain));                 // steps of short lists
                p_tri[1] += (unsigned)__popcll(__ballot(chain && t < 1e10));     // ... whose (first) triangle is hit
                p_tri[2] += (unsigned)__popcll(__ballot(!chain));                // steps behind culling boxes
                p_tri[3] += (unsigned)__popcll(__ballot(!chain && t < 1e10));    // ... whose (first) triangle is hit
#endif
            }
            RT_EV(9, run); RT_EV(10, run && (st & ST_FLAG));
            const int nsph = (int)__popcll(__ballot(is_sph));
            {
                int extra = 0; // a group's further triangles
#pragma unroll
                for (int k = 1; k < GROUP; k++) extra += (int)__popcll(__ballot(run && (cur >> 29) == 3u && ((cur >> 26) & 3u) + 2u > (unsigned)k));
                RT_SCOUNT(c_done, nrun - nsph + extra);
            }
            RT_SCOUNT(c_sphdone, nsph);
        } else {
            RT_MARK("shade");
            bool ev_ray = false, ev_shade = false, ev_scan = false;
            if (run) { // ---- traversal finished (or nothing started yet)
                // the launch's cold arguments, read here and nowhere else
                const ColdArgs &C = *cold;
                const rt_params &P = C.P;
                const DevCamera &cam = C.cam;
                const int sizex = cam.sizex, sizey = cam.sizey, samps = P.samps;
                const unsigned n_work = C.n_work;
                unsigned *const counter = as_global(C.counter);
                double *const rgb = as_global(C.rgb), *const subbuf = as_global(C.subbuf);
                int *const prim = C.prim ? as_global(C.prim) : nullptr;
                unsigned *const subcnt = as_global(C.subcnt);
                const TriShade *const c_tshade = as_global(C.tshade);
                const SphGeom *const c_sph = as_global(C.sph);
                const SphShade *const c_sshade = as_global(C.sshade);
                bool go_on = false, need_primary = false, need_work = false;
                d3 ro = mk(0, 0, 0), rd = mk(0, 0, 1);
                const unsigned long long wp = (unsigned long long)__double_as_longlong(ls[LS_PST * lss]);
                const unsigned long long wo = (unsigned long long)__double_as_longlong(ls[LS_OUT * lss]);
                const unsigned pst = (unsigned)wp;
                int smp = (int)(wp >> 32);
                unsigned outk = (unsigned)wo, pxy = (unsigned)(wo >> 32);
                int depth = (int)(pst & 0xFFFFF), np = (int)((pst >> 20) & 7), sub = (int)((pst >> 23) & 3);
                bool into = (pst >> 25) & 1, primary = (pst >> 26) & 1;
                const bool have_hit = (pst >> 27) & 1;
                bool emit_ok = !NEE || ((pst >> 28) & 1), shadow = NEE && ((pst >> 29) & 1), emit_one = NEE && ((pst >> 30) & 1);
                unsigned long long rng = 0;
                if (!TRACE) rng = (unsigned long long)__double_as_longlong(ls[LS_RNG * lss]);
                if (have_hit) {
                    const double2 a = sA[sl], b = sB[sl];
                    const d3 o = mk(a.x, a.y, b.x), d = mk(ls[LS_V * lss], ls[(LS_V + 1) * lss], ls[(LS_V + 2) * lss]);
                    const double to = ls[LS_BT * lss];
                    const int tpos = (int)((const unsigned *)&ls[LS_HA * lss])[0];
                    double ts;
                    int ids;
                    const long long wsi = SPH ? __double_as_longlong(ls[LS_SI * lss]) : -2ll;
                    if (wsi != -2ll) { ts = ls[LS_ST * lss]; ids = (int)wsi; } // found through the hierarchy
                    else { trace_spheres(c_sph, S.n_sph, o, d, ts, ids); ev_scan = true; }
                    const bool fo = to < 1e10, fs = ts < 1e10;
                    if (TRACE) {
                        if (fo && !(ts < to)) ev_shade = true;
                        as_global(C.tri_id)[outk] = fo ? c_tshade[tpos].gid : -1;
                        as_global(C.tri_t)[outk] = to;
                        as_global(C.sph_id)[outk] = fs ? ids : -1;
                        as_global(C.sph_t)[outk] = ts;
                        need_work = true;
                    } else if (NEE && shadow) { // the light adds what was set aside for it if it is what the ray reaches first
                        if (fs && (ts < to || !fo) && ids == (int)__double_as_longlong(ls[LS_COL * lss])) {
                            ls[LS_ACC * lss] += ls[LS_THR * lss]; ls[(LS_ACC + 1) * lss] += ls[(LS_THR + 1) * lss];
                            ls[(LS_ACC + 2) * lss] += ls[(LS_THR + 2) * lss];
                        }
                    } else if (fo || fs) { // raytrace.cpp:48-67
                        d3 x, n, f, lig;
                        int type;
                        d3 thr = mk(ls[LS_THR * lss], ls[(LS_THR + 1) * lss], ls[(LS_THR + 2) * lss]);
                        if (ts < to || !fo) {
                            const SphGeom g = c_sph[ids];
                            const SphShade sh = c_sshade[ids];
                            x = o + d * ts;
                            n = normv(x - mk(g.px, g.py, g.pz));
                            f = ld3(sh.col); type = sh.type; lig = emit_ok ? ld3(sh.lig) : mk(0, 0, 0);
                            if (NEE && emit_one && ids == (int)__double_as_longlong(ls[LS_COL * lss])) lig = ld3(sh.lig) * (double)C.n_lights;
                            if (primary && sub == 0 && smp == 0 && prim) prim[outk] = -1 - ids;
                        } else { // getn() returns the face normal (polygon.h:67-69)
                            const TriShade *sh = c_tshade + tpos;
                            x = o + d * to;
                            n = ld3(sh->nc);
                            f = ld3(sh->col); type = sh->type; lig = ld3(sh->lig);
                            if (TEX) { // getcol(u, v) of a textured triangle (raytrace.cpp:61,65)
                                const unsigned tt = (unsigned)type >> 2;
                                type &= 3;
                                if (tt) {
                                    const double2 *p = (const double2 *)pool_at(S.pool, 2u * S.g_list + 3u * (S.g_list >> 2) + 5u * (unsigned)tpos);
                                    const double2 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3], q4 = p[4];
                                    double u, v;
                                    tri_uv(mk(q0.x, q0.y, q1.x), mk(q1.y, q2.x, q2.y), mk(q3.x, q3.y, q4.x), o, d, u, v);
                                    const unsigned char *tb = as_global(C.tex);
                                    f = tex_color(tb, ((const TexTri *)(tb + TEX_OFF_TRI))[tt - 1], u, v);
                                }
                            }
                            if (primary && sub == 0 && smp == 0 && prim) prim[outk] = sh->gid;
                            ev_shade = true;
                        }
                        const d3 nl = dot(n, d) < 0 ? n : n * -1.0;
                        if (lig.x != 0 || lig.y != 0 || lig.z != 0) {
                            ls[LS_ACC * lss] += thr.x * lig.x; ls[(LS_ACC + 1) * lss] += thr.y * lig.y; ls[(LS_ACC + 2) * lss] += thr.z * lig.z;
                        }
                        const double pm = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z; // raytrace.cpp:69
                        ++depth;
                        go_on = true;
                        if (depth > P.hard_depth) go_on = false;
                        else if (depth > P.rr_depth) { // raytrace.cpp:70-74
                            if (P.rr_mode && rng_next(rng) < pm) f = f * (1 / pm);
                            else go_on = false;
                        }
                        if (go_on) {
                            thr = mult(thr, f);
                            emit_ok = true; emit_one = false;
                            if (type == RT_DIFF) { // raytrace.cpp:77-82
                                const double r1 = 2 * M_PI * rng_next(rng), r2 = rng_next(rng), r2s = sqrt(r2);
                                const d3 w = nl, u = normv(cross(fabs(w.x) > .1 ? mk(0, 1, 0) : mk(1, 0, 0), w)), vv = cross(w, u);
                                double sn, cs;
                                sincos_2pi(r1, sn, cs);
                                rd = normv(u * cs * r2s + vv * sn * r2s + w * sqrt(1 - r2));
                                ro = x - d * EPS;
                                if (NEE) { // one emissive sphere, picked uniformly, sampled over the cone it subtends from ro
                                    const int nlt = C.n_lights;
                                    const double uk = rng_next(rng), eps1 = rng_next(rng), eps2 = rng_next(rng);
                                    int k = (int)(uk * nlt);
                                    k = k > nlt - 1 ? nlt - 1 : k;
                                    const int target = as_global(C.lights)[k];
                                    const SphGeom g = c_sph[target];
                                    const d3 sv0 = mk(g.px, g.py, g.pz) - ro;
                                    const double dist2 = dot(sv0, sv0), rad2 = g.rad * g.rad;
                                    emit_ok = false;
                                    if (!(dist2 > rad2)) { // inside the light: the bounce ray stands in for the sample
                                        emit_one = true;
                                        ls[LS_COL * lss] = __longlong_as_double((long long)target);
                                    } else {
                                        const d3 sw = sv0 * (1 / sqrt(dist2));
                                        const d3 su = normv(cross(fabs(sw.x) > .1 ? mk(0, 1, 0) : mk(1, 0, 0), sw)), sv = cross(sw, su);
                                        const double cos_a_max = sqrt(1 - rad2 / dist2);
                                        const double cos_a = 1 - eps1 + eps1 * cos_a_max, sin_a = sqrt(1 - cos_a * cos_a);
                                        double sp2, cp2;
                                        sincos_2pi(2 * M_PI * eps2, sp2, cp2);
                                        const d3 l = normv(su * cp2 * sin_a + sv * sp2 * sin_a + sw * cos_a);
                                        const double cosl = dot(l, nl);
                                        if (cosl > 0) { // park the bounce ray, trace the shadow ray first
                                            double *q = ls + (size_t)(LS_PEND + PEND_WORDS * np) * lss;
                                            np++;
                                            q[0] = ro.x; q[lss] = ro.y; q[2 * lss] = ro.z;
                                            q[3 * lss] = rd.x; q[4 * lss] = rd.y; q[5 * lss] = rd.z;
                                            q[6 * lss] = thr.x; q[7 * lss] = thr.y; q[8 * lss] = thr.z;
                                            q[9 * lss] = __longlong_as_double((long long)((unsigned long long)(unsigned)depth | ((unsigned long long)into << 32) | (1ull << 33)));
                                            const double omega = 2 * M_PI * (1 - cos_a_max);
                                            thr = mult(thr, ld3(c_sshade[target].lig)) * (cosl * omega * M_1_PI * nlt);
                                            rd = l;
                                            shadow = true;
                                            ls[LS_COL * lss] = __longlong_as_double((long long)target);
                                        }
                                    }
                                }
                            } else if (type == RT_SPEC) { // raytrace.cpp:83-84
                                rd = d - n * 2.0 * dot(n, d);
                                ro = x - d * EPS;
                            } else { // raytrace.cpp:86-94
                                const d3 reflo = x - d * EPS, refld = d - n * 2.0 * dot(n, d);
                                const double nc = 1, nt = 1.7, nnt = into ? nc / nt : nt / nc, ddn = dot(d, nl);
                                const double cos2t = 1 - nnt * nnt * (1 - ddn * ddn);
                                if (cos2t < 0) { ro = reflo; rd = refld; } // total internal reflection
                                else {
                                    const d3 tdir = normv(d * nnt - n * ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))));
                                    const double a2 = nt - nc, b2 = nt + nc, R0 = a2 * a2 / (b2 * b2), c2 = 1 - (into ? -ddn : dot(tdir, n));
                                    const double Re = R0 + (1 - R0) * c2 * c2 * c2 * c2 * c2, Tr = 1 - Re, Pp = .25 + .5 * Re, RP = Re / Pp,
                                                 TP = Tr / (1 - Pp);
                                    const d3 refro = x + d * EPS;
                                    if (depth > P.split_depth) {
                                        if (rng_next(rng) < Pp) { thr = thr * RP; ro = reflo; rd = refld; }
                                        else { thr = thr * TP; ro = refro; rd = tdir; into = !into; }
                                    } else { // both branches: park the reflection, follow the refraction first
                                        double *q = ls + (size_t)(LS_PEND + PEND_WORDS * np) * lss;
                                        np++;
                                        const d3 pt = thr * Re;
                                        q[0] = reflo.x; q[lss] = reflo.y; q[2 * lss] = reflo.z;
                                        q[3 * lss] = refld.x; q[4 * lss] = refld.y; q[5 * lss] = refld.z;
                                        q[6 * lss] = pt.x; q[7 * lss] = pt.y; q[8 * lss] = pt.z;
                                        q[9 * lss] = __longlong_as_double((long long)((unsigned long long)(unsigned)depth | ((unsigned long long)into << 32)));
                                        thr = thr * Tr; ro = refro; rd = tdir; into = !into;
                                    }
                                }
                            }
                            ls[LS_THR * lss] = thr.x; ls[(LS_THR + 1) * lss] = thr.y; ls[(LS_THR + 2) * lss] = thr.z;
                        }
                    } else if (primary && sub == 0 && smp == 0 && prim) prim[outk] = RT_ID_MISS;
                    primary = false;
                    if (!TRACE && !go_on) {
                        if (np > 0) { // resume a parked glass branch
                            --np;
                            double *q = ls + (size_t)(LS_PEND + PEND_WORDS * np) * lss;
                            ro = mk(q[0], q[lss], q[2 * lss]); rd = mk(q[3 * lss], q[4 * lss], q[5 * lss]);
                            ls[LS_THR * lss] = q[6 * lss]; ls[(LS_THR + 1) * lss] = q[7 * lss]; ls[(LS_THR + 2) * lss] = q[8 * lss];
                            const unsigned long long w = (unsigned long long)__double_as_longlong(q[9 * lss]);
                            depth = (int)(unsigned)w; into = ((w >> 32) & 1) != 0;
                            emit_ok = !NEE || ((w >> 33) & 1) == 0; shadow = false; emit_one = false;
                            go_on = true;
                        } else { // the path is complete (raytrace.cpp:203-205)
                            const double is = 1.0 / samps;
                            d3 rsum = mk(ls[LS_RSUM * lss] + ls[LS_ACC * lss] * is, ls[(LS_RSUM + 1) * lss] + ls[(LS_ACC + 1) * lss] * is,
                                         ls[(LS_RSUM + 2) * lss] + ls[(LS_ACC + 2) * lss] * is);
                            if (++smp == samps) { // the sub-pixel is complete: its share of the pixel (raytrace.cpp:205)
                                smp = 0;
                                // The four sub-pixels of a pixel are separate pieces of work of ONE wavefront (see the
                                // hand-out below); whichever finishes last adds the shares in the reference's order, so
                                // the sum does not depend on who that is.  All of it stays inside one CU: the shares go
                                // to L2 and are read back from there (device-scope accesses, no cache maintenance).
                                double *sc = subbuf + ((size_t)outk * 4 + (size_t)sub) * 3;
                                __hip_atomic_store(sc, clamp01(rsum.x) * .25, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(sc + 1, clamp01(rsum.y) * .25, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                __hip_atomic_store(sc + 2, clamp01(rsum.z) * .25, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                rsum = mk(0, 0, 0);
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // the stores have left the CU
                                if (__hip_atomic_fetch_add(subcnt + outk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 3u) {
                                    const double *pc = subbuf + (size_t)outk * 12;
#pragma unroll
                                    for (int k = 0; k < 3; k++) {
                                        const double s0 = __hip_atomic_load(pc + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                                     s1 = __hip_atomic_load(pc + 3 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                                     s2 = __hip_atomic_load(pc + 6 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                                     s3 = __hip_atomic_load(pc + 9 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                        rgb[3 * (size_t)outk + k] = ((s0 + s1) + s2) + s3;
                                    }
                                }
                                need_work = true;
                            }
                            ls[LS_RSUM * lss] = rsum.x; ls[(LS_RSUM + 1) * lss] = rsum.y; ls[(LS_RSUM + 2) * lss] = rsum.z;
                            need_primary = !need_work;
                        }
                    }
                } else need_work = true; // first step of the slot: nothing traced yet

                // ---- slots without work take the next pixel (or ray) from the shared counter
                // (render: in whole pixels, i.e. multiples of four pieces, so that a pixel's sub-pixels stay in this
                // wavefront; what is left over waits in [stash_base, stash_base + stash_n) for the next slots)
                for (;;) {
                    const unsigned long long want = __ballot(need_work);
                    if (want == 0) break;
                    const unsigned k = (unsigned)__popcll(want);
                    const unsigned was_base = stash_get(0), was_n = stash_get(1);
                    const unsigned take = k <= was_n ? 0u : TRACE ? k : ((k - was_n + 15u) & ~15u); // (sixteen at a time: fewer trips to the counter)
                    unsigned base = 0;
                    const int leader = __ffsll((long long)want) - 1;
                    if (take != 0 && lane == leader) base = atomicAdd(counter, take);
                    base = __shfl(base, leader, 64);
                    if (lane == leader) {
                        if (k >= was_n) { stash_put(0, base + (k - was_n)); stash_put(1, take - (k - was_n)); }
                        else { stash_put(0, was_base + k); stash_put(1, was_n - k); }
                    }
                    if (need_work) {
                        const unsigned r = (unsigned)__popcll(want & lane_lt);
                        const unsigned w = r < was_n ? was_base + r : base + (r - was_n);
                        if (w >= n_work) { need_work = false; done = true; }
                        else if (TRACE) { outk = w; need_work = false; go_on = true; const double *rp = as_global(C.rays) + 6 * (size_t)w; ro = ld3(rp); rd = ld3(rp + 3); }
                        else {
                            const unsigned blk = w >> 8, li = (w >> 2) & 63; // a block of 8 x 8 pixels is 256 pieces: pixel, sub-pixel
                            const DevTile T = as_global(C.tiles)[as_global(C.blk_tile)[blk]]; // the tile this block belongs to
                            const int bq = (int)blk - T.blk_first;
                            const int lrow = (bq / T.blk_cols) * 8 + (int)(li >> 3), lcol = (bq % T.blk_cols) * 8 + (int)(li & 7);
                            if (lrow < T.rows && lcol < T.cols) { // ragged edge blocks have unused lanes
                                pxy = ((unsigned)(T.x0 + lrow) << 16) | (unsigned)(T.y0 + lcol);
                                outk = (unsigned)(T.out_off + (long long)lrow * T.cols + lcol);
                                sub = (int)(w & 3); smp = 0;
                                ls[LS_RSUM * lss] = 0; ls[(LS_RSUM + 1) * lss] = 0; ls[(LS_RSUM + 2) * lss] = 0;
                                need_work = false;
                                need_primary = true;
                            }
                        }
                    }
                }
                if (!TRACE && need_primary) { // raytrace.cpp:196-203
                    const d3 camo = ld3(cam.o), camd = ld3(cam.d), ccx = ld3(cam.cx), ccy = ld3(cam.cy);
                    const int sx = sub >> 1, sy = sub & 1, px = (int)(pxy >> 16), py = (int)(pxy & 0xFFFF);
                    const unsigned long long path =
                        ((((unsigned long long)px * (unsigned long long)sizey + (unsigned long long)py) * 2 + sx) * 2 + sy) *
                            (unsigned long long)samps + smp;
                    rng = path_state(P.seed, path);
                    double dx = 0, dy = 0;
                    if (P.jitter) {
                        const double r1 = 2 * rng_next(rng); dx = r1 < 1 ? sqrt(r1) - 1 : 1 - sqrt(2 - r1);
                        const double r2 = 2 * rng_next(rng); dy = r2 < 1 ? sqrt(r2) - 1 : 1 - sqrt(2 - r2);
                    }
                    const d3 dd = ccx * (((sx + .5 + dx) / 2 + px) / sizex - .5) + ccy * (((sy + .5 + dy) / 2 + py) / sizey - .5) + camd;
                    // raytrace.cpp:203: d is normalised in place before the origin is formed
                    // (g++ evaluates the constructor arguments right to left)
                    rd = normv(dd);
                    ro = camo + rd * 140.0;
                    ls[LS_ACC * lss] = 0; ls[(LS_ACC + 1) * lss] = 0; ls[(LS_ACC + 2) * lss] = 0;
                    ls[LS_THR * lss] = 1; ls[(LS_THR + 1) * lss] = 1; ls[(LS_THR + 2) * lss] = 1;
                    depth = 0; into = true; np = 0; primary = true;
                    emit_ok = true; shadow = false; emit_one = false;
                    go_on = true;
                }
                if (!TRACE) ls[LS_RNG * lss] = __longlong_as_double((long long)rng);
                const unsigned npst = (unsigned)depth | ((unsigned)np << 20) | ((unsigned)sub << 23) | ((unsigned)into << 25) |
                                      ((unsigned)primary << 26) | ((unsigned)go_on << 27) | (NEE ? ((unsigned)emit_ok << 28) | ((unsigned)shadow << 29) | ((unsigned)emit_one << 30) : 0u);
                ls[LS_PST * lss] = __longlong_as_double((long long)((unsigned long long)npst | ((unsigned long long)(unsigned)smp << 32)));
                ls[LS_OUT * lss] = __longlong_as_double((long long)((unsigned long long)outk | ((unsigned long long)pxy << 32)));
                if (go_on) { // KdTree::check on the root (raytrace.cpp:46)
                    const double ix = fabs(rd.x) < 1e-150 ? 1e300 : 1.0 / rd.x, iy = fabs(rd.y) < 1e-150 ? 1e300 : 1.0 / rd.y,
                                 iz = fabs(rd.z) < 1e-150 ? 1e300 : 1.0 / rd.z;
                    const float gf = ray_margin_factor(C.cen, C.reach, C.absn, ro.x, ro.y, ro.z, sqrt(rd.x * rd.x + rd.y * rd.y + rd.z * rd.z));
                    sA[sl] = make_double2(ro.x, ro.y);
                    sB[sl] = make_double2(ro.z, ix);
                    sC[sl] = make_double2(iy, iz);
                    ls[LS_V * lss] = rd.x; ls[(LS_V + 1) * lss] = rd.y; ls[(LS_V + 2) * lss] = rd.z;
                    ls[LS_BT * lss] = 1e20;
                    sT[sl] = 0x7F80u; // infinity
                    ls[LS_HA * lss] = __longlong_as_double(-1ll);                   // no triangle
                    sG[sl] = exp_up_put(gf);
                    // beyond the factor (4 * reach + 1 at upload) that the lists' working coefficients cover: the larger ones
                    st = gf <= (float)(3.9 * C.reach) ? 0u : ST_FAR;
                    { // (see the node step)
                        const double imax = fmax(fmax(fabs(ix), fabs(iy)), fabs(iz));
                        const double pad = S.padc + 4e-12 * (fabs(ro.x) + fabs(ro.y) + fabs(ro.z));
                        if ((imax < 4e4) & (pad < 3e-5)) st |= ST_TAME;
                    }
                    sp = 0; slo = 0;
                    if (SPH) { // many spheres: their hierarchy is searched after the kd-tree, from the bottom of the stack
                        const bool unit = fabs(dot(rd, rd) - 1.0) <= 1e-12; // what the boxes' inflation assumes
                        ls[LS_ST * lss] = 1e20;
                        ls[LS_SI * lss] = __longlong_as_double(unit ? -1ll : -2ll);
                        if (unit) { stk[0][sl] = S.sph_root; sp = 1; }
                    }
                    ev_ray = true;
                    // next stays 0: the root node
                } else want_pop = true; // sp is 0: the slot stays idle (done, or out of work)
            }
            RT_SCOUNT(c_rays, (unsigned)__popcll(__ballot(ev_ray)));
            RT_SCOUNT(c_shades, (unsigned)__popcll(__ballot(ev_shade)));
            RT_SCOUNT(c_scans, (unsigned)__popcll(__ballot(ev_scan)));
            if (c_iters >> 24) { flush(); flush_lanes(); } // (long before a 32-bit counter can overflow)
        }

        // ---- common tail for the lanes that ran: push, pop, publish the slot's next item
        RT_TICK(tk2);
        RT_MARK("tail");
#ifdef RT_DRAIN
        if (dr_dry == 0 && __ballot(done)) dr_dry = __builtin_amdgcn_s_memrealtime();
#endif
        RT_EV(11, run && push_item != 0); RT_EV(12, run && push_item != 0 && sp - slo == LDS_STACK); RT_EV(13, run && want_pop);
        RT_EV(17, run && want_pop && sp <= slo && sp != 0); // the popped level is in HBM
        if (run) {
            if (push_item != 0) push(push_item);
            // Pop; a marker (see the node step: "the near subtree is done and there is no own list to scan") that
            // comes up is dealt with on the spot, one per tail -- a second one in a row goes through a node step.
            if (want_pop) next = pop();
            RT_EV(14, is_marker(next)); RT_EV(15, next == IT_END);
            if (is_marker(next)) { // (without inner branches: the level below is read whether or not it is taken)
                const unsigned m = next, fk = (m >> 28) & 3u;
                const bool near_hit = (st & ST_FLAG) != 0, take_far = !near_hit && (fk == 1u || fk == 2u);
                vis_add += near_hit ? 0u : ((st >> ST_NSH) & 7u) + (fk == 3u ? 1u : 0u);
                st = (st & ~ST_N) | (((m >> IT_NSH) & 7u) << ST_NSH);
                st |= ((m >> 30) & 1u) << 8; // ST_FLAG: back in the enclosing frame
                const unsigned e = stk[(sp - 1) & (LDS_STACK - 1)][sl];
                const bool in_ring = sp > slo;
                next = take_far ? after_granule(m) : sp == 0 ? IT_END : in_ring ? e : IT_REFILL;
                sp -= (!take_far && in_ring) ? 1 : 0;
            }
            if (next == IT_END) { // the root's frame ends with the traversal
                vis_add += (st & ST_FLAG) ? 0u : (st >> ST_NSH) & 7u;
                st &= ~ST_N;
            }
        }
        l_vis += vis_add;
        // the slot goes on the stack of the kind of step it needs next
        {
            const unsigned nk = (!run || done) ? (unsigned)K_DONE : next == IT_END ? (unsigned)K_SHADE : is_marker(next) ? (unsigned)K_NODE : (next & IT_AFTER) ? (unsigned)K_CULL : item_kind(next);
            const unsigned long long m0 = __ballot(nk == K_NODE), m1 = __ballot(nk == K_CULL), m2 = __ballot(nk == K_TRI), m3 = __ballot(nk == K_SHADE);
            // (the place on each of the four stacks, then one of them: the masks and counts are scalars, and picking among
            // scalars per lane costs a move and a select each -- more than counting four times)
            auto rank = [](unsigned long long m) __attribute__((always_inline)) {
                return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            };
            const int i0 = 2 * wslot0 + q0 + rank(m0), i1 = 2 * wslot0 + SLOTS - 1 - q1 - rank(m1);
            const int i2 = 2 * wslot0 + SLOTS + q2 + rank(m2), i3 = 2 * wslot0 + 2 * SLOTS - 1 - q3 - rank(m3);
            const int qi = nk == K_NODE ? i0 : nk == K_CULL ? i1 : nk == K_TRI ? i2 : i3;
            if (nk != K_DONE)
                sQ[qi] = make_uint2((unsigned)slot | (((st & (ST_FLAG | ST_N | ST_FAR | ST_TAME)) | (unsigned)sp | ((unsigned)slo << ST_LO)) << 8), next == IT_END ? 0u : next);
            q0 += __popcll(m0); q1 += __popcll(m1); q2 += __popcll(m2); q3 += __popcll(m3);
        }
#ifdef RT_PROFILE
        RT_TICK(tk3);
        p_vote += tk1 - tk0; p_tail += tk3 - tk2;
#pragma unroll
        for (int k = 0; k < 4; k++) if (pick == k) { p_body[k] += tk2 - tk1; p_n[k]++; }
#endif
    }
#ifdef RT_PROFILE
    if (lane == 0) {
        atomicAdd(gstats + 11, p_vote); atomicAdd(gstats + 16, p_tail);
        for (int k = 0; k < 4; k++) { atomicAdd(gstats + 12 + k, p_body[k]); atomicAdd(gstats + 17 + k, p_n[k]); atomicAdd(gstats + 21 + k, p_tri[k]); }
        for (int k = 0; k < 3; k++) atomicAdd(gstats + 25 + k, p_load[k]);
    }
#endif

#ifdef RT_EVENTS
    if (lane == 0)
        for (int k = 0; k < N_EVENTS; k++) { atomicAdd(gstats + 11 + 2 * k, p_ev[k]); atomicAdd(gstats + 12 + 2 * k, p_evl[k]); }
#endif
#ifdef RT_DRAIN
    if (lane == 0) {
        const unsigned long long dr_end = __builtin_amdgcn_s_memrealtime();
        if (dr_dry == 0) dr_dry = dr_end;
        atomicMax(gstats + 11, ~0ull - dr_start); // (the counters start at zero: minima are kept as maxima of the complement)
        atomicMax(gstats + 12, dr_start);
        atomicMax(gstats + 13, ~0ull - dr_dry); atomicAdd(gstats + 14, dr_dry - dr_start); atomicMax(gstats + 15, dr_dry);
        atomicAdd(gstats + 16, dr_end - dr_start); atomicMax(gstats + 17, dr_end); atomicAdd(gstats + 18, 1ull);
        { const unsigned long long b = (dr_end - dr_dry) / 1000ull; atomicAdd(gstats + 19 + (b < DRAIN_BINS ? b : DRAIN_BINS - 1), 1ull); }
        for (int k = 0; k < 4; k++) { atomicAdd(gstats + 19 + DRAIN_BINS + k, (unsigned long long)dr_steps[k]); atomicAdd(gstats + 23 + DRAIN_BINS + k, (unsigned long long)dr_lanes[k]); }
        for (int k = 0; k < 6; k++) atomicAdd(gstats + 27 + DRAIN_BINS + k, (unsigned long long)dr_lone[k]);
        {
            const unsigned wid = blockIdx.x * WPG + wv;
            if (wid < (unsigned)DRAIN_WAVES) {
                unsigned *r = dr_waves[wid];
                r[0] = (unsigned)(dr_end - dr_dry);
                for (int k = 0; k < 4; k++) r[1 + k] = dr_steps[k];
                r[5] = dr_lone[0] + dr_lone[1] + dr_lone[2] + dr_lone[3]; r[6] = dr_lone[4]; r[7] = dr_lone[5];
            }
        }
    }
#endif
    // ---- work counters
    flush();
    for (int off = 32; off > 0; off >>= 1) l_tris += __shfl_down(l_tris, off, 64);
    if (lane == 0 && l_tris) atomicAdd(gstats + ST_TRIS, l_tris);
    flush_lanes();
#undef RT_SCOUNT
}

// Writes a launch's cold arguments to their place in device memory (they arrive as a kernel argument, so nothing is
// copied from host memory and the write is ordered with the launches on the stream).
__global__ void k_put_cold(ColdArgs a, ColdArgs *dst) { *dst = a; }

// A launch's tile table and block-to-tile table are written by kernels as well (the tiles arrive as kernel arguments,
// a chunk per launch): no host memory is read after the call returns, so a changed tile list never makes the host
// wait for the copy of the previous one.
constexpr int TILE_CHUNK = 96; // 3 KiB of kernel arguments
struct TileChunk { DevTile t[TILE_CHUNK]; };
__global__ void k_put_tiles(TileChunk c, int n, DevTile *dst) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = c.t[threadIdx.x];
}
// one workgroup per tile: the tile's 8x8 blocks all name it
__global__ void k_fill_blk(const DevTile *tiles, int *blk) {
    const DevTile t = tiles[blockIdx.x];
    const int n = ((t.rows + 7) / 8) * t.blk_cols;
    for (int b = (int)threadIdx.x; b < n; b += (int)blockDim.x) blk[t.blk_first + b] = (int)blockIdx.x;
}

__global__ void k_fp64(int op, long long n, const double *a, const double *b, double *out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = op == 0 ? a[i] / b[i] : sqrt(a[i]);
}

// ---- the allocation-free entry point's device memory (rt_render_tiles_noalloc)
// Everything a render launch needs beyond the scene and the caller's output buffers -- the work counter, the tile
// table, the block-to-tile table, the sub-pixel shares and their per-pixel counts -- at fixed offsets of ONE array
// in device memory that exists from the moment the library is loaded.  Each GPU holds its own copy.
constexpr size_t NA_MAX_PIXELS = RT_NOALLOC_MAX_PIXELS, NA_MAX_TILES = RT_NOALLOC_MAX_TILES, NA_MAX_BLOCKS = RT_NOALLOC_MAX_BLOCKS;
constexpr size_t na_align(size_t x) { return (x + 255) & ~(size_t)255; }
constexpr size_t NA_OFF_COUNTER = 0, NA_OFF_COLD = 256, NA_OFF_TILES = NA_OFF_COLD + na_align(sizeof(ColdArgs)), NA_OFF_BLK = NA_OFF_TILES + na_align(NA_MAX_TILES * sizeof(DevTile)),
                 NA_OFF_CNT = NA_OFF_BLK + na_align(NA_MAX_BLOCKS * sizeof(int)),
                 NA_OFF_SUB = NA_OFF_CNT + na_align(NA_MAX_PIXELS * sizeof(unsigned)),
                 NA_BYTES = NA_OFF_SUB + na_align(NA_MAX_PIXELS * 12 * sizeof(double));
static __device__ __align__(256) unsigned char scratch[NA_BYTES];

struct NaTiles { int n; rt_tile t[NA_MAX_TILES]; }; // passed by value: no table travels through host memory
static_assert(sizeof(NaTiles) <= 3072, "the tile list must fit the kernel argument buffer");

// One workgroup: writes the tile table and the block-to-tile table of the launch into the scratch array and clears
// the work counter (what launch_render() prepares on the host and copies over).
__global__ void k_na_setup(NaTiles T, unsigned char *scr) {
    DevTile *tiles = (DevTile *)(scr + NA_OFF_TILES);
    int *blk = (int *)(scr + NA_OFF_BLK);
    __shared__ int first[NA_MAX_TILES + 1];
    if (threadIdx.x == 0) {
        long long off = 0;
        int blocks = 0;
        for (int i = 0; i < T.n; i++) {
            DevTile d;
            d.x0 = T.t[i].x0; d.y0 = T.t[i].y0; d.rows = T.t[i].rows; d.cols = T.t[i].cols;
            d.out_off = off;
            d.blk_first = blocks;
            d.blk_cols = (d.cols + 7) / 8;
            tiles[i] = d;
            first[i] = blocks;
            off += (long long)d.rows * d.cols;
            blocks += ((d.rows + 7) / 8) * d.blk_cols;
        }
        first[T.n] = blocks;
        unsigned *counter = (unsigned *)(scr + NA_OFF_COUNTER);
        counter[0] = counter[1] = counter[2] = counter[3] = 0u;
    }
    __syncthreads();
    for (int i = 0; i < T.n; i++)
        for (int b = first[i] + (int)threadIdx.x; b < first[i + 1]; b += (int)blockDim.x) blk[b] = i;
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

// What ONE launch in flight owns beyond the scene: the per-slot state of its rays, its work counter, its cold arguments,
// its tile tables and the sub-pixels' shares.  A scene has two of these, so that two launches on two streams may overlap
// (one frame's last paths with the next frame's first: short frames gain most); the second is allocated when a second
// stream first shows up.
struct LaunchCtx {
    void *ovf = nullptr, *lstate = nullptr;
    ColdArgs *d_cold = nullptr; // where the launch's cold arguments go
    unsigned *d_counter = nullptr; // [0] block counter
    int *d_blk_tile = nullptr; // tile of every 8x8 block of the current tile list
    long long blk_cap = 0;
    DevTile *d_tiles = nullptr;
    int tiles_cap = 0;
    std::vector<DevTile> h_tiles;         // the tile list the device tables were written from
    double *d_sub = nullptr;     // shares of the sub-pixels of the frame being rendered [pixel][4][3]
    unsigned *d_subcnt = nullptr; // finished sub-pixels per pixel
    size_t sub_cap = 0;          // pixels both hold
    hipEvent_t ev_tables = nullptr;       // recorded behind the kernels that last wrote the tile tables
    hipStream_t tables_stream = nullptr;  // the stream they were queued on
    bool tables_valid = false;            // d_tiles / d_blk_tile hold what h_tiles says
    hipEvent_t ev_done = nullptr;         // recorded behind the last launch that used this set
    hipEvent_t ev0 = nullptr, ev1 = nullptr; // around that launch's render kernel, for rt_last_kernel_ms
    bool timed = false;                   // ev0 / ev1 have been recorded
    hipStream_t stream = nullptr;         // the stream of that launch
    bool allocated = false, used = false;
    unsigned long long seq = 0;           // when it was last used
};
constexpr int N_CTX = 2;

struct rt_scene {
    int dev = 0;
    DevScene d{};               // (its ovf / lstate are set per launch, from the launch's LaunchCtx)
    ColdArgs cold{};            // the scene's part of the cold arguments (shading records, scene extent)
    void *pool = nullptr, *tshade = nullptr, *tlist = nullptr, *sph = nullptr, *sshade = nullptr, *tex = nullptr;
    void *lhdr = nullptr, *lights = nullptr;
    LaunchCtx ctx[N_CTX];
    unsigned long long launches = 0;
    int ovf_depth = 0, n_slots = 0; // what sizes a LaunchCtx's per-slot arrays
    int tree_depth = 0;
    int grid = 0, grid_trace = 0, grid_lights = 0; // persistent workgroups of the render / explicit-ray / light-sampling kernel
    unsigned long long *d_stats = nullptr;
    double *d_stage_rgb = nullptr; // rt_render_tile's device-side output buffers, kept between calls
    int *d_stage_prim = nullptr;
    size_t stage_cap = 0;        // pixels both hold
    void *d_trace = nullptr;     // rt_trace_rays' device-side buffers (rays in, four result arrays out), kept between calls
    size_t trace_cap = 0;        // rays they hold
    unsigned char *scratch = nullptr; // this device's copy of the fixed array rt_render_tiles_noalloc works in
    bool deep = false;           // which build of the kernel this scene runs (KernelCfg)
    int slots = 0, wpg = 1;      // its slots per wavefront and wavefronts per workgroup
};

// A scene laid out for the device, in host memory: everything rt_upload_scene builds before it touches a GPU.  Built once,
// uploaded to as many GPUs as there are (from as many threads).
struct rt_prepared {
    std::vector<char> pool;
    std::vector<ListHdr> lhdr;
    std::vector<TriShade> tsh;
    std::vector<uint2> tls;
    std::vector<SphGeom> sg;
    std::vector<SphShade> ss;
    std::vector<unsigned char> texblk;
    std::vector<int> lights;
    DevScene d{};    // the scene's constants (the device pointers are filled per upload)
    ColdArgs cold{};
    int tree_depth = 0, cull_depth = 0;
    bool deep = false, has_tex = false;
};

using KernelFn = void (*)(DevScene, const ColdArgs *, unsigned long long *);
template <bool TRACE, int FEAT>
KernelFn pick_kernel2(bool sph, bool deep) {
    if (deep) return sph ? k_main<TRACE, true, FEAT, CFG_DEEP.slots, CFG_DEEP.ring, CFG_DEEP.wpg, CFG_DEEP.group>
                         : k_main<TRACE, false, FEAT, CFG_DEEP.slots, CFG_DEEP.ring, CFG_DEEP.wpg, CFG_DEEP.group>;
    return sph ? k_main<TRACE, true, FEAT, CFG_SHALLOW.slots, CFG_SHALLOW.ring, CFG_SHALLOW.wpg, CFG_SHALLOW.group>
               : k_main<TRACE, false, FEAT, CFG_SHALLOW.slots, CFG_SHALLOW.ring, CFG_SHALLOW.wpg, CFG_SHALLOW.group>;
}
// (explicit-ray queries never shade: they have only the plain instance)
template <bool TRACE>
KernelFn pick_kernel(bool sph, bool deep, bool tex, bool lights = false) {
#ifndef RT_NO_TEX // (experiment builds without the textured and light-sampling instances)
    if (!TRACE && lights) return pick_kernel2<false, 2>(sph, deep);
    if (!TRACE && tex) return pick_kernel2<false, 1>(sph, deep);
#endif
    return pick_kernel2<TRACE, 0>(sph, deep);
}

namespace {

// Culling hierarchy for one long own list: median splits of the triangle centroids along the
// widest axis, two levels per record, down to single triangles and groups (a range of up to KernelCfg::group triangles
// is not split: they share a box and a triangle step, which measured 7 % faster at 4K with pairs than a box and a step
// each).  Every record holds the boxes of
// its (up to four) children (exact bounds rounded OUTWARD to single precision) and the children as
// work items, so one 128-byte fetch decides four subtrees and a triangle is only fetched when its
// own box is met.  The root of a list's hierarchy is stored in the node's list record.
// `order[first..first+count)` is permuted into the hierarchy's left-to-right leaf order.
struct CullTri { double lo[3], hi[3], c[3]; int src; };

float f_down(double x) { float f = (float)x; if ((double)f > x) f = std::nextafterf(f, -INFINITY); return f; }
float f_up(double x) { float f = (float)x; if ((double)f < x) f = std::nextafterf(f, INFINITY); return f; }

struct CullBuild {
    std::vector<QuadRec> *out;
    std::vector<CullTri> *bt;
    unsigned cull_base, tri_base; // granules
    int first;                    // list position of the range's first triangle
    int max_depth;
    unsigned leaf_stride = 5;     // granules per leaf record
    bool by_src = false;          // leaves are addressed by their source index (spheres) instead of their list position
    int group = 1;                // the most triangles behind one box (KernelCfg::group); 1: every leaf has its own
};

// Splits the range [a, b) (b - a >= 2) in two and returns the split point: the plane between 16
// centroid bins, over the three axes, with the smallest sum of (surface area of the part's box) *
// (triangles in the part); the median along the widest axis where the centroids do not separate.
int cull_split(std::vector<CullTri> &bt, int a, int b) {
    double clo[3] = {1e300, 1e300, 1e300}, chi[3] = {-1e300, -1e300, -1e300};
    for (int i = a; i < b; i++)
        for (int k = 0; k < 3; k++) { clo[k] = std::min(clo[k], bt[i].c[k]); chi[k] = std::max(chi[k], bt[i].c[k]); }
#ifndef RT_SAH_BINS
#define RT_SAH_BINS 16
#endif
    constexpr int NB = RT_SAH_BINS;
    double best = 1e300, best_plane = 0;
    int best_ax = -1;
    if (b - a > 2) {
        for (int ax = 0; ax < 3; ax++) {
            const double w = chi[ax] - clo[ax];
            if (!(w > 0)) continue;
            struct Bin { double lo[3], hi[3]; int n; } bin[NB];
            for (int j = 0; j < NB; j++) { bin[j].n = 0; for (int k = 0; k < 3; k++) { bin[j].lo[k] = 1e300; bin[j].hi[k] = -1e300; } }
            const double sc = NB / w;
            for (int i = a; i < b; i++) {
                int j = (int)((bt[i].c[ax] - clo[ax]) * sc);
                j = j < 0 ? 0 : j >= NB ? NB - 1 : j;
                bin[j].n++;
                for (int k = 0; k < 3; k++) { bin[j].lo[k] = std::min(bin[j].lo[k], bt[i].lo[k]); bin[j].hi[k] = std::max(bin[j].hi[k], bt[i].hi[k]); }
            }
            double rarea[NB];
            int rn[NB];
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            int n = 0;
            auto area = [](const double *l, const double *h) {
                const double x = h[0] - l[0], y = h[1] - l[1], z = h[2] - l[2];
                return x * y + y * z + z * x;
            };
            for (int j = NB - 1; j > 0; j--) { // parts to the right of the plane below bin j
                n += bin[j].n;
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], bin[j].lo[k]); hi[k] = std::max(hi[k], bin[j].hi[k]); }
                rn[j] = n; rarea[j] = n ? area(lo, hi) : 0.0;
            }
            for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
            n = 0;
            for (int j = 1; j < NB; j++) {
                n += bin[j - 1].n;
                for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], bin[j - 1].lo[k]); hi[k] = std::max(hi[k], bin[j - 1].hi[k]); }
                if (n == 0 || rn[j] == 0) continue;
                const double cost = area(lo, hi) * n + rarea[j] * rn[j];
                if (cost < best) { best = cost; best_ax = ax; best_plane = clo[ax] + j / sc; }
            }
        }
    }
    if (best_ax >= 0) {
        const int ax = best_ax;
        const double plane = best_plane;
        const auto it = std::partition(bt.begin() + a, bt.begin() + b, [ax, plane](const CullTri &x) { return x.c[ax] < plane; });
        const int mid = (int)(it - bt.begin());
        if (mid > a && mid < b) return mid;
    }
    int ax = 0;
    if (chi[1] - clo[1] > chi[ax] - clo[ax]) ax = 1;
    if (chi[2] - clo[2] > chi[ax] - clo[ax]) ax = 2;
    const int mid = a + (b - a) / 2;
    std::nth_element(bt.begin() + a, bt.begin() + mid, bt.begin() + b, [ax](const CullTri &x, const CullTri &y) {
        return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.src < y.src);
    });
    return mid;
}

// fills `q` for the range [a, b) (b - a >= 2), builds the records below it, returns the union box
void cull_fill(CullBuild &B, int a, int b, int depth, QuadRec &q, double lo[3], double hi[3]) {
    if (depth > B.max_depth) B.max_depth = depth;
    std::vector<CullTri> &bt = *B.bt;
    int r[5], nr = 0; // child range boundaries
    r[nr++] = a;
    if (b - a > B.group) { // (a list of one triangle gets a record with one child, a list that fits a group one group)
        const int mid = cull_split(bt, a, b);
        const int keep = B.group + 1; // what fits a group stays together: one item behind one box
        if (mid - a >= keep) r[nr++] = cull_split(bt, a, mid);
        r[nr++] = mid;
        if (b - mid >= keep) r[nr++] = cull_split(bt, mid, b);
    }
    r[nr] = b;
    for (int k = 0; k < 3; k++) { lo[k] = 1e300; hi[k] = -1e300; }
    double cl[4][3], ch[4][3];
    memset(&q, 0, sizeof(q));
    for (int c = 0; c < nr; c++) {
        const int ca = r[c], cb = r[c + 1];
        if (cb - ca == 1) {
            for (int k = 0; k < 3; k++) { cl[c][k] = bt[ca].lo[k]; ch[c][k] = bt[ca].hi[k]; }
            q.item[c] = IT_TRI | (B.tri_base + B.leaf_stride * (unsigned)(B.by_src ? bt[ca].src : B.first + ca));
        } else if (cb - ca <= B.group) { // a group of triangles behind one box, neighbours in the pool
            for (int k = 0; k < 3; k++) { cl[c][k] = 1e300; ch[c][k] = -1e300; }
            for (int i = ca; i < cb; i++)
                for (int k = 0; k < 3; k++) { cl[c][k] = std::min(cl[c][k], bt[i].lo[k]); ch[c][k] = std::max(ch[c][k], bt[i].hi[k]); }
            q.item[c] = IT_GROUP | ((unsigned)(cb - ca - 2) << 26) | (unsigned)(B.first + ca);
        } else {
            const size_t me = B.out->size();
            B.out->emplace_back();
            QuadRec sub;
            cull_fill(B, ca, cb, depth + 1, sub, cl[c], ch[c]);
            (*B.out)[me] = sub;
            q.item[c] = IT_CULL | (B.cull_base + 4u * (unsigned)me);
        }
        for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], cl[c][k]); hi[k] = std::max(hi[k], ch[c][k]); }
    }
    // quantise: bytes chosen against the device's own expression fmaf(byte, step, base)
    for (int k = 0; k < 3; k++) {
        const float base = f_down(lo[k]), top = f_up(hi[k]);
        // 255 * step + base must reach `top` before rounding; then the rounded value is >= top >= hi
        float step = f_up(((double)top - (double)base) / 255.0);
        if (!(step >= 0.0f) || !std::isfinite(step)) step = 0.0f;
        for (int guard = 0; guard < 8 && step > 0.0f && (double)fmaf(255.0f, step, base) < hi[k]; guard++)
            step = std::nextafterf(step, INFINITY);
        q.base[k] = base; q.step[k] = step;
        for (int c = 0; c < nr; c++) {
            int b0 = 0, b1 = 255;
            if (step > 0.0f) {
                b0 = (int)std::floor((cl[c][k] - (double)base) / (double)step);
                b0 = std::max(0, std::min(255, b0));
                while (b0 > 0 && (double)fmaf((float)b0, step, base) > cl[c][k]) b0--;
                b1 = (int)std::ceil((ch[c][k] - (double)base) / (double)step);
                b1 = std::max(0, std::min(255, b1));
                while (b1 < 255 && (double)fmaf((float)b1, step, base) < ch[c][k]) b1++;
            } else { b0 = 0; b1 = 0; } // a flat union box: every plane is `base` itself (hi == lo == base is checked below)
            q.q[c][k][0] = (unsigned char)b0; q.q[c][k][1] = (unsigned char)b1;
        }
    }
}

// quantises the boxes of the (up to two) children of a node and of its own list; lo/hi[c] are exact bounds,
// has[c] says which exist (c = 2: the list, its box includes what its triangles' boxes were inflated by)
void kid_fill(KidRec &kr, const double lo[3][3], const double hi[3][3], const bool has[3]) {
    memset(&kr, 0, sizeof(kr));
    if (!has[0] && !has[1] && !has[2]) return;
    bool bad[3] = {false, false, false};
    for (int k = 0; k < 3; k++) {
        double l = 1e300, h = -1e300;
        for (int c = 0; c < 3; c++)
            if (has[c]) { l = std::min(l, lo[c][k]); h = std::max(h, hi[c][k]); }
        const float base = f_down(l), top = f_up(h);
        // the step keeps 16 bits: rounded up to the next such value, then raised until 255 steps reach the top
        float st = f_up(((double)top - (double)base) / 255.0);
        unsigned bits;
        memcpy(&bits, &st, 4);
        bits = (bits + 0xFFFFu) & 0xFFFF0000u;
        memcpy(&st, &bits, 4);
        if (!(st >= 0.0f) || !std::isfinite(st)) { st = 0.0f; bits = 0; }
        for (int guard = 0; guard < 8 && st > 0.0f && (double)fmaf(255.0f, st, base) < h; guard++) {
            bits += 0x10000u;
            memcpy(&st, &bits, 4);
        }
        kr.base[k] = base;
        kr.step16[k] = (unsigned short)(bits >> 16);
        for (int c = 0; c < 3; c++) {
            int b0 = 0, b1 = 255;
            if (has[c] && st > 0.0f) {
                b0 = (int)std::floor((lo[c][k] - (double)base) / (double)st);
                b0 = std::max(0, std::min(255, b0));
                while (b0 > 0 && (double)fmaf((float)b0, st, base) > lo[c][k]) b0--;
                b1 = (int)std::ceil((hi[c][k] - (double)base) / (double)st);
                b1 = std::max(0, std::min(255, b1));
                while (b1 < 255 && (double)fmaf((float)b1, st, base) < hi[c][k]) b1++;
                // a plane that cannot be placed outside the bound (non-finite or enormous coordinates) disables the box
                if (!((double)fmaf((float)b0, st, base) <= lo[c][k]) || !((double)fmaf((float)b1, st, base) >= hi[c][k])) bad[c] = true;
            } else if (has[c]) { // flat union box: every plane is `base`; usable only if that is the exact coordinate
                b0 = b1 = 0;
                if (!((double)base <= lo[c][k]) || !((double)base >= hi[c][k])) bad[c] = true;
            }
            kr.q[c][k][0] = (unsigned char)b0; kr.q[c][k][1] = (unsigned char)b1;
        }
    }
    if (bad[0] || bad[1]) kr.flags |= KID_NOKIDS;
    if (has[2] && !bad[2]) kr.flags |= KID_LIST;
}

// fills the list record of a list longer than SMALL_LIST and permutes order[first..first+count)
void build_list(const rt_scene_desc *D, int first, int count, std::vector<int> &order, std::vector<QuadRec> &cull, unsigned cull_base,
                unsigned tri_base, QuadRec &li, ListHdr &hdr, int &max_depth, double g0, double lbox_lo[3], double lbox_hi[3], int group) {
    std::vector<CullTri> bt((size_t)count);
    std::vector<double> kt((size_t)count); // margin coefficient 2 * (1e-9 L^2 + 1e-10) of each triangle, L its longest edge
    double lmax2 = 0;
    for (int j = 0; j < count; j++) {
        double tl2 = 0;
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
            tl2 = std::max(tl2, l2);
        }
        kt[j] = 2.0 * (1e-9 * tl2 + 1e-10) * 1.000001;
    }
    // A few long triangles must not inflate the boxes of all the others: the list's working coefficient is that
    // of its 75th-percentile triangle, and the boxes of the triangles above it are inflated here by the difference
    // times g0, the largest ray factor the working coefficient is used for (other rays use the list's maximum).
    double ktyp;
    {
        std::vector<double> srt(kt);
        const size_t at = (size_t)(0.75 * (double)(count - 1));
        std::nth_element(srt.begin(), srt.begin() + at, srt.end());
        ktyp = srt[at];
    }
    for (int j = 0; j < count; j++)
        if (kt[j] > ktyp) {
            const double grow = (kt[j] - ktyp) * g0 * 1.000001;
            for (int k = 0; k < 3; k++) { bt[j].lo[k] -= grow; bt[j].hi[k] += grow; }
        }
    CullBuild B = {&cull, &bt, cull_base, tri_base, first, 0};
    B.group = group;
    cull_fill(B, 0, count, 1, li, lbox_lo, lbox_hi);
    max_depth = std::max(max_depth, B.max_depth);
    for (int j = 0; j < count; j++) order[first + j] = bt[j].src;
    float kf = f_up(2.0 * (1e-9 * lmax2 + 1e-10) * 1.000001); // the list's largest margin coefficient, rounded up
    if (!(kf >= 0.0f)) kf = INFINITY;
    float kw = f_up(ktyp);
    if (!(kw >= 0.0f) || !std::isfinite(g0)) kw = kf;
    unsigned bits;
    memcpy(&bits, &kw, 4);
    bits = (bits + 0xFFu) & 0xFFFFFF00u; // rounded up to 16 mantissa bits
    memcpy(&kw, &bits, 4);
    unsigned r = 0;
    while (r < 255 && !(std::ldexp((double)kw, (int)r) >= (double)kf)) r++;
    hdr.meta = bits | r;
}

int validate(const rt_scene_desc *D, int *real_depth) {
    if (!D) return fail(RT_E_INVALID, "desc is NULL");
    if (D->n_nodes < 1) return fail(RT_E_INVALID, "n_nodes must be >= 1 (got %d)", D->n_nodes);
    if (D->n_nodes > (1 << 25)) return fail(RT_E_LIMIT, "n_nodes %d exceeds %d", D->n_nodes, 1 << 25);
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
    // Every node but the root has exactly one parent, and the longest root-to-leaf path must not exceed what the
    // caller states: tree_depth alone sizes the per-slot traversal stacks, which the kernel does not bounds-check.
    // (Children follow their parent in pre-order, so one forward pass gives every node's level.)
    {
        std::vector<int> level((size_t)D->n_nodes, 0);
        level[0] = 1;
        int deepest = 1;
        for (int i = 0; i < D->n_nodes; i++) {
            if (level[i] == 0) return fail(RT_E_INVALID, "node %d is not reachable from the root", i);
            const int kids[2] = {D->node_left[i], D->node_right[i]};
            for (int c = 0; c < 2; c++) {
                if (kids[c] < 0) continue;
                if (c == 1 && kids[1] == kids[0]) return fail(RT_E_INVALID, "node %d: both children are node %d", i, kids[c]);
                if (level[kids[c]] != 0) return fail(RT_E_INVALID, "node %d has more than one parent", kids[c]);
                level[kids[c]] = level[i] + 1;
                if (level[kids[c]] > deepest) deepest = level[kids[c]];
            }
        }
        if (D->tree_depth < deepest)
            return fail(RT_E_INVALID, "tree_depth is %d but the longest root-to-leaf path has %d nodes", D->tree_depth, deepest);
        if (real_depth) *real_depth = deepest;
    }
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
    if (D->n_textures < 0) return fail(RT_E_INVALID, "n_textures is negative");
    if (D->n_textures > 0) {
        if (!D->tex_data || !D->tex_offset || !D->tex_sizex || !D->tex_sizey || !D->tri_tex || !D->tri_ts)
            return fail(RT_E_INVALID, "texture arrays missing");
        for (int k = 0; k < D->n_textures; k++)
            if (D->tex_sizex[k] < 1 || D->tex_sizey[k] < 1 || D->tex_offset[k] < 0 || (long long)D->tex_sizex[k] * D->tex_sizey[k] > (1ll << 28))
                return fail(RT_E_INVALID, "texture %d: %d x %d pixels at offset %lld", k, D->tex_sizey[k], D->tex_sizex[k], (long long)D->tex_offset[k]);
        for (int i = 0; i < D->n_tris; i++)
            if (D->tri_tex[i] < -1 || D->tri_tex[i] >= D->n_textures) return fail(RT_E_INVALID, "triangle %d: texture %d of %d", i, D->tri_tex[i], D->n_textures);
    }
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
    if (cam->sizex > 65536 || cam->sizey > 65536) return fail(RT_E_LIMIT, "image size %d x %d exceeds 65536", cam->sizex, cam->sizey);
    if (p->samps < 1) return fail(RT_E_INVALID, "samps must be >= 1");
    if (p->split_depth < 0 || p->split_depth > MAX_SPLIT) return fail(RT_E_LIMIT, "split_depth must be in [0,%d]", MAX_SPLIT);
    if (p->rr_depth < 0) return fail(RT_E_INVALID, "rr_depth must be >= 0");
    if (p->hard_depth < 1 || p->hard_depth > 100000) return fail(RT_E_INVALID, "hard_depth must be in [1,100000]");
    if (p->light_mode != 0 && p->light_mode != 1) return fail(RT_E_INVALID, "light_mode must be 0 or 1");
    return RT_OK;
}

// The render kernel for this scene and these controls, and the largest grid its state arrays allow.
KernelFn render_kernel(const rt_scene *s, const rt_params *p, int *grid_cap) {
    const bool lights = p->light_mode == 1 && s->cold.n_lights > 0; // (without emissive spheres the two modes are the same)
    *grid_cap = lights ? s->grid_lights : s->grid;
    return pick_kernel<false>(s->d.sph_root != 0, s->deep, s->cold.tex != nullptr, lights);
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

// The device address of `scratch` on device `dev` (which must be the current device): in host code the array's name
// is only a handle for hipGetSymbolAddress, and every GPU has its own copy at its own address.  Looked up once per
// device and kept.
constexpr int NA_MAX_DEV = 64;
std::mutex g_scratch_mu;
unsigned char *g_scratch_ptr[NA_MAX_DEV];
int scratch_address(int dev, unsigned char **out) {
    if (dev < 0 || dev >= NA_MAX_DEV) return fail(RT_E_LIMIT, "device %d: rt_render_tiles_noalloc supports devices 0..%d", dev, NA_MAX_DEV - 1);
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void *ptr = nullptr;
        size_t bytes = 0;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        HIP_TRY(hipGetSymbolSize(&bytes, HIP_SYMBOL(scratch)));
        if (!ptr || bytes != NA_BYTES) return fail(RT_E_DEVICE, "scratch array: address %p, %zu bytes (expected %zu)", ptr, bytes, NA_BYTES);
        g_scratch_ptr[dev] = (unsigned char *)ptr;
    }
    *out = g_scratch_ptr[dev];
    return RT_OK;
}

// Per-slot state, work counter and cold-argument block of one set (the tables and the shares grow with the launches).
int alloc_ctx(rt_scene *s, LaunchCtx &c) {
    if (c.allocated) return RT_OK;
    hipError_t e = hipSuccess;
    if (s->ovf_depth > 0) e = hipMalloc(&c.ovf, (size_t)s->ovf_depth * s->n_slots * sizeof(unsigned));
    if (e == hipSuccess) e = hipMalloc(&c.lstate, (size_t)LS_WORDS * s->n_slots * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&c.d_counter, 16);
    if (e == hipSuccess) e = hipMalloc((void **)&c.d_cold, sizeof(ColdArgs));
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_tables, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreate(&c.ev0);
    if (e == hipSuccess) e = hipEventCreate(&c.ev1);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "device allocation failed: %s", hipGetErrorString(e));
    c.allocated = true;
    return RT_OK;
}

// The set of per-launch state for a launch on `stream`: the one last used on that stream (launches on one stream run one
// after the other anyway), else one not in use yet, else the one used longest ago -- whose launch this one then waits for.
// `first_only`: callers that may not allocate, or that use the first set by contract (explicit rays, the entry point that
// allocates nothing).  On return the stream is ordered behind the set's previous launch.
int acquire_ctx(rt_scene *s, hipStream_t stream, bool first_only, LaunchCtx **out) {
    int pick = -1;
    if (first_only) pick = 0;
    for (int k = 0; k < N_CTX && pick < 0; k++)
        if (s->ctx[k].used && s->ctx[k].stream == stream) pick = k;
    for (int k = 0; k < N_CTX && pick < 0; k++)
        if (!s->ctx[k].used) pick = k;
    if (pick < 0) {
        pick = 0;
        for (int k = 1; k < N_CTX; k++)
            if (s->ctx[k].seq < s->ctx[pick].seq) pick = k;
    }
    LaunchCtx &c = s->ctx[pick];
    if (!c.allocated) {
        const int rc = alloc_ctx(s, c);
        if (rc) return rc;
    }
    if (c.used && c.stream != stream) HIP_TRY(hipStreamWaitEvent(stream, c.ev_done, 0));
    c.used = true;
    c.stream = stream;
    c.seq = ++s->launches;
    *out = &c;
    return RT_OK;
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
        // (the 32-bit work counter also takes the overshoot of every slot that finds the work used up: leave room for it)
        if (blocks > 0xF00000LL) return fail(RT_E_LIMIT, "too many pixel blocks in one launch");
    }
    HIP_TRY(hipSetDevice(s->dev));
    LaunchCtx *cp = nullptr;
    rc = acquire_ctx(s, stream, false, &cp);
    if (rc) return rc;
    LaunchCtx &c = *cp;
    if (n_tiles > c.tiles_cap) {
        if (c.d_tiles) HIP_TRY(hipFree(c.d_tiles));
        c.d_tiles = nullptr;
        c.tiles_cap = 0;
        HIP_TRY(hipMalloc((void **)&c.d_tiles, sizeof(DevTile) * (size_t)n_tiles));
        c.tiles_cap = n_tiles;
        c.tables_valid = false;
    }
    if (blocks > c.blk_cap) {
        if (c.d_blk_tile) HIP_TRY(hipFree(c.d_blk_tile));
        c.d_blk_tile = nullptr;
        c.blk_cap = 0;
        HIP_TRY(hipMalloc((void **)&c.d_blk_tile, sizeof(int) * (size_t)blocks));
        c.blk_cap = blocks;
        c.tables_valid = false;
    }
    if (!c.tables_valid || c.h_tiles.size() != ht.size() || memcmp(c.h_tiles.data(), ht.data(), sizeof(DevTile) * ht.size()) != 0) {
        c.tables_valid = false; // until all the kernels are queued
        // (this stream is already ordered behind the set's previous launch, the last reader of the old tables)
        for (int first = 0; first < n_tiles; first += TILE_CHUNK) {
            TileChunk ch;
            const int n = std::min(TILE_CHUNK, n_tiles - first);
            memcpy(ch.t, ht.data() + first, sizeof(DevTile) * (size_t)n);
            hipLaunchKernelGGL(k_put_tiles, dim3(1), dim3(TILE_CHUNK), 0, stream, ch, n, c.d_tiles + first);
        }
        hipLaunchKernelGGL(k_fill_blk, dim3(n_tiles), dim3(64), 0, stream, (const DevTile *)c.d_tiles, c.d_blk_tile);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c.ev_tables, stream));
        c.h_tiles = ht;
        c.tables_stream = stream;
        c.tables_valid = true;
    } else if (stream != c.tables_stream) {
        // the cached tables were written on another stream: this launch must not start before they have landed
        HIP_TRY(hipStreamWaitEvent(stream, c.ev_tables, 0));
    }
    if ((size_t)off > c.sub_cap) {
        if (c.d_sub) HIP_TRY(hipFree(c.d_sub));
        if (c.d_subcnt) HIP_TRY(hipFree(c.d_subcnt));
        c.d_sub = nullptr; c.d_subcnt = nullptr; c.sub_cap = 0;
        HIP_TRY(hipMalloc((void **)&c.d_sub, (size_t)off * 12 * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&c.d_subcnt, (size_t)off * sizeof(unsigned)));
        c.sub_cap = (size_t)off;
    }
    HIP_TRY(hipMemsetAsync(c.d_subcnt, 0, (size_t)off * sizeof(unsigned), stream));
    HIP_TRY(hipMemsetAsync(c.d_counter, 0, 16, stream));
    const DevCamera dc = make_camera(cam);
    int grid_cap = 0;
    const KernelFn kern = render_kernel(s, p, &grid_cap);
    const int grid = (int)std::min<long long>(grid_cap, (blocks * 256 + s->slots * s->wpg - 1) / (s->slots * s->wpg));
    ColdArgs ca = s->cold;
    ca.cam = dc; ca.P = *p;
    ca.tiles = c.d_tiles; ca.blk_tile = c.d_blk_tile; ca.counter = c.d_counter;
    ca.rgb = d_rgb; ca.prim = d_prim; ca.subbuf = c.d_sub; ca.subcnt = c.d_subcnt;
    ca.n_work = (unsigned)(blocks * 256); ca.n_tiles = n_tiles;
    hipLaunchKernelGGL(k_put_cold, dim3(1), dim3(1), 0, stream, ca, c.d_cold);
    HIP_TRY(hipGetLastError());
    DevScene dsc = s->d;
    dsc.ovf = (unsigned *)c.ovf; dsc.lstate = (double *)c.lstate;
    HIP_TRY(hipEventRecord(c.ev0, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * s->wpg), 0, stream, dsc, (const ColdArgs *)c.d_cold, s->d_stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c.ev1, stream));
    HIP_TRY(hipEventRecord(c.ev_done, stream));
    c.timed = true;
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

int rt_prepare_scene(const rt_scene_desc *D, rt_prepared **out) {
    if (!out) return fail(RT_E_INVALID, "out is NULL");
    *out = nullptr;
    int tree_depth = 0; // the tree's real depth (a caller may state more, never less)
    int rc = validate(D, &tree_depth);
    if (rc) return rc;
    std::unique_ptr<rt_prepared> P(new (std::nothrow) rt_prepared());
    if (!P) return fail(RT_E_NOMEM, "out of host memory");
    try {
    // which build of the kernel the scene runs: it also decides how the culling hierarchies are built (KernelCfg::group)
    int deep_from = DEEP_TREE;
    if (const char *e = getenv("RT_HIP_DEEP_TREE")) deep_from = atoi(e); // tuning / test knob: 0 = always the 8-level ring
    const bool deep = tree_depth >= deep_from;
    const int group = deep ? CFG_DEEP.group : CFG_SHALLOW.group;
    static_assert(CFG_DEEP.group >= 2 && CFG_DEEP.group <= 4 && CFG_SHALLOW.group >= 2 && CFG_SHALLOW.group <= 4, "group items hold 2 to 4 triangles");
    if (D->n_tris > (int)GROUP_POS) return fail(RT_E_LIMIT, "n_tris %d exceeds %u", D->n_tris, GROUP_POS);

    // pool layout in 16-byte granules: nodes (4 each), their list records (4 each), their child- and list-box records (3 each),
    // triangles (5), cull records (4); how many cull records the hierarchies take is only known once they are built
    const size_t g_list = 4 * (size_t)D->n_nodes, g_kid = 2 * g_list, g_tri = g_kid + 3 * (size_t)D->n_nodes,
                 g_sph = g_tri + 5 * (size_t)D->n_tris, // copies of the spheres (2 granules each) for the sphere hierarchy's leaves
                 g_cull = (g_sph + 2 * (size_t)D->n_spheres + 6) & ~(size_t)3;
    if (g_cull + 4 > (size_t)IDX28) return fail(RT_E_LIMIT, "scene too large for the 28-bit record index");
    std::vector<char> &pool = P->pool;
    pool.assign(g_cull * 16, 0);
    std::vector<QuadRec> cull;
    std::vector<ListHdr> &lhdr = P->lhdr;
    lhdr.assign((size_t)D->n_nodes, ListHdr{0u, 0u});
    // the ray margin factor up to which a list's working coefficient applies: four times the scene's reach
    // (summed half extents of the vertex box + longest edge), i.e. every origin within about three scene sizes
    double g0 = 1.0;
    {
        double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, l2max = 0;
        for (int g = 0; g < D->n_tris; g++) {
            const double *v = D->tri_verts + 9 * (size_t)g;
            for (int e = 0; e < 3; e++) {
                const double *a = v + 3 * e, *b = v + 3 * ((e + 1) % 3);
                l2max = std::max(l2max, (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
                for (int k = 0; k < 3; k++) {
                    if (g == 0 && e == 0) lo[k] = hi[k] = a[k];
                    lo[k] = std::min(lo[k], a[k]); hi[k] = std::max(hi[k], a[k]);
                }
            }
        }
        g0 += 4.0 * (std::sqrt(l2max) + 0.5 * ((hi[0] - lo[0]) + (hi[1] - lo[1]) + (hi[2] - lo[2])));
    }
    std::vector<int> order((size_t)D->n_list); // device position -> entry of list_tri
    for (int i = 0; i < D->n_list; i++) order[i] = i;
    // The nodes' records and their lists' hierarchies, built in parallel: chunks of nodes are handed to worker threads,
    // each chunk collects its culling records in a vector of its own (numbered from zero), and afterwards the chunks
    // are laid end to end in node order and the record numbers shifted accordingly -- the same layout a single
    // thread walking the nodes in order produces.
    int cull_depth = 0;
    {
        constexpr int CHUNK = 2048;
        const int n_chunks = (D->n_nodes + CHUNK - 1) / CHUNK;
        std::vector<std::vector<QuadRec>> chunk_cull((size_t)n_chunks);
        std::vector<int> chunk_depth((size_t)n_chunks, 0);
        std::atomic<int> next_chunk(0);
        auto work = [&]() {
            for (;;) {
                const int c = next_chunk.fetch_add(1);
                if (c >= n_chunks) break;
                std::vector<QuadRec> &lc = chunk_cull[(size_t)c];
                for (int i = c * CHUNK; i < std::min(D->n_nodes, (c + 1) * CHUNK); i++) {
                    NodeRec n;
                    for (int k = 0; k < 3; k++) { n.mi[k] = D->node_box[6 * (size_t)i + k]; n.ma[k] = D->node_box[6 * (size_t)i + 3 + k]; }
                    n.split = D->node_split[i];
                    n.right = D->node_right[i] > 0 ? 4u * (unsigned)D->node_right[i] : 0u;
                    n.info = (unsigned)D->node_axis[i] | (D->node_left[i] >= 0 ? 4u : 0u) | ((unsigned)D->node_count[i] << 3);
                    QuadRec li;
                    memset(&li, 0, sizeof(li));
                    ListHdr &hdr = lhdr[i];
                    const int cnt = D->node_count[i], first = D->node_first[i];
                    double klo[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, khi[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
                    bool has[3] = {false, false, false};
                    if (cnt > SMALL_LIST) { // (disjoint ranges of `order` per list)
                        build_list(D, first, cnt, order, lc, (unsigned)g_cull, (unsigned)g_tri, li, hdr, chunk_depth[(size_t)c], g0, klo[2], khi[2], group);
                        has[2] = true;
                    } else {
                        hdr.meta = LIST_SHORT | (unsigned)cnt;
                        li.item[0] = IT_TRI | IT_CHAIN | ((unsigned)g_tri + 5u * (unsigned)first);
                    }
                    hdr.right = n.right;
                    {
                        KidRec kr;
                        const int kid[2] = {D->node_left[i] >= 0 ? i + 1 : -1, D->node_right[i] > 0 ? D->node_right[i] : -1};
                        for (int c2 = 0; c2 < 2; c2++) {
                            has[c2] = kid[c2] >= 0;
                            if (has[c2])
                                for (int k = 0; k < 3; k++) { klo[c2][k] = D->node_box[6 * (size_t)kid[c2] + k]; khi[c2][k] = D->node_box[6 * (size_t)kid[c2] + 3 + k]; }
                        }
                        kid_fill(kr, klo, khi, has);
                        if (cnt == 0) kr.flags |= KID_EMPTY;
                        kr.meta = hdr.meta;
                        if (cnt >= 1 && cnt <= group) kr.direct = li.item[0]; // the whole list is one triangle item
                        memcpy(&pool[(g_kid + 3 * (size_t)i) * 16], &kr, 48);
                    }
                    memcpy(&pool[(size_t)i * 64], &n, 64);
                    memcpy(&pool[(g_list + 4 * (size_t)i) * 16], &li, 64);
                }
            }
        };
        int n_threads = 16; // (a one-GPU job's share of the host; the machine may show many more CPUs)
        if (const char *e = getenv("OMP_NUM_THREADS")) { const int v = atoi(e); if (v >= 1 && v <= 256) n_threads = v; }
        n_threads = std::max(1, std::min(n_threads, std::min(n_chunks, (int)std::max(1u, std::thread::hardware_concurrency()))));
        if (n_threads == 1) work();
        else {
            std::vector<std::thread> pool_threads;
            for (int t = 0; t < n_threads; t++) pool_threads.emplace_back(work);
            for (auto &t : pool_threads) t.join();
        }
        // chunks end to end; a culling item names its record by granule, so every such item of a chunk moves by the
        // chunk's start (in the chunk's own records and in its nodes' list records)
        size_t total = 0;
        for (const auto &lc : chunk_cull) total += lc.size();
        cull.reserve(total + 1);
        auto shift_items = [](QuadRec &q, unsigned by) {
            for (int k = 0; k < 4; k++)
                if (q.item[k] != 0 && ((q.item[k] >> 29) & 3u) == 1u) q.item[k] += by;
        };
        for (int c = 0; c < n_chunks; c++) {
            const unsigned by = 4u * (unsigned)cull.size();
            std::vector<QuadRec> &lc = chunk_cull[(size_t)c];
            cull_depth = std::max(cull_depth, chunk_depth[(size_t)c]);
            if (lc.empty()) continue;
            if (by) {
                for (QuadRec &q : lc) shift_items(q, by);
                for (int i = c * CHUNK; i < std::min(D->n_nodes, (c + 1) * CHUNK); i++)
                    if (D->node_count[i] > SMALL_LIST) {
                        QuadRec li;
                        memcpy(&li, &pool[(g_list + 4 * (size_t)i) * 16], 64);
                        shift_items(li, by);
                        memcpy(&pool[(g_list + 4 * (size_t)i) * 16], &li, 64);
                    }
            }
            cull.insert(cull.end(), lc.begin(), lc.end());
            std::vector<QuadRec>().swap(lc);
        }
    }
    // Scenes with many spheres: the same kind of hierarchy over the spheres' boxes, searched after the kd-tree
    // (the reference scans all of them for every ray, sphere.cpp:3-14).  Its root is one more cull record.
    unsigned sph_root = 0;
    const size_t g_sphc = g_cull + 4 * cull.size();
    float sph_k = 0.0f, sph_c = 0.0f;
    double sph_cen[3] = {0, 0, 0}, sph_h = 0;
    if (D->n_spheres >= SPH_HIER_MIN) {
        std::vector<CullTri> bt((size_t)D->n_spheres);
        double rmin = 1e300, rmax = 0, cabs = 0;
        for (int i = 0; i < D->n_spheres; i++) {
            const double r = D->sph_rad[i];
            rmin = std::min(rmin, r); rmax = std::max(rmax, r);
            for (int k = 0; k < 3; k++) {
                bt[i].c[k] = D->sph_pos[3 * i + k];
                bt[i].lo[k] = bt[i].c[k] - fabs(r); bt[i].hi[k] = bt[i].c[k] + fabs(r);
                cabs = std::max(cabs, std::max(fabs(bt[i].lo[k]), fabs(bt[i].hi[k])));
            }
            bt[i].src = i;
            memcpy(&pool[(g_sph + 2 * (size_t)i) * 16], D->sph_pos + 3 * i, 24);
            memcpy(&pool[(g_sph + 2 * (size_t)i) * 16 + 24], &r, 8);
        }
        CullBuild B = {&cull, &bt, (unsigned)g_cull, (unsigned)g_sph, 0, 0};
        B.leaf_stride = 2; B.by_src = true;
        QuadRec root;
        double lo[3], hi[3];
        cull_fill(B, 0, D->n_spheres, 1, root, lo, hi);
        cull_depth = std::max(cull_depth, B.max_depth);
        sph_root = IT_CULL | IT_CHAIN | (unsigned)(g_cull + 4 * cull.size());
        cull.push_back(root);
        for (int k = 0; k < 3; k++) { sph_cen[k] = 0.5 * (lo[k] + hi[k]); sph_h += 0.5 * (hi[k] - lo[k]) * 1.000001 + 1e-300; }
        // inflation = sph_k * G^2 + sph_c: the acceptance bound of the exact test (see the cull step) with a factor
        // of two to spare, plus the single-precision slab test's slack for the spheres' own coordinates
        sph_k = f_up(1.1e-12 / rmin);
        if (!(rmin > 0) || !std::isfinite(sph_k)) sph_k = INFINITY;
        sph_c = f_up(1e-15 * rmax + 16.0 * 5.97e-8 * cabs);
    }
    const size_t n_cull = cull.size();
    const size_t g_end = g_cull + 4 * n_cull + 4;
    if (g_end > (size_t)IDX28) return fail(RT_E_LIMIT, "scene too large for the 28-bit record index");
    if ((size_t)D->n_nodes * 4 > ((size_t)1 << 26)) return fail(RT_E_LIMIT, "too many tree nodes for the 24-bit node field of a stack item");
    pool.resize(g_end * 16, 0);
    if (n_cull) memcpy(&pool[g_cull * 16], cull.data(), n_cull * sizeof(QuadRec));
    std::vector<unsigned char> last((size_t)D->n_list, 0); // device position is the last of its own list
    for (int i = 0; i < D->n_nodes; i++)
        if (D->node_count[i] > 0) last[(size_t)D->node_first[i] + D->node_count[i] - 1] = 1;
    double blo[3] = {0, 0, 0}, bhi[3] = {0, 0, 0}, lg2 = 0; // vertex box and longest edge of the scene
    std::vector<TriShade> &tsh = P->tsh;
    tsh.resize((size_t)D->n_tris);
    std::vector<TexTri> textri;
    std::vector<unsigned> list_first((size_t)D->n_list, 0); // by list position: where its own list starts
    for (int i = 0; i < D->n_nodes; i++)
        for (int k = 0; k < D->node_count[i]; k++) list_first[(size_t)D->node_first[i] + k] = (unsigned)D->node_first[i];
    std::vector<uint2> &tls = P->tls;
    tls.resize((size_t)D->n_list);
    for (int pos = 0; pos < D->n_list; pos++) {
        const int rank = order[pos];
        const int g = D->list_tri[rank];
        const double *v = D->tri_verts + 9 * (size_t)g;
        TriRec t;
        for (int k = 0; k < 3; k++) {
            t.p0[k] = v[k];
            t.e1[k] = v[3 + k] - v[k]; // polygon.h:39
            t.e2[k] = v[6 + k] - v[k]; // polygon.h:40
        }
        for (int e = 0; e < 3; e++) {
            const double *a = v + 3 * e, *b = v + 3 * ((e + 1) % 3);
            lg2 = std::max(lg2, (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
            for (int k = 0; k < 3; k++) {
                if (pos == 0 && e == 0) blo[k] = bhi[k] = a[k];
                blo[k] = std::min(blo[k], a[k]); bhi[k] = std::max(bhi[k], a[k]);
            }
        }
        t.pos = (int)((unsigned)pos | (last[pos] ? TRI_LAST : 0u));
        t.rank = rank;
        tls[pos] = make_uint2(list_first[rank], (unsigned)rank);
        memcpy(&pool[(g_tri + 5 * (size_t)pos) * 16], &t, sizeof(t));
        TriShade &h = tsh[pos];
        for (int k = 0; k < 3; k++) {
            h.nc[k] = D->tri_nc[3 * (size_t)g + k];
            h.col[k] = D->tri_col[3 * (size_t)g + k];
            h.lig[k] = D->tri_lig[3 * (size_t)g + k];
        }
        h.type = D->tri_type[g];
        h.gid = g;
        if (D->n_textures > 0 && D->tri_tex[g] >= 0) { // (images are laid out below, in index order)
            TexTri x;
            for (int k = 0; k < 6; k++) x.ts[k] = D->tri_ts[6 * (size_t)g + k];
            x.off = (unsigned long long)D->tri_tex[g]; // replaced by the image's offset in the block below
            x.sizex = D->tex_sizex[D->tri_tex[g]];
            x.sizey = D->tex_sizey[D->tri_tex[g]];
            textri.push_back(x);
            if (textri.size() >= (1u << 29)) return fail(RT_E_LIMIT, "too many textured triangles");
            h.type |= (int)((unsigned)textri.size() << 2);
        }
    }
    const size_t n_textri = textri.size();
    std::vector<unsigned char> &texblk = P->texblk;
    if (n_textri) {
        std::vector<unsigned long long> img_off((size_t)D->n_textures);
        size_t at = TEX_OFF_TRI + n_textri * sizeof(TexTri);
        for (int k = 0; k < D->n_textures; k++) { img_off[k] = at; at += 3 * (size_t)D->tex_sizex[k] * D->tex_sizey[k]; }
        texblk.resize((at + 15) & ~(size_t)15);
        double powtab[256];
        for (int b = 0; b < 256; b++) powtab[b] = std::pow((double)b / 256, 2.2); // polygon.h:87
        memcpy(texblk.data(), powtab, sizeof(powtab));
        for (TexTri &x : textri) x.off = img_off[(size_t)x.off];
        memcpy(texblk.data() + TEX_OFF_TRI, textri.data(), n_textri * sizeof(TexTri));
        for (int k = 0; k < D->n_textures; k++)
            memcpy(texblk.data() + img_off[k], D->tex_data + D->tex_offset[k], 3 * (size_t)D->tex_sizex[k] * D->tex_sizey[k]);
    }
    std::vector<SphGeom> &sg = P->sg;
    std::vector<SphShade> &ss = P->ss;
    sg.resize((size_t)D->n_spheres);
    ss.resize((size_t)D->n_spheres);
    for (int i = 0; i < D->n_spheres; i++) {
        sg[i].px = D->sph_pos[3 * i]; sg[i].py = D->sph_pos[3 * i + 1]; sg[i].pz = D->sph_pos[3 * i + 2];
        sg[i].rad = D->sph_rad[i];
        for (int k = 0; k < 3; k++) { ss[i].lig[k] = D->sph_lig[3 * i + k]; ss[i].col[k] = D->sph_col[3 * i + k]; }
        ss[i].type = D->sph_type[i];
        ss[i].pad0 = 0; ss[i].pad1 = 0;
    }

    for (int i = 0; i < D->n_spheres; i++) // the emissive spheres (rt_params.light_mode 1 samples them)
        if (ss[i].lig[0] != 0 || ss[i].lig[1] != 0 || ss[i].lig[2] != 0) P->lights.push_back(i);
    P->tree_depth = tree_depth;
    P->cull_depth = cull_depth;
    { // the wavefront-wide list scan's reserve (k_main): three words per level of the deepest hierarchy, generously
        const char *e = getenv("RT_HIP_COOP"); // tuning / test knob: 0 = every list is scanned by its slot alone
        const int reserve = 4 * cull_depth + 8;
        P->d.coop_reserve = (e && atoi(e) == 0) || reserve > 2 * 64 ? 0 : reserve;
    }
    P->deep = deep;
    P->has_tex = n_textri != 0;
    P->cold.n_lights = (int)P->lights.size();
    P->cold.reach = std::sqrt(lg2) * 1.000001;
    P->cold.absn = 0;
    for (int k = 0; k < 3; k++) {
        P->cold.cen[k] = 0.5 * (blo[k] + bhi[k]);
        P->cold.reach += 0.5 * (bhi[k] - blo[k]) * 1.000001;
        P->cold.absn += std::fabs(P->cold.cen[k]) + 0.5 * (bhi[k] - blo[k]);
    }
    double boxsum = 0;
    for (int i = 0; i < D->n_nodes; i++) {
        double b = 0;
        for (int k = 0; k < 6; k++) b += std::fabs(D->node_box[6 * (size_t)i + k]);
        if (!(b <= boxsum)) boxsum = b; // also takes a NaN, which then disables the shortcut
    }
    P->d.padc = 2.5e-5 + 4e-12 * boxsum;
    P->d.padcf = f_up(P->d.padc * 1.000002);
    P->d.g_list = (unsigned)g_list;
    P->d.g_tri = (unsigned)g_tri;
    P->d.g_sph = (unsigned)g_sph;
    P->d.g_sphc = (unsigned)g_sphc;
    P->d.sph_root = sph_root;
    P->d.sph_k = sph_k; P->d.sph_c = sph_c;
    for (int k = 0; k < 3; k++) P->d.sph_cen[k] = sph_cen[k];
    P->d.sph_h = sph_h;
    P->d.fslack = f_up(16.0 * 5.97e-8 * (P->cold.absn + P->cold.reach)); // see the cull step
    P->d.n_sph = D->n_spheres;
    } catch (const std::bad_alloc &) {
        return fail(RT_E_NOMEM, "out of host memory while laying the scene out");
    }
    *out = P.release();
    return RT_OK;
}

int rt_free_prepared(rt_prepared *p) {
    delete p;
    return RT_OK;
}

int rt_upload_prepared(int dev, const rt_prepared *P, rt_scene **out) {
    if (!out) return fail(RT_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!P) return fail(RT_E_INVALID, "prepared scene is NULL");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (dev < 0 || dev >= ndev) return fail(RT_E_INVALID, "device %d not in [0,%d)", dev, ndev);
    HIP_TRY(hipSetDevice(dev));
    const int tree_depth = P->tree_depth, cull_depth = P->cull_depth;
    const unsigned sph_root = P->d.sph_root;
    const bool has_tex = P->has_tex;
    int rc = RT_OK;
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_E_NOMEM, "out of host memory");
    s->dev = dev;
    s->tree_depth = tree_depth;
    rc = upload(&s->pool, P->pool);
    if (!rc) rc = upload(&s->lhdr, P->lhdr);
    if (!rc) rc = upload(&s->tshade, P->tsh);
    if (!rc) rc = upload(&s->tlist, P->tls);
    if (!rc) rc = upload(&s->sph, P->sg);
    if (!rc) rc = upload(&s->sshade, P->ss);
    if (!rc && has_tex) rc = upload(&s->tex, P->texblk);
    if (!rc && !P->lights.empty()) rc = upload(&s->lights, P->lights);
    // this device's copy of the array rt_render_tiles_noalloc works in: resolved here, once, so that entry point looks nothing up
    if (!rc) rc = scratch_address(dev, &s->scratch);
    if (rc) { rt_free_scene(s); return rc; }
    s->d = P->d;       // the scene's constants; the pointers follow
    s->cold = P->cold;
    s->cold.lights = (const int *)s->lights;

    s->deep = P->deep;
    const KernelCfg cfg = s->deep ? CFG_DEEP : CFG_SHALLOW;
    s->slots = cfg.slots;
    s->wpg = cfg.wpg;
    const int LDS_STACK = cfg.ring;
    // persistent grid: as many workgroups as the chip holds at this kernel's occupancy
    hipDeviceProp_t pr;
    {
        const hipError_t pe = hipGetDeviceProperties(&pr, dev);
        if (pe != hipSuccess || pr.multiProcessorCount < 1) {
            rt_free_scene(s);
            return fail(RT_E_DEVICE, "hipGetDeviceProperties(%d): %s", dev, hipGetErrorString(pe));
        }
    }
    // (the instances this scene can run -- render, explicit rays, render with light sampling -- each get the grid that fills the chip at
    // their own occupancy; the per-slot state arrays are sized for the larger one)
    int per_cu = 0, per_cu_t = 0, per_cu_l = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, pick_kernel<false>(sph_root != 0, s->deep, has_tex), WAVE * cfg.wpg, 0) != hipSuccess || per_cu < 1)
        per_cu = 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_t, pick_kernel<true>(sph_root != 0, s->deep, false), WAVE * cfg.wpg, 0) != hipSuccess || per_cu_t < 1)
        per_cu_t = 4;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_l, pick_kernel<false>(sph_root != 0, s->deep, has_tex, true), WAVE * cfg.wpg, 0) != hipSuccess || per_cu_l < 1)
        per_cu_l = 4;
    s->grid = pr.multiProcessorCount * per_cu;
    s->grid_trace = pr.multiProcessorCount * per_cu_t;
    s->grid_lights = pr.multiProcessorCount * per_cu_l;
    const int n_slots = std::max(std::max(s->grid, s->grid_trace), s->grid_lights) * s->wpg * slot_stride(s->slots);
    // one after item per tree level, the far child, one list's hierarchy walk or short list on top
    const int need_depth = tree_depth + 3 * cull_depth + 5 + (sph_root ? 1 : 0);
    const int ovf_depth = need_depth > LDS_STACK ? need_depth - LDS_STACK + 1 : 0; // levels that can move to HBM
    if ((long long)LS_WORDS * n_slots * (long long)sizeof(double) >= (1ll << 32)) { // (the kernel addresses a state word by a 32-bit byte offset)
        rt_free_scene(s);
        return fail(RT_E_LIMIT, "per-slot state of %d slots exceeds 4 GiB", n_slots);
    }
    if ((long long)ovf_depth * n_slots >= (1ll << 31)) {
        rt_free_scene(s);
        return fail(RT_E_LIMIT, "overflow stack of %d levels x %d slots exceeds the 31-bit index", ovf_depth, n_slots);
    }
    if (need_depth > (int)ST_SP) { // the stack pointer is kept in 8 bits
        rt_free_scene(s);
        return fail(RT_E_LIMIT, "tree too deep: traversal stack of %d levels (max %u)", need_depth, ST_SP);
    }
    s->ovf_depth = ovf_depth;
    s->n_slots = n_slots;
    if (alloc_ctx(s, s->ctx[0]) != RT_OK) { // the first set of per-launch state; a second stream brings the second
        rt_free_scene(s);
        return RT_E_DEVICE;
    }
    hipError_t e = hipMalloc((void **)&s->d_stats, N_STATS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(s->d_stats, 0, N_STATS * sizeof(unsigned long long));
    if (e != hipSuccess) {
        rt_free_scene(s);
        return fail(RT_E_DEVICE, "device allocation failed: %s", hipGetErrorString(e));
    }
    s->d.pool = (const char *)s->pool;
    s->cold.tshade = (const TriShade *)s->tshade;
    s->d.tlist = (const uint2 *)s->tlist;
    s->cold.sph = (const SphGeom *)s->sph;
    s->cold.sshade = (const SphShade *)s->sshade;
    s->cold.tex = has_tex ? (const unsigned char *)s->tex : nullptr;
    s->d.ovf = nullptr; s->d.lstate = nullptr; // per launch
    s->d.lhdr = (const ListHdr *)s->lhdr;
    s->d.ovf_depth = ovf_depth;
    s->d.n_slots = n_slots;
    *out = s;
    return RT_OK;
}

int rt_upload_scene(int dev, const rt_scene_desc *D, rt_scene **out) {
    if (!out) return fail(RT_E_INVALID, "out is NULL");
    *out = nullptr;
    rt_prepared *P = nullptr;
    int rc = rt_prepare_scene(D, &P);
    if (rc) return rc;
    rc = rt_upload_prepared(dev, P, out);
    rt_free_prepared(P);
    return rc;
}

int rt_free_scene(rt_scene *s) {
    if (!s) return RT_OK;
    (void)hipSetDevice(s->dev);
    void *ptrs[] = {s->pool, s->tshade, s->tlist, s->sph, s->sshade, s->tex, s->lights, s->lhdr, s->d_stats, s->d_stage_rgb, s->d_stage_prim, s->d_trace};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (LaunchCtx &c : s->ctx) {
        void *own[] = {c.ovf, c.lstate, c.d_counter, c.d_cold, c.d_tiles, c.d_sub, c.d_subcnt, c.d_blk_tile};
        for (void *p : own)
            if (p) (void)hipFree(p);
        if (c.ev_tables) (void)hipEventDestroy(c.ev_tables);
        if (c.ev_done) (void)hipEventDestroy(c.ev_done);
        if (c.ev0) (void)hipEventDestroy(c.ev0);
        if (c.ev1) (void)hipEventDestroy(c.ev1);
    }
    delete s;
    return RT_OK;
}

int rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_tile *tiles, int32_t n_tiles, const rt_params *p,
                           double *d_rgb, int32_t *d_prim_id, void *stream) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (!d_rgb) return fail(RT_E_INVALID, "d_rgb is NULL");
    return launch_render(s, cam, tiles, n_tiles, p, d_rgb, d_prim_id, (hipStream_t)stream);
}

int rt_render_tiles_noalloc(rt_scene *s, const rt_camera *cam, const rt_tile *tiles, int32_t n_tiles, const rt_params *p,
                            double *d_rgb, int32_t *d_prim_id, void *stream_) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (!d_rgb) return fail(RT_E_INVALID, "d_rgb is NULL");
    int rc = check_params(cam, p);
    if (rc) return rc;
    if (!tiles || n_tiles < 1) return fail(RT_E_INVALID, "no tiles");
    if ((size_t)n_tiles > NA_MAX_TILES) return fail(RT_E_LIMIT, "rt_render_tiles_noalloc: %d tiles exceed the maximum of %zu", n_tiles, NA_MAX_TILES);
    NaTiles T; // on the stack, handed to the set-up kernel by value
    T.n = n_tiles;
    long long off = 0, blocks = 0;
    for (int i = 0; i < n_tiles; i++) {
        const rt_tile &t = tiles[i];
        if (t.rows < 1 || t.cols < 1 || t.x0 < 0 || t.y0 < 0 || (long long)t.x0 + t.rows > cam->sizex ||
            (long long)t.y0 + t.cols > cam->sizey)
            return fail(RT_E_INVALID, "tile %d (%d,%d,%d,%d) is empty or outside the %d x %d image", i, t.x0, t.y0, t.rows,
                        t.cols, cam->sizex, cam->sizey);
        T.t[i] = t;
        off += (long long)t.rows * t.cols;
        blocks += (long long)((t.rows + 7) / 8) * ((t.cols + 7) / 8);
        if ((size_t)off > NA_MAX_PIXELS)
            return fail(RT_E_LIMIT, "rt_render_tiles_noalloc: more than %zu pixels in one call", NA_MAX_PIXELS);
        if ((size_t)blocks > NA_MAX_BLOCKS)
            return fail(RT_E_LIMIT, "rt_render_tiles_noalloc: more than %zu blocks of 8 x 8 pixels in one call", NA_MAX_BLOCKS);
    }
    for (int i = n_tiles; i < (int)NA_MAX_TILES; i++) T.t[i] = rt_tile{0, 0, 0, 0};
    HIP_TRY(hipSetDevice(s->dev));
    unsigned char *const scr = s->scratch; // looked up by rt_upload_scene: nothing is resolved here
    hipStream_t stream = (hipStream_t)stream_;
    LaunchCtx *cp = nullptr; // the rays' per-slot state is the first set's, which rt_upload_scene allocated
    rc = acquire_ctx(s, stream, true, &cp);
    if (rc) return rc;
    // the array keeps the previous call's contents: the per-pixel counts must start at zero (the set-up kernel
    // rewrites the tables and the work counter; the shares are written before they are read)
    HIP_TRY(hipMemsetAsync(scr + NA_OFF_CNT, 0, (size_t)off * sizeof(unsigned), stream));
    hipLaunchKernelGGL(k_na_setup, dim3(1), dim3(256), 0, stream, T, scr);
    HIP_TRY(hipGetLastError());
    const DevCamera dc = make_camera(cam);
    int grid_cap = 0;
    const KernelFn kern = render_kernel(s, p, &grid_cap);
    const int grid = (int)std::min<long long>(grid_cap, (blocks * 256 + s->slots * s->wpg - 1) / (s->slots * s->wpg));
    ColdArgs ca = s->cold;
    ca.cam = dc; ca.P = *p;
    ca.tiles = (const DevTile *)(scr + NA_OFF_TILES); ca.blk_tile = (const int *)(scr + NA_OFF_BLK); ca.counter = (unsigned *)(scr + NA_OFF_COUNTER);
    ca.rgb = d_rgb; ca.prim = d_prim_id; ca.subbuf = (double *)(scr + NA_OFF_SUB); ca.subcnt = (unsigned *)(scr + NA_OFF_CNT);
    ca.n_work = (unsigned)(blocks * 256); ca.n_tiles = n_tiles;
    hipLaunchKernelGGL(k_put_cold, dim3(1), dim3(1), 0, stream, ca, (ColdArgs *)(scr + NA_OFF_COLD));
    HIP_TRY(hipGetLastError());
    DevScene dsc = s->d;
    dsc.ovf = (unsigned *)cp->ovf; dsc.lstate = (double *)cp->lstate;
    HIP_TRY(hipEventRecord(cp->ev0, stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * s->wpg), 0, stream, dsc, (const ColdArgs *)(scr + NA_OFF_COLD), s->d_stats);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(cp->ev1, stream));
    HIP_TRY(hipEventRecord(cp->ev_done, stream));
    cp->timed = true;
    return RT_OK;
}

int rt_get_stats(rt_scene *s, rt_stats *stats) {
    if (!s || !stats) return fail(RT_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(s->dev));
    for (LaunchCtx &c : s->ctx) // every launch in flight adds to the same counters
        if (c.used) HIP_TRY(hipEventSynchronize(c.ev_done));
    unsigned long long h[N_STATS];
    HIP_TRY(hipMemcpy(h, s->d_stats, sizeof(h), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(s->d_stats, 0, sizeof(h)));
    stats->rays = h[0]; stats->node_visits = h[1]; stats->tri_tests = h[2]; stats->sph_tests = h[3]; stats->tri_shades = h[4];
    stats->tri_tests_done = h[5]; stats->cull_tests = h[6]; stats->sched_steps = h[7]; stats->sched_lanes = h[8];
    stats->node_fetches = h[9]; stats->sph_tests_done = h[10];
#ifdef RT_EVENTS
    {
        static const char *en[N_EVENTS] = {"node: record fetched", "node: exact check_aabb code", "node: refill", "node: marker", "node: box not met",
                                           "cull: after item", "cull: boxes tested", "cull: no box met", "cull: short list / none", "tri: step",
                                           "tri: frame has a hit", "tail: push", "tail: push moves a level to HBM", "tail: pop", "tail: marker popped",
                                           "tail: traversal ends", "node: goes on without a push", "tail: popped level is in HBM", "cull: list's first record meets no box", "cull: list of one item after a near subtree",
                                           "node: box beyond the best hit", "cull: a met box lies beyond the best hit"};
        const double iters = (double)h[ST_ITERS], rays = (double)h[ST_RAYS];
        for (int k = 0; k < N_EVENTS; k++)
            if (en[k][0])
                fprintf(stderr, "[events] %-34s steps %6.2f%% of all  lanes/ray %7.3f  lanes per such step %5.1f\n", en[k], 100.0 * h[11 + 2 * k] / iters,
                        h[12 + 2 * k] / rays, h[11 + 2 * k] ? (double)h[12 + 2 * k] / h[11 + 2 * k] : 0.0);
    }
#endif
#ifdef RT_DRAIN
    if (h[18]) { // (one launch between two calls: the extremes are those of that launch)
        const double t0 = (double)(~0ull - h[11]), w = (double)h[18], us = 0.01; // 100 MHz ticks
        fprintf(stderr, "[drain] wavefronts %.0f  last start %.1f us  counter dry: first %.1f mean %.1f last %.1f us  end: mean %.1f last %.1f us\n", w,
                ((double)h[12] - t0) * us, ((double)(~0ull - h[13]) - t0) * us, (double)h[14] / w * us, ((double)h[15] - t0) * us,
                (double)h[16] / w * us, ((double)h[17] - t0) * us);
        fprintf(stderr, "[drain] wavefronts by the time from dry to end, bins of 10 us:");
        for (int b = 0; b < DRAIN_BINS; b++) if (h[19 + b]) fprintf(stderr, " %d:%llu", b, h[19 + b]);
        fprintf(stderr, "\n[drain] after dry, per wavefront:");
        { static const char *kn[4] = {"node", "cull", "tri", "shade"};
          for (int k = 0; k < 4; k++) fprintf(stderr, " %s %.1f steps x %.1f lanes", kn[k], (double)h[19 + DRAIN_BINS + k] / w, h[19 + DRAIN_BINS + k] ? (double)h[23 + DRAIN_BINS + k] / h[19 + DRAIN_BINS + k] : 0.0); }
        {
            static unsigned hw[DRAIN_WAVES][8];
            if (hipMemcpyFromSymbol(hw, HIP_SYMBOL(dr_waves), sizeof(hw)) == hipSuccess) {
                std::vector<int> idx(DRAIN_WAVES);
                for (int i = 0; i < DRAIN_WAVES; i++) idx[i] = i;
                std::partial_sort(idx.begin(), idx.begin() + 6, idx.end(), [&](int a, int b) { return hw[a][0] > hw[b][0]; });
                for (int k = 0; k < 6; k++) {
                    const unsigned *r = hw[idx[k]];
                    fprintf(stderr, "\n[drain]   wavefront %4d: %7.1f us from dry to end; after dry node %u cull %u tri %u shade %u steps; %u steps for its last slot alone; %u list scans in %u batches",
                            idx[k], r[0] * 0.01, r[1], r[2], r[3], r[4], r[5], r[6], r[7]);
                }
            }
        }
        fprintf(stderr, "\n[drain] steps for a wavefront's last slot alone, all wavefronts: node %llu cull %llu tri %llu shade %llu; list scans by the wavefront %llu in %llu batches\n",
                h[27 + DRAIN_BINS], h[28 + DRAIN_BINS], h[29 + DRAIN_BINS], h[30 + DRAIN_BINS], h[31 + DRAIN_BINS], h[32 + DRAIN_BINS]);
    }
#endif
#ifdef RT_PROFILE
    {
        static const char *kn[4] = {"node", "cull", "tri", "shade"};
        const unsigned long long *q = h + 11; // vote, four bodies, tail (ticks); steps per kind; triangle step details
        const double steps = (double)(q[6] + q[7] + q[8] + q[9]);
        if (steps > 0) {
            fprintf(stderr, "[profile] steps %.0f  ticks/step: vote %.1f tail %.1f\n", steps, q[0] / steps, q[5] / steps);
            for (int k = 0; k < 4; k++)
                fprintf(stderr, "[profile]   %-5s steps %5.1f%%  body ticks/step %.1f  share of ticks %.1f%%\n", kn[k], 100.0 * q[6 + k] / steps,
                        q[6 + k] ? (double)q[1 + k] / q[6 + k] : 0.0, 100.0 * q[1 + k] / (double)(q[0] + q[5] + q[1] + q[2] + q[3] + q[4]));
            for (int k = 0; k < 3; k++)
                fprintf(stderr, "[profile]   %-5s ticks from the body's start to the arrival of its loads: %.1f per step\n", kn[k], q[6 + k] ? (double)h[25 + k] / q[6 + k] : 0.0);
            fprintf(stderr, "[profile]   triangle tests per ray: short lists %.1f (hit %.2f), behind boxes %.1f (hit %.2f)\n", (double)q[10] / h[0],
                    (double)q[11] / h[0], (double)q[12] / h[0], (double)q[13] / h[0]);
        }
    }
#endif
    return RT_OK;
}

int rt_last_kernel_ms(rt_scene *s, float *ms) {
    if (!s || !ms) return fail(RT_E_INVALID, "NULL argument");
    const LaunchCtx *last = nullptr; // the set of the launch queued last: every set has its own pair of events
    for (const LaunchCtx &c : s->ctx)
        if (c.timed && (!last || c.seq > last->seq)) last = &c;
    if (!last) return fail(RT_E_INVALID, "no render has been launched on this scene");
    HIP_TRY(hipSetDevice(s->dev));
    HIP_TRY(hipEventSynchronize(last->ev1));
    HIP_TRY(hipEventElapsedTime(ms, last->ev0, last->ev1));
    return RT_OK;
}

int rt_render_tile(rt_scene *s, const rt_camera *cam, const rt_tile *tile, const rt_params *p, double *rgb_out,
                   int32_t *prim_id_out, rt_stats *stats) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (!tile || !rgb_out) return fail(RT_E_INVALID, "tile or rgb_out is NULL");
    if (tile->rows < 1 || tile->cols < 1) return fail(RT_E_INVALID, "empty tile");
    HIP_TRY(hipSetDevice(s->dev));
    const size_t npix = (size_t)tile->rows * tile->cols;
    if (npix > s->stage_cap) { // the staging buffers grow to the largest tile seen and stay until rt_free_scene
        if (s->d_stage_rgb) (void)hipFree(s->d_stage_rgb);
        if (s->d_stage_prim) (void)hipFree(s->d_stage_prim);
        s->d_stage_rgb = nullptr; s->d_stage_prim = nullptr; s->stage_cap = 0;
        HIP_TRY(hipMalloc((void **)&s->d_stage_rgb, npix * 3 * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&s->d_stage_prim, npix * sizeof(int)));
        s->stage_cap = npix;
    }
    double *const d_rgb = s->d_stage_rgb;
    int *const d_prim = s->d_stage_prim;
    hipError_t e = hipSuccess;
    int rc = RT_OK;
    if (hipMemset(s->d_stats, 0, N_STATS * sizeof(unsigned long long)) != hipSuccess) rc = fail(RT_E_DEVICE, "hipMemset failed");
    if (!rc) rc = launch_render(s, cam, tile, 1, p, d_rgb, d_prim, nullptr);
    if (!rc) {
        e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(rgb_out, d_rgb, npix * 3 * sizeof(double), hipMemcpyDeviceToHost);
        if (e == hipSuccess && prim_id_out) e = hipMemcpy(prim_id_out, d_prim, npix * sizeof(int), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_E_DEVICE, "render failed: %s", hipGetErrorString(e));
    }
    rt_stats own; // the counters are fetched (and thereby cleared) whether or not the caller wants them
    if (!rc) rc = rt_get_stats(s, stats ? stats : &own);
    return rc;
}

int rt_trace_rays(rt_scene *s, int64_t n, const double *rays, int32_t *tri_id, double *tri_t, int32_t *sph_id, double *sph_t,
                  rt_stats *stats) {
    if (!s) return fail(RT_E_INVALID, "scene is NULL");
    if (n > 0x70000000LL) return fail(RT_E_LIMIT, "too many rays in one call");
    if (n < 0 || (n > 0 && (!rays || !tri_id || !tri_t || !sph_id || !sph_t))) return fail(RT_E_INVALID, "bad ray buffers");
    HIP_TRY(hipSetDevice(s->dev));
    if (n == 0) {
        if (stats) memset(stats, 0, sizeof(*stats));
        return RT_OK;
    }
    LaunchCtx *cp = nullptr;
    const int rc = acquire_ctx(s, nullptr, true, &cp);
    if (rc) return rc;
    // one allocation holds the rays and the four result arrays; it grows to the largest query seen and stays until rt_free_scene
    constexpr size_t PER_RAY = 6 * sizeof(double) + 2 * sizeof(double) + 2 * sizeof(int);
    if ((size_t)n > s->trace_cap) {
        if (s->d_trace) (void)hipFree(s->d_trace);
        s->d_trace = nullptr; s->trace_cap = 0;
        HIP_TRY(hipMalloc(&s->d_trace, (size_t)n * PER_RAY));
        s->trace_cap = (size_t)n;
    }
    double *const d_rays = (double *)s->d_trace, *const d_tt = d_rays + 6 * (size_t)n, *const d_st = d_tt + (size_t)n;
    int *const d_ti = (int *)(d_st + (size_t)n), *const d_si = d_ti + (size_t)n;
    hipError_t e = hipMemcpy(d_rays, rays, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(s->d_stats, 0, N_STATS * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(cp->d_counter, 0, 16);
    if (e == hipSuccess) {
        const long long want = (n + s->slots * s->wpg - 1) / (s->slots * s->wpg);
        const int grid = (int)std::min<long long>(s->grid_trace, want); // stay within the slot count the state arrays were sized for
        const KernelFn kern = pick_kernel<true>(s->d.sph_root != 0, s->deep, false);
        ColdArgs ca = s->cold;
        ca.P = rt_params{1, 0, 0, 0, 1, 0, 0, 0, 0};
        ca.counter = cp->d_counter; ca.n_work = (unsigned)n;
        ca.rays = d_rays; ca.tri_id = d_ti; ca.tri_t = d_tt; ca.sph_id = d_si; ca.sph_t = d_st;
        DevScene dsc = s->d;
        dsc.ovf = (unsigned *)cp->ovf; dsc.lstate = (double *)cp->lstate;
        hipLaunchKernelGGL(k_put_cold, dim3(1), dim3(1), 0, nullptr, ca, cp->d_cold);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVE * s->wpg), 0, nullptr, dsc, (const ColdArgs *)cp->d_cold, s->d_stats);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipEventRecord(cp->ev_done, nullptr);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(tri_id, d_ti, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(tri_t, d_tt, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sph_id, d_si, (size_t)n * sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(sph_t, d_st, (size_t)n * sizeof(double), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(RT_E_DEVICE, "rt_trace_rays: %s", hipGetErrorString(e));
    rt_stats own;
    return rt_get_stats(s, stats ? stats : &own);
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
