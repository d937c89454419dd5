/*
 * rt_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, CPU-only restatement of the reference's per-pixel path: camera ray,
 * kd-tree nearest-triangle query, brute-force sphere query, and the recursive
 * smallpt-style `radiance` integrator.  It exists so that the HIP path can be
 * checked against something that follows the reference line by line.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Reference lines restated (paths relative to the reference checkout):
 *   point.h:11-70          vector algebra, norm() = multiply by 1/sqrt
 *   polygon.h:23-27        face normal nc (6-term expansion, then normalised)
 *   polygon.h:37-60        Moller-Trumbore, two sided, det<eps rejected
 *   polygon.h:61-72        getn (returns nc unless the s[] guard fails)
 *   polygon.h:30-35,73-89  GetImageData / getcol (plain colour, or the texture image)
 *   sphere.h:13-21         ray/sphere roots filtered by >eps
 *   sphere.cpp:3-14        brute-force scan from the LAST sphere down, strict <
 *   kdtree.h:36-76         box init/update and check_aabb
 *   kdtree.cpp:31-96       median-split builder, leaf when <4, straddlers stay
 *   kdtree.cpp:98-142      check_node / check (near, own list, far)
 *   raytrace.cpp:41-95     radiance
 *   raytrace.cpp:155-156   camera basis
 *   raytrace.cpp:196-205   2x2 sub-pixel loop, tent filter, clamp, *0.25
 *   raytrace.cpp:28-39     norm (clamp) and drawPixel (gamma 1/2.2, BGR bytes)
 *
 * PARITY STATUS.  Everything up to and including the nearest-hit query
 * (builder, check_aabb, check, check_node, Polygon::intersect, nc, getn,
 * Sphere::intersect, sphere_intersect) is pinned bit-for-bit against the
 * reference's own object code (oracle/_ref, built from the reference's
 * kdtree.cpp / sphere.cpp / headers) by tests/test_oracle_vs_ref.py and by the
 * golden vectors under tests/golden/.  `radiance`, the pixel loop and drawPixel
 * live in raytrace.cpp, which needs OpenCV headers that this image does
 * not have, so that file cannot be compiled here: for those functions this file
 * is a careful restatement and their parity is UNPINNED.
 *
 * Two things the reference leaves to chance are fixed here by policy, and the HIP
 * path uses the same policy:
 *   - drand48() is the libc 48-bit LCG, but the reference seeds it from time()
 *     and shares one stream over all pixels.  Here every path (pixel, sub-pixel,
 *     sample) restarts that same LCG from a hashed 48-bit state, and draws are
 *     consumed in the order the reference's code consumes them for that path.
 *   - `radiance(refl)*Re + radiance(refr)*Tr` (raytrace.cpp:94) has unspecified
 *     evaluation order; g++ 11.4 evaluates the refraction call first.  The
 *     stream order here is refraction subtree first, then reflection subtree.
 * Additions that the reference does not have (all default to reference behaviour):
 *   rr_depth/rr_mode  (reference: roulette after depth 5)   -> bounded depth runs
 *   split_depth       (reference: both glass branches while depth <= 2)
 *   hard_depth        (reference: unbounded recursion)      -> safety cap
 *   jitter            (reference: always on)                -> id maps
 *   light_mode        (reference: none, light.h is unused)  -> 1 = explicit light sampling:
 *                     at every diffuse hit one emissive sphere, picked uniformly, is sampled
 *                     over the cone it subtends and a shadow ray decides whether it is seen;
 *                     the next hit then does not add an emissive sphere's emission again.
 *                     No counterpart in the reference, so this mode's parity is UNPINNED by
 *                     construction; it converges to the same image as mode 0.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define EPS 1e-5 /* point.h:7 */

typedef struct { double x, y, z; } v3;
typedef struct { v3 o, d; } ray_t;

static inline v3 V(double x, double y, double z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }      /* point.h:29 */
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }      /* point.h:34 */
static inline v3 vscale(v3 a, double b) { return V(a.x * b, a.y * b, a.z * b); }      /* point.h:39 */
static inline v3 vmult(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }     /* point.h:44 */
static inline double vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }   /* point.h:53 */
static inline v3 vcross(v3 a, v3 b) {                                                 /* point.h:58 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 vnorm(v3 a) {                                                        /* point.h:49 */
    return vscale(a, 1 / sqrt(a.x * a.x + a.y * a.y + a.z * a.z));
}

enum { DIFF = 0, SPEC = 1, REFR = 2 }; /* point.h:9 */

typedef struct {
    v3 p[3];   /* points3d */
    v3 n[3];   /* vertex normal minus position (polygon.h:21); zero when the mesh has none */
    v3 nc;     /* face normal */
    v3 col;    /* polygon colour */
    int obj;   /* owning object (its lig/type are used for shading, raytrace.cpp:66) */
    int tex;   /* texture image or -1 (Polygon::tex == NULL) */
    v3 ts[3];  /* texture coordinates of the vertices (Polygon::ts; z = 0) */
} otri;

typedef struct { unsigned char *data; int sizex, sizey; } otex; /* Polygon::tex/sizex/sizey: sizex = image columns, sizey = rows */

typedef struct { v3 lig; int type; int first, count; } oobj;
typedef struct { double rad; v3 pos, lig, col; int type; } osph;

typedef struct onode {
    struct onode *l, *r;
    double mi[3], ma[3];
    int w;
    double split;
    int n;       /* own list length */
    int *list;   /* global triangle ids, original (object, face) order */
} onode;

typedef struct {
    otri *tris; int ntris, captris;
    oobj *objs; int nobjs, capobjs;
    osph *sphs; int nsphs, capsphs;
    otex *texs; int ntexs, captexs;
    onode *root;
    double (*tmi)[3], (*tma)[3]; /* per-triangle boxes (KdTreeNode ctor, kdtree.h:29-34) */
    int nnodes, depth;
} oscene;

typedef struct {
    uint64_t rays, node_visits, tri_tests, sph_tests, tri_shades;
} ostats;

/* ------------------------------------------------------------------ scene */

oscene *orc_scene_new(void) { return (oscene *)calloc(1, sizeof(oscene)); }

static void free_node(onode *t) {
    if (!t) return;
    free_node(t->l); free_node(t->r); free(t->list); free(t);
}

void orc_scene_free(oscene *s) {
    if (!s) return;
    free_node(s->root);
    for (int i = 0; i < s->ntexs; i++) free(s->texs[i].data);
    free(s->tris); free(s->objs); free(s->sphs); free(s->texs); free(s->tmi); free(s->tma);
    free(s);
}

/* One mesh object: ntri triangles (9 doubles each), optional per-triangle colours
 * (3 doubles each, else `col` for all), optional vertex normals (9 doubles each). */
int orc_add_object(oscene *s, const double *verts, int ntri, const double *lig, const double *col,
                   const double *tri_cols, const double *vnormals, int type) {
    if (s->nobjs == s->capobjs) {
        s->capobjs = s->capobjs ? 2 * s->capobjs : 8;
        s->objs = (oobj *)realloc(s->objs, sizeof(oobj) * s->capobjs);
    }
    if (s->ntris + ntri > s->captris) {
        s->captris = 2 * (s->ntris + ntri) + 16;
        s->tris = (otri *)realloc(s->tris, sizeof(otri) * s->captris);
    }
    oobj *o = &s->objs[s->nobjs];
    o->lig = V(lig[0], lig[1], lig[2]); o->type = type; o->first = s->ntris; o->count = ntri;
    for (int i = 0; i < ntri; i++) {
        otri *t = &s->tris[s->ntris + i];
        for (int k = 0; k < 3; k++) {
            t->p[k] = V(verts[9 * i + 3 * k], verts[9 * i + 3 * k + 1], verts[9 * i + 3 * k + 2]);
            if (vnormals) { /* polygon.h:21  n[i] = n_[i] - points3d_[i] */
                v3 nn = V(vnormals[9 * i + 3 * k], vnormals[9 * i + 3 * k + 1], vnormals[9 * i + 3 * k + 2]);
                t->n[k] = vsub(nn, t->p[k]);
            } else t->n[k] = V(0, 0, 0);
        }
        const v3 a = t->p[0], b = t->p[1], c = t->p[2]; /* polygon.h:23-27 */
        t->nc = V((b.y * c.z + c.y * a.z + a.y * b.z - b.y * a.z - a.y * c.z - c.y * b.z),
                  (a.x * c.z + b.x * a.z + c.x * b.z - c.x * a.z - b.x * c.z - a.x * b.z),
                  (a.x * b.y + b.x * c.y + c.x * a.y - c.x * b.y - b.x * a.y - a.x * c.y));
        t->nc = vnorm(t->nc);
        t->col = tri_cols ? V(tri_cols[3 * i], tri_cols[3 * i + 1], tri_cols[3 * i + 2]) : V(col[0], col[1], col[2]);
        t->obj = s->nobjs;
        t->tex = -1;
        for (int k = 0; k < 3; k++) t->ts[k] = V(0, 0, 0);
    }
    s->ntris += ntri;
    s->nobjs++;
    return s->nobjs - 1;
}

/* One texture image: rows x cols BGR pixels, row-major, as loadImage stores it (opencvlib.cpp:21-29);
 * Polygon::sizex = cols, Polygon::sizey = rows (opencvlib.cpp:19-20). Returns the image index. */
int orc_add_texture(oscene *s, const unsigned char *bgr, int rows, int cols) {
    if (s->ntexs == s->captexs) {
        s->captexs = s->captexs ? 2 * s->captexs : 8;
        s->texs = (otex *)realloc(s->texs, sizeof(otex) * s->captexs);
    }
    otex *t = &s->texs[s->ntexs];
    t->data = (unsigned char *)malloc(3 * (size_t)rows * cols);
    memcpy(t->data, bgr, 3 * (size_t)rows * cols);
    t->sizex = cols; t->sizey = rows;
    return s->ntexs++;
}

/* Images and texture coordinates (6 doubles per triangle: ts[0..2].xy) for the triangles of object `obj`. */
void orc_set_textures(oscene *s, int obj, const int *tri_tex, const double *tri_ts) {
    const oobj *o = &s->objs[obj];
    for (int i = 0; i < o->count; i++) {
        otri *t = &s->tris[o->first + i];
        t->tex = tri_tex[i];
        for (int k = 0; k < 3; k++) t->ts[k] = V(tri_ts[6 * i + 2 * k], tri_ts[6 * i + 2 * k + 1], 0);
    }
}

int orc_add_sphere(oscene *s, double rad, const double *pos, const double *lig, const double *col, int type) {
    if (s->nsphs == s->capsphs) {
        s->capsphs = s->capsphs ? 2 * s->capsphs : 16;
        s->sphs = (osph *)realloc(s->sphs, sizeof(osph) * s->capsphs);
    }
    osph *p = &s->sphs[s->nsphs];
    p->rad = rad; p->pos = V(pos[0], pos[1], pos[2]); p->lig = V(lig[0], lig[1], lig[2]);
    p->col = V(col[0], col[1], col[2]); p->type = type;
    return s->nsphs++;
}

/* ------------------------------------------------------------------ builder */

static double dmin(double a, double b) { if (a < b) return a; else return b; } /* kdtree.h:19 */
static double dmax(double a, double b) { if (a > b) return a; else return b; } /* kdtree.h:24 */

/* kdtree.cpp:31-96 with the dead `com` arrays dropped (the live code never reads them). */
static onode *build(oscene *s, const int *pre, int n, int w, int depth) {
    onode *t = (onode *)calloc(1, sizeof(onode));
    s->nnodes++;
    if (depth > s->depth) s->depth = depth;
    t->w = w;
    for (int k = 0; k < 3; k++) { t->mi[k] = 1e10; t->ma[k] = -1e10; } /* kdtree.h:36-40 */
    for (int i = 0; i < n; i++)
        for (int k = 0; k < 3; k++) {                                  /* kdtree.h:52-55 */
            t->mi[k] = dmin(t->mi[k], s->tmi[pre[i]][k]);
            t->ma[k] = dmax(t->ma[k], s->tma[pre[i]][k]);
        }
    if (n < 4) { /* kdtree.cpp:36-40: leaf keeps everything; split stays unset (0 here) */
        t->n = n; t->list = (int *)malloc(sizeof(int) * (n ? n : 1));
        memcpy(t->list, pre, sizeof(int) * n);
        return t;
    }
    t->split = (t->mi[w] + t->ma[w]) * 0.5; /* kdtree.cpp:82 */
    int *vl = (int *)malloc(sizeof(int) * n), *vr = (int *)malloc(sizeof(int) * n);
    t->list = (int *)malloc(sizeof(int) * n);
    int nl = 0, nr = 0;
    for (int i = 0; i < n; i++) { /* kdtree.cpp:85-93 */
        if (s->tma[pre[i]][w] < t->split) vl[nl++] = pre[i];
        else if (s->tmi[pre[i]][w] > t->split) vr[nr++] = pre[i];
        else t->list[t->n++] = pre[i];
    }
    if (nl) t->l = build(s, vl, nl, (w + 1) % 3, depth + 1); /* kdtree.cpp:94 */
    if (nr) t->r = build(s, vr, nr, (w + 1) % 3, depth + 1); /* kdtree.cpp:95 */
    free(vl); free(vr);
    return t;
}

void orc_build(oscene *s) {
    free_node(s->root); s->root = NULL; s->nnodes = 0; s->depth = 0;
    free(s->tmi); free(s->tma);
    int n = s->ntris;
    s->tmi = (double (*)[3])malloc(sizeof(double[3]) * (n ? n : 1));
    s->tma = (double (*)[3])malloc(sizeof(double[3]) * (n ? n : 1));
    int *pre = (int *)malloc(sizeof(int) * (n ? n : 1));
    for (int i = 0; i < n; i++) { /* kdtree.h:29-50 */
        pre[i] = i;
        for (int k = 0; k < 3; k++) { s->tmi[i][k] = 1e10; s->tma[i][k] = -1e10; }
        for (int j = 0; j < 3; j++) {
            const v3 a = s->tris[i].p[j];
            s->tmi[i][0] = dmin(s->tmi[i][0], a.x); s->tmi[i][1] = dmin(s->tmi[i][1], a.y); s->tmi[i][2] = dmin(s->tmi[i][2], a.z);
            s->tma[i][0] = dmax(s->tma[i][0], a.x); s->tma[i][1] = dmax(s->tma[i][1], a.y); s->tma[i][2] = dmax(s->tma[i][2], a.z);
        }
    }
    /* raytrace.cpp:128 builds the tree even for zero objects; the root is then an empty leaf */
    s->root = build(s, pre, n, 0, 1);
    free(pre);
}

int orc_num_nodes(const oscene *s) { return s->nnodes; }
int orc_tree_depth(const oscene *s) { return s->depth; }
int orc_num_tris(const oscene *s) { return s->ntris; }

/* Pre-order dump (node, left subtree, right subtree; own list appended at the node), the
 * order of the reference's own flattening (cuda.cu:68-89). Arrays sized by orc_num_nodes/ntris. */
typedef struct { double *box, *split; int *axis, *left, *right, *first, *count, *list; int nn, nl; } dumpctx;
static int dump(const onode *t, dumpctx *d) {
    if (!t) return -1;
    int me = d->nn++;
    for (int k = 0; k < 3; k++) { d->box[6 * me + k] = t->mi[k]; d->box[6 * me + 3 + k] = t->ma[k]; }
    d->split[me] = t->split; d->axis[me] = t->w; d->first[me] = d->nl; d->count[me] = t->n;
    for (int i = 0; i < t->n; i++) d->list[d->nl++] = t->list[i];
    int l = dump(t->l, d), r = dump(t->r, d);
    d->left[me] = l; d->right[me] = r;
    return me;
}
void orc_tree_dump(const oscene *s, double *box, double *split, int *axis, int *left, int *right,
                   int *first, int *count, int *list) {
    dumpctx d = {box, split, axis, left, right, first, count, list, 0, 0};
    dump(s->root, &d);
}

/* ------------------------------------------------------------------ intersection */

/* polygon.h:37-60 */
static double tri_intersect(const otri *p, const ray_t *r, double *u, double *v) {
    v3 E1 = vsub(p->p[1], p->p[0]);
    v3 E2 = vsub(p->p[2], p->p[0]);
    v3 P = vcross(r->d, E2);
    double det = vdot(E1, P), t;
    v3 T;
    if (det > 0) T = vsub(r->o, p->p[0]);
    else { T = vsub(p->p[0], r->o); det = -det; }
    if (det < EPS) return 1e20;
    *u = vdot(T, P);
    if (*u < 0 || *u > det) return 1e20;
    v3 Q = vcross(T, E1);
    *v = vdot(r->d, Q);
    if (*v < 0 || *u + *v > det) return 1e20;
    t = vdot(E2, Q);
    double fInvDet = 1.0 / det;
    t *= fInvDet; *u *= fInvDet; *v *= fInvDet;
    return t;
}

/* polygon.h:61-72 */
static v3 tri_getn(const otri *p, v3 x) {
    v3 v[3], t[3], ans = V(0, 0, 0);
    double s[3], sum = 0;
    for (int i = 0; i < 3; i++) v[i] = vnorm(vsub(p->p[i], x));
    for (int i = 0; i < 3; i++) t[i] = vcross(v[i], v[(i + 1) % 3]);
    for (int i = 0; i < 3; i++) { s[i] = sqrt(vdot(t[i], t[i])); sum += s[i]; }
    sum *= 0.1;
    if (s[0] > sum || s[1] > sum || s[2] > sum || sum > 250) return p->nc;
    for (int i = 0; i < 3; i++) ans = vadd(ans, vscale(p->n[i], s[(i + 1) % 3]));
    return vscale(ans, (double)(1.0 / (s[0] + s[1] + s[2])));
}

/* polygon.h:30-35.  The reference indexes the image unchecked: coordinates outside [0, 1) can give a negative
 * remainder and an index outside the image (undefined behaviour there).  Policy here and in the HIP path: a value
 * that does not fit an int, or an index outside the image, reads as 0. */
static double image_data(const otex *t, double x, double y, int c) {
    const double fx = x * t->sizex + 0.5, fy = y * t->sizey + 0.5;
    if (!(fabs(fx) < 2147483648.0) || !(fabs(fy) < 2147483648.0)) return 0;
    const long long idx = (long long)((int)fx % t->sizex) * 3 * t->sizey + 3 * (long long)((int)fy % t->sizey) + c;
    if (idx < 0 || idx >= 3ll * t->sizex * t->sizey) return 0;
    return t->data[idx];
}

/* polygon.h:73-89 */
static v3 tri_getcol(const oscene *sc, const otri *p, double u, double v) {
    if (p->tex < 0) return p->col;
    const otex *t = &sc->texs[p->tex];
    v3 s = vadd(vadd(vscale(vsub(p->ts[1], p->ts[0]), u), vscale(vsub(p->ts[2], p->ts[0]), v)), p->ts[0]);
    v3 ans;
    ans.x = image_data(t, s.x, s.y, 2);
    ans.y = image_data(t, s.x, s.y, 1);
    ans.z = image_data(t, s.x, s.y, 0);
    ans.x = pow(ans.x / 256, 2.2); ans.y = pow(ans.y / 256, 2.2); ans.z = pow(ans.z / 256, 2.2);
    return ans;
}

/* sphere.h:13-21 */
static double sph_intersect(const osph *s, const ray_t *r) {
    v3 op = vsub(s->pos, r->o);
    double t, b = vdot(op, r->d), det = b * b - vdot(op, op) + s->rad * s->rad;
    if (det < 0) return 1e20;
    else det = sqrt(det);
    return (t = b - det) > EPS ? t : ((t = b + det) > EPS ? t : 1e20);
}

/* sphere.cpp:3-14 */
static int sphere_intersect(const oscene *sc, const ray_t *r, int *id, double *t, ostats *st) {
    double d;
    *t = 1e20;
    for (int i = sc->nsphs; i--;) {
        st->sph_tests++;
        if ((d = sph_intersect(&sc->sphs[i], r)) && d < *t) { *t = d; *id = i; }
    }
    return *t < 1e10;
}

/* kdtree.h:57-76 */
static int check_aabb(const onode *b, const ray_t *ray) {
    double s[3] = {ray->o.x, ray->o.y, ray->o.z}, v[3] = {ray->d.x, ray->d.y, ray->d.z};
    const double *mi = b->mi, *ma = b->ma;
    if (mi[0] < s[0] + EPS && s[0] < ma[0] + EPS && mi[1] < s[1] + EPS && s[1] < ma[1] + EPS &&
        mi[2] < s[2] + EPS && s[2] < ma[2] + EPS) return 1;
    for (int i = 0; i < 3; i++) {
        if (fabs(v[i]) > EPS) {
            double t = ((v[i] > 0 ? mi[i] : ma[i]) - s[i]) / v[i];
            if (t > EPS) {
                v3 pos = vadd(ray->o, vscale(ray->d, t));
                double p[3] = {pos.x, pos.y, pos.z};
                int j = (i + 1) % 3, k = (i + 2) % 3;
                if (mi[j] < p[j] + EPS && p[j] < ma[j] + EPS && mi[k] < p[k] + EPS && p[k] < ma[k] + EPS) return 1;
            }
        }
    }
    return 0;
}

/* kdtree.cpp:98-109; ids are global triangle ids (object offset + face, cuda.cu:80) */
static int check_node(const oscene *sc, const onode *t, const ray_t *ray, int *id, double *dis, ostats *st) {
    *dis = 1e20;
    for (int l = 0; l < t->n; l++) {
        double tt, u, v;
        st->tri_tests++;
        if ((tt = tri_intersect(&sc->tris[t->list[l]], ray, &u, &v)) < *dis && tt > EPS) { *dis = tt; *id = t->list[l]; }
    }
    return *dis < 1e10;
}

/* kdtree.cpp:111-142 */
static int kd_check(const oscene *sc, const onode *t, const ray_t *ray, int *id, double *dis, ostats *st) {
    *dis = 1e20;
    double dis1 = 1e20;
    st->node_visits++;
    if (!check_aabb(t, ray)) return 0;
    double s[3] = {ray->o.x, ray->o.y, ray->o.z}, v[3] = {ray->d.x, ray->d.y, ray->d.z};
    int id1 = -1, w = t->w;
    int hit = 0;
    if (s[w] < t->split) {
        if (t->l) hit = kd_check(sc, t->l, ray, id, dis, st);
        check_node(sc, t, ray, &id1, &dis1, st);
        if (dis1 < *dis) { *id = id1; *dis = dis1; }
        if (t->r && !hit && v[w] > 0) kd_check(sc, t->r, ray, &id1, &dis1, st);
        if (dis1 < *dis) { *id = id1; *dis = dis1; }
        return *dis < 1e10;
    }
    if (s[w] > t->split) {
        if (t->r) hit = kd_check(sc, t->r, ray, id, dis, st);
        check_node(sc, t, ray, &id1, &dis1, st);
        if (dis1 < *dis) { *id = id1; *dis = dis1; }
        if (t->l && !hit && v[w] < 0) kd_check(sc, t->l, ray, &id1, &dis1, st);
        if (dis1 < *dis) { *id = id1; *dis = dis1; }
        return *dis < 1e10;
    }
    check_node(sc, t, ray, &id1, &dis1, st);
    if (v[w] < 0 && t->l) kd_check(sc, t->l, ray, id, dis, st);
    if (v[w] > 0 && t->r) kd_check(sc, t->r, ray, id, dis, st);
    if (dis1 < *dis) { *id = id1; *dis = dis1; }
    return *dis < 1e10;
}

/* Nearest-hit queries for n rays (6 doubles each). tri_id = global triangle id or -1,
 * sph_id = sphere index or -1; t = 1e20 on a miss, exactly as the reference leaves it. */
void orc_trace(const oscene *sc, int n, const double *rays, int *tri_id, double *tri_t, int *sph_id, double *sph_t,
               ostats *st_out) {
    ostats st = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        ray_t r = {V(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), V(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5])};
        int id = -1, ids = -1; double to = 1e20, ts = 1e20;
        st.rays++;
        int fo = kd_check(sc, sc->root, &r, &id, &to, &st);
        int fs = sphere_intersect(sc, &r, &ids, &ts, &st);
        if (fo && !(ts < to)) st.tri_shades++;
        tri_id[i] = fo ? id : -1; tri_t[i] = to;
        sph_id[i] = fs ? ids : -1; sph_t[i] = ts;
    }
    if (st_out) *st_out = st;
}

/* Exposed pieces for known-answer tests against the reference's object code. */
double orc_tri_intersect(const oscene *sc, int gid, const double *ray, double *u, double *v) {
    ray_t r = {V(ray[0], ray[1], ray[2]), V(ray[3], ray[4], ray[5])};
    *u = 0; *v = 0;
    return tri_intersect(&sc->tris[gid], &r, u, v);
}
void orc_tri_getn(const oscene *sc, int gid, const double *x, double *out) {
    v3 n = tri_getn(&sc->tris[gid], V(x[0], x[1], x[2]));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}
void orc_tri_nc(const oscene *sc, int gid, double *out) {
    out[0] = sc->tris[gid].nc.x; out[1] = sc->tris[gid].nc.y; out[2] = sc->tris[gid].nc.z;
}
void orc_tri_getcol(const oscene *sc, int gid, double u, double v, double *out) {
    v3 c = tri_getcol(sc, &sc->tris[gid], u, v);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}
double orc_sph_intersect(const oscene *sc, int i, const double *ray) {
    ray_t r = {V(ray[0], ray[1], ray[2]), V(ray[3], ray[4], ray[5])};
    return sph_intersect(&sc->sphs[i], &r);
}

int orc_check_aabb(const double *box, const double *ray) {
    onode b; memset(&b, 0, sizeof(b));
    for (int k = 0; k < 3; k++) { b.mi[k] = box[k]; b.ma[k] = box[3 + k]; }
    ray_t r = {V(ray[0], ray[1], ray[2]), V(ray[3], ray[4], ray[5])};
    return check_aabb(&b, &r);
}

/* ------------------------------------------------------------------ RNG policy */

/* libc drand48 recurrence: X' = (0x5DEECE66D X + 0xB) mod 2^48, value X'/2^48. */
typedef struct { uint64_t x; } rng48;
static inline double rng_next(rng48 *r) {
    r->x = (r->x * 0x5DEECE66DULL + 0xBULL) & 0xFFFFFFFFFFFFULL;
    return (double)r->x * (1.0 / 281474976710656.0);
}
/* splitmix64 finaliser of (seed, path id) -> 48-bit start state */
static inline uint64_t path_state(uint64_t seed, uint64_t path) {
    uint64_t z = seed + 0x9E3779B97F4A7C15ULL * (path + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z >> 16;
}
uint64_t orc_path_state(uint64_t seed, uint64_t path) { return path_state(seed, path); }
double orc_rng_draw(uint64_t *state) { rng48 r = {*state}; double d = rng_next(&r); *state = r.x; return d; }

/* ------------------------------------------------------------------ integrator */

typedef struct {
    int32_t samps;       /* samples per sub-pixel (raytrace.cpp:198) */
    int32_t rr_depth;    /* reference 5: roulette once ++depth > rr_depth (raytrace.cpp:70) */
    int32_t rr_mode;     /* 1 = Russian roulette (reference); 0 = stop there and return the emission */
    int32_t split_depth; /* reference 2: trace both glass branches while depth <= split_depth (raytrace.cpp:93) */
    int32_t hard_depth;  /* stop (return emission) once ++depth > hard_depth; reference recursion is unbounded */
    int32_t jitter;      /* 1 = tent-filter draws (reference); 0 = dx=dy=0 and no draws consumed */
    uint64_t seed;
    int32_t light_mode;  /* 0 = reference (emission found by the paths alone); 1 = explicit light sampling */
} oparams;

typedef struct {
    const oscene *sc; const oparams *pr; rng48 rng; ostats st;
    int first_id; /* id of the first ray's hit (triangle gid >=0, sphere -1-i, miss INT32_MIN) */
    int nrays_path;
    const int *lights; int n_lights; /* emissive spheres, ascending index (light_mode 1) */
} octx;

/* Indices of the spheres with a non-zero emission, ascending; the caller frees. */
static int *light_list(const oscene *sc, int *n) {
    int *l = (int *)malloc(sizeof(int) * (sc->nsphs ? sc->nsphs : 1));
    *n = 0;
    for (int i = 0; i < sc->nsphs; i++)
        if (sc->sphs[i].lig.x != 0 || sc->sphs[i].lig.y != 0 || sc->sphs[i].lig.z != 0) l[(*n)++] = i;
    return l;
}

#define MISS_ID (-2147483647 - 1)

/* Explicit light sampling at a diffuse hit (light_mode 1; no counterpart in the reference).  xo is the
 * origin the bounce ray starts from, nl the shading normal.  Three draws, always: which light, then the
 * point in the cone.  Returns 1 if the light could be sampled (no emissive sphere's emission may then
 * be added by the bounce ray), 0 if xo is inside it (the bounce ray then stands in for the sample: it adds
 * that sphere's emission x number of lights, and no other sphere's); *which is the light; *e is its
 * radiance x cosine x solid angle / pi x number of lights when the shadow ray's nearest hit is that
 * light, else 0. */
static int sample_light(octx *c, v3 xo, v3 nl, v3 *e, int *which) {
    const oscene *sc = c->sc;
    *e = V(0, 0, 0);
    int k = (int)(rng_next(&c->rng) * c->n_lights);
    if (k > c->n_lights - 1) k = c->n_lights - 1;
    const double eps1 = rng_next(&c->rng), eps2 = rng_next(&c->rng);
    const int target = c->lights[k];
    const osph *s = &sc->sphs[target];
    *which = target;
    const v3 sv0 = vsub(s->pos, xo);
    const double dist2 = vdot(sv0, sv0), rad2 = s->rad * s->rad;
    if (!(dist2 > rad2)) return 0;
    const v3 sw = vscale(sv0, 1 / sqrt(dist2));
    const v3 su = vnorm(vcross(fabs(sw.x) > .1 ? V(0, 1, 0) : V(1, 0, 0), sw)), sv = vcross(sw, su);
    const double cos_a_max = sqrt(1 - rad2 / dist2);
    const double cos_a = 1 - eps1 + eps1 * cos_a_max, sin_a = sqrt(1 - cos_a * cos_a), phi = 2 * M_PI * eps2;
    const v3 l = vnorm(vadd(vadd(vscale(vscale(su, cos(phi)), sin_a), vscale(vscale(sv, sin(phi)), sin_a)), vscale(sw, cos_a)));
    const double cosl = vdot(l, nl);
    if (!(cosl > 0)) return 1; /* below the horizon: no contribution, no shadow ray */
    ray_t sr = {xo, l};
    double ts, to = 1e20;
    int ids = -1, idt = -1;
    c->st.rays++;
    c->nrays_path++;
    const int fo = kd_check(sc, sc->root, &sr, &idt, &to, &c->st);
    const int fs = sphere_intersect(sc, &sr, &ids, &ts, &c->st);
    if (fs && (ts < to || !fo) && ids == target) {
        const double omega = 2 * M_PI * (1 - cos_a_max);
        *e = vscale(s->lig, cosl * omega * M_1_PI * c->n_lights);
    }
    return 1;
}

/* raytrace.cpp:41-95.  emit (light_mode 1 only; always 1 otherwise) says what an emissive SPHERE at the hit
 * adds: 1 its emission; 0 nothing (the ray leaves a diffuse hit where a light was sampled explicitly);
 * 2 its emission x number of lights if it is sphere `target`, else nothing (that light could not be sampled). */
static v3 radiance(octx *c, ray_t r, int depth, int into, int emit, int target) {
    const oscene *sc = c->sc;
    double ts, to = 1e20;
    int ids = -1, idt = -1;
    int fs, fo;
    c->st.rays++;
    fo = kd_check(sc, sc->root, &r, &idt, &to, &c->st);
    fs = sphere_intersect(sc, &r, &ids, &ts, &c->st);
    int primary = (c->nrays_path++ == 0);
    if (!fo && !fs) { if (primary) c->first_id = MISS_ID; return V(0, 0, 0); }
    v3 x, n, nl, f, lig;
    int type;
    if (ts < to || !fo) {
        const osph *obj = &sc->sphs[ids];
        x = vadd(r.o, vscale(r.d, ts)); n = vnorm(vsub(x, obj->pos)); nl = (vdot(n, r.d) < 0 ? n : vscale(n, -1));
        f = obj->col; type = obj->type; lig = emit == 1 ? obj->lig : emit == 2 && ids == target ? vscale(obj->lig, c->n_lights) : V(0, 0, 0);
        if (primary) c->first_id = -1 - ids;
    } else {
        double u, v;
        const otri *p = &sc->tris[idt];
        tri_intersect(p, &r, &u, &v);
        x = vadd(r.o, vscale(r.d, to));
        n = tri_getn(p, x);
        nl = (vdot(n, r.d) < 0 ? n : vscale(n, -1));
        f = tri_getcol(sc, p, u, v); /* raytrace.cpp:65 */
        type = sc->objs[p->obj].type; lig = sc->objs[p->obj].lig;
        if (primary) c->first_id = idt;
        c->st.tri_shades++;
    }

    double p = f.x > f.y && f.x > f.z ? f.x : f.y > f.z ? f.y : f.z; /* max refl */
    ++depth;
    if (depth > c->pr->hard_depth) return lig; /* not in the reference: bounds the recursion */
    if (depth > c->pr->rr_depth) {
        if (c->pr->rr_mode && rng_next(&c->rng) < p) f = vscale(f, 1 / p);
        else return lig;
    }

    if (type == DIFF) {
        double r1 = 2 * M_PI * rng_next(&c->rng), r2 = rng_next(&c->rng), r2s = sqrt(r2);
        v3 w = nl, u = vnorm(vcross(fabs(w.x) > .1 ? V(0, 1, 0) : V(1, 0, 0), w)), v = vcross(w, u);
        v3 d = vnorm(vadd(vadd(vscale(vscale(u, cos(r1)), r2s), vscale(vscale(v, sin(r1)), r2s)), vscale(w, sqrt(1 - r2))));
        ray_t nr = {vsub(x, vscale(r.d, EPS)), d};
        if (c->pr->light_mode && c->n_lights > 0) {
            v3 e;
            int which;
            const int sampled = sample_light(c, nr.o, nl, &e, &which);
            return vadd(lig, vmult(f, vadd(e, radiance(c, nr, depth, into, sampled ? 0 : 2, which))));
        }
        return vadd(lig, vmult(f, radiance(c, nr, depth, into, 1, -1)));
    } else if (type == SPEC) {
        ray_t nr = {vsub(x, vscale(r.d, EPS)), vsub(r.d, vscale(vscale(n, 2), vdot(n, r.d)))};
        return vadd(lig, vmult(f, radiance(c, nr, depth, into, 1, -1)));
    }

    ray_t reflRay = {vsub(x, vscale(r.d, EPS)), vsub(r.d, vscale(vscale(n, 2), vdot(n, r.d)))};
    double nc = 1, nt = 1.7, nnt = into ? nc / nt : nt / nc, ddn = vdot(r.d, nl), cos2t;
    if ((cos2t = 1 - nnt * nnt * (1 - ddn * ddn)) < 0) /* total internal reflection */
        return vadd(lig, vmult(f, radiance(c, reflRay, depth, into, 1, -1)));
    v3 tdir = vnorm(vsub(vscale(r.d, nnt), vscale(n, ((into ? 1 : -1) * (ddn * nnt + sqrt(cos2t))))));
    double a = nt - nc, b = nt + nc, R0 = a * a / (b * b), cc = 1 - (into ? -ddn : vdot(tdir, n));
    double Re = R0 + (1 - R0) * cc * cc * cc * cc * cc, Tr = 1 - Re, P = .25 + .5 * Re, RP = Re / P, TP = Tr / (1 - P);
    ray_t refrRay = {vadd(x, vscale(r.d, EPS)), tdir};
    if (depth > c->pr->split_depth) {
        if (rng_next(&c->rng) < P) return vadd(lig, vmult(f, vscale(radiance(c, reflRay, depth, into, 1, -1), RP)));
        else return vadd(lig, vmult(f, vscale(radiance(c, refrRay, depth, !into, 1, -1), TP)));
    }
    /* both branches: refraction subtree consumes the stream first (g++ evaluation order) */
    v3 Lt = radiance(c, refrRay, depth, !into, 1, -1);
    v3 Lr = radiance(c, reflRay, depth, into, 1, -1);
    return vadd(lig, vmult(f, vadd(vscale(Lr, Re), vscale(Lt, Tr))));
}

static inline double clamp01(double x) { return x < 0 ? 0 : x > 1 ? 1 : x; } /* raytrace.cpp:28-31 */

typedef struct {
    double o[3], d[3]; /* camera position and (unnormalised) direction, raytrace.cpp:150-151 */
    int32_t sizex, sizey; /* rows, columns (raytrace.cpp:148, 217) */
} ocamera;

/* raytrace.cpp:155-156 */
void orc_camera_basis(const ocamera *cam, double *camd, double *cx, double *cy) {
    v3 d = vnorm(V(cam->d[0], cam->d[1], cam->d[2]));
    v3 vy = V(cam->sizey * .5035 / cam->sizex, 0, 0), vx = vscale(vnorm(vcross(vy, d)), -.5035);
    camd[0] = d.x; camd[1] = d.y; camd[2] = d.z;
    cx[0] = vx.x; cx[1] = vx.y; cx[2] = vx.z; cy[0] = vy.x; cy[1] = vy.y; cy[2] = vy.z;
}

/* The camera ray of raytrace.cpp:203 for an (unnormalised) direction d, as render_pixel forms it. */
void orc_primary_ray(const double *camo, const double *d, double *out) {
    v3 dn = vnorm(V(d[0], d[1], d[2]));
    v3 o = vadd(V(camo[0], camo[1], camo[2]), vscale(dn, 140));
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = dn.x; out[4] = dn.y; out[5] = dn.z;
}

/* One pixel (row x, column y) of raytrace.cpp:196-205. */
static void render_pixel(const oscene *sc, const ocamera *cam, const oparams *pr, const int *lights, int n_lights, int x, int y,
                         double *rgb, int *prim_id, ostats *st) {
    v3 camo = V(cam->o[0], cam->o[1], cam->o[2]);
    v3 camd = vnorm(V(cam->d[0], cam->d[1], cam->d[2]));
    v3 cy = V(cam->sizey * .5035 / cam->sizex, 0, 0), cx = vscale(vnorm(vcross(cy, camd)), -.5035);
    int sizex = cam->sizex, sizey = cam->sizey, samps = pr->samps;
    v3 col = V(0, 0, 0);
    octx c; c.sc = sc; c.pr = pr; memset(&c.st, 0, sizeof(c.st));
    c.lights = lights; c.n_lights = n_lights;
    for (int sx = 0; sx < 2; sx++)
        for (int sy = 0; sy < 2; sy++) {
            v3 r = V(0, 0, 0);
            for (int s = 0; s < samps; s++) {
                uint64_t path = ((((uint64_t)x * (uint64_t)sizey + (uint64_t)y) * 2 + sx) * 2 + sy) * (uint64_t)samps + s;
                c.rng.x = path_state(pr->seed, path);
                double dx = 0, dy = 0;
                if (pr->jitter) {
                    double r1 = 2 * rng_next(&c.rng); dx = r1 < 1 ? sqrt(r1) - 1 : 1 - sqrt(2 - r1);
                    double r2 = 2 * rng_next(&c.rng); dy = r2 < 1 ? sqrt(r2) - 1 : 1 - sqrt(2 - r2);
                }
                v3 d = vadd(vadd(vscale(cx, ((sx + .5 + dx) / 2 + x) / sizex - .5),
                                 vscale(cy, ((sy + .5 + dy) / 2 + y) / sizey - .5)), camd);
                /* raytrace.cpp:203 `Ray(cam.o + d * 140, d.norm())`: g++ evaluates the arguments
                 * right to left, so d is normalised in place BEFORE the origin is formed (pinned by
                 * the ref_primary_ray known-answer test against the reference's own types). */
                v3 dn = vnorm(d);
                ray_t pr_ray = {vadd(camo, vscale(dn, 140)), dn};
                c.nrays_path = 0; c.first_id = MISS_ID;
                v3 L = radiance(&c, pr_ray, 0, 1, 1, -1);
                if (sx == 0 && sy == 0 && s == 0 && prim_id) *prim_id = c.first_id;
                r = vadd(r, vscale(L, 1.0 / samps));
            }
            col = vadd(col, vscale(V(clamp01(r.x), clamp01(r.y), clamp01(r.z)), .25));
        }
    rgb[0] = col.x; rgb[1] = col.y; rgb[2] = col.z;
    st->rays += c.st.rays; st->node_visits += c.st.node_visits; st->tri_tests += c.st.tri_tests; st->sph_tests += c.st.sph_tests;
    st->tri_shades += c.st.tri_shades;
}

/* Render the tile rows [x0,x0+rows) x columns [y0,y0+cols) into rgb[rows*cols*3] (linear,
 * after the per-sub-pixel clamp and *0.25, i.e. `col[]` of raytrace.cpp:205) and prim_id. */
typedef struct {
    const oscene *sc; const ocamera *cam; const oparams *pr; int x0, y0, rows, cols;
    double *rgb; int *prim_id; int next_row; pthread_mutex_t mu; ostats st;
    int *lights; int n_lights;
} job_t;

static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    ostats st = {0, 0, 0, 0, 0};
    const int npix = j->rows * j->cols, chunk = 4;
    for (;;) {
        pthread_mutex_lock(&j->mu);
        int k0 = j->next_row; /* next unrendered pixel (row-major) */
        j->next_row += chunk;
        pthread_mutex_unlock(&j->mu);
        if (k0 >= npix) break;
        for (int k = k0; k < k0 + chunk && k < npix; k++)
            render_pixel(j->sc, j->cam, j->pr, j->lights, j->n_lights, j->x0 + k / j->cols, j->y0 + k % j->cols, j->rgb + 3 * (size_t)k,
                         j->prim_id ? j->prim_id + k : NULL, &st);
    }
    pthread_mutex_lock(&j->mu);
    j->st.rays += st.rays; j->st.node_visits += st.node_visits; j->st.tri_tests += st.tri_tests; j->st.sph_tests += st.sph_tests;
    j->st.tri_shades += st.tri_shades;
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

void orc_render_tile(const oscene *sc, const ocamera *cam, const oparams *pr, int x0, int y0, int rows, int cols,
                     double *rgb, int *prim_id, ostats *st_out, int nthreads) {
    job_t j; memset(&j, 0, sizeof(j));
    j.sc = sc; j.cam = cam; j.pr = pr; j.x0 = x0; j.y0 = y0; j.rows = rows; j.cols = cols; j.rgb = rgb; j.prim_id = prim_id;
    j.lights = light_list(sc, &j.n_lights);
    pthread_mutex_init(&j.mu, NULL);
    if (nthreads <= 1) worker(&j);
    else {
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
        for (int i = 0; i < nthreads; i++) pthread_create(&th[i], NULL, worker, &j);
        for (int i = 0; i < nthreads; i++) pthread_join(th[i], NULL);
        free(th);
    }
    pthread_mutex_destroy(&j.mu);
    free(j.lights);
    if (st_out) *st_out = j.st;
}

/* radiance() for explicit rays, each with its own restarted stream (path id = index). */
void orc_radiance(const oscene *sc, const oparams *pr, int n, const double *rays, double *rgb, ostats *st_out) {
    octx c; c.sc = sc; c.pr = pr; memset(&c.st, 0, sizeof(c.st));
    int nl = 0, *lights = light_list(sc, &nl);
    c.lights = lights; c.n_lights = nl;
    for (int i = 0; i < n; i++) {
        ray_t r = {V(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), V(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5])};
        c.rng.x = path_state(pr->seed, (uint64_t)i);
        c.nrays_path = 0;
        v3 L = radiance(&c, r, 0, 1, 1, -1);
        rgb[3 * i] = L.x; rgb[3 * i + 1] = L.y; rgb[3 * i + 2] = L.z;
    }
    free(lights);
    if (st_out) *st_out = c.st;
}

/* raytrace.cpp:33-39: clamp, gamma 1/2.2, round, BGR byte order. n pixels. */
void orc_draw_pixels(const double *rgb, size_t n, unsigned char *bgr) {
    for (size_t i = 0; i < n; i++) {
        double r = clamp01(rgb[3 * i]), g = clamp01(rgb[3 * i + 1]), b = clamp01(rgb[3 * i + 2]);
        bgr[3 * i + 2] = (unsigned char)(int)(pow(r, 1 / 2.2) * 255 + .5);
        bgr[3 * i + 1] = (unsigned char)(int)(pow(g, 1 / 2.2) * 255 + .5);
        bgr[3 * i + 0] = (unsigned char)(int)(pow(b, 1 / 2.2) * 255 + .5);
    }
}
