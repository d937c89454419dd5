// ref_l3_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Thin extern "C" shim over the REFERENCE's own acceleration / intersection layer.
// It is compiled together with the reference's kdtree.cpp and sphere.cpp *where they
// lie* (see oracle/Makefile: -I$(REF), sources named by path, nothing copied), against
// the CUDA runtime headers that ship inside this image's triton package -- point.h,
// polygon.h and kdtree.h include <cuda.h>/<cuda_runtime.h> only for the
// __host__/__device__ qualifiers.  The result, oracle/_ref/libref_l3.so, is the real
// reference object code for:
//     KdTree::KdTree / create      (kdtree.cpp:6-96)
//     KdTree::check / check_node   (kdtree.cpp:98-142)
//     KdTreeNode::check_aabb       (kdtree.h:57-76)
//     Polygon ctor (nc) / intersect / getn / GetImageData / getcol   (polygon.h:17-89)
//     Sphere::intersect, sphere_intersect    (sphere.h:13-21, sphere.cpp:3-14)
//     Point3D / Ray                          (point.h)
//     Object::readfile -- the OBJ / MTL reader (object.cpp:9-286) with Matrix::Rx/Ry/Rz and
//     Matrix * Point3D (matrix.h:24-52, matrix.cpp): object.cpp and matrix.cpp include no OpenCV
//     header, only the declaration of loadImage, which is defined below as a function that ends
//     the process -- it is reached only for a material with an image (map_Kd / map_Ks), which
//     the OBJ files read through this shim do not have.
// raytrace.cpp (radiance, main) and opencvlib.cpp are NOT part of this build: they need OpenCV,
// which the image lacks.
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "point.h"
#include "polygon.h"
#include "object.h"
#include "sphere.h"
#include "kdtree.h"
#include "rt_binding.h" // integration/rt_binding.h: the reference-side flattening, compiled here against the reference's own headers

struct RefTex { std::vector<unsigned char> data; int sizex, sizey; };
struct RefScene {
    std::vector<RefTex *> texs; // stable addresses: Polygons keep raw pointers into them
    std::vector<Object> objs;
    std::vector<Sphere> spheres;
    KdTree *tree;
    RefScene() : tree(NULL) {}
};

// opencvlib.h's loadImage (opencvlib.cpp:11-30 needs OpenCV): images are outside what this build can pin.
void loadImage(unsigned char *&, int &, int &, const char file[], double) {
    fprintf(stderr, "ref_l3_harness: loadImage(%s) reached -- this build reads OBJ files without images only\n", file);
    abort();
}

static Ray mkray(const double *r) { return Ray(Point3D(r[0], r[1], r[2]), Point3D(r[3], r[4], r[5])); }

extern "C" {

void *ref_new() { return new RefScene(); }
void ref_free(void *h) { delete (RefScene *)h; } // the reference's tree is never freed (kdtree.h:111-115)

// One Object with ntri Polygons built by the reference's Polygon constructor.
// vnormals (9 doubles per triangle) may be NULL -> zeros, texture is always NULL.
int ref_add_object(void *h, const double *verts, int ntri, const double *lig, const double *col,
                   const double *tri_cols, const double *vnormals, int type) {
    RefScene *s = (RefScene *)h;
    s->objs.push_back(Object());
    Object &o = s->objs.back();
    o.lig = Point3D(lig[0], lig[1], lig[2]);
    o.col = Point3D(col[0], col[1], col[2]);
    o.type = (RType)type;
    for (int i = 0; i < ntri; i++) {
        Point3D p[3], n[3], ts[3];
        for (int k = 0; k < 3; k++) {
            p[k] = Point3D(verts[9 * i + 3 * k], verts[9 * i + 3 * k + 1], verts[9 * i + 3 * k + 2]);
            // the constructor stores n_[k] - points3d_[k]; pass p[k] itself for "no normals" so
            // the stored value is exactly zero, as for a mesh whose loader gave none
            n[k] = vnormals ? Point3D(vnormals[9 * i + 3 * k], vnormals[9 * i + 3 * k + 1], vnormals[9 * i + 3 * k + 2]) : p[k];
        }
        Point3D c = tri_cols ? Point3D(tri_cols[3 * i], tri_cols[3 * i + 1], tri_cols[3 * i + 2]) : o.col;
        o.polys.push_back(Polygon(p, n, ts, o.lig, c, o.type, NULL, 0, 0));
    }
    return (int)s->objs.size() - 1;
}

// One image as loadImage leaves it (opencvlib.cpp:19-29): rows x cols BGR bytes, sizex = cols, sizey = rows.
int ref_add_texture(void *h, const unsigned char *bgr, int rows, int cols) {
    RefScene *s = (RefScene *)h;
    RefTex *t = new RefTex();
    t->data.assign(bgr, bgr + 3 * (size_t)rows * cols);
    t->sizex = cols; t->sizey = rows;
    s->texs.push_back(t);
    return (int)s->texs.size() - 1;
}

// As ref_add_object, with the constructor's texture arguments: tri_tex[i] = image or -1, tri_ts[i][6] = ts[0..2].xy.
int ref_add_object_tex(void *h, const double *verts, int ntri, const double *lig, const double *col,
                       const double *tri_cols, const double *vnormals, int type, const int *tri_tex, const double *tri_ts) {
    RefScene *s = (RefScene *)h;
    s->objs.push_back(Object());
    Object &o = s->objs.back();
    o.lig = Point3D(lig[0], lig[1], lig[2]);
    o.col = Point3D(col[0], col[1], col[2]);
    o.type = (RType)type;
    for (int i = 0; i < ntri; i++) {
        Point3D p[3], n[3], ts[3];
        for (int k = 0; k < 3; k++) {
            p[k] = Point3D(verts[9 * i + 3 * k], verts[9 * i + 3 * k + 1], verts[9 * i + 3 * k + 2]);
            n[k] = vnormals ? Point3D(vnormals[9 * i + 3 * k], vnormals[9 * i + 3 * k + 1], vnormals[9 * i + 3 * k + 2]) : p[k];
            ts[k] = Point3D(tri_ts[6 * i + 2 * k], tri_ts[6 * i + 2 * k + 1], 0);
        }
        Point3D c = tri_cols ? Point3D(tri_cols[3 * i], tri_cols[3 * i + 1], tri_cols[3 * i + 2]) : o.col;
        RefTex *t = tri_tex[i] >= 0 ? s->texs[tri_tex[i]] : NULL;
        o.polys.push_back(Polygon(p, n, ts, o.lig, c, o.type, t ? &t->data[0] : NULL, t ? t->sizex : 0, t ? t->sizey : 0));
    }
    return (int)s->objs.size() - 1;
}

// raytrace.cpp:110-127: one more Object, its lig / col / type set BEFORE readfile, then the reference's own reader.
// `mtllib` names are opened relative to the working directory (object.cpp:55).  Returns the object's index, -1 if
// readfile returned false (the object is then removed again).
int ref_readfile(void *h, const char *path, double times, const double *loc, const double *rot, const double *lig,
                 const double *col, int type) {
    RefScene *s = (RefScene *)h;
    s->objs.push_back(Object());
    Object &o = s->objs.back();
    o.lig = Point3D(lig[0], lig[1], lig[2]);
    o.col = Point3D(col[0], col[1], col[2]);
    o.type = (RType)type;
    if (!o.readfile(std::string(path), times, Point3D(loc[0], loc[1], loc[2]), Point3D(rot[0], rot[1], rot[2]))) {
        s->objs.pop_back();
        return -1;
    }
    return (int)s->objs.size() - 1;
}

int ref_obj_num_polys(void *h, int no) { return (int)((RefScene *)h)->objs[no].polys.size(); }

// What readfile made of the file: per Polygon its points3d (9), col (3), lig (3), ts[0..2].xy (6), nc (3), type and
// whether it has an image.
void ref_obj_dump(void *h, int no, double *verts, double *cols, double *ligs, double *ts, double *nc, int *type, int *has_tex) {
    const Object &o = ((RefScene *)h)->objs[no];
    for (size_t i = 0; i < o.polys.size(); i++) {
        const Polygon &p = o.polys[i];
        for (int k = 0; k < 3; k++) {
            verts[9 * i + 3 * k] = p.points3d[k].x; verts[9 * i + 3 * k + 1] = p.points3d[k].y; verts[9 * i + 3 * k + 2] = p.points3d[k].z;
            ts[6 * i + 2 * k] = p.ts[k].x; ts[6 * i + 2 * k + 1] = p.ts[k].y;
        }
        cols[3 * i] = p.col.x; cols[3 * i + 1] = p.col.y; cols[3 * i + 2] = p.col.z;
        ligs[3 * i] = p.lig.x; ligs[3 * i + 1] = p.lig.y; ligs[3 * i + 2] = p.lig.z;
        nc[3 * i] = p.nc.x; nc[3 * i + 1] = p.nc.y; nc[3 * i + 2] = p.nc.z;
        type[i] = (int)p.type;
        has_tex[i] = p.tex != NULL;
    }
}

void ref_poly_getcol(void *h, int no, int nv, double u, double v, double *out) {
    RefScene *s = (RefScene *)h;
    Point3D c = s->objs[no].polys[nv].getcol(u, v);
    out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

int ref_add_sphere(void *h, double rad, const double *pos, const double *lig, const double *col, int type) {
    RefScene *s = (RefScene *)h;
    s->spheres.push_back(Sphere(rad, Point3D(pos[0], pos[1], pos[2]), Point3D(lig[0], lig[1], lig[2]),
                                Point3D(col[0], col[1], col[2]), (RType)type));
    return (int)s->spheres.size() - 1;
}

void ref_build(void *h) { // raytrace.cpp:128
    RefScene *s = (RefScene *)h;
    s->tree = new KdTree(s->objs, 0);
}

// KdTree::check (raytrace.cpp:46). Returns the bool; no/nv/t are what the call left behind.
int ref_check(void *h, const double *ray, int *no, int *nv, double *t) {
    RefScene *s = (RefScene *)h;
    Ray r = mkray(ray);
    int a = -1, b = -1; double d = 1e20;
    bool f = s->tree->check(s->objs, r, a, b, d);
    *no = a; *nv = b; *t = d;
    return f ? 1 : 0;
}

// sphere_intersect (raytrace.cpp:47)
int ref_sphere_intersect(void *h, const double *ray, int *id, double *t) {
    RefScene *s = (RefScene *)h;
    Ray r = mkray(ray);
    int i = -1; double d = 1e20;
    bool f = sphere_intersect(s->spheres.empty() ? NULL : &s->spheres[0], (int)s->spheres.size(), r, i, d);
    *id = i; *t = d;
    return f ? 1 : 0;
}

double ref_sph_intersect(void *h, int i, const double *ray) {
    RefScene *s = (RefScene *)h;
    return s->spheres[i].intersect(mkray(ray));
}

double ref_poly_intersect(void *h, int no, int nv, const double *ray, double *u, double *v) {
    RefScene *s = (RefScene *)h;
    double uu = 0, vv = 0;
    double t = s->objs[no].polys[nv].intersect(mkray(ray), uu, vv);
    *u = uu; *v = vv;
    return t;
}

void ref_poly_getn(void *h, int no, int nv, const double *x, double *out) {
    RefScene *s = (RefScene *)h;
    Point3D n = s->objs[no].polys[nv].getn(Point3D(x[0], x[1], x[2]));
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

void ref_poly_nc(void *h, int no, int nv, double *out) {
    RefScene *s = (RefScene *)h;
    const Point3D &n = s->objs[no].polys[nv].nc;
    out[0] = n.x; out[1] = n.y; out[2] = n.z;
}

int ref_check_aabb(const double *box /* mi[3], ma[3] */, const double *ray) {
    KdTreeNode b;
    for (int k = 0; k < 3; k++) { b.mi[k] = box[k]; b.ma[k] = box[3 + k]; }
    return b.check_aabb(mkray(ray)) ? 1 : 0;
}

// Pre-order walk over the reference's tree objects (the order of its own flattening,
// cuda.cu:68-89). List entries are reported as (no, nv) pairs.
static int count_nodes(const KdTree *t) { return t ? 1 + count_nodes(t->l) + count_nodes(t->r) : 0; }
int ref_num_nodes(void *h) { return count_nodes(((RefScene *)h)->tree); }

struct Dump { double *box, *split; int *axis, *left, *right, *first, *count, *list_no, *list_nv; int nn, nl; };
static int dump(const KdTree *t, Dump &d) {
    if (!t) return -1;
    int me = d.nn++;
    for (int k = 0; k < 3; k++) { d.box[6 * me + k] = t->aabb.mi[k]; d.box[6 * me + 3 + k] = t->aabb.ma[k]; }
    // leaves never set `split` (kdtree.cpp:36-40); report 0 for them so dumps are comparable
    d.split[me] = (t->l || t->r || t->node.size() >= 4) ? t->split : 0.0;
    d.axis[me] = t->w; d.first[me] = d.nl; d.count[me] = (int)t->node.size();
    for (size_t i = 0; i < t->node.size(); i++) { d.list_no[d.nl] = t->node[i].no; d.list_nv[d.nl] = t->node[i].nv; d.nl++; }
    int l = dump(t->l, d), r = dump(t->r, d);
    d.left[me] = l; d.right[me] = r;
    return me;
}
void ref_tree_dump(void *h, double *box, double *split, int *axis, int *left, int *right, int *first, int *count,
                   int *list_no, int *list_nv) {
    Dump d = {box, split, axis, left, right, first, count, list_no, list_nv, 0, 0};
    dump(((RefScene *)h)->tree, d);
}

// The expression of raytrace.cpp:203, `Ray(cam.o + d * 140, d.norm())`, written with the
// reference's own Point3D/Ray so that the compiler's argument evaluation order (which decides
// whether the origin uses d before or after the in-place normalisation) can be observed.
void ref_primary_ray(const double *camo, const double *dvec, double *out) {
    Point3D o(camo[0], camo[1], camo[2]), d(dvec[0], dvec[1], dvec[2]);
    Ray cam(o, Point3D(0, 0, -1));
    Ray r = Ray(cam.o + d * 140, d.norm());
    out[0] = r.o.x; out[1] = r.o.y; out[2] = r.o.z; out[3] = r.d.x; out[4] = r.d.y; out[5] = r.d.z;
}

// The description integration/rt_binding.h makes of this scene's reference objects.  The RtFlat stays alive with the
// scene (leaked with it, like the reference's tree); returns a pointer to its rt_scene_desc.
const void *ref_flatten_desc(void *h) {
    RefScene *s = (RefScene *)h;
    RtFlat *f = new RtFlat();
    return rt_flatten(s->objs, s->spheres, s->tree, *f);
}

// Vector helpers of point.h for known-answer tests.
void ref_vec_norm(const double *a, double *out) {
    Point3D p(a[0], a[1], a[2]); p.norm();
    out[0] = p.x; out[1] = p.y; out[2] = p.z;
}

} // extern "C"
