// rt_binding.h -- what a maintainer of the reference adds to raytrace.cpp to call librt_hip.so instead of its
// disabled CUDA path: flattening of the reference's OWN structures (vector<Object> of Polygons, vector<Sphere>,
// KdTree) into the plain arrays of rt_scene_desc.  It includes the reference's headers ("kdtree.h", "object.h",
// "sphere.h"), so it only compiles inside the reference's tree (or with -I<reference>); nothing of the product
// includes it.  oracle/Makefile compiles it into oracle/_ref together with the reference's kdtree.cpp, and
// tests/test_oracle_vs_ref.py checks that the description it produces equals the host library's bit for bit.
//
// Order and ids are those of the reference's own flattening, travKdTree (cuda.cu:68-89): nodes in pre-order, own
// lists appended in that order, global triangle id = triangles of earlier objects + face index (cuda.cu:80,158-159).
#ifndef RT_BINDING_H
#define RT_BINDING_H

#include <algorithm>
#include <cstdint>
#include <vector>

#include "kdtree.h"
#include "object.h"
#include "sphere.h"
#include "rt_hip.h"

struct RtFlat {
    std::vector<double> box, split, verts, nc, col, lig, ts, srad, spos, slig, scol;
    std::vector<int32_t> axis, left, right, first, count, list, type, tex, stype;
    std::vector<uint8_t> tex_data;
    std::vector<int64_t> tex_offset;
    std::vector<int32_t> tex_sizex, tex_sizey;
    std::vector<const unsigned char *> tex_ptr; // images already seen (Polygons of one material share the pointer)
    int depth = 0;
    rt_scene_desc desc;
};

static inline int rt_flatten_node(const KdTree *t, const std::vector<int> &sum, RtFlat &f, int depth) {
    if (!t) return -1;
    const int me = (int)f.axis.size();
    f.depth = std::max(f.depth, depth);
    for (int k = 0; k < 3; k++) f.box.push_back(t->aabb.mi[k]);
    for (int k = 0; k < 3; k++) f.box.push_back(t->aabb.ma[k]);
    const bool leaf = !t->l && !t->r && t->node.size() < 4; // kdtree.cpp:36-40: leaves never set `split`
    f.split.push_back(leaf ? 0.0 : t->split);
    f.axis.push_back(t->w); f.left.push_back(-1); f.right.push_back(-1);
    f.first.push_back((int32_t)f.list.size()); f.count.push_back((int32_t)t->node.size());
    for (size_t i = 0; i < t->node.size(); i++) f.list.push_back(sum[t->node[i].no] + t->node[i].nv); // cuda.cu:80
    const int l = rt_flatten_node(t->l, sum, f, depth + 1), r = rt_flatten_node(t->r, sum, f, depth + 1);
    f.left[me] = l; f.right[me] = r;
    return me;
}

// Fills `f` (which must outlive every use of f.desc) from the reference's globals objs, spheres, kdtree
// (raytrace.cpp:24-26) and returns &f.desc, ready for rt_upload_scene.
static inline const rt_scene_desc *rt_flatten(const std::vector<Object> &objs, const std::vector<Sphere> &spheres, const KdTree *tree, RtFlat &f) {
    std::vector<int> sum(objs.size() + 1, 0); // cuda.cu:158-159
    for (size_t i = 0; i < objs.size(); i++) sum[i + 1] = sum[i] + (int)objs[i].polys.size();
    rt_flatten_node(tree, sum, f, 1);
    for (size_t i = 0; i < objs.size(); i++)
        for (size_t j = 0; j < objs[i].polys.size(); j++) {
            const Polygon &p = objs[i].polys[j];
            for (int k = 0; k < 3; k++) { f.verts.push_back(p.points3d[k].x); f.verts.push_back(p.points3d[k].y); f.verts.push_back(p.points3d[k].z); }
            f.nc.push_back(p.nc.x); f.nc.push_back(p.nc.y); f.nc.push_back(p.nc.z);
            f.col.push_back(p.col.x); f.col.push_back(p.col.y); f.col.push_back(p.col.z);
            f.lig.push_back(objs[i].lig.x); f.lig.push_back(objs[i].lig.y); f.lig.push_back(objs[i].lig.z); // raytrace.cpp:66
            f.type.push_back((int32_t)objs[i].type);
            for (int k = 0; k < 3; k++) { f.ts.push_back(p.ts[k].x); f.ts.push_back(p.ts[k].y); }
            int img = -1;
            if (p.tex) { // polygon.h:14-15: image pointer and sizes; the bytes are sizey rows of sizex BGR pixels (opencvlib.cpp:21-29)
                for (size_t k = 0; k < f.tex_ptr.size(); k++) if (f.tex_ptr[k] == p.tex) img = (int)k;
                if (img < 0) {
                    img = (int)f.tex_ptr.size();
                    f.tex_ptr.push_back(p.tex);
                    f.tex_offset.push_back((int64_t)f.tex_data.size());
                    f.tex_sizex.push_back(p.sizex); f.tex_sizey.push_back(p.sizey);
                    f.tex_data.insert(f.tex_data.end(), p.tex, p.tex + 3 * (size_t)p.sizex * p.sizey);
                }
            }
            f.tex.push_back(img);
        }
    for (size_t i = 0; i < spheres.size(); i++) {
        const Sphere &s = spheres[i];
        f.srad.push_back(s.rad);
        f.spos.push_back(s.pos.x); f.spos.push_back(s.pos.y); f.spos.push_back(s.pos.z);
        f.slig.push_back(s.lig.x); f.slig.push_back(s.lig.y); f.slig.push_back(s.lig.z);
        f.scol.push_back(s.col.x); f.scol.push_back(s.col.y); f.scol.push_back(s.col.z);
        f.stype.push_back((int32_t)s.type);
    }
    rt_scene_desc &d = f.desc;
    d = rt_scene_desc();
    d.n_nodes = (int32_t)f.axis.size(); d.node_box = f.box.data(); d.node_split = f.split.data();
    d.node_axis = f.axis.data(); d.node_left = f.left.data(); d.node_right = f.right.data();
    d.node_first = f.first.data(); d.node_count = f.count.data();
    d.n_list = (int32_t)f.list.size(); d.list_tri = f.list.data(); d.tree_depth = f.depth;
    d.n_tris = (int32_t)f.type.size(); d.tri_verts = f.verts.data(); d.tri_nc = f.nc.data();
    d.tri_col = f.col.data(); d.tri_lig = f.lig.data(); d.tri_type = f.type.data();
    d.n_spheres = (int32_t)f.stype.size(); d.sph_rad = f.srad.data(); d.sph_pos = f.spos.data();
    d.sph_lig = f.slig.data(); d.sph_col = f.scol.data(); d.sph_type = f.stype.data();
    d.n_textures = (int32_t)f.tex_offset.size(); d.tex_data = f.tex_data.data(); d.tex_offset = f.tex_offset.data();
    d.tex_sizex = f.tex_sizex.data(); d.tex_sizey = f.tex_sizey.data(); d.tri_tex = f.tex.data(); d.tri_ts = f.ts.data();
    return &d;
}

#endif // RT_BINDING_H
