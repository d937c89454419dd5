// rt_host.cpp -- host-side scene library (librt_host.so): scene model, scene-file / OBJ
// readers, kd-tree builder + flattening, tone mapping and BMP output.  No GPU code here.
//
// Reference behaviour restated (not its code):
//   scene file grammar          raytrace.cpp:106-151
//   OBJ / MTL reader            object.cpp:9-286 (see rth_add_obj_file_ex)
//   texture images              opencvlib.cpp:11-30 (BMP / PPM reader and a bilinear resize instead of OpenCV)
//   mesh transform              object.cpp:22-23,138-140 and matrix.h:24-52, matrix.cpp:21-37
//   face normal nc              polygon.h:23-27
//   kd-tree builder             kdtree.cpp:6-96, kdtree.h:29-55
//   pre-order flattening + ids  cuda.cu:68-89,158-159
//   drawPixel                   raytrace.cpp:28-39
// The builder works on index ranges instead of copying node vectors per level, and skips the
// three sorted event arrays the reference fills but never reads (kdtree.cpp:10-21), so it also
// handles meshes of tens of millions of triangles; the tree it produces is the same.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "rt_host.h"

namespace {

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

struct P3 { double x, y, z; };

double dmin(double a, double b) { if (a < b) return a; else return b; } // kdtree.h:19
double dmax(double a, double b) { if (a > b) return a; else return b; } // kdtree.h:24

struct Mesh {
    int64_t first, count;
    double lig[3];
    int type;
};

} // namespace

struct rth_scene {
    // triangles in global-id order: object by object, face by face
    std::vector<double> verts, nc, col, lig;
    std::vector<int32_t> type;
    std::vector<Mesh> meshes;
    std::vector<double> sph_rad, sph_pos, sph_lig, sph_col;
    std::vector<int32_t> sph_type;
    // texture images back to back (rows of BGR pixels) and, per triangle, its image (-1: none) and ts[0..2].xy
    std::vector<uint8_t> tex_data;
    std::vector<int64_t> tex_offset;
    std::vector<int32_t> tex_sizex, tex_sizey; // Polygon::sizex = columns, Polygon::sizey = rows (opencvlib.cpp:19-20)
    std::vector<int32_t> tri_tex;
    std::vector<double> tri_ts;
    // flattened tree
    std::vector<double> node_box, node_split;
    std::vector<int32_t> node_axis, node_left, node_right, node_first, node_count, list_tri;
    int32_t tree_depth = 0;
    bool built = false;
    rt_scene_desc desc{};
};

namespace {

void refresh_desc(rth_scene *s) {
    rt_scene_desc &d = s->desc;
    d.n_nodes = (int32_t)s->node_axis.size();
    d.node_box = s->node_box.data();
    d.node_split = s->node_split.data();
    d.node_axis = s->node_axis.data();
    d.node_left = s->node_left.data();
    d.node_right = s->node_right.data();
    d.node_first = s->node_first.data();
    d.node_count = s->node_count.data();
    d.n_list = (int32_t)s->list_tri.size();
    d.list_tri = s->list_tri.data();
    d.tree_depth = s->tree_depth;
    d.n_tris = (int32_t)s->type.size();
    d.tri_verts = s->verts.data();
    d.tri_nc = s->nc.data();
    d.tri_col = s->col.data();
    d.tri_lig = s->lig.data();
    d.tri_type = s->type.data();
    d.n_spheres = (int32_t)s->sph_type.size();
    d.sph_rad = s->sph_rad.data();
    d.sph_pos = s->sph_pos.data();
    d.sph_lig = s->sph_lig.data();
    d.sph_col = s->sph_col.data();
    d.sph_type = s->sph_type.data();
    d.n_textures = (int32_t)s->tex_offset.size();
    d.tex_data = s->tex_data.data();
    d.tex_offset = s->tex_offset.data();
    d.tex_sizex = s->tex_sizex.data();
    d.tex_sizey = s->tex_sizey.data();
    d.tri_tex = s->tri_tex.data();
    d.tri_ts = s->tri_ts.data();
}

// polygon.h:23-27: six-term expansion, then norm() = multiply by 1/sqrt (point.h:49-51)
void face_normal(const double *v, double *out) {
    const P3 a = {v[0], v[1], v[2]}, b = {v[3], v[4], v[5]}, c = {v[6], v[7], v[8]};
    const double nx = (b.y * c.z + c.y * a.z + a.y * b.z - b.y * a.z - a.y * c.z - c.y * b.z);
    const double ny = (a.x * c.z + b.x * a.z + c.x * b.z - c.x * a.z - b.x * c.z - a.x * b.z);
    const double nz = (a.x * b.y + b.x * c.y + c.x * a.y - c.x * b.y - b.x * a.y - a.x * c.y);
    const double k = 1 / std::sqrt(nx * nx + ny * ny + nz * nz);
    out[0] = nx * k; out[1] = ny * k; out[2] = nz * k;
}

// 4x4 matrices of matrix.h / matrix.cpp, restricted to what the loader uses
struct M4 { double a[4][4]; };
M4 zero4() { M4 m; memset(&m, 0, sizeof(m)); return m; }
M4 mul4(const M4 &a, const M4 &b) { // matrix.cpp:21-29 (loop order i, k, j; sums start at 0)
    M4 r = zero4();
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 4; k++)
            for (int j = 0; j < 4; j++) r.a[i][j] += a.a[i][k] * b.a[k][j];
    return r;
}
M4 rotx(double c, double s) { M4 m = zero4(); m.a[0][0] = m.a[3][3] = 1; m.a[1][1] = m.a[2][2] = c; m.a[1][2] = m.a[2][1] = s; m.a[1][2] *= -1; return m; }
M4 roty(double c, double s) { M4 m = zero4(); m.a[1][1] = m.a[3][3] = 1; m.a[0][0] = m.a[2][2] = c; m.a[0][2] = m.a[2][0] = s; m.a[2][0] *= -1; return m; }
M4 rotz(double c, double s) { M4 m = zero4(); m.a[2][2] = m.a[3][3] = 1; m.a[0][0] = m.a[1][1] = c; m.a[0][1] = m.a[1][0] = s; m.a[0][1] *= -1; return m; }
P3 apply4(const M4 &m, P3 p) { // matrix.cpp:31-37: 4x1 product, w = 1
    const double v[4] = {p.x, p.y, p.z, 1};
    double r[3];
    for (int i = 0; i < 3; i++) {
        double acc = 0;
        for (int k = 0; k < 4; k++) acc += m.a[i][k] * v[k];
        r[i] = acc;
    }
    return P3{r[0], r[1], r[2]};
}

int type_from_word(const char *w) { // raytrace.cpp:123-125
    if (w[0] == 'D') return RT_DIFF;
    if (w[0] == 'S') return RT_SPEC;
    return RT_REFR;
}

} // namespace

extern "C" {

const char *rth_last_error(void) { return g_err.c_str(); }

rth_scene *rth_scene_new(void) { return new rth_scene(); }
void rth_scene_free(rth_scene *s) { delete s; }

int rth_add_mesh(rth_scene *s, const double *verts, int64_t n_tris, const double *lig, const double *col,
                 const double *tri_cols, int type) {
    if (!s || n_tris < 0 || (n_tris > 0 && !verts) || !lig || !col) return fail(RT_E_INVALID, "rth_add_mesh: bad arguments");
    if (type < 0 || type > 2) return fail(RT_E_INVALID, "rth_add_mesh: type %d", type);
    if ((int64_t)s->type.size() + n_tris > 0x7FFFFFF0LL) return fail(RT_E_LIMIT, "too many triangles");
    Mesh m;
    m.first = (int64_t)s->type.size();
    m.count = n_tris;
    for (int k = 0; k < 3; k++) m.lig[k] = lig[k];
    m.type = type;
    s->verts.insert(s->verts.end(), verts, verts + 9 * n_tris);
    const size_t base = s->nc.size();
    s->nc.resize(base + 3 * (size_t)n_tris);
    s->col.resize(base + 3 * (size_t)n_tris);
    s->lig.resize(base + 3 * (size_t)n_tris);
    for (int64_t i = 0; i < n_tris; i++) {
        face_normal(verts + 9 * i, &s->nc[base + 3 * i]);
        for (int k = 0; k < 3; k++) {
            s->col[base + 3 * i + k] = tri_cols ? tri_cols[3 * i + k] : col[k];
            s->lig[base + 3 * i + k] = lig[k];
        }
    }
    s->type.insert(s->type.end(), (size_t)n_tris, type);
    s->tri_tex.insert(s->tri_tex.end(), (size_t)n_tris, -1);
    s->tri_ts.insert(s->tri_ts.end(), 6 * (size_t)n_tris, 0.0);
    s->meshes.push_back(m);
    s->built = false;
    return (int)s->meshes.size() - 1;
}

int rth_add_texture(rth_scene *s, const uint8_t *bgr, int32_t rows, int32_t cols) {
    if (!s || !bgr || rows < 1 || cols < 1) return fail(RT_E_INVALID, "rth_add_texture: bad arguments");
    if ((int64_t)rows * cols > (int64_t)1 << 28) return fail(RT_E_LIMIT, "rth_add_texture: image of %d x %d pixels is too large", rows, cols);
    const size_t before = s->tex_data.size();
    try { s->tex_data.insert(s->tex_data.end(), bgr, bgr + 3 * (size_t)rows * cols); }
    catch (const std::exception &) { return fail(RT_E_NOMEM, "rth_add_texture: out of memory"); }
    s->tex_offset.push_back((int64_t)before);
    s->tex_sizex.push_back(cols);
    s->tex_sizey.push_back(rows);
    if (s->built) refresh_desc(s); // images are not part of the tree
    return (int)s->tex_offset.size() - 1;
}

int rth_set_mesh_textures(rth_scene *s, int mesh, const int32_t *tri_tex, const double *tri_ts) {
    if (!s || !tri_tex || !tri_ts || mesh < 0 || (size_t)mesh >= s->meshes.size()) return fail(RT_E_INVALID, "rth_set_mesh_textures: bad arguments");
    const Mesh &m = s->meshes[mesh];
    for (int64_t i = 0; i < m.count; i++) {
        if (tri_tex[i] < -1 || tri_tex[i] >= (int32_t)s->tex_offset.size())
            return fail(RT_E_INVALID, "rth_set_mesh_textures: triangle %lld names image %d of %zu", (long long)i, tri_tex[i], s->tex_offset.size());
        s->tri_tex[(size_t)(m.first + i)] = tri_tex[i];
        for (int k = 0; k < 6; k++) s->tri_ts[6 * (size_t)(m.first + i) + k] = tri_ts[6 * i + k];
    }
    return RT_OK;
}

} // extern "C"

namespace {

// An image file as rows x cols BGR bytes, top row first (what cv::imread returns for a colour image).
// IMG_NOFILE: the file cannot be opened (the reference then leaves the material untextured, opencvlib.cpp:13-14);
// IMG_FORMAT: it exists but is not one of the formats read here; IMG_LIMIT: its header states a size beyond the limits.
enum { IMG_OK = 0, IMG_NOFILE = 1, IMG_FORMAT = 2, IMG_LIMIT = 3 };
constexpr int64_t IMG_MAX_SIDE = 65536, IMG_MAX_PIXELS = (int64_t)1 << 28; // (rth_add_texture's own limit)
int read_image_checked(const char *path, std::vector<uint8_t> &bgr, int &rows, int &cols) {
    FILE *f = fopen(path, "rb");
    if (!f) return IMG_NOFILE;
    std::vector<uint8_t> raw;
    uint8_t chunk[65536];
    size_t n;
    while ((n = fread(chunk, 1, sizeof(chunk), f)) > 0) raw.insert(raw.end(), chunk, chunk + n);
    fclose(f);
    auto le32 = [&](size_t o) { return (uint32_t)raw[o] | ((uint32_t)raw[o + 1] << 8) | ((uint32_t)raw[o + 2] << 16) | ((uint32_t)raw[o + 3] << 24); };
    if (raw.size() >= 54 && raw[0] == 'B' && raw[1] == 'M') { // BMP, uncompressed 24 or 32 bits per pixel
        const uint32_t off = le32(10), hdr = le32(14);
        const int32_t w = (int32_t)le32(18), h = (int32_t)le32(22);
        const int bpp = raw[28] | (raw[29] << 8);
        const uint32_t comp = le32(30);
        if (hdr < 40 || w < 1 || h == 0 || (bpp != 24 && bpp != 32) || (comp != 0 && !(comp == 3 && bpp == 32))) return IMG_FORMAT;
        // the header is not trusted: sizes are bounded before anything is computed from them
        const int64_t ah64 = h < 0 ? -(int64_t)h : (int64_t)h;
        if (w > IMG_MAX_SIDE || ah64 > IMG_MAX_SIDE || (int64_t)w * ah64 > IMG_MAX_PIXELS) return IMG_LIMIT;
        const int ah = (int)ah64;
        const size_t stride = (((size_t)w * bpp + 31) / 32) * 4;
        if ((size_t)off + stride * ah > raw.size()) return IMG_FORMAT;
        rows = ah; cols = w;
        bgr.resize(3 * (size_t)rows * cols);
        for (int r = 0; r < rows; r++) {
            const uint8_t *src = &raw[off + stride * (size_t)(h < 0 ? r : rows - 1 - r)]; // bottom-up unless the height is negative
            for (int c = 0; c < cols; c++)
                for (int k = 0; k < 3; k++) bgr[3 * ((size_t)r * cols + c) + k] = src[(size_t)c * (bpp / 8) + k];
        }
        return IMG_OK;
    }
    if (raw.size() >= 11 && raw[0] == 'P' && raw[1] == '6') { // binary PPM: RGB bytes
        size_t p = 2;
        long v[3];
        for (int k = 0; k < 3; k++) {
            for (;;) { // white space and comment lines
                while (p < raw.size() && isspace(raw[p])) p++;
                if (p < raw.size() && raw[p] == '#') { while (p < raw.size() && raw[p] != '\n') p++; continue; }
                break;
            }
            long x = 0;
            bool any = false;
            while (p < raw.size() && raw[p] >= '0' && raw[p] <= '9') { x = x * 10 + (raw[p] - '0'); p++; any = true; if (x > 1 << 28) return IMG_LIMIT; }
            if (!any) return IMG_FORMAT;
            v[k] = x;
        }
        p++; // the single white-space byte after maxval
        if (v[0] < 1 || v[1] < 1 || v[2] < 1 || v[2] > 255) return IMG_FORMAT;
        if (v[0] > IMG_MAX_SIDE || v[1] > IMG_MAX_SIDE || (int64_t)v[0] * v[1] > IMG_MAX_PIXELS) return IMG_LIMIT;
        if (p + 3 * (size_t)v[0] * (size_t)v[1] > raw.size()) return IMG_FORMAT;
        cols = (int)v[0]; rows = (int)v[1];
        bgr.resize(3 * (size_t)rows * cols);
        for (size_t i = 0; i < (size_t)rows * cols; i++) { bgr[3 * i] = raw[p + 3 * i + 2]; bgr[3 * i + 1] = raw[p + 3 * i + 1]; bgr[3 * i + 2] = raw[p + 3 * i]; }
        return IMG_OK;
    }
    return IMG_FORMAT;
}
// (allocation failures stay inside: nothing is thrown through the C boundary)
int read_image(const char *path, std::vector<uint8_t> &bgr, int &rows, int &cols) {
    try { return read_image_checked(path, bgr, rows, cols); }
    catch (const std::exception &) { return IMG_LIMIT; }
}

// cv::resize(src, dst, Size(dc, dr)) with its default bilinear interpolation for 8-bit images, as OpenCV publishes
// it: sample position (d + 0.5) * scale - 0.5, weights as 11-bit fixed point, horizontal pass in int, vertical pass
// (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  Equal sizes copy.
void resize_bilinear(const std::vector<uint8_t> &src, int sr, int sc, std::vector<uint8_t> &dst, int dr, int dc) {
    dst.resize(3 * (size_t)dr * dc);
    if (sr == dr && sc == dc) { dst = src; return; }
    const double scale_x = 1.0 / ((double)dc / sc), scale_y = 1.0 / ((double)dr / sr);
    std::vector<int> xofs((size_t)dc), yofs((size_t)dr);
    std::vector<short> alpha(2 * (size_t)dc), beta(2 * (size_t)dr);
    auto coef = [](float w) { long r = lrintf(w * 2048.0f); return (short)(r < -32768 ? -32768 : r > 32767 ? 32767 : r); };
    for (int dx = 0; dx < dc; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sc - 1) { fx = 0; sx = sc - 1; }
        xofs[dx] = sx;
        alpha[2 * dx] = coef(1.f - fx); alpha[2 * dx + 1] = coef(fx);
    }
    for (int dy = 0; dy < dr; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        const int sy = (int)std::floor(fy);
        fy -= sy;
        yofs[dy] = sy;
        beta[2 * dy] = coef(1.f - fy); beta[2 * dy + 1] = coef(fy);
    }
    std::vector<int> row0(3 * (size_t)dc), row1(3 * (size_t)dc);
    for (int dy = 0; dy < dr; dy++) {
        const int y0 = std::min(std::max(yofs[dy], 0), sr - 1), y1 = std::min(std::max(yofs[dy] + 1, 0), sr - 1);
        const uint8_t *s0 = &src[3 * (size_t)y0 * sc], *s1 = &src[3 * (size_t)y1 * sc];
        for (int dx = 0; dx < dc; dx++) {
            const int sx = xofs[dx], sx1 = std::min(sx + 1, sc - 1);
            for (int k = 0; k < 3; k++) {
                row0[3 * dx + k] = s0[3 * sx + k] * alpha[2 * dx] + s0[3 * sx1 + k] * alpha[2 * dx + 1];
                row1[3 * dx + k] = s1[3 * sx + k] * alpha[2 * dx] + s1[3 * sx1 + k] * alpha[2 * dx + 1];
            }
        }
        const int b0 = beta[2 * dy], b1 = beta[2 * dy + 1];
        for (int x = 0; x < 3 * dc; x++) {
            const int v = (((b0 * (row0[x] >> 4)) >> 16) + ((b1 * (row1[x] >> 4)) >> 16) + 2) >> 2;
            dst[3 * (size_t)dy * dc + x] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
        }
    }
}

} // namespace

extern "C" {

} // extern "C"

namespace {
// rth_load_texture_file; *missing says whether the failure is "the file cannot be opened"
int load_texture(rth_scene *s, const char *path, double times, bool *missing) {
    *missing = false;
    std::vector<uint8_t> img, out;
    int rows = 0, cols = 0;
    const int st = read_image(path, img, rows, cols);
    if (st == IMG_NOFILE) { *missing = true; return fail(RT_E_INVALID, "Can't open image %s", path); }
    if (st == IMG_LIMIT) return fail(RT_E_LIMIT, "image %s: more than %lld pixels a side or %lld in all", path, (long long)IMG_MAX_SIDE, (long long)IMG_MAX_PIXELS);
    if (st != IMG_OK) return fail(RT_E_INVALID, "Can't decode image %s: only uncompressed 24/32-bit BMP and binary PPM (P6) are read", path);
    const double fc = cols * times, fr = rows * times; // opencvlib.cpp:16
    if (!(fc >= 1) || !(fr >= 1)) return fail(RT_E_INVALID, "%s: scaled to %d x %d pixels", path, (int)fr, (int)fc);
    if (fc > (double)IMG_MAX_SIDE || fr > (double)IMG_MAX_SIDE || fc * fr > (double)IMG_MAX_PIXELS)
        return fail(RT_E_LIMIT, "image %s scaled by %g exceeds %lld pixels", path, times, (long long)IMG_MAX_PIXELS);
    const int dc = (int)fc, dr = (int)fr;
    try { resize_bilinear(img, rows, cols, out, dr, dc); }
    catch (const std::exception &) { return fail(RT_E_NOMEM, "image %s: out of memory", path); }
    return rth_add_texture(s, out.data(), dr, dc);
}
} // namespace

extern "C" {

int rth_load_texture_file(rth_scene *s, const char *path, double times) {
    if (!s || !path) return fail(RT_E_INVALID, "rth_load_texture_file: bad arguments");
    bool missing = false;
    return load_texture(s, path, times, &missing);
}

int rth_add_sphere(rth_scene *s, double rad, const double *pos, const double *lig, const double *col, int type) {
    if (!s || !pos || !lig || !col) return fail(RT_E_INVALID, "rth_add_sphere: bad arguments");
    if (type < 0 || type > 2) return fail(RT_E_INVALID, "rth_add_sphere: type %d", type);
    s->sph_rad.push_back(rad);
    for (int k = 0; k < 3; k++) { s->sph_pos.push_back(pos[k]); s->sph_lig.push_back(lig[k]); s->sph_col.push_back(col[k]); }
    s->sph_type.push_back(type);
    if (s->built) refresh_desc(s); // spheres are not part of the tree
    return (int)s->sph_type.size() - 1;
}

int64_t rth_num_tris(const rth_scene *s) { return s ? (int64_t)s->type.size() : 0; }
int32_t rth_num_spheres(const rth_scene *s) { return s ? (int32_t)s->sph_type.size() : 0; }

int rth_build(rth_scene *s) {
    if (!s) return fail(RT_E_INVALID, "rth_build: scene is NULL");
    const int64_t n = (int64_t)s->type.size();
    // per-triangle boxes (KdTreeNode(const Polygon&, no, nv), kdtree.h:29-50)
    std::vector<double> tmi(3 * (size_t)n), tma(3 * (size_t)n);
    for (int64_t i = 0; i < n; i++) {
        double mi[3] = {1e10, 1e10, 1e10}, ma[3] = {-1e10, -1e10, -1e10};
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++) {
                const double a = s->verts[9 * i + 3 * j + k];
                mi[k] = dmin(mi[k], a);
                ma[k] = dmax(ma[k], a);
            }
        for (int k = 0; k < 3; k++) { tmi[3 * i + k] = mi[k]; tma[3 * i + k] = ma[k]; }
    }
    s->node_box.clear(); s->node_split.clear(); s->node_axis.clear(); s->node_left.clear(); s->node_right.clear();
    s->node_first.clear(); s->node_count.clear(); s->list_tri.clear();
    s->list_tri.reserve((size_t)n);
    s->tree_depth = 0;

    std::vector<int32_t> idx((size_t)n), tl, tr;
    for (int64_t i = 0; i < n; i++) idx[i] = (int32_t)i;
    struct Work { int64_t begin, end; int axis, depth, parent; bool right; };
    std::vector<Work> stack;
    stack.push_back(Work{0, n, 0, 1, -1, false}); // raytrace.cpp:128: new KdTree(objs, 0), also for an empty scene
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        const int me = (int)s->node_axis.size();
        if (me >= 0x0FFFFFFF) return fail(RT_E_LIMIT, "kd-tree has too many nodes");
        if (w.parent >= 0) (w.right ? s->node_right : s->node_left)[w.parent] = me;
        if (w.depth > s->tree_depth) s->tree_depth = w.depth;
        double mi[3] = {1e10, 1e10, 1e10}, ma[3] = {-1e10, -1e10, -1e10}; // kdtree.h:36-40
        for (int64_t i = w.begin; i < w.end; i++)
            for (int k = 0; k < 3; k++) { // kdtree.h:52-55
                mi[k] = dmin(mi[k], tmi[3 * (size_t)idx[i] + k]);
                ma[k] = dmax(ma[k], tma[3 * (size_t)idx[i] + k]);
            }
        for (int k = 0; k < 3; k++) s->node_box.push_back(mi[k]);
        for (int k = 0; k < 3; k++) s->node_box.push_back(ma[k]);
        s->node_axis.push_back(w.axis);
        s->node_left.push_back(-1);
        s->node_right.push_back(-1);
        s->node_first.push_back((int32_t)s->list_tri.size());
        const int64_t cnt = w.end - w.begin;
        if (cnt < 4) { // kdtree.cpp:36-40: leaf; the reference leaves `split` unset there
            s->node_split.push_back(0.0);
            for (int64_t i = w.begin; i < w.end; i++) s->list_tri.push_back(idx[i]);
            s->node_count.push_back((int32_t)cnt);
            continue;
        }
        const int ax = w.axis;
        const double split = (mi[ax] + ma[ax]) * 0.5; // kdtree.cpp:82
        s->node_split.push_back(split);
        tl.clear(); tr.clear();
        int32_t own = 0;
        for (int64_t i = w.begin; i < w.end; i++) { // kdtree.cpp:85-93, order preserved in all three parts
            const int32_t t = idx[i];
            if (tma[3 * (size_t)t + ax] < split) tl.push_back(t);
            else if (tmi[3 * (size_t)t + ax] > split) tr.push_back(t);
            else { s->list_tri.push_back(t); own++; }
        }
        s->node_count.push_back(own);
        const int64_t nl = (int64_t)tl.size(), nr = (int64_t)tr.size();
        if (nl) memcpy(&idx[w.begin], tl.data(), sizeof(int32_t) * (size_t)nl);
        if (nr) memcpy(&idx[w.begin + nl], tr.data(), sizeof(int32_t) * (size_t)nr);
        const int nax = (ax + 1) % 3; // kdtree.cpp:94-95
        // right is pushed first so that the left subtree is emitted next (pre-order, cuda.cu:85)
        if (nr) stack.push_back(Work{w.begin + nl, w.begin + nl + nr, nax, w.depth + 1, me, true});
        if (nl) stack.push_back(Work{w.begin, w.begin + nl, nax, w.depth + 1, me, false});
    }
    s->built = true;
    refresh_desc(s);
    return RT_OK;
}

const rt_scene_desc *rth_desc(const rth_scene *s) {
    if (!s || !s->built) { fail(RT_E_INVALID, "rth_desc: call rth_build first"); return nullptr; }
    return &s->desc;
}

int rth_add_obj_file(rth_scene *s, const char *path, double scale, const double *loc, const double *rot_deg,
                     const double *lig, const double *col, int type) {
    return rth_add_obj_file_ex(s, path, scale, loc, rot_deg, lig, col, type, 0u);
}

int rth_add_obj_file_ex(rth_scene *s, const char *path, double scale, const double *loc, const double *rot_deg,
                        const double *lig, const double *col, int type, unsigned flags) {
    if (!s || !path || !loc || !rot_deg || !lig || !col) return fail(RT_E_INVALID, "rth_add_obj_file: bad arguments");
    FILE *fp = fopen(path, "r");
    if (!fp) return fail(RT_E_INVALID, "Can't open %s", path); // object.cpp:25-28
    const double k = M_PI / 180; // object.cpp:22
    const double rx = rot_deg[0] * k, ry = rot_deg[1] * k, rz = rot_deg[2] * k;
    const M4 t = mul4(mul4(rotx(std::cos(rx), std::sin(rx)), roty(std::cos(ry), std::sin(ry))), rotz(std::cos(rz), std::sin(rz)));
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);

    std::vector<P3> points;
    std::vector<P3> vt(1, P3{0, 0, 0});                   // object.cpp:21: texture coordinates are 1-based, slot 0 is (0, 0)
    std::vector<std::string> name(1);
    std::vector<P3> cols(1, P3{col[0], col[1], col[2]}); // object.cpp:17-18: slot 0 = the object's colour
    std::vector<int> tex(1, -1);                          // object.cpp:15: slot 0 has no image
    std::vector<int> face, face_t, face_mtl;
    int ntex = 0, lines = 0;
    char buf[512], str[512];
    int rc = RT_OK;
    // one vertex reference of a face: v, v/t, v//n or v/t/n (object.cpp:195-249); returns false if it has no vertex index
    auto parse_ref = [](const char *tok, int &v, int &tt) {
        tt = 0;
        if (sscanf(tok, "%d", &v) != 1) return false;
        const char *sl = strchr(tok, '/');
        if (sl && sl[1] != '/' && sl[1] != 0) { if (sscanf(sl + 1, "%d", &tt) != 1) tt = 0; }
        return true;
    };
    while (rc == RT_OK && fscanf(fp, "%511s", buf) != EOF) {
        lines++;
        switch (buf[0]) {
        case 'u': // usemtl: the LAST material of that name wins, unknown names select slot 0 (object.cpp:44-49)
            if (!fgets(buf, sizeof(buf), fp)) break;
            str[0] = 0;
            sscanf(buf, "%511s", str);
            ntex = 0;
            for (size_t i = 0; i < name.size(); i++) if (name[i] == str) ntex = (int)i;
            break;
        case 'm': { // mtllib (object.cpp:51-129)
            if (!fgets(buf, sizeof(buf), fp)) break;
            str[0] = 0;
            sscanf(buf, "%511s", str);
            FILE *fm = fopen((dir + str).c_str(), "r");
            if (!fm) fm = fopen(str, "r");
            if (!fm) break; // the reference dereferences NULL here; a missing library is skipped instead
            while (fscanf(fm, "%511s", buf) != EOF) {
                if (buf[0] == 'n') { // newmtl
                    if (!fgets(buf, sizeof(buf), fm)) break;
                    str[0] = 0;
                    sscanf(buf, "%511s", str);
                    name.push_back(str);
                    cols.push_back(P3{col[0], col[1], col[2]});
                    tex.push_back(-1);
                } else if (buf[0] == 'K' && ((buf[1] == 'd' && type == RT_DIFF) || (buf[1] == 's' && type != RT_DIFF))) {
                    P3 v; // Kd for diffuse objects, Ks otherwise (object.cpp:75-106)
                    if (fscanf(fm, "%lf%lf%lf", &v.x, &v.y, &v.z) != 3) {
                        rc = fail(RT_E_INVALID, "Error: Wrong Number of Values(Should be 3). at Line %d", lines);
                        break;
                    }
                    cols.back() = v;
                } else if ((!strcmp(buf, "map_Kd") && type == RT_DIFF) || (!strcmp(buf, "map_Ks") && type != RT_DIFF)) {
                    // object.cpp:112-119: the image of the material read last, scaled by the object's scale; a file
                    // that cannot be read leaves the material untextured (opencvlib.cpp:13-14)
                    if (!fgets(buf, sizeof(buf), fm)) break;
                    str[0] = 0;
                    sscanf(buf, "%511s", str);
                    // (a file that exists but cannot be decoded here -- PNG, JPEG, 8-bit BMP, which cv::imread reads -- is an
                    // error that names the file: rendering it with the flat colour would silently differ from the reference)
                    bool missing = false;
                    int img = load_texture(s, (dir + str).c_str(), scale, &missing);
                    if (img < 0 && missing) img = load_texture(s, str, scale, &missing);
                    if (img < 0 && !missing) { rc = img; break; }
                    tex.back() = img < 0 ? -1 : img;
                } else if (!fgets(buf, sizeof(buf), fm)) break;
            }
            fclose(fm);
            break;
        }
        case 'v':
            if (buf[1] == '\0') { // vertex: scale, rotate, translate (object.cpp:135-141)
                P3 v;
                if (fscanf(fp, "%lf%lf%lf", &v.x, &v.y, &v.z) != 3) {
                    rc = fail(RT_E_INVALID, "Error: Wrong Number of Values(Should be 3). at Line %d", lines);
                    break;
                }
                v.x *= scale; v.y *= scale; v.z *= scale;
                v = apply4(t, v);
                points.push_back(P3{v.x + loc[0], v.y + loc[1], v.z + loc[2]});
            } else if (buf[1] == 't' && buf[2] == '\0') { // texture coordinate: two values (object.cpp:164-171)
                P3 v{0, 0, 0};
                if (fscanf(fp, "%lf%lf", &v.x, &v.y) != 2) {
                    rc = fail(RT_E_INVALID, "Error: Wrong Number of Values(Should be 3). at Line %d", lines);
                    break;
                }
                vt.push_back(v);
            } else if (!fgets(buf, sizeof(buf), fp)) { /* vn (the normals never reach the image, polygon.h:67-69) and anything else */ }
            break;
        case 'f': { // the first three vertices of the face (object.cpp:186-257)
            int v[3] = {0, 0, 0}, tt[3] = {0, 0, 0};
            bool ok = true;
            for (int j = 0; j < 3 && ok; j++) {
                if (fscanf(fp, "%511s", str) != 1) { ok = false; break; }
                ok = parse_ref(str, v[j], tt[j]);
            }
            if (!ok) { rc = fail(RT_E_INVALID, "Error: Wrong Face at Line %d", lines); break; }
            auto resolve = [&](int &vv, int &t2) { // OBJ's relative indices, when asked for
                if (flags & RTH_OBJ_NEGATIVE) {
                    if (vv < 0) vv = (int)points.size() + vv + 1;
                    if (t2 < 0) t2 = (int)vt.size() + t2;
                }
            };
            for (int j = 0; j < 3; j++) resolve(v[j], tt[j]);
            for (int j = 0; j < 3; j++) { face.push_back(v[j] - 1); face_t.push_back(tt[j]); }
            face_mtl.push_back(ntex);
            if (!fgets(buf, sizeof(buf), fp)) buf[0] = 0; // the rest of the line: further vertices, if any
            if (flags & RTH_OBJ_TRIANGULATE) { // a fan (v0, v_k, v_k+1); the reference drops everything past the third vertex
                int pv = v[2], pt = tt[2];
                for (char *tok = strtok(buf, " \t\r\n"); tok; tok = strtok(nullptr, " \t\r\n")) {
                    int nv, nt;
                    if (!parse_ref(tok, nv, nt)) break;
                    resolve(nv, nt);
                    face.push_back(v[0] - 1); face.push_back(pv - 1); face.push_back(nv - 1);
                    face_t.push_back(tt[0]); face_t.push_back(pt); face_t.push_back(nt);
                    face_mtl.push_back(ntex);
                    pv = nv; pt = nt;
                }
            }
            break;
        }
        default: // '#' comments and everything else: eat the rest of the line
            if (!fgets(buf, sizeof(buf), fp)) { /* EOF */ }
            break;
        }
    }
    fclose(fp);
    if (rc) return rc;
    const size_t nf = face_mtl.size();
    std::vector<double> verts(9 * nf), tcols(3 * nf), ts(6 * nf);
    std::vector<int32_t> ttex(nf);
    bool any_tex = false;
    for (size_t i = 0; i < nf; i++) {
        for (int j = 0; j < 3; j++) {
            const int p = face[3 * i + j];
            if (p < 0 || (size_t)p >= points.size()) return fail(RT_E_INVALID, "Error face."); // object.cpp:271-277
            verts[9 * i + 3 * j] = points[p].x; verts[9 * i + 3 * j + 1] = points[p].y; verts[9 * i + 3 * j + 2] = points[p].z;
            const int ti = face_t[3 * i + j];
            // (the reference indexes vt[] unchecked, object.cpp:280; an index outside the file's coordinates is an error here)
            if (ti < 0 || (size_t)ti >= vt.size()) return fail(RT_E_INVALID, "Error face.");
            ts[6 * i + 2 * j] = vt[ti].x; ts[6 * i + 2 * j + 1] = vt[ti].y;
        }
        const P3 c = cols[face_mtl[i]];
        tcols[3 * i] = c.x; tcols[3 * i + 1] = c.y; tcols[3 * i + 2] = c.z;
        ttex[i] = tex[face_mtl[i]];
        any_tex = any_tex || ttex[i] >= 0;
    }
    const int mesh = rth_add_mesh(s, verts.data(), (int64_t)nf, lig, col, tcols.data(), type);
    if (mesh >= 0 && (any_tex || vt.size() > 1)) { // (coordinates are kept even where no image uses them, as Polygon::ts is)
        const int r2 = rth_set_mesh_textures(s, mesh, ttex.data(), ts.data());
        if (r2 < 0) return r2;
    }
    return mesh;
}

int rth_load_scene_file(rth_scene *s, const char *path, rth_settings *out) { return rth_load_scene_file_ex(s, path, out, 0u); }

int rth_load_scene_file_ex(rth_scene *s, const char *path, rth_settings *out, unsigned obj_flags) {
    if (!s || !path || !out) return fail(RT_E_INVALID, "rth_load_scene_file: bad arguments");
    FILE *fi = fopen(path, "r");
    if (!fi) return fail(RT_E_INVALID, "Can't open %s", path);
    std::string dir(path);
    const size_t slash = dir.find_last_of('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    int rc = RT_OK, n = 0, m = 0;
    char buf[256], buf2[256];
    if (fscanf(fi, "%d", &n) != 1 || n < 0) rc = fail(RT_E_INVALID, "%s: object count", path);
    for (int i = 0; rc == RT_OK && i < n; i++) { // raytrace.cpp:110-127
        double times, loc[3], rot[3], lig[3], col[3];
        if (fscanf(fi, "%255s%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf%255s", buf, &times, &loc[0], &loc[1], &loc[2], &rot[0], &rot[1],
                   &rot[2], &lig[0], &lig[1], &lig[2], &col[0], &col[1], &col[2], buf2) != 15) {
            rc = fail(RT_E_INVALID, "%s: object %d is malformed", path, i);
            break;
        }
        std::string obj = buf[0] == '/' ? std::string(buf) : dir + buf;
        const int r = rth_add_obj_file_ex(s, obj.c_str(), times, loc, rot, lig, col, type_from_word(buf2), obj_flags);
        if (r < 0) rc = r;
    }
    if (rc == RT_OK && (fscanf(fi, "%d", &m) != 1 || m < 0)) rc = fail(RT_E_INVALID, "%s: sphere count", path);
    for (int i = 0; rc == RT_OK && i < m; i++) { // raytrace.cpp:131-146
        double rad, pos[3], lig[3], col[3];
        if (fscanf(fi, "%lf%lf%lf%lf%lf%lf%lf%lf%lf%lf%255s", &rad, &pos[0], &pos[1], &pos[2], &lig[0], &lig[1], &lig[2], &col[0],
                   &col[1], &col[2], buf) != 11) {
            rc = fail(RT_E_INVALID, "%s: sphere %d is malformed", path, i);
            break;
        }
        rth_add_sphere(s, rad, pos, lig, col, type_from_word(buf));
    }
    if (rc == RT_OK) { // raytrace.cpp:148-151
        rth_settings &o = *out;
        if (fscanf(fi, "%d%d%d%d%d%d", &o.sizex, &o.sizey, &o.c_samps, &o.g_samps, &o.num_per_node, &o.cu_stack_size) != 6 ||
            fscanf(fi, "%lf%lf%lf", &o.view[0], &o.view[1], &o.view[2]) != 3 ||
            fscanf(fi, "%lf%lf%lf", &o.dir[0], &o.dir[1], &o.dir[2]) != 3)
            rc = fail(RT_E_INVALID, "%s: settings lines are malformed", path);
    }
    fclose(fi);
    return rc;
}

void rth_tonemap(const double *rgb, size_t npix, uint8_t *bgr) {
    for (size_t i = 0; i < npix; i++) {
        double c[3];
        for (int k = 0; k < 3; k++) { const double x = rgb[3 * i + k]; c[k] = x < 0 ? 0 : x > 1 ? 1 : x; } // raytrace.cpp:28-31
        bgr[3 * i + 2] = (uint8_t)(int)(std::pow(c[0], 1 / 2.2) * 255 + .5); // raytrace.cpp:36-38
        bgr[3 * i + 1] = (uint8_t)(int)(std::pow(c[1], 1 / 2.2) * 255 + .5);
        bgr[3 * i + 0] = (uint8_t)(int)(std::pow(c[2], 1 / 2.2) * 255 + .5);
    }
}

int rth_write_bmp(const char *path, const uint8_t *bgr, int32_t rows, int32_t cols) {
    if (!path || !bgr || rows < 1 || cols < 1) return fail(RT_E_INVALID, "rth_write_bmp: bad arguments");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(RT_E_INVALID, "Can't open %s for writing", path);
    const uint32_t stride = ((uint32_t)cols * 3 + 3) & ~3u, size = 54 + stride * (uint32_t)rows;
    uint8_t h[54];
    memset(h, 0, sizeof(h));
    h[0] = 'B'; h[1] = 'M';
    auto put32 = [&](int o, uint32_t v) { h[o] = v & 255; h[o + 1] = (v >> 8) & 255; h[o + 2] = (v >> 16) & 255; h[o + 3] = (v >> 24) & 255; };
    put32(2, size); put32(10, 54); put32(14, 40); put32(18, (uint32_t)cols); put32(22, (uint32_t)rows);
    h[26] = 1; h[28] = 24; put32(34, stride * (uint32_t)rows);
    bool ok = fwrite(h, 1, 54, f) == 54;
    std::vector<uint8_t> line(stride, 0);
    for (int32_t r = rows - 1; ok && r >= 0; r--) { // BMP rows are stored bottom-up
        memcpy(line.data(), bgr + (size_t)r * cols * 3, (size_t)cols * 3);
        ok = fwrite(line.data(), 1, stride, f) == stride;
    }
    fclose(f);
    return ok ? RT_OK : fail(RT_E_INVALID, "short write to %s", path);
}

} // extern "C"
