/*
 * rt_host.h -- C ABI of the host-side scene library (librt_host.so, no GPU dependency).
 *
 * It holds what the reference keeps on the CPU around the offloaded path: the scene model
 * (vector<Object> of Polygons + vector<Sphere>, raytrace.cpp:24-25), the `test.txt` scene file
 * and OBJ readers (raytrace.cpp:106-151, object.cpp:9-286), the median-split kd-tree builder
 * (kdtree.cpp:6-96), its pre-order flattening into the arrays of rt_scene_desc
 * (cuda.cu:68-89), and the image output (drawPixel, raytrace.cpp:33-39; BMP instead of
 * OpenCV's imwrite).  Functions return 0 on success or a negative code; rth_last_error()
 * gives the calling thread's last message.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;

/* The numbers of the scene file's settings lines (raytrace.cpp:148-151). */
typedef struct rth_settings {
    int32_t sizex, sizey;     /* rows, columns */
    int32_t c_samps, g_samps; /* CPU / GPU samples per sub-pixel */
    int32_t num_per_node;     /* ranks per node (unused here) */
    int32_t cu_stack_size;    /* CUDA stack bytes (unused here) */
    double view[3], dir[3];   /* camera position and direction */
} rth_settings;

rth_scene *rth_scene_new(void);
void rth_scene_free(rth_scene *s);

/* One Object (raytrace.cpp:112-126) given as triangles: verts[n_tris][9]; tri_cols (optional,
 * [n_tris][3]) plays the role of the per-material colour (object.cpp:283 cols[ntex]), otherwise
 * `col` applies to every triangle.  lig/type belong to the object.  Returns the object index. */
int rth_add_mesh(rth_scene *s, const double *verts, int64_t n_tris, const double *lig, const double *col,
                 const double *tri_cols, int type);

/* One texture image as cv::imread hands it to loadImage (opencvlib.cpp:15): rows x cols pixels of BGR bytes,
 * row-major, top row first.  Returns the image index used by rth_set_mesh_textures. */
int rth_add_texture(rth_scene *s, const uint8_t *bgr, int32_t rows, int32_t cols);

/* loadImage (opencvlib.cpp:11-30) without OpenCV: reads a 24- or 32-bit uncompressed BMP or a binary PPM (P6),
 * resizes it to (int)(cols*times) x (int)(rows*times) and adds it.  times == 1 copies the image, as cv::resize does
 * for equal sizes; other factors follow cv::resize's bilinear (INTER_LINEAR) fixed-point arithmetic as published,
 * which no OpenCV build was available to check against.  Returns the image index; RT_E_INVALID if the file cannot be
 * read (the reference then leaves the material untextured, which is what rth_add_obj_file does with this result). */
int rth_load_texture_file(rth_scene *s, const char *path, double times);

/* Texture coordinates and images for the triangles of object `mesh` (index from rth_add_mesh): tri_tex[n] = image
 * index or -1 (Polygon::tex == NULL), tri_ts[n][6] = Polygon::ts[0..2] (x, y). */
int rth_set_mesh_textures(rth_scene *s, int mesh, const int32_t *tri_tex, const double *tri_ts);

/* One Sphere (raytrace.cpp:133-145). Returns the sphere index. */
int rth_add_sphere(rth_scene *s, double rad, const double *pos, const double *lig, const double *col, int type);

/* Object::readfile (object.cpp:9-286) for a Wavefront OBJ with its MTL library: scale, rotate (degrees,
 * Rx*Ry*Rz), translate; per-material colour (Kd for DIFF objects, Ks otherwise) and texture (map_Kd / map_Ks,
 * scaled by the object's scale as the reference does); triangles only (a 4th vertex on an `f` line is dropped,
 * as there).  Image paths are tried relative to the OBJ's directory, then as given. */
int rth_add_obj_file(rth_scene *s, const char *path, double scale, const double *loc, const double *rot_deg,
                     const double *lig, const double *col, int type);

/* The same reader with extensions the reference does not have (flags == 0 is rth_add_obj_file):
 *   RTH_OBJ_TRIANGULATE  faces of more than three vertices become a fan of triangles (v0, v_k, v_k+1)
 *   RTH_OBJ_NEGATIVE     negative (relative) vertex / texture indices are resolved, as the OBJ format defines them */
#define RTH_OBJ_TRIANGULATE 1u
#define RTH_OBJ_NEGATIVE 2u
int rth_add_obj_file_ex(rth_scene *s, const char *path, double scale, const double *loc, const double *rot_deg,
                        const double *lig, const double *col, int type, unsigned flags);

/* The reference's scene file (raytrace.cpp:106-151). OBJ paths are resolved relative to the
 * scene file's directory. */
int rth_load_scene_file(rth_scene *s, const char *path, rth_settings *out);
/* ... reading its OBJ files with rth_add_obj_file_ex(flags). */
int rth_load_scene_file_ex(rth_scene *s, const char *path, rth_settings *out, unsigned obj_flags);

/* kd-tree build (kdtree.cpp:6-96) + pre-order flattening. Must be called after the last add. */
int rth_build(rth_scene *s);

/* Valid after rth_build until the scene is changed or freed. */
const rt_scene_desc *rth_desc(const rth_scene *s);

int64_t rth_num_tris(const rth_scene *s);
int32_t rth_num_spheres(const rth_scene *s);

/* drawPixel (raytrace.cpp:33-39) for npix pixels: clamp, gamma 1/2.2, round; BGR byte order. */
void rth_tonemap(const double *rgb, size_t npix, uint8_t *bgr);

/* 24-bit BMP, rows x cols, from BGR bytes in row-major top-down order. */
int rth_write_bmp(const char *path, const uint8_t *bgr, int32_t rows, int32_t cols);

const char *rth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_H */
