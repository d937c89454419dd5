/*
 * rt_hip.h -- C ABI of the MI355X ray-tracing core (librt_hip.so).
 *
 * This is the boundary a RayTrace host program binds instead of the reference's disabled
 * CUDA offload (cuda_raytrace.h:10-13, kdtree.cuh:6-12).  Plain C: pointers, sizes and POD
 * structs only.  Every function returns 0 on success or a negative RT_E_* code; the message
 * for the calling thread's last failure is available from rt_last_error().
 *
 * Image axes follow the reference: x is the ROW and sizex the number of rows, y is the
 * COLUMN and sizey the number of columns (raytrace.cpp:36, 217).
 *
 * Primitive ids reported by this library:
 *     triangle : global id = (triangles of all earlier objects) + face index  (cuda.cu:80,158-159)
 *     sphere   : -1 - sphere index
 *     miss     : RT_ID_MISS
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_E_INVALID (-1)  /* bad argument / inconsistent scene description */
#define RT_E_DEVICE (-2)   /* HIP runtime error */
#define RT_E_NOMEM (-3)
#define RT_E_LIMIT (-4)    /* a size exceeds what the kernels are built for */

#define RT_ID_MISS INT32_MIN

/* Material types, point.h:9 */
#define RT_DIFF 0
#define RT_SPEC 1
#define RT_REFR 2

/*
 * Flattened scene, produced on the host (librt_host.so: rth_scene_desc) from the reference's
 * data model: vector<Object> of Polygons, vector<Sphere>, KdTree.  All arrays stay owned by the
 * caller and are only read during rt_upload_scene.
 *
 * Nodes are in PRE-ORDER (node, left subtree, right subtree) -- the order of the reference's own
 * flattening, travKdTree (cuda.cu:68-89) -- so a left child, when present, is always node+1.
 * Own lists (KdTree::node) are concatenated in the same order; every triangle appears in exactly
 * one own list because straddling triangles stay on the interior node (kdtree.cpp:85-93).
 */
typedef struct rt_scene_desc {
    int32_t n_nodes;           /* >= 1: an empty scene still has the empty root leaf */
    const double *node_box;    /* [n_nodes][6]  mi.xyz, ma.xyz      KdTree::aabb  (kdtree.cpp:34-35) */
    const double *node_split;  /* [n_nodes]     KdTree::split       (kdtree.cpp:82); unused for leaves */
    const int32_t *node_axis;  /* [n_nodes]     KdTree::w           (kdtree.cpp:33) */
    const int32_t *node_left;  /* [n_nodes]     index or -1         KdTree::l */
    const int32_t *node_right; /* [n_nodes]     index or -1         KdTree::r */
    const int32_t *node_first; /* [n_nodes]     first own-list entry */
    const int32_t *node_count; /* [n_nodes]     own-list length     KdTree::node.size() */
    int32_t n_list;            /* total own-list entries, == n_tris */
    const int32_t *list_tri;   /* [n_list]      global triangle ids */
    int32_t tree_depth;        /* nodes on the longest root-to-leaf path */

    int32_t n_tris;
    const double *tri_verts;   /* [n_tris][9]   Polygon::points3d */
    const double *tri_nc;      /* [n_tris][3]   Polygon::nc         (polygon.h:23-27) */
    const double *tri_col;     /* [n_tris][3]   Polygon::col        (getcol, untextured: polygon.h:75) */
    const double *tri_lig;     /* [n_tris][3]   owning Object::lig  (raytrace.cpp:66) */
    const int32_t *tri_type;   /* [n_tris]      owning Object::type (raytrace.cpp:66) */

    int32_t n_spheres;
    const double *sph_rad;     /* [n_spheres]    Sphere::rad */
    const double *sph_pos;     /* [n_spheres][3] Sphere::pos */
    const double *sph_lig;     /* [n_spheres][3] Sphere::lig */
    const double *sph_col;     /* [n_spheres][3] Sphere::col */
    const int32_t *sph_type;   /* [n_spheres]    Sphere::type */

    /* Textures: Polygon::getcol with an image (polygon.h:30-35,73-89).  n_textures == 0 (what a zero-initialised
     * description says): every triangle has its plain colour tri_col, and the pointers below may be NULL. */
    int32_t n_textures;
    const uint8_t *tex_data;   /* the images back to back, each as loadImage leaves it in Object::tex[k]
                                  (opencvlib.cpp:21-29): tex_sizey rows of tex_sizex BGR pixels, 3 bytes each */
    const int64_t *tex_offset; /* [n_textures]  first byte of image k in tex_data */
    const int32_t *tex_sizex;  /* [n_textures]  Polygon::sizex = image columns (opencvlib.cpp:19) */
    const int32_t *tex_sizey;  /* [n_textures]  Polygon::sizey = image rows    (opencvlib.cpp:20) */
    const int32_t *tri_tex;    /* [n_tris]      image of the triangle, -1 = none (Polygon::tex == NULL) */
    const double *tri_ts;      /* [n_tris][6]   Polygon::ts[0..2] (x, y): the face's texture coordinates (object.cpp:280);
                                  read only where tri_tex >= 0 */
} rt_scene_desc;

/* Camera as the scene file gives it (raytrace.cpp:148-151); the basis vectors cx, cy and the
 * normalised direction are derived inside the library exactly as raytrace.cpp:155-156 does. */
typedef struct rt_camera {
    double o[3];
    double d[3];
    int32_t sizex; /* rows */
    int32_t sizey; /* columns */
} rt_camera;

/* Rows [x0, x0+rows) x columns [y0, y0+cols) of the image. */
typedef struct rt_tile {
    int32_t x0, y0, rows, cols;
} rt_tile;

/*
 * Integrator controls.  reference behaviour = {samps, 5, 1, 2, <large>, 1, seed, 0, 0}.
 *   rr_depth    Russian roulette starts once ++depth > rr_depth       (raytrace.cpp:70; reference 5)
 *   rr_mode     1: roulette with survival probability max(f)          (raytrace.cpp:71-74)
 *               0: the path ends there and returns the emission -- a hard depth bound
 *   split_depth glass traces BOTH branches while depth <= split_depth (raytrace.cpp:93; reference 2), max 4
 *   hard_depth  the path ends once ++depth > hard_depth; the reference recursion has no such bound
 *   jitter      1: tent-filter sub-pixel jitter from two draws        (raytrace.cpp:200-201)
 *               0: dx = dy = 0 and no draws are consumed
 *   seed        every path (pixel, sub-pixel, sample) restarts the drand48 recurrence from a state
 *               hashed from (seed, path index), so results do not depend on tiling or GPU count
 *   light_mode  0: the reference's integrator -- emission is found by the paths alone
 *               1: explicit light sampling ("shadow rays"); NO counterpart in the reference (its light.h is
 *                  never used).  At every diffuse hit one emissive sphere, picked uniformly, is sampled over
 *                  the cone it subtends (three more draws, after the bounce's two) and a shadow ray -- a ray
 *                  like any other in the counters -- decides whether it is seen; the bounce ray then adds no
 *                  emissive sphere's emission at its hit.  Emissive triangles are never sampled and always
 *                  add.  Converges to the same image as mode 0, with less noise for small lights.
 */
typedef struct rt_params {
    int32_t samps;
    int32_t rr_depth;
    int32_t rr_mode;
    int32_t split_depth;
    int32_t hard_depth;
    int32_t jitter;
    uint64_t seed;
    int32_t light_mode;
    int32_t reserved; /* 0 */
} rt_params;

/* Work counters. A ray is one nearest-hit query = one radiance() invocation of the reference. */
typedef struct rt_stats {
    uint64_t rays;        /* radiance() invocations            (raytrace.cpp:41) */
    uint64_t node_visits; /* KdTree::check invocations         (kdtree.cpp:111) */
    uint64_t tri_tests;   /* Polygon::intersect in check_node  (kdtree.cpp:105): what the reference would run */
    uint64_t sph_tests;   /* Sphere::intersect                 (sphere.cpp:8) */
    uint64_t tri_shades;  /* rays whose chosen hit is a triangle (raytrace.cpp:58-67) */
    /* what the kernel actually executed (long own lists are culled conservatively, results unchanged).  These two and
     * the scheduler's figures below depend on which boxes the best hit so far lets the kernel drop, i.e. on the order
     * in which a list is scanned: in the kernel build for deep trees, where a wavefront's last slots may be scanned by
     * all its lanes together, they can differ by a few from run to run.  The five counters above never do. */
    uint64_t tri_tests_done; /* exact triangle tests performed */
    uint64_t cull_tests;     /* culling records (four boxes each) tested */
    /* the wavefront scheduler: steps executed and lanes that took part in them
     * (sched_lanes / (64 * sched_steps) = fraction of lanes doing useful work per step) */
    uint64_t sched_steps;
    uint64_t sched_lanes;
    /* records actually fetched: node visits that loaded the node's records (a visit decided in the parent's
     * step from the child's box loads nothing), and spheres tested one by one (through the sphere hierarchy,
     * or by the scan when the scene has no hierarchy) */
    uint64_t node_fetches;
    uint64_t sph_tests_done;
} rt_stats;

typedef struct rt_scene rt_scene; /* opaque, one per (scene, device) */

/* replaces check_device() (cuda_raytrace.h:10, device_query.cu:5-14): number of GPUs, or <0 */
int rt_device_count(void);

/* replaces print_device_info() (cuda_raytrace.h:11, device_query.cu:16-64): writes a
 * NUL-terminated description of device `dev` into buf (at most cap bytes) */
int rt_device_info(int dev, char *buf, size_t cap);

/* replaces cudaSetDevice + cudaDeviceSetLimit + copyToDevice (raytrace.cpp:165-171,
 * cuda_raytrace.h:12, cuda.cu:91-176): validates desc, lays the scene out in HBM on device `dev` */
int rt_upload_scene(int dev, const rt_scene_desc *desc, rt_scene **out);

int rt_free_scene(rt_scene *s);

/*
 * rt_upload_scene in two steps, for hosts that drive several GPUs: copyToDevice (cuda.cu:91-176) ran once per MPI
 * rank, each of which had flattened the tree for itself (travKdTree, cuda.cu:68-89).  rt_prepare_scene does all the
 * host-side work of an upload -- validation, the device layout of nodes, own lists and their culling hierarchies,
 * triangles, spheres, textures -- without touching a GPU; rt_upload_prepared copies that layout to one device and
 * may be called for different devices from different threads at the same time (the prepared scene is only read).
 * rt_upload_scene(dev, desc, out) is rt_prepare_scene + rt_upload_prepared + rt_free_prepared.  `desc` is only read
 * during rt_prepare_scene.
 */
typedef struct rt_prepared rt_prepared; /* opaque, host memory only */
int rt_prepare_scene(const rt_scene_desc *desc, rt_prepared **out);
int rt_upload_prepared(int dev, const rt_prepared *prep, rt_scene **out);
int rt_free_prepared(rt_prepared *prep);

/*
 * replaces the per-pixel cuda_radiance() loop (cuda_raytrace.h:13, raytrace.cpp:186-192,
 * cuda.cu:412-427) for a whole tile: one kernel launch, synchronous, HOST output buffers.
 *   rgb_out     [rows*cols*3] linear colour after the per-sub-pixel clamp and *0.25 average,
 *               i.e. the value the reference accumulates into col[] (raytrace.cpp:205)
 *   prim_id_out optional [rows*cols]: primitive hit by the pixel's first camera ray
 *               (sub-pixel 0,0, sample 0)
 *   stats       optional; the scene's work counters are cleared before the launch and after it (what earlier
 *               rt_render_tiles_device launches accumulated and rt_get_stats has not fetched is discarded)
 */
int rt_render_tile(rt_scene *s, const rt_camera *cam, const rt_tile *tile, const rt_params *p,
                   double *rgb_out, int32_t *prim_id_out, rt_stats *stats);

/*
 * Same work for a list of tiles in ONE launch with DEVICE output buffers, asynchronous on
 * `stream` (a hipStream_t, NULL = default stream).  Tile k's pixels are written contiguously
 * (row-major inside the tile) starting at pixel offset sum_{j<k} rows_j*cols_j, so a GPU's share
 * of the frame is one packed buffer ready for a gather.  d_prim_id may be NULL.
 * Work counters accumulate over launches until rt_get_stats fetches and clears them.
 * One host thread at a time may call into one rt_scene (different rt_scene objects are independent), but the
 * launches it queues on DIFFERENT streams may run side by side on the GPU: a scene keeps two sets of per-launch
 * state (the second is allocated when a second stream first shows up), so one frame's last paths overlap the
 * next frame's first -- worth +16 % on a 540 x 1920 share of the 4K frame.  A launch on a third stream waits
 * for the set used longest ago.  Launches on one stream run one after the other, as on any stream.
 */
int rt_render_tiles_device(rt_scene *s, const rt_camera *cam, const rt_tile *tiles, int32_t n_tiles,
                           const rt_params *p, double *d_rgb, int32_t *d_prim_id, void *stream);

/*
 * The same work as rt_render_tiles_device for callers that cannot allocate device memory: tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  The call allocates and frees nothing
 * and looks nothing up: the work counter, the tile tables and the per-pixel sub-pixel buffers that
 * rt_render_tiles_device keeps in (growing) allocations of the scene live in one fixed array in device memory
 * that each GPU holds from library load until the process ends (about 25 MiB; rt_upload_scene resolves its address
 * on the scene's device), which bounds a call to RT_NOALLOC_MAX_TILES tiles, RT_NOALLOC_MAX_PIXELS pixels and
 * RT_NOALLOC_MAX_BLOCKS blocks of 8 x 8 pixels (a ragged tile edge takes a block of its own); more is rejected
 * with RT_E_LIMIT.  Results are identical to rt_render_tiles_device's.  Every call on a GPU uses that GPU's one
 * array: calls that overlap on one GPU -- from any scene, on any stream -- race; order them on one stream or
 * synchronise between them.  The call records events on `stream` (for rt_get_stats and rt_last_kernel_ms), so it
 * is not meant to be captured into a HIP graph.
 */
#define RT_NOALLOC_MAX_TILES 64
#define RT_NOALLOC_MAX_PIXELS (1 << 18)
#define RT_NOALLOC_MAX_BLOCKS 8192
int rt_render_tiles_noalloc(rt_scene *s, const rt_camera *cam, const rt_tile *tiles, int32_t n_tiles,
                            const rt_params *p, double *d_rgb, int32_t *d_prim_id, void *stream);

/* Work counters summed over every launch on this scene since the previous rt_get_stats call
 * (or since upload); synchronises the last stream used, then clears the counters. */
int rt_get_stats(rt_scene *s, rt_stats *stats);

/* Milliseconds the render kernel queued LAST on this scene took on its stream (every set of per-launch state has
 * its own pair of HIP events around its kernel, so a launch overlapping on another stream does not disturb the
 * figure); synchronises on the closing event. */
int rt_last_kernel_ms(rt_scene *s, float *ms);

/*
 * Nearest-hit queries for explicit rays (6 doubles each: origin, direction) -- the pair of calls
 * at raytrace.cpp:46-47, KdTree::check and sphere_intersect.  Host buffers, synchronous.
 *   tri_id/tri_t  global triangle id or -1, and the distance the query left behind (1e20 on a miss)
 *   sph_id/sph_t  sphere index or -1, and its distance (1e20 on a miss)
 * Like rt_render_tile it clears the scene's work counters before and after.
 */
int rt_trace_rays(rt_scene *s, int64_t n, const double *rays, int32_t *tri_id, double *tri_t,
                  int32_t *sph_id, double *sph_t, rt_stats *stats);

/* IEEE-754 check of the device's double-precision divide (op 0: a/b) and square root (op 1:
 * sqrt(a)), the only non-trivial operations bit-exact ids depend on.  Host buffers. */
int rt_debug_fp64(int dev, int op, int64_t n, const double *a, const double *b, double *out);

/* Message for the calling thread's most recent failing call ("" if none). */
const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
