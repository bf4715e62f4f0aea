/*
 * rt_capi.h -- the drop-in boundary: a C ABI for "render every pixel of a
 * rectangle" on an AMD MI355X (gfx950).
 *
 * The reference (ccelio/TileCodeRayTracer) has no plugin/FFI seam: main() takes
 * no arguments and the renderer is the direct call
 *     pixels[x][z] = calculatePixel(createEyeRay(x/W, z/H), 0)
 * inside raytrace_main()'s pixel loop (src/RayTracer.cpp:904-923).  This
 * header cuts the seam exactly there.  Above it sits the reference's C++
 * object model (Scene / SceneObject / Camera, mirrored in
 * tilecoderaytracer_amd/csrc/host/); below it are plain-old-data tables and
 * hand-written HIP kernels.  No C++ or torch types cross this boundary.
 *
 * Conventions kept from the reference: every call returns int, 0 = ok,
 * non-zero = error (Scene::initialize / init_log / raytrace_main,
 * src/Scene.cpp:386, src/RayTracer.cpp:862-873, 2027-2031); nothing throws or
 * aborts across the ABI; rt_last_error() gives the text for the calling thread.
 *
 * There is NO CPU fallback behind this ABI.  If no HIP device is usable every
 * call that needs one fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_CAPI_H_
#define RT_CAPI_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_CAPI_VERSION 2      /* 2: rt_multi_*, rt_chunk_bounds, rt_get_timeline; the second pass ("defer") is gone */

enum {
    RT_OK = 0,
    RT_ERR_INVALID = 1,      /* bad argument / malformed description        */
    RT_ERR_NO_DEVICE = 2,    /* no usable HIP device                        */
    RT_ERR_HIP = 3,          /* a HIP runtime call failed, or a kernel reported trouble (see "help_spin_limit") */
    RT_ERR_CAPACITY = 4,     /* more than 4 096 objects, a bounce stack beyond 8 GB, or an LDS option that does not fit */
    RT_ERR_RCCL = 5          /* an RCCL call failed (rt_render_multi)       */
};

/* primitive kinds: SceneSphere / SceneInfinitePlane / SceneFinitePlane
 * (src/SceneSphere.h:10, src/SceneInfinitePlane.h:16, src/SceneFinitePlane.h:16) */
enum { RT_KIND_SPHERE = 0, RT_KIND_INFINITE_PLANE = 1, RT_KIND_FINITE_PLANE = 2 };

/* One scene object, flattened: the members of SceneObject
 * (src/SceneObject.h:189-199), its ObjMaterial (src/ObjMaterial.h:69-79) and
 * the derived geometry each primitive's constructor computes
 * (src/SceneSphere.cpp:44-48, src/SceneInfinitePlane.cpp:11-26,
 * src/SceneFinitePlane.cpp:18-80).  Objects are listed in Scene index order;
 * that order is observable (nearest-hit ties, per-light clamp order).  At most
 * 4 096 objects per scene (RT_ERR_CAPACITY beyond; the reference's Scene holds
 * 3 999, src/Scene.h:8). */
typedef struct rt_object_desc {
    int32_t kind;                 /* RT_KIND_*                                 */
    int32_t is_light;             /* SceneObject::isaLightSource               */
    int32_t texture;              /* index into rt_scene_desc.textures, -1 none */
    float   intensity;            /* SceneObject::intensity                    */
    float   origin[3];            /* SceneObject::origin (sphere centre, light position, infinite-plane origin) */
    float   color[3];             /* ObjMaterial::myColor                      */
    float   diffuse, specular, reflective;
    float   radius, radius_squared;               /* sphere                    */
    float   plane_origin[3];                      /* finite plane              */
    float   normal[3], vertical[3], horizontal[3], reverse_normal[3]; /* planes */
    float   v_distance, h_distance;               /* finite plane              */
    float   distance_to_origin;                   /* planes                    */
} rt_object_desc;

/* Texture_CheckerBoard (src/Texture_CheckerBoard.h:13-71): the only texture
 * the reference has. */
typedef struct rt_texture_desc {
    float light[3], dark[3];
    float width, height;
} rt_texture_desc;

typedef struct rt_scene_desc {
    int32_t               n_objects;
    const rt_object_desc *objects;
    int32_t               n_textures;
    const rt_texture_desc *textures;
    /* Scene::scene_object_start_index / final_index (src/Scene.h:41-42): on
     * x86 the shadow scan covers [shadow_begin, shadow_end)
     * (src/RayTracer.cpp:716-722).  Scene::initialize() sets [0, count). */
    int32_t               shadow_begin, shadow_end;
    float                 null_color[3];   /* NULL_COLOR, src/RayTracer.h:52 */
} rt_scene_desc;

/* The members of Camera that createEyeRay reads (src/Camera.cpp:71-84). */
typedef struct rt_camera_desc {
    float screen_width, screen_height, screen_halfwidth, screen_halfheight;
    float screen_origin[3], vector_horizontal[3], vector_vertical[3], eye_origin[3];
} rt_camera_desc;

typedef struct rt_scene rt_scene;     /* opaque; owns device copies of the tables */

typedef struct rt_timing {
    double   last_kernel_ms;    /* render kernel of the last rt_render* call (HIP events on its stream) */
    double   sum_kernel_ms;     /* accumulated since rt_reset_timing                                    */
    uint64_t launches;          /* kernel launches accumulated                                          */
    double   last_upload_ms;    /* scene-table upload in rt_scene_create                                */
    double   last_download_ms;  /* device->host copy in rt_render (0 for rt_render_device)              */
} rt_timing;

typedef struct rt_launch_info {
    int32_t block_threads;      /* threads per workgroup                                        */
    int32_t lds_bytes;          /* dynamic LDS per workgroup (scene tables + bounce stack)      */
    int32_t scene_lds_bytes;    /* of which scene tables                                        */
    int32_t grid_blocks;        /* workgroups of the last launch                                */
    int32_t tile_x, tile_z;     /* pixels per wavefront tile (tile_x * tile_z == 64)            */
    char    kernel[48];         /* name of the __global__ function the last launch ran (its first pass) */
} rt_launch_info;

/* Replaces: the Scene the reference keeps in the global my_scene
 * (src/RayTracer.h:50) -- copies the description, uploads the tables to
 * `device`.  The caller keeps ownership of *desc. */
int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out);
int rt_scene_destroy(rt_scene *scene);

/* Replaces: raytrace_main()'s pixel loop (src/RayTracer.cpp:904-923) for
 * columns [x0, x1) and all z of a W x H image, recursion limit max_depth
 * (MAX_RECURSION_LEVEL, src/rt_project_parameters.h:73).  out_rgb is HOST
 * memory, packed fp32: out_rgb[((x-x0)*H + z)*3 + c] -- the pixels[x][z]
 * order of src/RayTracer.h:44.  dx = (float)x / W, dz = (float)z / H use the
 * global W, H, so a strip is bit-identical to the same columns of a full
 * render.  Synchronous. */
int rt_render(rt_scene *scene, const rt_camera_desc *cam, int W, int H,
              int x0, int x1, int max_depth, float *out_rgb);

/* Same, but d_out_rgb is DEVICE memory on the scene's device and the kernel
 * is enqueued on hip_stream (a hipStream_t; NULL = the null stream) without
 * synchronising.  Used by the multi-GPU path and by callers that keep the
 * framebuffer in HBM. */
int rt_render_device(rt_scene *scene, const rt_camera_desc *cam, int W, int H,
                     int x0, int x1, int max_depth, void *d_out_rgb, void *hip_stream);

/* Replaces: the reference's static partitioning (strategy 1,
 * src/RayTracer.cpp:904-923 with CORE_NUM > 1) across the GPUs of one node,
 * single process: contiguous x-strips, one per GPU, gathered to device 0 with
 * ncclGather over xGMI, then copied to out_rgb (host, whole image). */
int rt_render_multi(const rt_scene_desc *desc, const rt_camera_desc *cam, int W, int H,
                    int max_depth, int ngpu, float *out_rgb);

/* The same with everything that can survive from frame to frame kept in a handle: the per-GPU
 * scenes, two streams per GPU, the strip buffers, the image on device 0 and the RCCL communicator
 * (rt_render_multi = create + render + destroy).  rt_multi_render renders and sends every strip in
 * `chunks` column chunks (1..64): chunk k travels to device 0 on the GPU's communication stream
 * (ncclSend / ncclRecv) while its compute stream renders chunk k+1 -- within one frame, the way the
 * reference's ranks write into the shared `pixels` while they render (src/RayTracer.cpp:904-923,
 * 1188-1193).  Every chunk is a kernel launch of its own: 1 is right where a strip's kernel is long
 * next to its transfer, 4..8 where the transfer is as long as the kernel (the built-in scene). */
typedef struct rt_multi rt_multi;
int rt_multi_create(const rt_scene_desc *desc, int ngpu, rt_multi **out);
int rt_multi_render(rt_multi *multi, const rt_camera_desc *cam, int W, int H, int max_depth,
                    int chunks, float *out_rgb);
int rt_multi_set_option(rt_multi *multi, const char *key, int value);   /* rt_set_option on every GPU's scene */
int rt_multi_destroy(rt_multi *multi);

/* Chunk k of `chunks` column chunks of columns [x0, x1): [*a, *b), about equal widths, inner
 * boundaries a multiple of `align` columns from x0 (trailing chunks may be empty).  Returns 0, or
 * 1 on bad arguments. */
int rt_chunk_bounds(int x0, int x1, int chunks, int k, int align, int *a, int *b);

/* The partition rt_render_multi uses: strip g of ngpu equal x-strips of ceil(W / ngpu) columns is
 * columns [*x0, *x1) (trailing strips may be short or empty); returns the strip width, which is also
 * the column stride of the strips in the gathered buffer (rank g at g * width: only trailing strips
 * are short, so columns [0, W) are contiguous at its start).  Returns 0 on bad arguments. */
int rt_strip_bounds(int W, int ngpu, int g, int *x0, int *x1);

/* Diagnostic "counting build" of rt_render (same arithmetic and control flow,
 * plus work counters; slower).  stats[k], k < RT_STATS_COUNT:
 *   0 nearest-hit rays (lanes)        1 shadow rays (lanes)
 *   2 nearest-hit scans (wavefronts)  3 shadow scans (wavefronts)
 *   4 sphere tests issued (wavefronts) 5 plane tests issued (wavefronts)
 *   6 cluster box tests issued (wavefronts)
 *   7 sphere tests the lane itself needed (lanes)
 *   8, 9, 10 shader cycles wavefronts spent in nearest-hit scans, in shadow
 *     scans, and on whole tiles (each wavefront counts its own resident time,
 *     so these are comparable with each other, not with wall time)
 *   11, 12, 13 the same for the winner's collision record, the light loop
 *     (shadow scans included) and the reflection step
 *   14, 15, 16 shadow scans: items left by the bundle cull, leaves some lane
 *     needed, and (summed over scans) the most leaves one lane needed
 *   17, 18, 19, 20 nearest-hit scans (wavefronts) by how many of the 64 lanes traced
 *     a ray: 1-16, 17-32, 33-48, 49-64 (what bounce compaction could merge)
 *   21, 22 nearest-hit scans whose bundle cull was skipped (ray directions of both signs on
 *     every axis), and the cluster box tests issued in them
 *   23, 24 sphere tests of cluster leaves issued in those scans, and in all nearest-hit scans
 * wave_cycles (may be NULL) receives, per wavefront tile in row-major order
 * (tile = tile_row * tiles_x + tile_col), six words {shader cycles the
 * wavefront was resident, sphere tests it issued, box tests it issued, scans
 * it ran, start and end time on the 100 MHz constant clock}, up to
 * n_wave_cycles words.  out_rgb may be NULL.  The reference has no
 * counterpart (its gprof figures are quoted in SURVEY.md section 3.3). */
#define RT_STATS_COUNT 25
int rt_render_stats(rt_scene *scene, const rt_camera_desc *cam, int W, int H, int x0, int x1,
                    int max_depth, float *out_rgb, uint64_t *stats, int n_stats,
                    uint64_t *wave_cycles, int n_wave_cycles);


/* Learn where this scene's launches of ONE shape start handing out their tile rows (speed only).  Renders that shape once with
 * the counting build (about three times a frame's time; no pixels are returned) and keeps, per macro row (four tile rows), the
 * longest tile and the rows' sums; every later rt_render / rt_render_device with the same W, H, x0, x1, max_depth and tile shape
 * starts its queues a little before the row of the longest tile, sweeping up or down, instead of by the start-row rule
 * ("first_row" -1) -- if that measured faster: the call times the rule's sweep and the two learned ones and keeps a learned one
 * only if it beats the rule by 3 %.  A launch that is short of tiles -- one GPU's strip of a multi-GPU frame --
 * ends waiting for its longest tiles, and which they are is a matter of the scene and the camera (the previous frame knows).
 * The timed frames (about twenty) go into the handle's own buffer (they count in rt_get_timing:
 * rt_reset_timing afterwards).  rt_set_option("learned_order", 0) forgets it; so does learning another shape.
 * Replaces nothing in the reference (its workers pull pixels in index order, src/RayTracer.cpp:956-992). */
int rt_learn_tile_order(rt_scene *scene, const rt_camera_desc *camera, int W, int H, int x0, int x1, int max_depth);
/* Diagnostic BUILDS only (make -C tilecoderaytracer_amd/csrc variant NAME=timeline DEFS=-DRT_TIMELINE=1; the product
 * library refuses the option: the few instructions it takes cost the render kernels registers): with option
 * "timeline" = 1 every launch records, per wavefront tile in row-major order
 * (tile = tile_row * tiles_x + tile_col, rt_launch_info's tile shape), four words: {start, end on the GPU's
 * 100 MHz constant clock, workgroup * 16 + wavefront that rendered it, 1 if it was rendered as a HEAVY tile};
 * this copies up to n_words of the last launch's record (waits for the launch). */
int rt_get_timeline(rt_scene *scene, uint64_t *out, int n_words);

int rt_get_timing(const rt_scene *scene, rt_timing *out);
int rt_reset_timing(rt_scene *scene);
int rt_get_launch_info(const rt_scene *scene, rt_launch_info *out);

/* Tuning knobs (speed only, never results).  key:
 *   "tile_z"        wavefront tile height, 1,2,4,...,64 (width = 64 / height)
 *   "block_threads" 0 = auto, else 64, 128, 192 or 256 (a value beyond the launch bounds of
 *                   the kernel a launch picks -- 320..512 -- is refused by that launch)
 *   "stack"         bounce stack: 0 auto, 1 LDS, 2 HBM
 *   "pairs"         scenes with clustered sphere runs: 0 = every needed leaf is tested for
 *                   the whole wavefront (round 1's route); 1 (default) = the (ray, leaf)
 *                   pairs that the per-lane box tests leave are compacted into full
 *                   wavefront rounds
 *   "tables"        where the kernel reads the scene tables: 1 = LDS (staged once per
 *                   workgroup; at most 160 KiB), 2 = global memory through the L2 (any
 *                   size), 0 = automatic (LDS up to 80 KiB)
 *   "grid_mult"     persistent grid = occupancy x CUs x this; 0 = no persistence
 *   "first_row"     where the tile queues start, thousandths of the image
 *                   height (from there upwards; rows wrap around); -1 = automatic:
 *                   row 0 upwards, or -- scenes with a horizon and clustered sphere
 *                   runs -- from a little above the horizon row downwards (tiles in
 *                   order of decreasing cost)
 *   "cull"          0 = the plain scans of the reference: every object one item in
 *                   Scene index order, no wavefront-level culling, no
 *                   nearest-first early exit, no sphere clustering, no
 *                   axis-aligned route (the slow baseline the fast path is
 *                   checked against, pixel for pixel, in tests/)
 *   "help"          scenes with clustered sphere runs: 1 = a wavefront that has run out
 *                   of tiles stays and tests candidate leaves of its workgroup's long
 *                   shadow scans (a desk in LDS, a shared cursor over the candidates;
 *                   blocking is an OR, so who tests which leaf cannot change a pixel):
 *                   shortens the end of a GPU's strip of a frame; -1 (default) =
 *                   automatic: on for launches of at most three quarters of the image's
 *                   width (a whole frame pays 1 % for the owners' looks at the desk and
 *                   ends well without help); 0 = such wavefronts leave; 2..64 = on, and
 *                   a scan asks for help from this many candidate leaves on (default 8;
 *                   tests use 2)
 *   "heavy"         scenes with clustered sphere runs under a horizon (with "help" on): the tiles
 *                   of the band of tile rows along the horizon line -- each keeps a wavefront
 *                   busy for a millisecond -- are rendered first, one per WORKGROUP (one
 *                   wavefront renders, the others share its shadow scans from the first on);
 *                   -1 (default) = automatic: when the launch renders a strip of at most a
 *                   third of the image's width (one GPU's share on three or more), a band of
 *                   0.25 % of the image height either side of the line; 0 = off; k = always,
 *                   k - 1 tile rows either side
 *   "tile_prio"     a wavefront's priority on its SIMD follows the bounce level of its tile (the
 *                   tiles whose rays go on bouncing are the long ones, and a launch short of tiles
 *                   waits for them): -1 (default) = automatic, for strips of at most three fifths
 *                   of the image's width; 0 = off; 1 = on
 *   "help_spin_limit" the bound of an owner's wait for helpers to leave its desk (default
 *                   2^22 polls); -1 makes every such wait count as timed out: the owner then
 *                   tests the leaves itself (same pixels), its workgroup stops helping, and
 *                   the next rt_render / rt_get_timing returns RT_ERR_HIP once (tests)
 *   "timeline"      diagnostic builds: 1 = launches record per tile when and by whom it was rendered
 *                   (rt_get_timeline); the product library accepts 0 only
 *   "fast"          scenes without clustered runs: 1 (default) = one kind-sorted item list
 *                   with direct test records (FAST tables), 0 = the two item tables
 *   "primary"       FAST tables: 1 (default) = the scan of the camera rays culls by the pixel rectangle
 *                   every item's box projects to (computed per launch from the camera; scenes of up to
 *                   64 items), 0 = by the bundle of rays like every other scan
 *   "tight_planes"  0 = plane items get the (much larger) padding of sphere items
 *   "aa_planes"     0 switches the axis-aligned rectangle route off
 *   "cluster_leaf"  spheres per leaf of a clustered run (default -1 = by the run's length: 16 below 512
 *                   spheres, 20 below 896, 24 below 3 000, else 32; 0 = no clustering) */
int rt_set_option(rt_scene *scene, const char *key, int value);

int         rt_device_count(int *count);
int         rt_capi_version(void);
const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_CAPI_H_ */
