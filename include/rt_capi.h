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
 * This header is the whole drop-in surface: scenes, renders, the multi-GPU
 * partition, timing, errors.  Speed-only options, the counting build and the
 * diagnostic calls live in rt_capi_tuning.h (its own version number); nothing
 * a maintainer of the reference needs is there.
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

#define RT_CAPI_VERSION 4      /* 3: strips cut by measured cost (rt_multi_render chunks = 0, rt_balance_strips, rt_suggest_chunks,
                                * rt_multi_set_bounds, rt_multi_get_info); options, counters and calibration moved to rt_capi_tuning.h
                                * 4: rt_shared_image_* (one image in one GPU's HBM that the other GPUs' processes render into); rt_multi_render measures
                                *    a direct-store transport beside RCCL (rt_multi_info.transport, .trial_frame_ms) */

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
 * single process: contiguous x-strips, one per GPU, each GPU's columns sent to
 * device 0 over its own xGMI link (ncclSend / ncclRecv) while it renders the next
 * ones, then copied to out_rgb (host, whole image).  The strips are cut by
 * measured cost and sent in as many column chunks as the measurement suggests
 * (rt_multi_render with chunks = 0, below): create + render + destroy. */
int rt_render_multi(const rt_scene_desc *desc, const rt_camera_desc *cam, int W, int H,
                    int max_depth, int ngpu, float *out_rgb);

/* The same with everything that can survive from frame to frame kept in a handle: the per-GPU
 * scenes, two streams per GPU, the strip buffers, the image on device 0, the RCCL communicator and
 * the partition.  rt_multi_render renders and sends every strip in `chunks` column chunks: chunk k
 * travels to device 0 on the GPU's communication stream (ncclSend / ncclRecv) while its compute
 * stream renders chunk k+1 -- within one frame, the way the reference's ranks write into the shared
 * `pixels` while they render (src/RayTracer.cpp:904-923, 1188-1193).  Every chunk is a kernel launch
 * of its own: 1 is right where a strip's kernel is long next to its transfer, 4..8 where the
 * transfer is as long as the kernel (the built-in scene).
 *   chunks = 1..64  that many, on equal strips (rt_strip_bounds) or the strips of rt_multi_set_bounds
 *   chunks = 0      automatic.  The first such call for a frame shape (W, H, max_depth, camera) renders
 *                   three extra frames on equal strips -- one to warm up, the kernels alone, the
 *                   transfers alone -- and cuts the strips so that device 0, which receives and sends
 *                   nothing, renders as long as a peer needs to render and ship its columns
 *                   (rt_balance_strips), with rt_suggest_chunks's chunk count; later calls of the same
 *                   shape reuse the cut.  The image is the same for every partition.
 *                   Where every GPU may address device 0's memory (peer access over xGMI) the same call also measures
 *                   the DIRECT transport -- every GPU's kernel stores its strip straight into the image on device 0,
 *                   as the reference's ranks write into the one `pixels` array; no strip buffer, no transfer, strips of
 *                   equal measured kernel time -- renders two frames of each transport on its own cut, compares sampled columns of both
 *                   images with GPU 0's own rendering, and keeps the faster of those that are right (rt_multi_info.transport,
 *                   .trial_frame_ms, .trial_image_ok).
 * A failed frame leaves nothing queued on any GPU and no RCCL group open; after a failed RCCL call
 * the handle refuses further frames (RT_ERR_RCCL): destroy it. */
#define RT_MULTI_MAX_GPUS 16
typedef struct rt_multi rt_multi;
enum { RT_MULTI_TRANSPORT_AUTO = 0, RT_MULTI_TRANSPORT_RCCL = 1, RT_MULTI_TRANSPORT_DIRECT = 2 };
typedef struct rt_multi_info {                      /* of the last rt_multi_render */
    int32_t ngpu, chunks;
    int32_t balanced;                               /* 1: the strips were cut by measured cost */
    int32_t transport;                              /* RT_MULTI_TRANSPORT_RCCL: strip buffers + ncclSend / ncclRecv; _DIRECT: the kernels stored into device 0's image */
    int32_t bounds[RT_MULTI_MAX_GPUS + 1];          /* GPU g rendered columns [bounds[g], bounds[g + 1]) */
    double  kernel_ms[RT_MULTI_MAX_GPUS];           /* per GPU: its kernels of that frame (HIP events) */
    double  frame_ms;                               /* host clock: first enqueue until everything was on device 0 */
    double  measured_kernel_ms[RT_MULTI_MAX_GPUS];  /* what the cut was computed from: every GPU's equal strip ... */
    double  measured_gather_ms;                     /* ... and the equal strips' transfers on their own */
    int32_t trial_image_ok[2];                      /* ... and whether that transport's image equalled GPU 0's own rendering on sampled columns
                                                     * (1 / 0; -1: not tried): a faster transport with a wrong image is not chosen */
    double  trial_frame_ms[2];                      /* the automatic choice of transport: the faster of two frames of each on its own cut, [0] RCCL, [1] direct (0: not tried) */
} rt_multi_info;
int rt_multi_create(const rt_scene_desc *desc, int ngpu, rt_multi **out);
int rt_multi_render(rt_multi *multi, const rt_camera_desc *cam, int W, int H, int max_depth,
                    int chunks, float *out_rgb);
/* strips given by the caller for images W wide: GPU g renders columns [bounds[g], bounds[g + 1]), bounds[0] = 0,
 * bounds[ngpu] = W, non-decreasing (empty strips allowed); `chunks` is what rt_multi_render(chunks = 0) then uses.
 * bounds = NULL: back to equal strips / the measured cut. */
int rt_multi_set_bounds(rt_multi *multi, int W, const int *bounds, int chunks);
int rt_multi_get_info(const rt_multi *multi, rt_multi_info *out);
int rt_multi_destroy(rt_multi *multi);

/* The cut (pure arithmetic, no device): ngpu contiguous strips of a W-column image, in GPU order, that minimise the frame
 * time when GPU 0 only renders and every other GPU renders and sends -- max(R, S) + min(R, S) / chunks for a peer whose
 * strip takes R to render and S to send (chunks = 1: R + S).  A column costs what its strip cost in the measurement:
 * kernel_ms[g] spread evenly over columns [measured_bounds[g], measured_bounds[g + 1]); sending one column takes
 * send_ms_per_column.  out_bounds: ngpu + 1 ints.  The same arithmetic, bit for bit, as balanced_bounds() of
 * tilecoderaytracer_amd/distributed.py (bench.py's one-process-per-GPU path).  Returns 0, or 1 on bad arguments.
 * Replaces: `dz = SCREEN_VERTICAL_RESOLUTION / CORE_NUM` (src/RayTracer.cpp:904-912), which assumes equal cost. */
int rt_balance_strips(int W, int ngpu, const int *measured_bounds, const double *kernel_ms,
                      double send_ms_per_column, int chunks, int *out_bounds);
/* column chunks per strip: 1 while a strip's transfer is short next to its kernel, up to `most` where it is as long or
 * longer: round(4 send_ms / kernel_ms) clamped to [1, most] */
int rt_suggest_chunks(double kernel_ms, double send_ms, int most);

/* Chunk k of `chunks` column chunks of columns [x0, x1): [*a, *b), about equal widths, inner
 * boundaries a multiple of `align` columns from x0 (trailing chunks may be empty).  Returns 0, or
 * 1 on bad arguments. */
int rt_chunk_bounds(int x0, int x1, int chunks, int k, int align, int *a, int *b);

/* The equal partition (what the measurement starts from): strip g of ngpu equal x-strips of ceil(W / ngpu) columns is
 * columns [*x0, *x1) (trailing strips may be short or empty); returns the strip width, which is also
 * the column stride of the strips in the gathered buffer (rank g at g * width: only trailing strips
 * are short, so columns [0, W) are contiguous at its start).  Returns 0 on bad arguments. */
int rt_strip_bounds(int W, int ngpu, int g, int *x0, int *x1);

/* Replaces: the one `pixels` array that every rank of the reference writes into while it renders
 * (src/RayTracer.h:44; src/RayTracer.cpp:904-923, 1188-1196: the Tilera tiles share that memory) -- for one process per GPU.
 * The process that owns the frame creates it in ITS GPU's HBM (rt_shared_image_create: `bytes` of device memory, a
 * hipMalloc of its own, and the 64-byte handle that names it); the handle travels to the other processes of the node by
 * whatever channel they have (a pipe, a file, torch.distributed); each maps the image (rt_shared_image_open, on the GPU it
 * renders with) and passes `mapped + (size_t)x0 * H * 3 * sizeof(float)` as rt_render_device's d_out_rgb: the kernel's
 * pixel stores travel over xGMI into the owner's HBM as they are issued -- no strip buffer, no gather step, nothing for
 * the owner's GPU to do.  A strip is complete, and visible to the owner, when the stream it was rendered on has drained
 * (hipStreamSynchronize / an event): tell the owner then, by the same channel.  The owner renders its own columns through
 * the pointer rt_shared_image_create gave it.  close: a mapping of rt_shared_image_open; destroy: the owner's allocation,
 * after every mapping is closed.  int returns as everywhere (RT_OK, else rt_last_error()). */
#define RT_SHARED_HANDLE_BYTES 64
int rt_shared_image_create(int device, uint64_t bytes, void **d_image, unsigned char handle[RT_SHARED_HANDLE_BYTES]);
int rt_shared_image_open(int device, const unsigned char handle[RT_SHARED_HANDLE_BYTES], void **d_image);
int rt_shared_image_close(int device, void *d_image);
int rt_shared_image_destroy(int device, void *d_image);

int rt_get_timing(const rt_scene *scene, rt_timing *out);
int rt_reset_timing(rt_scene *scene);

int         rt_device_count(int *count);
int         rt_capi_version(void);
const char *rt_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_CAPI_H_ */
