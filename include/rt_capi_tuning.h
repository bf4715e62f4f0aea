/*
 * rt_capi_tuning.h -- the part of libtcrt.so's C ABI that is NOT the drop-in
 * surface: speed-only options, the counting build, launch diagnostics and the
 * one calibration call.  Nothing here changes a pixel (every option is covered
 * by a bit-exactness test), nothing here has a counterpart in the reference, and
 * INTEGRATION.md sections 1-2 do not need it.  Versioned on its own
 * (RT_CAPI_TUNING_VERSION / rt_capi_tuning_version()): the drop-in surface of
 * rt_capi.h can stay put while knobs come and go.
 */
#ifndef RT_CAPI_TUNING_H_
#define RT_CAPI_TUNING_H_

#include "rt_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RT_CAPI_TUNING_VERSION 1

typedef struct rt_launch_info {
    int32_t block_threads;      /* threads per workgroup                                        */
    int32_t lds_bytes;          /* dynamic LDS per workgroup (scene tables + bounce stack)      */
    int32_t scene_lds_bytes;    /* of which scene tables                                        */
    int32_t grid_blocks;        /* workgroups of the last launch                                */
    int32_t tile_x, tile_z;     /* pixels per wavefront tile (tile_x * tile_z == 64)            */
    char    kernel[48];         /* name of the __global__ function the last launch ran (its first pass) */
} rt_launch_info;

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

int rt_get_launch_info(const rt_scene *scene, rt_launch_info *out);

/* Tuning knobs (speed only, never results).  key:
 *   "tile_z"        wavefront tile height, 1,2,4,...,64 (width = 64 / height)
 *   "block_threads" 0 = auto (256; 512 for scenes with clustered sphere runs whose tables are so large that
 *                   four-wavefront workgroups would leave LDS room for fewer than six wavefronts per SIMD),
 *                   else a multiple of 64 up to 1024 (a value beyond the launch bounds of the kernel a
 *                   launch picks -- 256, or 512 for the clustered-scene kernels and the counting builds --
 *                   is refused by that launch)
 *   "wide"          scenes with clustered sphere runs: which kernel (-1 automatic: the 96-register one when
 *                   LDS admits fewer than six wavefronts per SIMD anyway; 0 the 80-register one; 1 the other)
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
 *   "svox"          scenes with clustered sphere runs, at most 64 shadow items and two lights: SHADOW
 *                   VOXELS -- the host lays a grid over the region the leaves occupy (plus a few
 *                   ever larger cells beyond it on every side) and notes, per voxel and light, which
 *                   leaves can block the segment from ANY point of the voxel to that light (every
 *                   slack of the float sphere test included); a shadow scan ORs its lanes' voxel
 *                   masks and drops every other candidate leaf before the per-ray box tests.
 *                   -1 (default) = automatic: 4 096 voxels, scenes with at least 24 leaves;
 *                   0 = no table; n = at most n voxels, from 4 leaves on
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

/* rt_set_option on every GPU's scene -- and the handle's own option "transport": RT_MULTI_TRANSPORT_AUTO (0, the default: measured,
 * include/rt_capi.h at rt_multi_render), _RCCL (1) or _DIRECT (2: refused unless every GPU has peer access to device 0) */
int rt_multi_set_option(rt_multi *multi, const char *key, int value);

int rt_capi_tuning_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_CAPI_TUNING_H_ */
