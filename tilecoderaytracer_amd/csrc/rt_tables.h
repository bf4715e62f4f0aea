/*
 * rt_tables.h -- the device scene format: what rt_scene_create() packs from an
 * rt_scene_desc and what every workgroup stages into LDS.
 *
 * The image is an array of 16-byte "quads" (float4).  Sections, in order:
 *
 *   geometry   one record per object; Scene index order, except that the
 *              spheres of a clustered run are stored leaf by leaf
 *                sphere          1 quad : {cx, cy, cz, r^2}
 *                infinite plane  5 quads: q0 {n.xyz, distance_to_origin}
 *                finite plane    5 quads  q1 {anchor.xyz, h_distance}
 *                                         q2 {horizontal.xyz, v_distance}
 *                                         q3 {vertical.xyz, 0}
 *                                         q4 {reverseNormal.xyz, 0}
 *                (anchor = SceneObject::origin for the infinite plane,
 *                 plane_origin for the finite plane: the point texture/bounds
 *                 coordinates are measured from)
 *   aa         2 quads per axis-aligned finite plane (its fast test record):
 *                {dto, sn, sh, sv}, {po_a, po_b, h_dist, v_dist}
 *                in coordinates permuted to (normal, horizontal, vertical) axis
 *   cidx       one u32 per clustered sphere: its Scene index (member order)
 *   near items, shadow items   2 quads per ITEM, see below
 *   lights     2 quads per light, Scene index order:
 *                {origin.xyz, intensity}, {colour.rgb, bits(object index)}
 *   materials  2 quads per DISTINCT (ObjMaterial, intensity, light flag) row:
 *                {colour.rgb, diffuse}, {specular, reflective, intensity,
 *                 bits(is_light | (texture+1) << 1)}
 *   textures   2 quads per checkerboard: {light.rgb, width}, {dark.rgb, height}
 *   objinfo    one u32 per object (4 per quad):
 *                bits 0-15 geometry offset (quads), 16-17 kind, 20-31 material row
 *
 * Clustered sphere runs.  A run of >= 4*leaf consecutive spheres is regrouped
 * (k-d median split of the centres) into spatial leaves of <= leaf spheres.  A leaf's
 * box is axis-aligned, inflated, and contains every member sphere.  Members are visited out of Scene order; the nearest hit
 * stays exact because ties are broken on the Scene index (the lexicographic
 * minimum of (distance, index) is what an in-order scan with a strict `<`
 * computes).
 *
 * Items.  Both scans of the reference -- getCollision over all objects
 * (src/RayTracer.cpp:50-89) and the shadow scan over the non-light objects of
 * the scan range (src/RayTracer.cpp:709-739) -- walk a table of ITEMS: one per
 * object that is not in a clustered run, in Scene index order, then one per
 * LEAF of each clustered run.  2 quads per item:
 *   {box CENTRE.xyz, bits(kind | count << 8 | geometry quad offset << 16)},
 *   {box HALF-EXTENT.xyz, bits(Scene index | quad offset of the full 5-quad plane record << 12)}
 * (the box is [centre - half, centre + half]; an axis the item is unbounded on: centre 0, half-extent infinity)
 * kind = RT_KIND_SPHERE / _INFINITE_PLANE / _FINITE_PLANE;
 *        RT_KIND_FINITE_AA + normal axis for an axis-aligned rectangle (geometry
 *        offset = its aa record);
 *        RT_KIND_SPHERE_LEAF for a leaf (count = its members, geometry offset =
 *        its first member sphere, second word = u32 index of its members'
 *        Scene indices).
 * The box (inflated on the host) contains the object; an infinite plane's box
 * is all of space.  The wavefront culls 64 item boxes at once, one per lane
 * (rt_kernel.hip: nearest_hit_items, in_shade).
 */
#ifndef RT_TABLES_H_
#define RT_TABLES_H_

#include <stdint.h>

#define RT_SPHERE_QUADS 1
#define RT_PLANE_QUADS  5
#define RT_LIGHT_QUADS  2
#define RT_MAT_QUADS    2
#define RT_TEX_QUADS    2

/* item kinds: RT_KIND_* of rt_capi.h (0 sphere, 1 infinite plane, 2 finite
 * plane), plus axis-aligned finite planes: kind = 4 + axis of the normal (0 x, 1 y, 2 z).
 * Their 2-quad test record lists the two in-plane axes in cyclic order after the
 * normal's: {dto, sign_n, sign_a, sign_b}, {origin_a, origin_b, extent_a, extent_b}.
 * The full 5-quad record of each plane is kept too (winner record, non-finite rays). */
#define RT_KIND_FINITE_AA 4
/* one leaf of a clustered sphere run as an item: count = its members, geometry
 * offset = its first member sphere, second word = u32 index of its members'
 * Scene indices */
#define RT_KIND_SPHERE_LEAF 10
#define RT_AA_QUADS 2
/* bit 4 of an item's first word: the item is a finite plane, whose box is padded for a plane's rounding only
 * and whose slack in the wavefront culls is RT_PLANE_SLACK of the distance instead of RT_SPHERE_SLACK */
#define RT_ITEM_TIGHT 16u

#define RT_MAX_GEOM_QUADS 65535   /* 16-bit geometry offset in objinfo */
#define RT_MAX_MATERIALS  4095    /* 12-bit material row in objinfo    */
#define RT_MAX_OBJECTS    4096    /* 12-bit Scene index in the item tables */
#define RT_MAX_LDS_BYTES  (160 * 1024)
#ifndef RT_LDS_TABLE_BYTES
#define RT_LDS_TABLE_BYTES (80 * 1024)   /* automatic: larger tables than this stay in global memory (rt_render_kernel_large) */
#endif
#ifndef RT_STACK_LDS_SHARE
#define RT_STACK_LDS_SHARE 7      /* tables + the LDS part of the bounce stack must fit this many times per CU */
#endif

#define RT_TILE_STATS 6          /* counting build: words per wavefront tile {cycles, sphere tests, box tests, scans, start, end (100 MHz clock)} */

/* tile queues: one per XCD, heads RT_QUEUE_STRIDE words apart (own cache lines);
 * a macro tile is RT_MACRO_ROWS vertically adjacent wavefront tiles */
#define RT_TILE_QUEUES 8
#define RT_QUEUE_STRIDE 32
/* a launch's block of counters: the eight queue heads and the HEAVY tiles' head, each on its own cache line.  Every launch
 * zeroes the block its scene's NEXT launch will use (RtParams::next_counters) */
#define RT_COUNTER_WORDS ((RT_TILE_QUEUES + 1) * RT_QUEUE_STRIDE)
#ifndef RT_MACRO_ROWS
#define RT_MACRO_ROWS 4
#endif
#define RT_GETREG_XCC_ID ((3 << 11) | (0 << 6) | 20)   /* s_getreg_b32 HW_REG_XCC_ID, bits [3:0] */

#define RT_HELP_LEAVES 8
#define RT_HELP_SPIN_LIMIT (1 << 22)
#define RT_HEAVY_PERMILLE10 25        /* automatic half-width of the HEAVY band: 0.25 % of the image height (2 tile rows of 4 pixels at 4096) */
/* the workgroup's HELP desk: words of LDS (rt_kernel.hip) */
enum { RT_DESK_STATE = 0, RT_DESK_CURSOR, RT_DESK_INSIDE, RT_DESK_FINISHED, RT_DESK_MASK_LO, RT_DESK_MASK_HI,
       RT_DESK_BASE, RT_DESK_VERDICT_LO, RT_DESK_VERDICT_HI, RT_DESK_BROKEN, RT_DESK_DEDICATED, RT_DESK_PHASE, RT_DESK_WORDS = 12 };

#define RT_NEAR_CULL_MIN_ITEMS 8     /* below this many items the nearest scan skips the bundle cull */
#define RT_SHADOW_CULL_MIN_ITEMS 8   /* below this many shadow items the wavefront skips the bundle-box cull */

#ifndef RT_ORDER_MIN_CANDIDATES
#define RT_ORDER_MIN_CANDIDATES 4    /* from this many candidates on the nearest scan takes them nearest first */
#endif

#define RT_STACK_ENTRY_BYTES 16   /* {local.rgb, bits(object index | texsel << 16)} per bounce level per lane */

#define RT_PRIMARY_ITEMS 64          /* scenes with more FAST items than this have no PRIMARY table */

/* relative growth of a leaf's box that must contain every member sphere the reference's float arithmetic can report from a
 * given origin (derived at box_needed(), rt_kernel.hip); the host's SHADOW VOXELS use the same figure */
#ifndef RT_SPHERE_SLACK
#define RT_SPHERE_SLACK 1.5e-3f
#endif
/* SHADOW VOXELS (below): at most this many voxels, lights and shadow items */
#ifndef RT_SVOX_MAX_CELLS
#define RT_SVOX_MAX_CELLS 4096
#endif
/* cells beyond the core, per axis and side: cell j holds the points whose distance d beyond the core, in core cells, has
 * 16^j <= 1 + 15 d < 16^(j+1) -- four cells reach 4 369 core cells out; further away every item may matter anyway */
#define RT_SVOX_TAIL 4
#define RT_SVOX_MAX_LIGHTS 2        /* one quad per voxel: a 64-bit mask per light */
#define RT_SVOX_MAX_ITEMS 64
#define RT_SVOX_MIN_LEAVES 24        /* automatic: scenes with fewer leaves get no table (option "svox" n: from 4 leaves on) */

typedef struct RtParams {
    /* camera (src/Camera.cpp:71-84) */
    float so[3], ch[3], cv[3], eye[3];
    float sw, sh, shw, shh;
    float null_color[3];
    int32_t W, H, x0, x1, max_depth;
    /* tables */
    int32_t n_lights;
    int32_t n_shadow_items, shadow_items_off;            /* shadow item table (quads), see below */
    int32_t n_near_items, near_items_off;                /* nearest-hit item table                      */
    int32_t n_clusters;                                  /* leaves of clustered sphere runs       */
    int32_t near_first_leaf, shadow_first_leaf;          /* item tables: the leaf items are the ones from this index on (the tables' ends if none) */
    int32_t image_quads;                 /* quads staged into LDS */
    int32_t lights_off, mat_off, tex_off, objinfo_off;   /* quad offsets */
    /* tiling: a wavefront renders tile_x x tile_z pixels, tile_x * tile_z == 64 */
    int32_t tile_z_log2;
    int32_t tiles_z;                     /* wavefront tiles along z */
    int32_t tiles_x;                     /* wavefront tiles along x */
    int32_t n_tiles;                     /* total wavefront tiles   */
    int32_t stack_off;                   /* quad offset of the bounce stack's LDS levels: behind the tables, or 0 when the tables stay in global memory */
    int32_t stack_stride;                /* threads that keep a bounce stack per workgroup (all of them) */
    int32_t stack_lds_levels;            /* bounce levels below this keep their stack entries in LDS (behind the tables), the others in HBM */
    int32_t first_macro_row;             /* the tile queues start at this macro row and wrap around ... */
    int32_t rows_downwards;              /* ... upwards (0) or downwards (1)                           */
    /* HELP (rt_kernel.hip): the workgroup's desk, RT_DESK_WORDS words of LDS at quad desk_off; help_rays_quads != 0:
     * the launch carries 128 quads of global memory per workgroup for the published rays (0: no helping) */
    int32_t desk_off, help_rays_quads;
    int32_t help_leaves;                 /* a shadow scan with this many candidate leaves asks for help (RT_HELP_LEAVES; option "help") */
    int32_t help_spin_limit;             /* the owner's bounded wait for helpers to leave its desk (RT_HELP_SPIN_LIMIT; < 0: every wait counts as timed out -- tests) */
    /* HEAVY tiles (rt_kernel.hip): the tiles within heavy_half tile rows of the horizon line -- tile row
     * (heavy_row0_q16 + tile column * heavy_slope_q16) >> 16 -- are rendered first, one per WORKGROUP: wavefront 0
     * renders, the others serve its shadow scans at the desk from the first scan on.  heavy_half < 0: none. */
    int32_t heavy_half, heavy_row0_q16, heavy_slope_q16;
    /* OLD TILES FIRST (rt_kernel.hip): a wavefront's priority on its SIMD rises with the age of its tile (0: off) */
    int32_t tile_prio;
    int32_t cull;                        /* 0: plain in-order scans (no bundle cull, no nearest-first exit); option "cull" */
    /* FAST tables (scenes without clustered sphere runs, option "fast"): see below */
    int32_t n_fast_items, n_fast_shadow; /* items in all; the first n_fast_shadow are the shadow scan's */
    int32_t fast_box_off, fast_rec_off;  /* quad offsets: 2 box quads and 2 record quads per item */
    int32_t fast_ctl_off;                /* u32 offset (4 per quad): one control word per item */
    /* PRIMARY table (FAST tables, at most RT_PRIMARY_ITEMS items; rt_capi.hip: primary_table()): per item of the FAST list, for
     * THIS launch's camera and image, the rectangle of pixels whose camera ray can reach the item's box and a lower bound of the
     * distance at which it does: the nearest-hit scan of the camera rays needs no bundle bounds, no reciprocals and no box
     * arithmetic -- a tile is a pixel rectangle.  n_primary = items (0: no table, the camera rays take the general cull);
     * primary_off = its place in LDS (quads, behind the image; one quad per item: {x_lo | x_hi << 16 (int16 pixels),
     * z_lo | z_hi << 16, bits of the entry distance, 0}). */
    int32_t n_primary, primary_off;
    /* SHADOW VOXELS (scenes with clustered sphere runs, at most RT_SVOX_MAX_ITEMS shadow items and RT_SVOX_MAX_LIGHTS lights;
     * rt_capi.hip: shadow_voxels(); option "svox").  A grid of svox_n[0] x [1] x [2] voxels over the part of space the leaves
     * occupy (the core), continued on every axis and side by RT_SVOX_TAIL cells that grow 16-fold each (N = n + 2 RT_SVOX_TAIL
     * cells per axis); per voxel and light one 64-bit mask: bit i set = shadow item i can block the segment from SOME
     * point of the voxel to that light (plain items: always set; a leaf: its box, grown by the slack the float sphere test needs
     * from anywhere in the voxel, meets the hull of voxel and light).  A shadow scan ORs the masks of its lanes' voxels (a lane
     * outside the grid: all ones) and ANDs the result into the candidates its bundle cull left: the bundle is one box around
     * all 64 shading points, the voxels follow each ray.  svox_off = quad offset of the table in the image's GLOBAL copy, behind
     * the part that is staged into LDS (0: no table); one quad per voxel (light 0's mask, light 1's), voxel = (z * Ny + y) * Nx + x, the
     * per-axis cell numbers as svox_axis_cell() (rt_kernel.hip) finds them from (P.k - svox_lo[k]) * svox_scale[k] */
    int32_t svox_off;
    int32_t svox_n[3];
    float svox_lo[3], svox_scale[3];
    uint32_t primary[RT_PRIMARY_ITEMS * 4];
    /* one word of host memory the kernel can write (rt_scene's sticky device error): set when a HELP wait timed out */
    uint64_t error_word;
    /* the block of counters (RT_COUNTER_WORDS u32) this scene's next launch will use: this launch zeroes its heads */
    uint64_t next_counters;
    /* diagnostic (option "timeline"): 0, or device memory for RT_TIMELINE_WORDS u64 per wavefront tile, row-major:
     * {start, end (100 MHz constant clock), workgroup * 16 + wavefront, 1 if rendered as a HEAVY tile} */
    uint64_t timeline;
} RtParams;
#define RT_TIMELINE_WORDS 4

/* FAST tables.  Scenes without clustered sphere runs (the reference's built-in Scene: 32 objects) are walked
 * through ONE item list that serves both scans: first the objects of the shadow scan (the non-light objects of
 * the scan range), then the others; within each part sorted by kind.  Neither scan depends on the order: the
 * shadow verdict is an OR (src/RayTracer.cpp:727-729), the nearest hit the minimum of (distance, Scene index)
 * (:71-80).  Per item i:
 *   box      2 quads {centre.xyz, bits(kind | RT_ITEM_TIGHT)}, {half-extent.xyz, control word}: what the wavefront culls test
 *   record   2 quads, all the exact test reads, so that a candidate costs ONE LDS round trip:
 *              sphere          {c.xyz, r^2}, {-}
 *              infinite plane  {n.xyz, distance_to_origin}, {-}
 *              AA rectangle    {dto, sn, sa, sb}, {origin_a, origin_b, extent_a, extent_b}   (as RT_KIND_FINITE_AA)
 *              finite plane    {n.xyz, distance_to_origin}, {bits(quad offset of its full record), -, -, -}
 *   control  one u32: kind | Scene index << 8, read with a SCALAR load from the image in global memory (the
 *            wavefront's dispatch on the kind and the index for ties never touch a vector register) */
#define RT_FAST_BOX_QUADS 2
#define RT_FAST_REC_QUADS 2

#endif /* RT_TABLES_H_ */
