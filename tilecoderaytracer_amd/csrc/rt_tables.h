/*
 * rt_tables.h -- the device scene format: what rt_scene_create() packs from an
 * rt_scene_desc and what every workgroup stages into LDS.
 *
 * The image is an array of 16-byte "quads" (float4).  Sections, in order:
 *
 *   geometry   one record per object, Scene index order
 *                sphere          1 quad : {cx, cy, cz, r^2}
 *                infinite plane  5 quads: q0 {n.xyz, distance_to_origin}
 *                finite plane    5 quads  q1 {anchor.xyz, h_distance}
 *                                         q2 {horizontal.xyz, v_distance}
 *                                         q3 {vertical.xyz, 0}
 *                                         q4 {reverseNormal.xyz, 0}
 *                (anchor = SceneObject::origin for the infinite plane,
 *                 plane_origin for the finite plane: the point texture/bounds
 *                 coordinates are measured from)
 *   lights     2 quads per light, Scene index order:
 *                {origin.xyz, intensity}, {colour.rgb, bits(object index)}
 *   materials  2 quads per DISTINCT (ObjMaterial, intensity, light flag) row:
 *                {colour.rgb, diffuse}, {specular, reflective, intensity,
 *                 bits(is_light | (texture+1) << 1)}
 *   textures   2 quads per checkerboard: {light.rgb, width}, {dark.rgb, height}
 *   objinfo    one u32 per object (4 per quad):
 *                bits 0-15 geometry offset (quads), 16-17 kind, 20-31 material row
 *   cidx       one u32 per clustered sphere: its Scene index (member order)
 *
 * Clustered sphere runs.  A run of >= 4*leaf spheres is regrouped into spatial
 * leaves of <= leaf spheres; the spheres' geometry quads are stored leaf by
 * leaf (objinfo still finds each one) and a cluster table (between geometry
 * and lights) holds 2 quads per leaf:
 *                {box lo.xyz, bits(member geometry offset | member count << 16)},
 *                {box hi.xyz, bits(first slot in the run's cidx table)}
 * Leaves come out of the split in spatial order; every `group` consecutive
 * leaves form a GROUP with its own box (group table after the leaf table, 2
 * quads: {lo.xyz, bits(quad offset of its first leaf record)}, {hi.xyz,
 * bits(leaf count)}).  A clustered run lists its groups.  Each (inflated,
 * axis-aligned) box contains every member sphere.  Nearest-hit stays exact
 * because ties are broken on the Scene index (lexicographic min of (distance,
 * index) is what an in-order scan with a strict `<` computes).
 *
 * Shadow items.  The shadow scan (src/RayTracer.cpp:709-739) walks a table of
 * ITEMS: one per non-light object of the scan range, except that a clustered
 * sphere run contributes one item per GROUP of leaves.  2 quads per item:
 *   {box lo.xyz, bits(kind | count << 8 | geometry quad offset << 16)},
 *   {box hi.xyz, bits(Scene index | quad offset of the plane's full 5-quad record << 12)}
 *   (for a group of a clustered run the second word is the u32 index of the
 *    run's Scene-index table instead)
 * kind = RT_KIND_SPHERE / _INFINITE_PLANE / _FINITE_PLANE, RT_KIND_SPHERE_CLUSTERED
 * for a group (count = its leaves, geometry offset = its first leaf record), RT_KIND_FINITE_AA + class for an axis-aligned
 * rectangle (geometry offset = its AA test record).  The box (inflated on the
 * host) contains the object; an infinite plane's box is all of space.  The
 * wavefront tests 64 item boxes at once, one per lane (rt_kernel.hip, in_shade).
 * The nearest-hit scan has the same kind of table over ALL objects, lights
 * included (nearest_hit_items).
 *
 * The object list is additionally described as RUNS of consecutive objects of
 * one kind and one light flag (an int4 each, kept in global memory and read
 * with scalar loads because the run index is wave-uniform), used by the
 * nearest-hit scan.  In-order runs preserve Scene index order, so "first
 * strictly-smaller distance wins" (src/RayTracer.cpp:75-78) needs no tie-break
 * there; class-sorted and clustered runs break ties on the Scene index.
 */
#ifndef RT_TABLES_H_
#define RT_TABLES_H_

#include <stdint.h>

#define RT_SPHERE_QUADS 1
#define RT_PLANE_QUADS  5
#define RT_LIGHT_QUADS  2
#define RT_MAT_QUADS    2
#define RT_TEX_QUADS    2
#define RT_CLUSTER_QUADS 2

/* run kinds: RT_KIND_* of rt_capi.h (0 sphere, 1 infinite plane, 2 finite
 * plane) plus a long sphere run regrouped into spatial clusters */
#define RT_KIND_SPHERE_CLUSTERED 3
/* axis-aligned finite planes, six classes by axis permutation: kind = 4 + class,
 * class = 2*normal_axis + (horizontal_axis == (normal_axis+1)%3 ? 0 : 1).
 * Test records (2 quads): {dto, sn, sh, sv}, {po_a, po_b, h_dist, v_dist} in
 * the permuted coordinates (n, a, b); `first` = u32 index of the run's Scene
 * index table.  The full 5-quad record of each plane is kept too (winner
 * record, non-finite rays). */
#define RT_KIND_FINITE_AA 4
#define RT_AA_QUADS 2

#define RT_MAX_GEOM_QUADS 65535   /* 16-bit geometry offset in objinfo */
#define RT_MAX_MATERIALS  4095    /* 12-bit material row in objinfo    */
#define RT_MAX_LDS_BYTES  (160 * 1024)

#define RT_TILE_STATS 6          /* counting build: words per wavefront tile {cycles, sphere tests, box tests, scans, start, end (100 MHz clock)} */

/* tile queues: one per XCD, heads RT_QUEUE_STRIDE words apart (own cache lines);
 * a macro tile is RT_MACRO_ROWS vertically adjacent wavefront tiles */
#define RT_TILE_QUEUES 8
#define RT_QUEUE_STRIDE 32
#define RT_MACRO_ROWS 4
#define RT_GETREG_XCC_ID ((3 << 11) | (0 << 6) | 20)   /* s_getreg_b32 HW_REG_XCC_ID, bits [3:0] */

#define RT_NEAR_CULL_MIN_ITEMS 8     /* below this many items the nearest scan skips the bundle cull */
#define RT_SHADOW_CULL_MIN_ITEMS 8   /* below this many shadow items the wavefront skips the bundle-box cull */

#define RT_STACK_ENTRY_BYTES 16   /* {local.rgb, bits(object index | texsel << 16)} per bounce level per lane */

typedef struct RtRun {
    int32_t kind;       /* RT_KIND_*                               */
    int32_t count;      /* objects in the run; clustered: groups   */
    int32_t first;      /* Scene index of the first object; clustered: u32 index of the run's cidx table */
    int32_t geom_off;   /* quad offset of the first object's record; clustered: of the group table       */
} RtRun;

typedef struct RtParams {
    /* camera (src/Camera.cpp:71-84) */
    float so[3], ch[3], cv[3], eye[3];
    float sw, sh, shw, shh;
    float null_color[3];
    int32_t W, H, x0, x1, max_depth;
    /* tables */
    int32_t n_runs, n_lights;
    int32_t n_shadow_items, shadow_items_off;            /* shadow item table (quads), see below */
    int32_t n_near_items, near_items_off, near_items_on; /* nearest-hit item table; on = use it         */
    int32_t n_clusters;                                  /* leaves of clustered sphere runs       */
    int32_t image_quads;                 /* quads staged into LDS */
    int32_t lights_off, mat_off, tex_off, objinfo_off;   /* quad offsets */
    /* tiling: a wavefront renders tile_x x tile_z pixels, tile_x * tile_z == 64 */
    int32_t tile_z_log2;
    int32_t tiles_z;                     /* wavefront tiles along z */
    int32_t tiles_x;                     /* wavefront tiles along x */
    int32_t n_tiles;                     /* total wavefront tiles   */
    int32_t stack_in_lds;                /* bounce stack in LDS (behind the tables) instead of HBM */
} RtParams;

#endif /* RT_TABLES_H_ */
