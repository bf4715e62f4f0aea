/*
 * rt_capi.hip -- implementation of the C ABI in include/rt_capi.h:
 * packs an rt_scene_desc into the device scene format (rt_tables.h), uploads
 * it, launches rt_render_kernel and times it with HIP events on the launch
 * stream.  No CPU rendering path exists here: without a HIP device every
 * entry point that needs one fails.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/rt_capi_tuning.h"
#include "rt_tables.h"

#define RT_DECLARE_KERNEL(name)                                                                                   \
    extern "C" __global__ void name(const RtParams p, const float4 *__restrict__ image, float *__restrict__ out, \
                                    unsigned int *__restrict__ tile_counter, float4 *__restrict__ bounce_stack,  \
                                    unsigned int *__restrict__ help_area)
#define RT_DECLARE_STATS_KERNEL(name)                                                                             \
    extern "C" __global__ void name(const RtParams p, const float4 *__restrict__ image, float *__restrict__ out, \
                                    unsigned int *__restrict__ tile_counter, float4 *__restrict__ bounce_stack,  \
                                    unsigned long long *__restrict__ stats_out, unsigned int *__restrict__ help_area)
RT_DECLARE_KERNEL(rt_render_kernel);                  /* FAST tables (scenes without clustered runs) */
RT_DECLARE_KERNEL(rt_render_kernel_items);            /* the two item tables, no clustered runs      */
RT_DECLARE_KERNEL(rt_render_kernel_large);            /* tables in global memory                     */
RT_DECLARE_KERNEL(rt_render_kernel_clusters);         /* clustered sphere runs, six wavefronts per SIMD */
RT_DECLARE_KERNEL(rt_render_kernel_clusters_wide);    /* ... five */
RT_DECLARE_STATS_KERNEL(rt_render_kernel_stats);      /* the counting builds */
RT_DECLARE_STATS_KERNEL(rt_render_kernel_fast_stats);

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorNoDevice ? RT_ERR_NO_DEVICE : RT_ERR_HIP,               \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    } while (0)

struct Quad { float v[4]; };

float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct EventPair { hipEvent_t start, stop; bool pending; };
constexpr int kEventRing = 64;
constexpr int kCounterWords = RT_COUNTER_WORDS;   /* 8 queue heads + the HEAVY tiles', own cache lines */

} // namespace

struct rt_scene {
    int device = 0;
    /* the caller's description, copied */
    std::vector<rt_object_desc> objects;
    std::vector<rt_texture_desc> textures;
    int shadow_begin = 0, shadow_end = 0;
    float null_color[3] = {0.75f, 0.75f, 0.75f};
    /* packed tables (host copies) */
    std::vector<Quad> image;
    RtParams base{};              /* table offsets filled at create */
    /* device copies */
    void *d_image = nullptr;
    /* scratch framebuffer for rt_render (host destination) */
    void *d_fb = nullptr;
    size_t d_fb_bytes = 0;
    /* options */
    int tile_z_log2 = -1;         /* wavefront tile height: -1 = auto (see launch()), else log2 */
    int block_threads_opt = 0;    /* 0 = auto */
    int stack_opt = 0;            /* bounce stack: 0 = auto, 1 = LDS, 2 = HBM */
    int tile_prio_opt = -1;       /* OLD TILES FIRST: -1 = automatic (strips of at most a third of the width), 0 off, 1 on */
    int first_row_permille = -1;  /* the tile queues start this far up the image (speed only); -1 = horizon_start() */
    int help_opt = -1;            /* clustered scenes: wavefronts out of tiles help their workgroup's long shadow scans (0: they leave; -1: on for strips) */
    int heavy_opt = -1;           /* HEAVY tiles (the band of tile rows along the horizon line, one per workgroup, first): -1 = automatic, 0 = off, k = k - 1 rows either side */
    int help_spin_opt = RT_HELP_SPIN_LIMIT;   /* the owner's bounded wait at its desk; -1: every wait counts as timed out (tests) */
    unsigned int *h_error = nullptr;          /* pinned host word the kernels can write: a HELP wait timed out */
    int timeline_opt = 0;                     /* diagnostic: every launch records per tile when and by whom it was rendered */
    unsigned long long *d_timeline = nullptr;
    size_t timeline_words = 0, timeline_valid = 0;
    int wide_opt = -1;            /* experiments: which clustered-scene kernel (-1 automatic, 0 the 80-register one, 1 the 96-register one) */
    int pairs_opt = 1;            /* scenes with clustered runs: the kernel that compacts (ray, leaf) pairs (0: the plain kernel) */
    int tables_opt = 0;           /* where the kernel reads the tables: 0 = automatic, 1 = LDS, 2 = global memory (any size) */
    int grid_mult = 1;            /* grid = occupancy * CUs * this; 0 = one workgroup per 4 tiles (no persistence) */
    int aa_planes = 1;            /* class-sorted fast path for axis-aligned finite planes             */
    int primary_opt = 1;          /* FAST tables: the camera rays' scan culls by projected pixel rectangles (PRIMARY table); 0: the bundle cull */
    int fast_opt = 1;             /* scenes without clustered runs: the kind-sorted item list with direct records (FAST tables); 0: the two item tables */
    int tight_planes = 1;         /* plane items: boxes padded for a plane's rounding only (RT_ITEM_TIGHT); 0: the sphere padding */
    int cluster_leaf = -1;        /* spheres per cluster leaf for long sphere runs; 0 = no clustering, -1 = by the run's length (auto_leaf()) */
    int cull_opt = 1;             /* 0: plain in-order scans -- no bundle cull, no nearest-first exit, no clustering, no AA route */
    int svox_opt = -1;            /* SHADOW VOXELS (scenes with clustered runs): -1 = automatic (RT_SVOX_MAX_CELLS voxels), 0 = none, n = at most n voxels */
    int n_clusters = 0;
    /* tile queue heads, one per in-flight launch (same ring as the events) */
    unsigned int *d_counters = nullptr;
    /* HELP: 2 KB per workgroup for the rays a wavefront publishes at its workgroup's desk */
    /* LEARNED START ROW (rt_learn_tile_order): per macro row of one launch shape, its longest tile and its tiles' sum (cycles of
     * the counting build) */
    std::vector<double> row_peak, row_sum;     /* empty: nothing learned */
    int order_key[6] = {0, 0, 0, 0, 0, 0};     /* W, H, x0, x1, max_depth, tile_z_log2 of the launch they were learned from */
    int learned_sweep = -1;                    /* what rt_learn_tile_order measured to be fastest: -1 the rule, 0 from that row upwards, 1 downwards */
    void *d_help = nullptr;
    size_t d_help_bytes = 0;
    /* bounce stack in HBM: grid_blocks x (max_depth + 1) x block_threads entries of 16 B */
    void *d_stack = nullptr;
    size_t d_stack_bytes = 0;
    hipStream_t last_stream = nullptr;
    bool has_last_stream = false;
    int n_cus = 0;
    /* timing */
    EventPair ev[kEventRing];
    int ev_next = 0;
    bool ev_ready = false;
    rt_timing timing{};
    rt_launch_info launch{};
    std::mutex mu;
};

namespace {

/* ---- sphere clusters (rt_tables.h "clustered sphere runs") ----------------
 * A long run of spheres is regrouped into spatial leaves of <= leaf spheres
 * (k-d median split of the centres).  Each leaf gets a bounding ball that
 * contains every member sphere; the kernel skips a leaf only when a
 * conservative test proves no member can be a candidate hit, so results are
 * unchanged (rt_kernel.hip, cluster_needed()). */
struct Leaf { std::vector<int> members; float lo[3], hi[3]; };

void make_leaf(const rt_object_desc *objs, const std::vector<int> &ids, std::vector<Leaf> &out) {
    Leaf L;
    L.members = ids;
    /* axis-aligned box around every member sphere (centre +- |radius|) */
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i : ids)
        for (int k = 0; k < 3; ++k) {
            const double r = std::fabs((double)objs[i].radius);
            lo[k] = std::min(lo[k], (double)objs[i].origin[k] - r);
            hi[k] = std::max(hi[k], (double)objs[i].origin[k] + r);
        }
    /* inflate by 1 % of the largest extent + 1e-4, then round outwards to float */
    const double pad = 0.01 * std::max(hi[0] - lo[0], std::max(hi[1] - lo[1], hi[2] - lo[2])) + 1e-4;
    for (int k = 0; k < 3; ++k) {
        L.lo[k] = std::nextafter((float)(lo[k] - pad), -INFINITY);
        L.hi[k] = std::nextafter((float)(hi[k] + pad), INFINITY);
    }
    out.push_back(std::move(L));
}

void split_leaves(const rt_object_desc *objs, std::vector<int> ids, int leaf, std::vector<Leaf> &out) {
    if ((int)ids.size() <= leaf) { make_leaf(objs, ids, out); return; }
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i : ids)
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::min(lo[k], (double)objs[i].origin[k]);
            hi[k] = std::max(hi[k], (double)objs[i].origin[k]);
        }
    int axis = 0;
    for (int k = 1; k < 3; ++k) if (hi[k] - lo[k] > hi[axis] - lo[axis]) axis = k;
    /* split so that the left part is a whole number of leaves */
    const size_t n_leaves = (ids.size() + (size_t)leaf - 1) / (size_t)leaf;
    const size_t mid = (n_leaves / 2) * (size_t)leaf;
    std::nth_element(ids.begin(), ids.begin() + (long)mid, ids.end(), [&](int a, int b) {
        if (objs[a].origin[axis] != objs[b].origin[axis]) return objs[a].origin[axis] < objs[b].origin[axis];
        return a < b;
    });
    split_leaves(objs, std::vector<int>(ids.begin(), ids.begin() + (long)mid), leaf, out);
    split_leaves(objs, std::vector<int>(ids.begin() + (long)mid, ids.end()), leaf, out);
}

/* Spheres per leaf of a clustered run of n spheres.  A scan pays a box test per leaf it looks at and a member test per sphere
 * of the leaves it opens: few large leaves for a long run, many small ones for a short one.  Measured on n x n sphere grids,
 * 4096^2 depth 4, frame ms at 16 / 20 / 24 / 32 per leaf (scripts/sweep_gpu.py, profiles/r03_experiments.txt 16): 400 spheres
 * 2.79 / 2.80 / 2.87 / 3.05; 576: 3.24 / 3.20 / 3.22 / 3.50; 784: 4.05 / 3.69 / 3.76 / 4.12; 1 024: 4.55 / 4.44 / 4.39 / 4.89;
 * 2 304: 14.7 / 13.5 / 12.8 / 14.6; 3 969 (tables in global memory): 18.9 / 17.2 / 17.1 / 16.3.
 * Leaves of 32 and more never go through the (ray, leaf) pair compaction (five bits for the member count), which only the
 * clustered-scene kernels have: the largest size is for tables that are read from global memory anyway (two mirrors, 1024^2
 * depth 50: 0.692 -> 0.626 ms there, but 0.445 -> 0.649 ms with its tables forced into LDS). */
int auto_leaf(int n, bool tables_in_lds) { return n < 512 ? 16 : (n < 896 ? 20 : ((n < 3000 || tables_in_lds) ? 24 : 32)); }

/* An item's box as the kernel reads it (rt_tables.h): CENTRE and HALF-EXTENT per axis, such that [c - h, c + h] in real
 * arithmetic contains [lo, hi]; an axis the item is unbounded on: centre 0, half-extent infinity.  (The culls' slab tests are
 * c (1/d) -+ h |1/d| in this form: no minima or maxima per axis, which issue at half rate.) */
void centre_half(double lo, double hi, float *c, float *h) {
    if (!std::isfinite(lo) || !std::isfinite(hi)) { *c = 0.0f; *h = INFINITY; return; }
    const float centre = (float)(0.5 * (lo + hi));
    const double need = std::max(hi - (double)centre, (double)centre - lo);
    *c = centre;
    *h = std::nextafter((float)std::max(need, 0.0), INFINITY);
    if ((double)*h < need) *h = std::nextafter(*h, INFINITY);
}
double box_lo(const Quad &centre, const Quad &half, int k) { return (double)centre.v[k] - (double)half.v[k]; }    /* (-infinity for an unbounded axis) */
double box_hi(const Quad &centre, const Quad &half, int k) { return (double)centre.v[k] + (double)half.v[k]; }

/* axis of a +-unit axis vector (other components exactly +-0), or -1 */
int unit_axis(const float v[3], float *sign) {
    for (int k = 0; k < 3; ++k)
        if (std::fabs(v[k]) == 1.0f && v[(k + 1) % 3] == 0.0f && v[(k + 2) % 3] == 0.0f) { *sign = v[k]; return k; }
    return -1;
}

/* class of an axis-aligned finite plane (rt_tables.h), or -1 */
int aa_class(const rt_object_desc &o, float *sn, float *sh, float *sv, int *a_axis, int *b_axis) {
    if (o.kind != RT_KIND_FINITE_PLANE) return -1;
    const int kn = unit_axis(o.normal, sn), kh = unit_axis(o.horizontal, sh), kv = unit_axis(o.vertical, sv);
    if (kn < 0 || kh < 0 || kv < 0 || kn == kh || kn == kv || kh == kv) return -1;
    const float vals[6] = {o.plane_origin[0], o.plane_origin[1], o.plane_origin[2], o.h_distance, o.v_distance,
                           o.distance_to_origin};
    for (float f : vals) if (!std::isfinite(f)) return -1;
    if (o.h_distance < 0.0f || o.v_distance < 0.0f) return -1;
    /* The record lists the two in-plane axes in cyclic order after the normal's
     * ((kn+1)%3, then (kn+2)%3), whichever of them is "horizontal": the bounds test
     * treats both alike, so the kernel needs the normal's axis only. */
    if (kh != (kn + 1) % 3) { std::swap(*sh, *sv); *a_axis = kv; *b_axis = kh; return kn + 3; }
    *a_axis = kh; *b_axis = kv;
    return kn;
}

bool all_finite(const rt_object_desc &o) {
    return std::isfinite(o.origin[0]) && std::isfinite(o.origin[1]) && std::isfinite(o.origin[2]) &&
           std::isfinite(o.radius);
}

/* SHADOW VOXELS (rt_tables.h, RtParams::svox_*).  The shadow scans of a wavefront cull the scene's items against ONE box around
 * its 64 shading points; after a bounce those lie all over the scene and the box is large: 9 candidate leaves per scan on the
 * 1 024-sphere grid where 2.3 are needed by any ray, each of them a box test for the whole wavefront.  What a single shading
 * point needs depends on where it is and on the light only -- neither changes from frame to frame -- so the host answers the
 * question once per scene for every VOXEL of a grid laid over the region the leaves occupy: which shadow items can block the
 * segment from some point of the voxel to light l?  A scan then ORs its lanes' voxel masks and ANDs them into the candidates.
 *
 * Exactness.  A leaf may be left out of a voxel's mask only if no member sphere can be reported by the reference's float test
 * (src/SceneSphere.cpp:50-116) for a shadow ray that starts in the voxel.  box_needed() (rt_kernel.hip) derives what that takes:
 * the ray must meet the leaf's box grown by RT_SPHERE_SLACK of the L1 distance `far` from the ray's origin to the box's farthest
 * corner, between its origin and the light (give or take 2.4e-7 far).  Here: the box is grown by that slack for the LARGEST far any
 * point of the voxel has, plus 1e-5 far + 1e-5 for what separates the float ray from the exact segment (the direction's
 * normalisation, the distance to the light: a few 1e-7 of the length each); the voxel is grown by 1e-4 of a cell for the rounding
 * of the kernel's voxel index (its two float operations err by less than 1e-5 cells); and the set of segments from the voxel to
 * the light is the hull of a box and a point: at parameter s the box of half-extent (1 - s) e around c + s (L - c), so
 * "meets the grown box" is an intersection of intervals in s -- exact in real arithmetic, evaluated in double.  A shading point
 * outside the grid (or a NaN) gets all ones.  Plain items (planes, single spheres) are always set: the bundle cull's business. */
struct ShadowVoxels {
    std::vector<uint64_t> masks;          /* [cell][RT_SVOX_MAX_LIGHTS] */
    int n[3] = {0, 0, 0};                 /* core cells per axis (the grid has n + 2 RT_SVOX_TAIL per axis) */
    float lo[3] = {0, 0, 0}, scale[3] = {0, 0, 0};
};

bool hull_meets_box(const double c[3], const double e[3], const double L[3], const double lo[3], const double hi[3]) {
    double s_lo = 0.0, s_hi = 1.0;
    for (int k = 0; k < 3; ++k) {
        const double m = 0.5 * (lo[k] + hi[k]), h = 0.5 * (hi[k] - lo[k]);
        const double a = c[k] - m, g = L[k] - c[k], r = h + e[k];
        /* |a + s g| <= r - s e:   s (g + e) <= r - a   and   s (e - g) <= r + a */
        const double A[2] = {g + e[k], e[k] - g}, B[2] = {r - a, r + a};
        for (int j = 0; j < 2; ++j) {
            if (A[j] > 0.0) s_hi = std::min(s_hi, B[j] / A[j]);
            else if (A[j] < 0.0) s_lo = std::max(s_lo, B[j] / A[j]);
            else if (B[j] < 0.0) return false;
        }
        if (s_lo > s_hi + 1e-9) return false;
    }
    return true;
}

/* One axis of the grid, as svox_axis_cell() in rt_kernel.hip numbers it: RT_SVOX_TAIL cells below the core, growing 16-fold each
 * (distance d beyond the core in cells: cell j holds 16^j <= 1 + 15 d < 16^(j+1)), n core cells, RT_SVOX_TAIL above.  The bounds
 * are what the kernel's float arithmetic can put into the cell: (x - lo) * scale, the distance beyond the core and 1 + 15 d are
 * rounded once each (relative 6e-8 each; x - lo absolutely by half an ulp of the larger operand), so every cell gets 1e-5 of 1 + 15 d,
 * 1e-4 of a core cell and 1e-6 of the coordinates' magnitude on either side. */
void svox_axis_bounds(float lo_f, float scale_f, int n, std::vector<double> &cell_lo, std::vector<double> &cell_hi) {
    const double lo = lo_f, scale = scale_f;
    const int N = n + 2 * RT_SVOX_TAIL;
    cell_lo.assign((size_t)N, 0.0);
    cell_hi.assign((size_t)N, 0.0);
    auto beyond = [&](int j, double *d_lo, double *d_hi) {       /* tail cell j: the distances beyond the core, in cells */
        *d_lo = (std::pow(16.0, j) * (1.0 - 1e-5) - 1.0) / 15.0 - 1e-4;
        *d_hi = (std::pow(16.0, j + 1) * (1.0 + 1e-5) - 1.0) / 15.0 + 1e-4;
    };
    for (int idx = 0; idx < N; ++idx) {
        double a, b;
        if (idx < RT_SVOX_TAIL) {
            double d_lo, d_hi;
            beyond(RT_SVOX_TAIL - 1 - idx, &d_lo, &d_hi);
            a = lo - d_hi / scale; b = lo - d_lo / scale;
        } else if (idx < RT_SVOX_TAIL + n) {
            const double i = idx - RT_SVOX_TAIL;
            a = lo + (i - 1e-4) / scale; b = lo + (i + 1.0 + 1e-4) / scale;
        } else {
            double d_lo, d_hi;
            beyond(idx - RT_SVOX_TAIL - n, &d_lo, &d_hi);
            a = lo + ((double)n + d_lo) / scale; b = lo + ((double)n + d_hi) / scale;
        }
        cell_lo[(size_t)idx] = a - 1e-6 * (std::fabs(a) + std::fabs(lo));
        cell_hi[(size_t)idx] = b + 1e-6 * (std::fabs(b) + std::fabs(lo));
    }
}

/* items: 2 quads each ({lo.xyz, bits}, {hi.xyz, word}); the leaves are the items from first_leaf on; lights: RT_LIGHT_QUADS each */
bool shadow_voxels(const std::vector<Quad> &items, int first_leaf, const std::vector<Quad> &lights, int max_cells, int min_leaves, ShadowVoxels *out) {
    const int n_items = (int)(items.size() / 2), n_lights = (int)(lights.size() / RT_LIGHT_QUADS);
    const int n_leaves = n_items - first_leaf;
    if (n_items > RT_SVOX_MAX_ITEMS || n_lights < 1 || n_lights > RT_SVOX_MAX_LIGHTS || n_leaves < min_leaves) return false;
    /* the core: around the leaves, and on an axis where the other items' bounded sides reach further -- a ceiling above the
     * field -- out to those, unless that would more than double the largest extent.  Shading points beyond it -- the ground
     * in front of a field of spheres, out to the horizon -- fall into the tail cells. */
    double glo[3] = {1e300, 1e300, 1e300}, ghi[3] = {-1e300, -1e300, -1e300};
    for (int i = first_leaf; i < n_items; ++i)
        for (int k = 0; k < 3; ++k) {
            const double a = box_lo(items[(size_t)2 * i], items[(size_t)2 * i + 1], k), b = box_hi(items[(size_t)2 * i], items[(size_t)2 * i + 1], k);
            if (!std::isfinite(a) || !std::isfinite(b)) return false;
            glo[k] = std::min(glo[k], a);
            ghi[k] = std::max(ghi[k], b);
        }
    const double largest = std::max(ghi[0] - glo[0], std::max(ghi[1] - glo[1], ghi[2] - glo[2]));
    if (!(largest > 0.0) || !std::isfinite(largest)) return false;
    for (int k = 0; k < 3; ++k) {
        double lo = glo[k], hi = ghi[k];
        for (int i = 0; i < first_leaf; ++i) {
            const double a = box_lo(items[(size_t)2 * i], items[(size_t)2 * i + 1], k), b = box_hi(items[(size_t)2 * i], items[(size_t)2 * i + 1], k);
            if (std::isfinite(a)) lo = std::min(lo, a);
            if (std::isfinite(b)) hi = std::max(hi, b);
        }
        if (hi - lo <= 2.0 * largest) { glo[k] = lo; ghi[k] = hi; }
        glo[k] -= 1e-3 * largest;
        ghi[k] += 1e-3 * largest;
    }
    for (int l = 0; l < n_lights; ++l)
        for (int k = 0; k < 3; ++k)
            if (!std::isfinite(lights[(size_t)l * RT_LIGHT_QUADS].v[k])) return false;
    /* cubic core cells, as small as the budget allows */
    if (max_cells < (2 * RT_SVOX_TAIL + 1) * (2 * RT_SVOX_TAIL + 1) * (2 * RT_SVOX_TAIL + 1)) return false;
    double cell = std::cbrt((ghi[0] - glo[0]) * (ghi[1] - glo[1]) * (ghi[2] - glo[2]) / (double)max_cells);
    int n[3], N[3];
    for (int tries = 0;; ++tries) {
        long long cells = 1;
        for (int k = 0; k < 3; ++k) {
            n[k] = (int)std::min(256.0, std::max(1.0, std::ceil((ghi[k] - glo[k]) / cell)));
            N[k] = n[k] + 2 * RT_SVOX_TAIL;
            cells *= N[k];
        }
        if (cells <= max_cells) break;
        if (tries > 400) return false;
        cell *= 1.03;
    }
    std::vector<double> clo[3], chi[3];
    for (int k = 0; k < 3; ++k) {
        out->n[k] = n[k];
        out->lo[k] = (float)glo[k];
        out->scale[k] = (float)((double)n[k] / (ghi[k] - glo[k]));
        if (!std::isfinite(out->lo[k]) || !std::isfinite(out->scale[k]) || !(out->scale[k] > 0.0f)) return false;
        svox_axis_bounds(out->lo[k], out->scale[k], n[k], clo[k], chi[k]);
    }
    const uint64_t plain = first_leaf >= 64 ? ~0ull : ((1ull << first_leaf) - 1ull);
    /* two masks per voxel whatever the number of lights (a quad: one load) */
    out->masks.assign((size_t)N[0] * N[1] * N[2] * RT_SVOX_MAX_LIGHTS, plain);
    double L[RT_SVOX_MAX_LIGHTS][3];
    for (int l = 0; l < n_lights; ++l)
        for (int k = 0; k < 3; ++k) L[l][k] = lights[(size_t)l * RT_LIGHT_QUADS].v[k];
    for (int z = 0; z < N[2]; ++z)
        for (int y = 0; y < N[1]; ++y)
            for (int x = 0; x < N[0]; ++x) {
                const double vlo[3] = {clo[0][(size_t)x], clo[1][(size_t)y], clo[2][(size_t)z]};
                const double vhi[3] = {chi[0][(size_t)x], chi[1][(size_t)y], chi[2][(size_t)z]};
                double c[3], e[3];
                for (int k = 0; k < 3; ++k) { c[k] = 0.5 * (vlo[k] + vhi[k]); e[k] = 0.5 * (vhi[k] - vlo[k]); }
                uint64_t *m = &out->masks[(((size_t)z * N[1] + y) * N[0] + x) * RT_SVOX_MAX_LIGHTS];
                for (int i = first_leaf; i < n_items; ++i) {
                    double far = 0.0, blo[3], bhi[3];
                    for (int k = 0; k < 3; ++k) {
                        blo[k] = box_lo(items[(size_t)2 * i], items[(size_t)2 * i + 1], k);
                        bhi[k] = box_hi(items[(size_t)2 * i], items[(size_t)2 * i + 1], k);
                        far += std::max(std::max(std::fabs(blo[k] - vlo[k]), std::fabs(blo[k] - vhi[k])),
                                        std::max(std::fabs(bhi[k] - vlo[k]), std::fabs(bhi[k] - vhi[k])));
                    }
                    const double grow = ((double)RT_SPHERE_SLACK + 1e-5) * far + 1e-5;
                    for (int k = 0; k < 3; ++k) { blo[k] -= grow; bhi[k] += grow; }
                    for (int l = 0; l < n_lights; ++l)
                        if (hull_meets_box(c, e, L[l], blo, bhi)) m[l] |= 1ull << i;
                }
            }
    return true;
}

/* Build the LDS image + run lists from the stored description. */
int pack_scene(rt_scene *s) {
    const int n = (int)s->objects.size();
    const rt_object_desc *objs = s->objects.data();
    const int sb = s->shadow_begin, se = s->shadow_end;
    /* option "cull" = 0: every object is a plain item in Scene index order */
    const int cluster_leaf_opt = s->cull_opt ? s->cluster_leaf : 0;
    const bool aa_planes = s->cull_opt && s->aa_planes;

    std::vector<Quad> geom, lights, mats, texs;
    std::vector<uint32_t> objinfo((size_t)n, 0u), cidx;
    std::vector<int> geom_off((size_t)n, 0), mat_of((size_t)n, 0);
    std::map<std::vector<uint32_t>, int> mat_index;

    /* materials (de-duplicated bit-wise), lights */
    for (int i = 0; i < n; ++i) {
        const rt_object_desc &o = objs[i];
        const uint32_t mbits = (o.is_light ? 1u : 0u) | ((uint32_t)(o.texture + 1) << 1);
        Quad m0 = {{o.color[0], o.color[1], o.color[2], o.diffuse}};
        Quad m1 = {{o.specular, o.reflective, o.intensity, bits_to_float(mbits)}};
        std::vector<uint32_t> key(8);
        std::memcpy(key.data(), m0.v, 16);
        std::memcpy(key.data() + 4, m1.v, 16);
        auto it = mat_index.find(key);
        if (it == mat_index.end()) {
            const int mi = (int)mat_index.size();
            if (mi > RT_MAX_MATERIALS) return fail(RT_ERR_CAPACITY, "too many distinct materials");
            mat_index.emplace(key, mi);
            mats.push_back(m0);
            mats.push_back(m1);
            mat_of[(size_t)i] = mi;
        } else {
            mat_of[(size_t)i] = it->second;
        }
        if (o.is_light) {
            lights.push_back({{o.origin[0], o.origin[1], o.origin[2], o.intensity}});
            lights.push_back({{o.color[0], o.color[1], o.color[2], bits_to_float((uint32_t)i)}});
        }
    }
    for (const rt_texture_desc &x : s->textures) {
        texs.push_back({{x.light[0], x.light[1], x.light[2], x.width}});
        texs.push_back({{x.dark[0], x.dark[1], x.dark[2], x.height}});
    }

    /* runs of consecutive objects of one kind and one light flag, in index order */
    struct Span { int kind, first, count; bool light; };
    std::vector<Span> spans;
    for (int i = 0; i < n; ++i) {
        const bool light = objs[i].is_light != 0;
        if (!spans.empty() && spans.back().kind == objs[i].kind && spans.back().light == light)
            ++spans.back().count;
        else
            spans.push_back(Span{objs[i].kind, i, 1, light});
    }

    auto emit_geometry = [&](int i) {
        const rt_object_desc &o = objs[i];
        geom_off[(size_t)i] = (int)geom.size();
        if (o.kind == RT_KIND_SPHERE) {
            geom.push_back({{o.origin[0], o.origin[1], o.origin[2], o.radius_squared}});
        } else {
            const float *anchor = (o.kind == RT_KIND_INFINITE_PLANE) ? o.origin : o.plane_origin;
            geom.push_back({{o.normal[0], o.normal[1], o.normal[2], o.distance_to_origin}});
            geom.push_back({{anchor[0], anchor[1], anchor[2], o.h_distance}});
            geom.push_back({{o.horizontal[0], o.horizontal[1], o.horizontal[2], o.v_distance}});
            geom.push_back({{o.vertical[0], o.vertical[1], o.vertical[2], 0.0f}});
            geom.push_back({{o.reverse_normal[0], o.reverse_normal[1], o.reverse_normal[2], 0.0f}});
        }
    };

    int n_clusters = 0;
    /* Cluster/idx offsets are patched once the section bases are known. */
    std::vector<int> aa_all;                         /* axis-aligned finite planes (Scene indices) */
    std::vector<int> aa_rec_of((size_t)n, -1);       /* their AA test record (quad offset within aa_recs) */
    std::vector<int> aa_cls_of((size_t)n, -1);
    std::vector<char> clustered((size_t)n, 0);
    struct LeafItem { float lo[3], hi[3]; uint32_t member_off, count, cidx_slot; bool in_shadow; };
    std::vector<LeafItem> leaf_items;
    std::vector<Quad> shadow_items, near_items;
    std::vector<Quad> aa_recs;

    for (const Span &sp : spans) {
        const int first = sp.first, last = sp.first + sp.count;             /* [first, last) */
        /* part of the span inside the shadow scan range (lights never cast shadows) */
        const int s0 = std::max(first, sb), s1 = std::min(last, se);
        const bool in_shadow_all = !sp.light && s0 == first && s1 == last;
        const bool in_shadow_none = sp.light || s0 >= s1;
        const int cluster_leaf = cluster_leaf_opt < 0 ? auto_leaf(sp.count, s->tables_opt == 1) : cluster_leaf_opt;
        bool cluster = sp.kind == RT_KIND_SPHERE && cluster_leaf > 0 && sp.count >= 4 * cluster_leaf &&
                       (in_shadow_all || in_shadow_none);
        if (cluster)
            for (int i = first; i < last; ++i) cluster = cluster && all_finite(objs[i]);
        if (cluster) {
            std::vector<int> ids((size_t)sp.count);
            for (int i = 0; i < sp.count; ++i) ids[(size_t)i] = first + i;
            std::vector<Leaf> leaves;
            split_leaves(objs, ids, cluster_leaf, leaves);
            /* the leaves come out of the k-d split in spatial order; each becomes one item of both scans */
            for (const Leaf &L : leaves) {
                const int member_off = (int)geom.size();
                const int slot = (int)cidx.size();
                for (int i : L.members) { emit_geometry(i); cidx.push_back((uint32_t)i); clustered[(size_t)i] = 1; }
                LeafItem li;
                for (int k = 0; k < 3; ++k) { li.lo[k] = L.lo[k]; li.hi[k] = L.hi[k]; }
                li.member_off = (uint32_t)member_off;
                li.count = (uint32_t)L.members.size();
                li.cidx_slot = (uint32_t)slot;
                li.in_shadow = in_shadow_all;
                leaf_items.push_back(li);
            }
            n_clusters += (int)leaves.size();
        } else if (sp.kind == RT_KIND_FINITE_PLANE && aa_planes) {
            /* axis-aligned members leave the in-order run for the class-sorted tables built below */
            for (int i = first; i < last; ++i) emit_geometry(i);
            int i = first;
            while (i < last) {
                float sn, sh, sv; int ka, kb;
                const bool aa = aa_class(objs[i], &sn, &sh, &sv, &ka, &kb) >= 0;
                int j = i;
                while (j < last) {
                    const bool aj = aa_class(objs[j], &sn, &sh, &sv, &ka, &kb) >= 0;
                    if (aj != aa) break;
                    ++j;
                }
                if (aa) {
                    for (int k = i; k < j; ++k) aa_all.push_back(k);
                }
                i = j;
            }
        } else {
            for (int i = first; i < last; ++i) emit_geometry(i);
        }
        if (geom.size() > RT_MAX_GEOM_QUADS) return fail(RT_ERR_CAPACITY, "geometry table too large");
    }
    /* axis-aligned rectangles: their two-quad test records (rt_tables.h) */
    for (int i : aa_all) {
        float sn, sh, sv; int ka, kb;
        const int cls = aa_class(objs[i], &sn, &sh, &sv, &ka, &kb);
        const rt_object_desc &o = objs[i];
        aa_rec_of[(size_t)i] = (int)aa_recs.size();
        aa_cls_of[(size_t)i] = cls % 3;
        const bool swapped = cls >= 3;                   /* first in-plane axis is the plane's "vertical" */
        aa_recs.push_back({{o.distance_to_origin, sn, sh, sv}});
        aa_recs.push_back({{o.plane_origin[ka], o.plane_origin[kb], swapped ? o.v_distance : o.h_distance,
                            swapped ? o.h_distance : o.v_distance}});
    }
    for (int i = 0; i < n; ++i)
        objinfo[(size_t)i] = (uint32_t)geom_off[(size_t)i] | ((uint32_t)objs[i].kind << 16) |
                             ((uint32_t)mat_of[(size_t)i] << 20);

    /* assemble the image (into temporaries: a failure below leaves the handle as it was) */
    RtParams b;
    std::memset(&b, 0, sizeof(b));
    std::vector<Quad> image;
    image.insert(image.end(), geom.begin(), geom.end());
    const int aa_off = (int)image.size();
    image.insert(image.end(), aa_recs.begin(), aa_recs.end());
    /* Scene-index tables of the clustered / class-sorted runs */
    const int cidx_off = (int)image.size();
    image.resize(image.size() + (cidx.size() + 3) / 4, Quad{{0, 0, 0, 0}});
    if (!cidx.empty()) std::memcpy(image[(size_t)cidx_off].v, cidx.data(), cidx.size() * 4);

    /* FAST tables (rt_tables.h): scenes without clustered runs get one kind-sorted item list with direct test
     * records instead of the two item tables (and the sections only those refer to) */
    /* (only while the tables go to LDS: the large-scene kernel reads the item tables) */
    const size_t fast_quads = geom.size() + (size_t)n * (RT_FAST_BOX_QUADS + RT_FAST_REC_QUADS) + lights.size() + mats.size() +
                              texs.size() + 2 * (((size_t)n + 3) / 4);
    const bool fast = s->cull_opt && s->fast_opt && n_clusters == 0 && n > 0 && s->tables_opt != 2 &&
                      fast_quads * 16 <= (s->tables_opt == 1 ? (size_t)RT_MAX_LDS_BYTES : (size_t)RT_LDS_TABLE_BYTES);
    std::vector<uint32_t> fast_ctl;
    std::vector<Quad> fast_boxes, fast_recs;
    int fast_n_shadow = 0;
    /* Item tables (rt_tables.h): one item per object that is not part of a clustered run, in
     * Scene order, then one per leaf (or group) of each clustered run.  `near_items` covers every object,
     * `shadow_items` the non-light objects of the shadow scan range. */
    {
        const float INF = INFINITY;
        /* Padding.  A sphere's box also has to hold what the reference's coarse float sphere test reports
         * (box_needed() in rt_kernel.hip): 1 % of its extent here plus the kernel's distance-proportional
         * RT_SPHERE_SLACK.  A plane item (RT_ITEM_TIGHT) gets 2e-5 + 2e-5 of the box's magnitude: 170 ulp of the
         * largest coordinate.  What it has to cover is the part of the hit point's error that scales with WHERE the
         * rectangle is: p = t d + o is rounded twice per component (2 ulp of |p_k|) and the bounds test
         * (p - plane_origin).h, 0 <= x <= h_dist adds a few ulp of the rectangle's coordinates -- well under 20 ulp
         * in all.  The part that scales with the distance TRAVELLED -- t d_k and o_k cancel when a ray comes from
         * far away, and t itself carries the relative error of n.o + dto: about 1e-6 of the distance for origins
         * 1e4-6e4 away, grazing or not -- is the kernel's per-axis RT_PLANE_SLACK, 1e-5 of the distance on that
         * axis (an axis on which the ray hardly moves has a hit-point error that small, too: the error of t is
         * multiplied by d_k).  tests/scene_gen.py's far-origin grazing scenes and scripts/fuzz_gpu.py's `far` mode
         * exercise exactly that against the oracle.  (With the sphere padding a 14-unit wall was 0.28 thick and,
         * e.g., a light 0.01 in front of it made it a candidate of every shadow scan towards that light; at 170 ulp
         * a plane is not even a candidate of the shadow rays that START on it, 1e-3 in front of it --
         * src/SceneFinitePlane.h:11 --, in scenes up to a few tens of units across.) */
        auto box_item = [&](std::vector<Quad> &out, const double lo[3], const double hi[3], uint32_t bits,
                            uint32_t word1, bool unbounded) {
            double ext = 0.0, mag = 0.0;
            for (int k = 0; k < 3; ++k) {
                if (!std::isfinite(lo[k]) || !std::isfinite(hi[k])) continue;       /* an axis the item is unbounded on */
                ext = std::max(ext, hi[k] - lo[k]);
                mag = std::max(mag, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
            }
            const double pad = (bits & RT_ITEM_TIGHT) ? 2e-5 + 2e-5 * mag : 1e-4 + 1e-4 * mag + 1e-2 * ext;
            Quad q0, q1;
            for (int k = 0; k < 3; ++k) {
                const bool ok = !unbounded && std::isfinite(lo[k]) && std::isfinite(hi[k]) && std::isfinite(pad);
                centre_half(ok ? (double)std::nextafter((float)(lo[k] - pad), -INF) : -(double)INF,
                            ok ? (double)std::nextafter((float)(hi[k] + pad), INF) : (double)INF, &q0.v[k], &q1.v[k]);
            }
            q0.v[3] = bits_to_float(bits);
            q1.v[3] = bits_to_float(word1);
            out.push_back(q0);
            out.push_back(q1);
        };
        auto object_item = [&](std::vector<Quad> &out, int i) {
            const rt_object_desc &o = objs[i];
            double lo[3], hi[3];
            const uint32_t full = (uint32_t)geom_off[(size_t)i];
            const uint32_t word1 = (uint32_t)i | (full << 12);          /* Scene index | full record offset */
            if (o.kind == RT_KIND_SPHERE) {
                const double r = std::fabs((double)o.radius);
                for (int k = 0; k < 3; ++k) { lo[k] = (double)o.origin[k] - r; hi[k] = (double)o.origin[k] + r; }
                box_item(out, lo, hi, (uint32_t)RT_KIND_SPHERE | (full << 16), word1, false);
            } else if (o.kind == RT_KIND_INFINITE_PLANE) {
                /* An infinite plane whose normal is a +-unit axis vector is a SLAB: bounded on that axis (at
                 * x_k = -dto * sign: sign * x_k + dto = 0), unbounded on the others, and tight like a finite
                 * plane -- the hit point's k component is t d_k + o_k with t = (-dto - o_k sign) / (d_k sign)
                 * (src/SceneInfinitePlane.cpp:40-55), off the plane by a few ulp of |o_k - x_k| + |x_k|.  Any
                 * other infinite plane is unbounded on every axis: a candidate of every scan. */
                float sign = 0.0f;
                const int axis = (s->tight_planes && std::isfinite(o.distance_to_origin)) ? unit_axis(o.normal, &sign) : -1;
                for (int k = 0; k < 3; ++k) { lo[k] = -(double)INF; hi[k] = (double)INF; }
                if (axis >= 0) lo[axis] = hi[axis] = -(double)o.distance_to_origin * (double)sign;
                box_item(out, lo, hi, (uint32_t)RT_KIND_INFINITE_PLANE | (axis >= 0 ? (uint32_t)RT_ITEM_TIGHT : 0u) | (full << 16),
                         word1, axis < 0);
            } else {
                /* The hit region is {p on the plane : 0 <= (p-po).h <= h_dist, 0 <= (p-po).v <= v_dist}
                 * (src/SceneFinitePlane.cpp:117-124).  h and v need be neither orthogonal to each
                 * other nor to the normal (axis constructor with arbitrary vectors), so the
                 * corners come from solving  u.n = 0, u.h = x, u.v = y  for u = p - po. */
                bool unbounded = false;
                for (int k = 0; k < 3; ++k) { lo[k] = 1e300; hi[k] = -1e300; }
                {
                    const double n[3] = {o.normal[0], o.normal[1], o.normal[2]};
                    const double h[3] = {o.horizontal[0], o.horizontal[1], o.horizontal[2]};
                    const double v[3] = {o.vertical[0], o.vertical[1], o.vertical[2]};
                    auto cross3 = [](const double a[3], const double c[3], double r[3]) {
                        r[0] = a[1] * c[2] - a[2] * c[1]; r[1] = a[2] * c[0] - a[0] * c[2]; r[2] = a[0] * c[1] - a[1] * c[0];
                    };
                    double hv[3], vn[3], nh[3];
                    cross3(h, v, hv); cross3(v, n, vn); cross3(n, h, nh);
                    const double det = n[0] * hv[0] + n[1] * hv[1] + n[2] * hv[2];
                    if (!(std::fabs(det) > 1e-6) || !std::isfinite(det)) {
                        unbounded = true;
                    } else {
                        for (int a = 0; a < 2; ++a)
                            for (int c = 0; c < 2; ++c) {
                                const double x = a * (double)o.h_distance, y = c * (double)o.v_distance;
                                for (int k = 0; k < 3; ++k) {          /* u = (x (v x n) + y (n x h)) / det */
                                    const double u = (x * vn[k] + y * nh[k]) / det;
                                    const double pk = (double)o.plane_origin[k] + u;
                                    lo[k] = std::min(lo[k], pk);
                                    hi[k] = std::max(hi[k], pk);
                                }
                            }
                    }
                }
                if (unbounded) { for (int k = 0; k < 3; ++k) { lo[k] = hi[k] = 0.0; } }
                const uint32_t tight = s->tight_planes ? (uint32_t)RT_ITEM_TIGHT : 0u;
                if (aa_rec_of[(size_t)i] >= 0)
                    box_item(out, lo, hi, (uint32_t)(RT_KIND_FINITE_AA + aa_cls_of[(size_t)i]) | tight |
                                              ((uint32_t)(aa_off + aa_rec_of[(size_t)i]) << 16), word1, unbounded);
                else
                    box_item(out, lo, hi, (uint32_t)RT_KIND_FINITE_PLANE | tight | (full << 16), word1, unbounded);
            }
        };
        for (int i = 0; i < n; ++i)
            if (!clustered[(size_t)i]) object_item(near_items, i);
        auto leaf_item = [&](std::vector<Quad> &out, const LeafItem &l) {
            Quad q0 = {{0, 0, 0, bits_to_float((uint32_t)RT_KIND_SPHERE_LEAF | (l.count << 8) | (l.member_off << 16))}};
            Quad q1 = {{0, 0, 0, bits_to_float((uint32_t)(cidx_off * 4) + l.cidx_slot)}};
            for (int k = 0; k < 3; ++k) centre_half((double)l.lo[k], (double)l.hi[k], &q0.v[k], &q1.v[k]);
            out.push_back(q0);
            out.push_back(q1);
        };
        b.near_first_leaf = (int)(near_items.size() / 2);
        for (const LeafItem &l : leaf_items) leaf_item(near_items, l);
        for (int i = sb; i < se; ++i)
            if (!objs[i].is_light && !clustered[(size_t)i]) object_item(shadow_items, i);
        b.shadow_first_leaf = (int)(shadow_items.size() / 2);
        for (const LeafItem &l : leaf_items)
            if (l.in_shadow) leaf_item(shadow_items, l);
        if (fast) {
            auto kind_rank = [&](int i) {
                if (aa_rec_of[(size_t)i] >= 0) return aa_cls_of[(size_t)i];                 /* 0..2: AA rectangles by normal axis */
                return objs[i].kind == RT_KIND_SPHERE ? 3 : objs[i].kind == RT_KIND_FINITE_PLANE ? 4 : 5;
            };
            std::vector<int> order;
            for (int part = 0; part < 2; ++part)
                for (int rank = 0; rank < 6; ++rank)
                    for (int i = 0; i < n; ++i) {
                        const bool in_shadow = i >= sb && i < se && !objs[i].is_light;
                        if ((part == 0) == in_shadow && kind_rank(i) == rank) order.push_back(i);
                    }
            for (int i : order) {
                const rt_object_desc &o = objs[i];
                if (i >= sb && i < se && !o.is_light) ++fast_n_shadow;
                std::vector<Quad> item;
                object_item(item, i);
                const uint32_t kind = aa_rec_of[(size_t)i] >= 0 ? (uint32_t)(RT_KIND_FINITE_AA + aa_cls_of[(size_t)i]) : (uint32_t)o.kind;
                const uint32_t ctl = kind | ((uint32_t)i << 8);
                uint32_t bits0; std::memcpy(&bits0, &item[0].v[3], 4);
                item[0].v[3] = bits_to_float(kind | (bits0 & RT_ITEM_TIGHT));
                item[1].v[3] = bits_to_float(ctl);
                fast_boxes.push_back(item[0]);
                fast_boxes.push_back(item[1]);
                fast_ctl.push_back(ctl);
                if (aa_rec_of[(size_t)i] >= 0) {
                    fast_recs.push_back(aa_recs[(size_t)aa_rec_of[(size_t)i]]);
                    fast_recs.push_back(aa_recs[(size_t)aa_rec_of[(size_t)i] + 1]);
                } else if (o.kind == RT_KIND_SPHERE) {
                    fast_recs.push_back({{o.origin[0], o.origin[1], o.origin[2], o.radius_squared}});
                    fast_recs.push_back({{0, 0, 0, 0}});
                } else {
                    fast_recs.push_back({{o.normal[0], o.normal[1], o.normal[2], o.distance_to_origin}});
                    fast_recs.push_back({{bits_to_float((uint32_t)geom_off[(size_t)i]), 0, 0, 0}});
                }
            }
        }
    }
    if (fast) {
        image.assign(geom.begin(), geom.end());                /* no leaves, no aa section, no cidx */
        b.n_fast_items = (int)fast_ctl.size();
        b.n_fast_shadow = fast_n_shadow;
        b.fast_box_off = (int)image.size();
        image.insert(image.end(), fast_boxes.begin(), fast_boxes.end());
        b.fast_rec_off = (int)image.size();
        image.insert(image.end(), fast_recs.begin(), fast_recs.end());
        near_items.clear();
        shadow_items.clear();
    }
    b.near_items_off = (int)image.size();
    b.n_near_items = fast ? b.n_fast_items : (int)(near_items.size() / 2);
    image.insert(image.end(), near_items.begin(), near_items.end());
    b.shadow_items_off = (int)image.size();
    b.n_shadow_items = fast ? b.n_fast_shadow : (int)(shadow_items.size() / 2);
    image.insert(image.end(), shadow_items.begin(), shadow_items.end());
    b.lights_off = (int)image.size();
    image.insert(image.end(), lights.begin(), lights.end());
    b.mat_off = (int)image.size();
    image.insert(image.end(), mats.begin(), mats.end());
    b.tex_off = (int)image.size();
    image.insert(image.end(), texs.begin(), texs.end());
    b.objinfo_off = (int)image.size();
    image.resize(image.size() + ((size_t)n + 3) / 4, Quad{{0, 0, 0, 0}});
    if (n > 0) std::memcpy(image[(size_t)b.objinfo_off].v, objinfo.data(), (size_t)n * 4);
    if (fast) {
        b.fast_ctl_off = (int)image.size() * 4;
        image.resize(image.size() + (fast_ctl.size() + 3) / 4, Quad{{0, 0, 0, 0}});
        std::memcpy(image[(size_t)b.fast_ctl_off / 4].v, fast_ctl.data(), fast_ctl.size() * 4);
    }
    if (image.empty()) image.push_back(Quad{{0, 0, 0, 0}});   /* keep uploads non-empty */
    b.image_quads = (int)image.size();
    /* SHADOW VOXELS: behind the staged part; read from the global copy by the clustered-scene kernels */
    b.svox_off = 0;
    if (!fast && s->cull_opt && s->svox_opt != 0 && n_clusters > 0) {
        ShadowVoxels sv;
        /* automatic: from RT_SVOX_MIN_LEAVES leaves on.  With fewer the bundle cull leaves little to take away -- the 256-sphere
         * grid (16 leaves) at depth 8: 3.8 -> 2.6 candidates per scan, frame 3.94 -> 4.03 ms with the table; the 1 024-sphere grid
         * (43 leaves): 8.9 -> 4.4, 3.87 -> 3.71 ms (profiles/r04_experiments.txt 8) */
        if (shadow_voxels(shadow_items, b.shadow_first_leaf, lights, s->svox_opt > 0 ? s->svox_opt : RT_SVOX_MAX_CELLS,
                          s->svox_opt > 0 ? 4 : RT_SVOX_MIN_LEAVES, &sv)) {
            b.svox_off = (int)image.size();
            image.resize(image.size() + (sv.masks.size() + 1) / 2, Quad{{0, 0, 0, 0}});
            std::memcpy(image[(size_t)b.svox_off].v, sv.masks.data(), sv.masks.size() * 8);
            for (int k = 0; k < 3; ++k) { b.svox_n[k] = sv.n[k]; b.svox_lo[k] = sv.lo[k]; b.svox_scale[k] = sv.scale[k]; }
        }
    }
    b.n_clusters = n_clusters;
    b.n_lights = (int)(lights.size() / RT_LIGHT_QUADS);
    for (int c = 0; c < 3; ++c) b.null_color[c] = s->null_color[c];
    s->image.swap(image);
    s->base = b;
    s->n_clusters = n_clusters;
    return RT_OK;
}

/* validate + copy the caller's description into the handle */
int adopt_desc(const rt_scene_desc *desc, rt_scene *s) {
    const int n = desc->n_objects;
    if (n < 0) return fail(RT_ERR_INVALID, "n_objects < 0");
    if (n > 0 && !desc->objects) return fail(RT_ERR_INVALID, "objects is NULL");
    if (n > RT_MAX_OBJECTS)
        return fail(RT_ERR_CAPACITY, "more than " + std::to_string(RT_MAX_OBJECTS) + " objects (the reference's Scene holds 3 999, src/Scene.h:8)");
    if (desc->n_textures < 0 || (desc->n_textures > 0 && !desc->textures))
        return fail(RT_ERR_INVALID, "bad textures");
    if (desc->shadow_begin < 0 || desc->shadow_end < desc->shadow_begin || desc->shadow_end > n)
        return fail(RT_ERR_INVALID, "shadow range must satisfy 0 <= begin <= end <= n_objects");
    for (int i = 0; i < n; ++i) {
        const rt_object_desc &o = desc->objects[i];
        if (o.kind != RT_KIND_SPHERE && o.kind != RT_KIND_INFINITE_PLANE && o.kind != RT_KIND_FINITE_PLANE)
            return fail(RT_ERR_INVALID, "object " + std::to_string(i) + ": unknown kind");
        if (o.texture < -1 || o.texture >= desc->n_textures)
            return fail(RT_ERR_INVALID, "object " + std::to_string(i) + ": texture index out of range");
    }
    s->objects.assign(desc->objects, desc->objects + n);
    s->textures.assign(desc->textures, desc->textures + desc->n_textures);
    s->shadow_begin = desc->shadow_begin;
    s->shadow_end = desc->shadow_end;
    for (int c = 0; c < 3; ++c) s->null_color[c] = desc->null_color[c];
    return RT_OK;
}

/* (re)upload the packed tables */
int upload_scene(rt_scene *s) {
    HIP_TRY(hipSetDevice(s->device));
    if (s->d_image) { HIP_TRY(hipFree(s->d_image)); s->d_image = nullptr; }
    hipEvent_t t0, t1;
    HIP_TRY(hipEventCreate(&t0));
    HIP_TRY(hipEventCreate(&t1));
    HIP_TRY(hipEventRecord(t0, nullptr));
    const size_t image_bytes = s->image.size() * sizeof(Quad);
    HIP_TRY(hipMalloc(&s->d_image, image_bytes));
    HIP_TRY(hipMemcpy(s->d_image, s->image.data(), image_bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipEventRecord(t1, nullptr));
    HIP_TRY(hipEventSynchronize(t1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
    s->timing.last_upload_ms = ms;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return RT_OK;
}

int ensure_events(rt_scene *s) {
    if (s->ev_ready) return RT_OK;
    for (int i = 0; i < kEventRing; ++i) {
        HIP_TRY(hipEventCreate(&s->ev[i].start));
        HIP_TRY(hipEventCreate(&s->ev[i].stop));
        s->ev[i].pending = false;
    }
    s->ev_ready = true;
    return RT_OK;
}

int drain_event(rt_scene *s, int i) {
    EventPair &e = s->ev[i];
    if (!e.pending) return RT_OK;
    HIP_TRY(hipEventSynchronize(e.stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e.start, e.stop));
    s->timing.last_kernel_ms = ms;
    s->timing.sum_kernel_ms += ms;
    s->timing.launches += 1;
    e.pending = false;
    return RT_OK;
}

/* Where the tile queues start (thousandths of the image height; the rows wrap
 * around).  Scheduling only.  The most expensive tiles of a frame should not be
 * the last ones handed out, and with clustered sphere runs in the scene there is
 * one place where they are known to be: where primary rays graze an infinite
 * plane.  Hit points thousands of units away make every leaf of every run a
 * candidate of their shadow rays (the reference's float sphere test is that
 * coarse out there, box_needed() in rt_kernel.hip), so those few pixel rows cost
 * 10-100 x the median tile -- a single wavefront works milliseconds on one.
 * Starting just below the horizon row (image centre column) puts them first:
 * 15 % on the 1 024-sphere grid frame, more on the strips of a multi-GPU frame.
 * Without clustered runs the natural order (bottom row first) is kept. */
/* the height dz (fraction of the image, may lie outside [0, 1]) at which the pixel column at dx looks along infinite
 * plane `o`: its rays graze the plane there -- the plane's horizon line; false if the column never does */
bool horizon_dz(const rt_object_desc &o, const rt_camera_desc *cam, double dx, double *dz) {
    /* direction of the pixel at (dx, dz) (src/Camera.cpp:71-84), dotted with n: A + B dz */
    double a = 0.0, b = 0.0;
    for (int c = 0; c < 3; ++c) {
        const double at_dx = (double)cam->screen_origin[c] +
                             (double)cam->vector_horizontal[c] * (dx * cam->screen_width - cam->screen_halfwidth) -
                             (double)cam->vector_vertical[c] * cam->screen_halfheight - (double)cam->eye_origin[c];
        a += at_dx * o.normal[c];
        b += (double)cam->vector_vertical[c] * cam->screen_height * o.normal[c];
    }
    if (!(std::fabs(b) > 0.0) || !std::isfinite(a / b)) return false;
    *dz = -a / b;
    return true;
}

/* the infinite plane whose horizon line crosses the image's centre column lowest (nullptr: none does) */
const rt_object_desc *horizon_plane(const rt_scene *s, const rt_camera_desc *cam, double *dz_centre) {
    const rt_object_desc *best = nullptr;
    for (const rt_object_desc &o : s->objects) {
        double dz;
        if (o.kind != RT_KIND_INFINITE_PLANE || !horizon_dz(o, cam, 0.5, &dz)) continue;
        if (dz > 0.0 && dz < 1.0 && (!best || dz < *dz_centre)) { best = &o; *dz_centre = dz; }
    }
    return best;
}

int horizon_start(const rt_scene *s, const rt_camera_desc *cam) {
    if (s->n_clusters <= 0) return 0;
    double dz = 0.0;
    if (!horizon_plane(s, cam, &dz)) return 0;
    return std::max(0, (int)(dz * 1000.0) - 8);
}

/* PRIMARY table (rt_tables.h): for every item of the FAST list, the rectangle of pixels of a W x H image whose camera ray can
 * reach the item, and a lower bound of the distance at which it does.
 *
 * Exactness.  The kernel's culls rest on: a hit the reference's float test reports lies inside the item's box grown by the
 * slack of RT_CULL_SLACK (rt_kernel.hip; the box is already padded on the host).  The camera ray of pixel (x, z) leaves the eye
 * through the screen point so + ch a + cv b with a = x/W sw - shw, b = z/H sh - shh (src/Camera.cpp:71-84), evaluated in floats:
 * off the exact point by a few ulp of the coordinates involved.  So the hit is on a line from the eye through a point within
 * `err` of that pixel's exact screen point, and inside the grown box.  The grown box is convex and -- when all eight corners are
 * in front of the eye -- its central projection onto the screen plane is the hull of the projected corners: their bounding
 * rectangle in (a, b), turned into pixels and widened by two pixels plus `err` in pixels, contains every pixel that can hit the
 * item.  A box that reaches behind the camera is clipped a little in front of the eye first (see below: nothing a ray can reach is
 * lost), an unbounded one is cut 70 000 away (the rays end at 65 535), one entirely behind the camera gets no pixel, and an item
 * with the eye inside its grown box gets the whole image.  The entry distance is the Euclidean distance from the eye to the grown box less the scans' tolerance (a ray's
 * parameter is its distance: |d| = 1 to 2e-7); 0 for items that contain the eye, which are always tested.
 * Returns false when no table can be made (too many items, a degenerate camera, an image too large for 16-bit pixels). */
bool primary_table(const rt_scene *s, const rt_camera_desc *cam, int W, int H, uint32_t *out /* [n][4] */) {
    const int n = s->base.n_fast_items;
    if (n <= 0 || n > RT_PRIMARY_ITEMS || W > 30000 || H > 30000) return false;
    double eye[3], w0[3], ch[3], cv[3];
    for (int k = 0; k < 3; ++k) {
        eye[k] = cam->eye_origin[k]; w0[k] = (double)cam->screen_origin[k] - eye[k];
        ch[k] = cam->vector_horizontal[k]; cv[k] = cam->vector_vertical[k];
    }
    const double sw = cam->screen_width, sh = cam->screen_height, shw = cam->screen_halfwidth, shh = cam->screen_halfheight;
    if (!(sw > 0.0) || !(sh > 0.0) || !std::isfinite(sw) || !std::isfinite(sh) || !std::isfinite(shw) || !std::isfinite(shh)) return false;
    /* M = [w0 | ch | cv]; c - eye = u0 (w0 + a ch + b cv) with a = u1/u0, b = u2/u0 */
    const double m[3][3] = {{w0[0], ch[0], cv[0]}, {w0[1], ch[1], cv[1]}, {w0[2], ch[2], cv[2]}};
    const double det = m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
                       m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
    double scale = 0.0;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) scale = std::max(scale, std::fabs(m[r][c]));
    if (!std::isfinite(det) || !(std::fabs(det) > 1e-9 * scale * scale * scale) || scale == 0.0) return false;
    double inv[3][3];
    inv[0][0] = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) / det; inv[0][1] = (m[0][2] * m[2][1] - m[0][1] * m[2][2]) / det; inv[0][2] = (m[0][1] * m[1][2] - m[0][2] * m[1][1]) / det;
    inv[1][0] = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) / det; inv[1][1] = (m[0][0] * m[2][2] - m[0][2] * m[2][0]) / det; inv[1][2] = (m[0][2] * m[1][0] - m[0][0] * m[1][2]) / det;
    inv[2][0] = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) / det; inv[2][1] = (m[0][1] * m[2][0] - m[0][0] * m[2][1]) / det; inv[2][2] = (m[0][0] * m[1][1] - m[0][1] * m[1][0]) / det;
    /* float error of a pixel's screen point (a few ulp of every term's magnitude), in units of a and b, in pixels */
    double mag = 0.0, len_h = 0.0, len_v = 0.0;
    for (int k = 0; k < 3; ++k) {
        mag += std::fabs((double)cam->screen_origin[k]) + std::fabs(eye[k]) + std::fabs(ch[k]) * (std::fabs(shw) + sw) + std::fabs(cv[k]) * (std::fabs(shh) + sh);
        len_h += ch[k] * ch[k]; len_v += cv[k] * cv[k];
    }
    len_h = std::sqrt(len_h); len_v = std::sqrt(len_v);
    if (!(len_h > 0.0) || !(len_v > 0.0)) return false;
    const double err = 16.0 * 1.2e-7 * mag;                       /* 16 roundings' worth */
    const double margin_x = 2.0 + std::ceil(err / len_h * (double)W / sw), margin_z = 2.0 + std::ceil(err / len_v * (double)H / sh);
    if (!(margin_x < 1000.0) || !(margin_z < 1000.0)) return false;
    const Quad *boxes = s->image.data() + s->base.fast_box_off;
    for (int i = 0; i < n; ++i) {
        const Quad &b0 = boxes[2 * i], &b1 = boxes[2 * i + 1];
        uint32_t bits; std::memcpy(&bits, &b0.v[3], 4);
        /* an axis the item is unbounded on (infinite planes): the rays end at 65535 (the reference's infinity), so 70 000 either
         * side of the eye is as good as unbounded */
        double lo[3], hi[3], far[3], far_sum = 0.0;
        for (int k = 0; k < 3; ++k) {
            lo[k] = std::isfinite(box_lo(b0, b1, k)) ? box_lo(b0, b1, k) : eye[k] - 7.0e4;
            hi[k] = std::isfinite(box_hi(b0, b1, k)) ? box_hi(b0, b1, k) : eye[k] + 7.0e4;
            lo[k] = std::max(lo[k], eye[k] - 7.0e4); hi[k] = std::min(hi[k], eye[k] + 7.0e4);
            far[k] = std::max(std::fabs(lo[k] - eye[k]), std::fabs(hi[k] - eye[k]));
            far_sum += far[k];
        }
        int x_lo = -32768, x_hi = 32767, z_lo = -32768, z_hi = 32767;         /* the whole image, always tested: the fallback */
        float entry = 0.0f;
        {
            /* the kernel's slack (RT_CULL_SLACK: 1.5e-3 of the L1 distance for sphere-like items, 1e-5 per axis for planes), twice over */
            double d2 = 0.0;
            bool eye_inside = true;
            for (int k = 0; k < 3; ++k) {
                const double ex = 2.0 * (((bits & RT_ITEM_TIGHT) ? 1.0e-5 * far[k] : 1.5e-3 * far_sum) + 1.0e-4);
                lo[k] -= ex; hi[k] += ex;
                const double dk = std::max(std::max(lo[k] - eye[k], eye[k] - hi[k]), 0.0);
                if (dk > 0.0) eye_inside = false;
                d2 += dk * dk;
            }
            const double dist = std::sqrt(d2);
            if (!eye_inside && dist > 0.0 && lo[0] <= hi[0] && lo[1] <= hi[1] && lo[2] <= hi[2]) {
                /* In the camera's coordinates u (c - eye = u0 w0 + u1 ch + u2 cv) a ray of pixel (a, b) is u = lambda (1, a, b):
                 * what it can reach of the box has u0 = lambda > 0.  The part of the box with u0 < eps lies within
                 * eps (|w0| + |a ch| + |b cv|) of the eye for the image's pixels (|a|, |b| <= the screen's half sizes + 1); with
                 * eps below dist / that length it is empty of reachable points, so the box may be clipped at u0 = eps before it
                 * is projected -- which keeps the projection finite for boxes that reach behind the camera. */
                double reach = 0.0;
                for (int k = 0; k < 3; ++k) reach += std::fabs(w0[k]) + std::fabs(ch[k]) * (std::fabs(shw) + sw + 1.0) + std::fabs(cv[k]) * (std::fabs(shh) + sh + 1.0);
                const double eps = std::min(0.5, 0.5 * dist / reach);
                double u[8][3];
                for (int c = 0; c < 8; ++c) {
                    const double v[3] = {((c & 1) ? hi[0] : lo[0]) - eye[0], ((c & 2) ? hi[1] : lo[1]) - eye[1], ((c & 4) ? hi[2] : lo[2]) - eye[2]};
                    for (int r = 0; r < 3; ++r) u[c][r] = inv[r][0] * v[0] + inv[r][1] * v[1] + inv[r][2] * v[2];
                }
                double a_lo = 1e300, a_hi = -1e300, b_lo = 1e300, b_hi = -1e300;
                int kept = 0;
                auto project = [&](double u0, double u1, double u2) {
                    a_lo = std::min(a_lo, u1 / u0); a_hi = std::max(a_hi, u1 / u0);
                    b_lo = std::min(b_lo, u2 / u0); b_hi = std::max(b_hi, u2 / u0);
                    ++kept;
                };
                for (int c = 0; c < 8; ++c) {
                    if (u[c][0] >= eps) project(u[c][0], u[c][1], u[c][2]);
                    for (int axis = 0; axis < 3; ++axis) {                      /* the edges from c towards higher corners */
                        const int o2 = c | (1 << axis);
                        if (o2 == c) continue;
                        const double p0 = u[c][0], p1 = u[o2][0];
                        if ((p0 < eps) != (p1 < eps)) {                         /* the edge crosses u0 = eps */
                            const double f = (eps - p0) / (p1 - p0);
                            project(eps, u[c][1] + f * (u[o2][1] - u[c][1]), u[c][2] + f * (u[o2][2] - u[c][2]));
                        }
                    }
                }
                const double e = dist - 1.0e-4 * dist - 1.0e-6;
                entry = e > 0.0 ? std::nextafter((float)e, 0.0f) : 0.0f;              /* rounded towards 0 */
                if (!(entry > 0.0f) || !std::isfinite(entry)) entry = 0.0f;
                if (kept == 0) {
                    x_lo = 1; x_hi = 0; z_lo = 1; z_hi = 0;                            /* all of it behind the camera: no pixel */
                } else if (std::isfinite(a_lo) && std::isfinite(a_hi) && std::isfinite(b_lo) && std::isfinite(b_hi)) {
                    auto clamp16 = [](double v) { return (int)std::max(-32768.0, std::min(32767.0, v)); };
                    /* (a relative widening for the interpolated points and the division) */
                    const double wa = 1e-9 * (std::fabs(a_lo) + std::fabs(a_hi)), wb = 1e-9 * (std::fabs(b_lo) + std::fabs(b_hi));
                    x_lo = clamp16(std::floor((a_lo - wa + shw) / sw * (double)W - margin_x));
                    x_hi = clamp16(std::ceil((a_hi + wa + shw) / sw * (double)W + margin_x));
                    z_lo = clamp16(std::floor((b_lo - wb + shh) / sh * (double)H - margin_z));
                    z_hi = clamp16(std::ceil((b_hi + wb + shh) / sh * (double)H + margin_z));
                }
            }
        }
        uint32_t ebits; std::memcpy(&ebits, &entry, 4);
        out[4 * i + 0] = ((uint32_t)x_lo & 0xFFFFu) | ((uint32_t)x_hi << 16);
        out[4 * i + 1] = ((uint32_t)z_lo & 0xFFFFu) | ((uint32_t)z_hi << 16);
        out[4 * i + 2] = ebits;
        out[4 * i + 3] = 0u;
    }
    return true;
}

/* Workgroup size and where the bounce stack goes.  The stack is 16 B per level
 * per thread.  As many of its lowest levels as fit share LDS with the scene
 * tables while RT_STACK_LDS_SHARE workgroups per CU still fit in the 160 KiB
 * (nearly every reflection chain uses the first levels, few the deep ones); the
 * rest lives in HBM.  Option "stack": 1 = all of it in LDS, 2 = all in HBM. */
int primary_quads(const rt_scene *s) {
    return (s->primary_opt && s->base.n_fast_items > 0 && s->base.n_fast_items <= RT_PRIMARY_ITEMS) ? s->base.n_fast_items : 0;
}

int choose_block(const rt_scene *s, int max_depth, bool counting, int *block, int *lds_bytes, int *stack_lds_levels, bool *global_tables,
                 int block_override = 0) {
    /* Tables in LDS (staged once per workgroup), or -- large scenes -- left in global memory and read
     * through the L2 (rt_render_kernel_large): automatic beyond RT_LDS_TABLE_BYTES, where LDS would hold
     * fewer than two workgroups per CU; beyond 160 KiB it is the only way.  Option "tables". */
    /* (FAST tables carry this launch's PRIMARY table behind the image: one quad per item) */
    size_t scene_bytes = ((size_t)s->base.image_quads + (primary_quads(s) > 0 ? (size_t)primary_quads(s) : 0)) * 16;
    /* the counting build has no global-memory variant: automatic means LDS for it whenever the tables fit at all */
    *global_tables = s->tables_opt == 2 ||
                     (s->tables_opt == 0 && scene_bytes > (counting ? (size_t)RT_MAX_LDS_BYTES : (size_t)RT_LDS_TABLE_BYTES));
    if (!*global_tables && scene_bytes > RT_MAX_LDS_BYTES)
        return fail(RT_ERR_CAPACITY, "option tables=1: the scene tables do not fit in LDS (160 KiB)");
    if (*global_tables) scene_bytes = 0;
    *block = block_override ? block_override : (s->block_threads_opt ? s->block_threads_opt : 256);
    const double per_level = (double)RT_STACK_ENTRY_BYTES * (double)*block;
    /* levels 0 .. max_depth - 1 can push an entry (the last level's reflection is folded where it is found: rt_kernel.hip) */
    const double levels = (double)max_depth;
    double in_lds = 0.0;
    if (s->stack_opt == 1) {
        in_lds = levels;
        if ((double)scene_bytes + levels * per_level > (double)RT_MAX_LDS_BYTES)
            return fail(RT_ERR_CAPACITY, "stack option: tables + bounce stack exceed 160 KiB LDS");
    } else if (s->stack_opt == 0) {
        /* the clustered-scene kernels run six wavefronts per SIMD (80 registers, no spills), the others seven */
        const int share256 = (s->n_clusters > 0 && s->pairs_opt && s->cull_opt) ? 6 : RT_STACK_LDS_SHARE;
        const int share = std::max(1, share256 * 256 / *block);       /* (workgroups per CU: the same wavefronts in larger ones) */
        const double room = (double)(RT_MAX_LDS_BYTES / share) - (double)scene_bytes;
        in_lds = room > 0.0 ? std::floor(room / per_level) : 0.0;
        if (in_lds > levels) in_lds = levels;
    }
    *stack_lds_levels = (int)in_lds;
    *lds_bytes = (int)((double)scene_bytes + in_lds * per_level);
    if (*lds_bytes < 16) *lds_bytes = 16;
    return RT_OK;
}

/* what a finished kernel told the host: reported once, by the first synchronous point that sees it */
int device_report(rt_scene *s) {
    if (s->h_error && *reinterpret_cast<volatile unsigned int *>(s->h_error) != 0u) {
        *s->h_error = 0u;
        return fail(RT_ERR_HIP, "a wavefront's wait for the helpers at its workgroup's desk timed out (HELP): the image is complete "
                                "and exact (the owner tested the leaves itself), but this should never happen");
    }
    return RT_OK;
}

int launch(rt_scene *s, const rt_camera_desc *cam, int W, int H, int x0, int x1, int max_depth,
           float *d_out, hipStream_t stream, unsigned long long *d_stats = nullptr) {
    if (!cam) return fail(RT_ERR_INVALID, "camera is NULL");
    if (W <= 0 || H <= 0) return fail(RT_ERR_INVALID, "W and H must be positive");
    if (x0 < 0 || x1 > W || x0 > x1) return fail(RT_ERR_INVALID, "need 0 <= x0 <= x1 <= W");
    if (max_depth < 0) return fail(RT_ERR_INVALID, "max_depth < 0");
    if (!d_out && x1 > x0) return fail(RT_ERR_INVALID, "output pointer is NULL");
    if ((double)(x1 - x0) * (double)H * 3.0 > 2.0e9 * 4.0)
        return fail(RT_ERR_INVALID, "strip too large");

    int block = 0, lds_bytes = 0, stack_lds_levels = 0;
    bool global_tables = false;
    int rc = choose_block(s, max_depth, d_stats != nullptr, &block, &lds_bytes, &stack_lds_levels, &global_tables);
    if (rc) return rc;
    /* Scenes with clustered runs whose tables are large (the 1 024-sphere grid: 28 KB): five workgroups of four wavefronts are
     * all that LDS admits per CU, five wavefronts per SIMD.  Workgroups of EIGHT wavefronts share one copy of the tables among
     * twice as many: three of them fit with room for bounce-stack levels, six wavefronts per SIMD in the 80-register kernel
     * (grid-32 frame 4.49 -> 4.36 ms, longest of 8 strips 0.98 -> 0.93 ms on one box, profiles/r04_experiments.txt 1).  Only
     * where it raises the occupancy: with small tables the larger workgroup gains one LDS stack level and loses in strips
     * (seven of eight wavefronts stand at the desk of a HEAVY tile). */
    /* Small tables and a deep recursion (the 256-sphere grid at depth 8), WHOLE frames: eight-wavefront workgroups at the same six
     * wavefronts per SIMD keep one bounce-stack level more in LDS (one copy of the tables less per CU), and every level that
     * stays out of HBM takes a row per workgroup out of a working set that is as large as the XCD's L2 (HBM traffic of that
     * frame 7.0 -> 5.x times the algorithmic bytes at the same frame time; strips keep four wavefronts per workgroup: seven
     * helpers at the desk of a HEAVY tile lose more than the level gains).  profiles/r04_experiments.txt 4 */
    if (!d_stats && !global_tables && s->block_threads_opt == 0 && s->n_clusters > 0 && s->pairs_opt && s->wide_opt < 0) {
        const bool few_waves = (RT_MAX_LDS_BYTES / (size_t)lds_bytes) * 4 < 24;
        const bool whole_frame = (long long)(x1 - x0) * 4 > (long long)W * 3;
        if (few_waves || (whole_frame && stack_lds_levels < max_depth)) {
            int block2 = 0, lds2 = 0, levels2 = 0;
            bool global2 = false;
            if (choose_block(s, max_depth, false, &block2, &lds2, &levels2, &global2, 512) == RT_OK && !global2 &&
                (RT_MAX_LDS_BYTES / (size_t)lds2) * 8 >= 24 && (few_waves || levels2 > stack_lds_levels)) {
                block = block2; lds_bytes = lds2; stack_lds_levels = levels2;
            }
        }
    }
    if (global_tables && d_stats)
        return fail(RT_ERR_CAPACITY, s->tables_opt == 2
                        ? "the counting build keeps the tables in LDS: set option tables to 0 or 1 for it"
                        : "the counting build keeps the tables in LDS: this scene's exceed 160 KiB");

    RtParams p = s->base;
    for (int c = 0; c < 3; ++c) {
        p.so[c] = cam->screen_origin[c];
        p.ch[c] = cam->vector_horizontal[c];
        p.cv[c] = cam->vector_vertical[c];
        p.eye[c] = cam->eye_origin[c];
    }
    p.sw = cam->screen_width; p.sh = cam->screen_height;
    p.shw = cam->screen_halfwidth; p.shh = cam->screen_halfheight;
    p.W = W; p.H = H; p.x0 = x0; p.x1 = x1; p.max_depth = max_depth;
    p.stack_lds_levels = stack_lds_levels;
    p.stack_stride = block;
    p.n_primary = 0;
    p.primary_off = s->base.image_quads;
    if (!global_tables && primary_quads(s) > 0 && primary_table(s, cam, W, H, p.primary)) p.n_primary = primary_quads(s);
    /* (the LDS place of the table is reserved whether or not this camera admits one) */
    p.stack_off = global_tables ? 0 : s->base.image_quads + primary_quads(s);
    p.cull = s->cull_opt;
    /* Wavefront tile shape (speed only).  4 x 16 (x by z) makes every lane-row's
     * stores whole 64-byte sectors (16 pixels x 12 B = 192 B, aligned): measured
     * WRITE_SIZE = 1.08 x the framebuffer bytes vs 1.27 x for 16 x 4.  On the
     * sphere-grid scenes the wider 16 x 4 tile diverges less and is 5-7 % faster,
     * while on small scenes the two run alike; hence the default. */
    const int tile_z_log2 = s->tile_z_log2 >= 0 ? s->tile_z_log2 : (s->objects.size() <= 128 ? 4 : 2);
    p.tile_z_log2 = tile_z_log2;
    const int tile_z = 1 << tile_z_log2, tile_x = 64 >> tile_z_log2;
    const long long tiles_z = ((long long)H + tile_z - 1) / tile_z;
    const long long tiles_x = ((long long)(x1 - x0) + tile_x - 1) / tile_x;
    const long long n_tiles = tiles_z * tiles_x;
    if (n_tiles > 0x7fffffffLL) return fail(RT_ERR_INVALID, "too many tiles");
    p.tiles_z = (int)tiles_z;
    p.tiles_x = (int)tiles_x;
    {
        const long long macro_rows = (tiles_z + RT_MACRO_ROWS - 1) / RT_MACRO_ROWS;
        /* Automatic order for scenes with a horizon and clustered sphere runs: start a little ABOVE the
         * horizon row and go DOWN -- the horizon rows are the most expensive of the frame, the rows
         * below them (the ground, with the spheres on it) get cheaper towards the bottom, and the rows
         * above the horizon, which come last after the wrap-around, are the cheapest: the queues then
         * hand out tiles roughly in order of decreasing cost, which keeps the tail of a frame -- or of
         * a GPU's strip of it -- short.  "first_row" given: from there upwards, as before. */
        const int horizon = horizon_start(s, cam);                    /* thousandths of the image height; 8 below the horizon row; 0 = none */
        const bool automatic = s->first_row_permille < 0 && horizon > 0;
        const int permille = s->first_row_permille >= 0 ? s->first_row_permille : (automatic ? std::min(999, horizon + 8 + 30) : 0);
        p.rows_downwards = automatic ? 1 : 0;
        p.first_macro_row = (int)std::min(macro_rows - 1, macro_rows * (long long)permille / 1000);
        if (p.first_macro_row < 0) p.first_macro_row = 0;
    }
    p.n_tiles = (int)n_tiles;
    const int waves_per_block = block / 64;
    const long long blocks_all = (n_tiles + waves_per_block - 1) / waves_per_block;

    s->launch.block_threads = block;
    s->launch.lds_bytes = lds_bytes;
    s->launch.scene_lds_bytes = global_tables ? 0 : (s->base.image_quads + primary_quads(s)) * 16;
    s->launch.tile_x = tile_x;
    s->launch.tile_z = tile_z;
    s->launch.grid_blocks = 0;
    if (blocks_all == 0) return RT_OK;

    HIP_TRY(hipSetDevice(s->device));
    rc = ensure_events(s);
    if (rc) return rc;
    /* persistent grid: as many workgroups as the chip holds at once (by the
     * occupancy query; a larger grid would also be correct, its surplus
     * workgroups simply find the queue empty), never more than there are tiles */
    if (!s->d_counters) {
        HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_counters),
                          (size_t)kEventRing * kCounterWords * sizeof(unsigned int)));
        /* zeroed once; from then on every launch zeroes the block the next launch of this scene will use (rt_kernel.hip) */
        HIP_TRY(hipMemset(s->d_counters, 0, (size_t)kEventRing * kCounterWords * sizeof(unsigned int)));
        HIP_TRY(hipDeviceSynchronize());          /* (the launches may be on streams that do not wait for the null stream) */
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, s->device));
        s->n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    /* the kernel: FAST tables, item tables, the one for clustered scenes (in the register budget that fits the
     * occupancy LDS allows), or the large-scene one */
    /* HELP (rt_kernel.hip): the clustered-scene kernels keep a desk of a few LDS words behind tables and stack */
    const bool clusters_kernel = !d_stats && !global_tables && s->n_clusters > 0 && s->pairs_opt;
    p.desk_off = 0;
    p.help_rays_quads = 0;
    p.help_leaves = s->help_opt >= 2 ? s->help_opt : RT_HELP_LEAVES;
    /* automatic: for launches of at most three quarters of the image's width.  The owners' look at the desk before every long
     * shadow scan costs a whole frame 1-1.4 % (grid-32 4.55 -> 4.49 ms, grid-16 d8 4.22 -> 4.17 without it), and a whole frame
     * has tiles enough to end well without help; a strip does not (longest of 2 / 4 strips of the grid-32 frame: 2.79 / 1.53 ms
     * with help, 3.10 / 2.28 without).  profiles/r03_experiments.txt 16 */
    const bool help_wanted = s->help_opt > 0 || (s->help_opt < 0 && (long long)(x1 - x0) * 4 <= (long long)W * 3);
    if (clusters_kernel && help_wanted && block > 64) {
        const int desk_off = p.stack_off + stack_lds_levels * block;
        const int with_desk = (desk_off + (RT_DESK_WORDS * 4 + 15) / 16) * 16;
        if ((size_t)with_desk <= RT_MAX_LDS_BYTES) {
            p.desk_off = desk_off;
            p.help_rays_quads = 128;
            lds_bytes = with_desk;
            s->launch.lds_bytes = lds_bytes;
        }
    }
    /* HEAVY tiles (rt_kernel.hip, render_body): with HELP on, the band of tile rows along the horizon line */
    p.heavy_half = -1;
    p.heavy_row0_q16 = p.heavy_slope_q16 = 0;
    p.help_spin_limit = s->help_spin_opt;
    p.timeline = 0;
    if (s->timeline_opt) {
        const size_t words = (size_t)n_tiles * RT_TIMELINE_WORDS;
        if (words > s->timeline_words) {
            HIP_TRY(hipSetDevice(s->device));
            HIP_TRY(hipDeviceSynchronize());
            if (s->d_timeline) { HIP_TRY(hipFree(s->d_timeline)); s->d_timeline = nullptr; s->timeline_words = 0; }
            HIP_TRY(hipMalloc(reinterpret_cast<void **>(&s->d_timeline), words * sizeof(unsigned long long)));
            s->timeline_words = words;
        }
        HIP_TRY(hipMemsetAsync(s->d_timeline, 0, words * sizeof(unsigned long long), stream));
        s->timeline_valid = words;
        p.timeline = (uint64_t)(uintptr_t)s->d_timeline;
    }
    p.error_word = (uint64_t)(uintptr_t)s->h_error;
    /* automatic: only when the launch renders a strip of at most a third of the image's width -- one GPU's share on three
     * or more.  There the strip waits for its horizon tiles (4096^2, 1 024-sphere grid, longest of 8 strips: 1.39 -> 1.05 ms
     * with the band, 4 strips 1.68 -> 1.63 ms); a whole frame has enough other tiles to run beside them, and giving three
     * of a workgroup's four wavefronts to one tile only costs it throughput (4.83 -> 5.02 ms; with a band of 0.9 % of the
     * height 5.32 ms).  profiles/r03_experiments.txt */
    /* (automatic: for strips of up to three fifths of the width -- halves re-cut by cost included.  Longest of 2 strips with /
     * without: grid-32 2.48 / 2.66 ms, grid-16 d8 2.36 / 2.44, built-in 0.410 / 0.428; of 4: 1.50 / 1.52, 1.38 / 1.44, 0.274 /
     * 0.323; whole frames lose 0.3-0.8 % to it) */
    p.tile_prio = s->tile_prio_opt >= 0 ? s->tile_prio_opt : ((long long)(x1 - x0) * 5 <= (long long)W * 3 ? 1 : 0);
    const bool heavy_wanted = s->heavy_opt > 0 || (s->heavy_opt < 0 && (long long)(x1 - x0) * 3 <= (long long)W);
    if (p.help_rays_quads != 0 && heavy_wanted) {
        double dz_centre = 0.0;
        if (const rt_object_desc *plane = horizon_plane(s, cam, &dz_centre)) {
            /* tile row (as a real number) of the line at the centre of tile column c: linear in c */
            auto row_at = [&](double c, double *row) {
                double dz;
                if (!horizon_dz(*plane, cam, ((double)x0 + (c + 0.5) * tile_x) / (double)W, &dz)) return false;
                *row = dz * (double)H / (double)tile_z;
                return true;
            };
            double r0, r1;
            const double c1 = (double)std::max<long long>(tiles_x - 1, 1);
            if (row_at(0.0, &r0) && row_at(c1, &r1) && std::fabs(r0) < 30000.0 && std::fabs(r1) < 30000.0) {
                p.heavy_half = s->heavy_opt > 0 ? s->heavy_opt - 1
                                                : (int)std::ceil((double)RT_HEAVY_PERMILLE10 * 1e-4 * (double)H / (double)tile_z);
                p.heavy_row0_q16 = (int)std::floor(r0 * 65536.0);
                p.heavy_slope_q16 = (int)std::lround((r1 - r0) / c1 * 65536.0);
            }
        }
    }
    /* LEARNED START ROW (rt_learn_tile_order): the queues start a little before the macro row that held the longest tile of the
     * counting frame -- outside the HEAVY band, whose tiles have their own queue -- and sweep towards the side where most of the
     * frame's cost lies (rows in image order: long and short tiles stay interleaved on the SIMDs; rows sorted by cost measured
     * slower, profiles/r03_experiments.txt 23) */
    {
        const int key[6] = {W, H, x0, x1, max_depth, tile_z_log2};
        const long long macro_rows = (tiles_z + RT_MACRO_ROWS - 1) / RT_MACRO_ROWS;
        if (!d_stats && s->first_row_permille < 0 && s->learned_sweep >= 0 && s->row_peak.size() == (size_t)macro_rows &&
            std::equal(key, key + 6, s->order_key)) {
            long long best = -1;
            double weighted = 0.0, total = 0.0;
            for (long long m = 0; m < macro_rows; ++m) {
                bool in_band = false;
                if (p.heavy_half >= 0) {
                    const int mid_col = (int)(tiles_x / 2);
                    const int line = (p.heavy_row0_q16 + mid_col * p.heavy_slope_q16) >> 16;
                    const long long lo = (line - p.heavy_half - 1) / RT_MACRO_ROWS, hi = (line + p.heavy_half + 1) / RT_MACRO_ROWS;
                    in_band = m >= lo && m <= hi;
                }
                weighted += s->row_sum[(size_t)m] * (double)m;
                total += s->row_sum[(size_t)m];
                if (!in_band && (best < 0 || s->row_peak[(size_t)m] > s->row_peak[(size_t)best])) best = m;
            }
            if (best >= 0 && total > 0.0) {
                (void)weighted;
                const bool upwards = s->learned_sweep == 0;
                p.rows_downwards = upwards ? 0 : 1;
                const long long margin = std::max<long long>(1, macro_rows / 64);
                p.first_macro_row = (int)std::min(macro_rows - 1, std::max<long long>(0, upwards ? best - margin : best + margin));
            }
        }
    }
    /* the 96-register kernel when LDS leaves room for fewer than six wavefronts per SIMD anyway (24 per CU) */
    const bool clusters_wide = s->wide_opt >= 0 ? s->wide_opt != 0
                                                : (RT_MAX_LDS_BYTES / (size_t)lds_bytes) * (size_t)(block / 64) < 24;
    const bool fast_tables = s->base.n_fast_items > 0;
    struct Kernel { const void *fn; const char *name; };
#define RT_KERNEL(k) Kernel{(const void *)k, #k}
    const Kernel first_kernel = d_stats ? (fast_tables ? RT_KERNEL(rt_render_kernel_fast_stats) : RT_KERNEL(rt_render_kernel_stats))
                                : global_tables ? RT_KERNEL(rt_render_kernel_large)
                                : (s->n_clusters > 0 && s->pairs_opt)
                                      ? (clusters_wide ? RT_KERNEL(rt_render_kernel_clusters_wide) : RT_KERNEL(rt_render_kernel_clusters))
                                : fast_tables ? RT_KERNEL(rt_render_kernel)
                                : RT_KERNEL(rt_render_kernel_items);
    const void *first = first_kernel.fn;
    std::snprintf(s->launch.kernel, sizeof(s->launch.kernel), "%s", first_kernel.name);
    {
        /* a workgroup larger than the kernel was compiled for (__launch_bounds__) must never be launched */
        hipFuncAttributes attr;
        HIP_TRY(hipFuncGetAttributes(&attr, first));
        if (block > attr.maxThreadsPerBlock)
            return fail(RT_ERR_INVALID, "block_threads " + std::to_string(block) + " exceeds the launch bounds of " +
                                            first_kernel.name + " (" + std::to_string(attr.maxThreadsPerBlock) + ")");
    }
    int per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, first, block, (size_t)lds_bytes));
    if (per_cu < 1) per_cu = 1;
    const long long blocks = s->grid_mult > 0
        ? std::min(blocks_all, (long long)per_cu * (long long)s->n_cus * (long long)s->grid_mult)
        : blocks_all;
    s->launch.grid_blocks = (int)blocks;
    HIP_TRY(hipFuncSetAttribute(first, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    /* bounce stack: one slice per workgroup of the persistent grid */
    {
        const double need_d = stack_lds_levels >= max_depth ? 16.0
                                           : (double)blocks * (double)block * (double)(max_depth + 1) * RT_STACK_ENTRY_BYTES;
        if (need_d > 8.0e9)
            return fail(RT_ERR_CAPACITY, "max_depth too large: the bounce stack would exceed 8 GB of HBM");
        const size_t need = (size_t)need_d;
        /* the stack (and nothing else) is shared by successive launches of this handle:
         * launches must be stream-ordered, so fence when the caller switches streams */
        if (s->has_last_stream && s->last_stream != stream) HIP_TRY(hipStreamSynchronize(s->last_stream));
        s->last_stream = stream;
        s->has_last_stream = true;
        if (need > s->d_stack_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (s->d_stack) { HIP_TRY(hipFree(s->d_stack)); s->d_stack = nullptr; s->d_stack_bytes = 0; }
            HIP_TRY(hipMalloc(&s->d_stack, need));
            s->d_stack_bytes = need;
        }
    }
    if (p.help_rays_quads != 0) {
        const size_t need = (size_t)blocks * (size_t)p.help_rays_quads * 16;
        if (need > s->d_help_bytes) {
            HIP_TRY(hipDeviceSynchronize());
            if (s->d_help) { HIP_TRY(hipFree(s->d_help)); s->d_help = nullptr; s->d_help_bytes = 0; }
            HIP_TRY(hipMalloc(&s->d_help, need));
            s->d_help_bytes = need;
        }
    }
    /* This launch's block of counters is at zero (the invariant: the block of slot ev_next is, whenever a launch of this scene
     * starts -- all of them at allocation, and every launch zeroes the next slot's while it runs; launches of one scene are
     * stream-ordered, above).  The next slot's old launch, 63 launches ago, must be over before this one writes its block. */
    const int slot = s->ev_next, next_slot = (slot + 1) % kEventRing;
    rc = drain_event(s, slot);            /* ring wrapped: account for the old launch first */
    if (rc == RT_OK) rc = drain_event(s, next_slot);
    if (rc) return rc;
    unsigned int *counter = s->d_counters + (size_t)slot * kCounterWords;
    p.next_counters = (uint64_t)(uintptr_t)(s->d_counters + (size_t)next_slot * kCounterWords);
    HIP_TRY(hipEventRecord(s->ev[slot].start, stream));
    const float4 *image_arg = reinterpret_cast<const float4 *>(s->d_image);
    float4 *stack_arg = reinterpret_cast<float4 *>(s->d_stack);
    unsigned int *list_arg = reinterpret_cast<unsigned int *>(s->d_help);      /* the clustered-scene kernels' HELP areas (unused by the others) */
    {
        void *args6[] = {&p, &image_arg, &d_out, &counter, &stack_arg, &list_arg};
        void *args7[] = {&p, &image_arg, &d_out, &counter, &stack_arg, &d_stats, &list_arg};
        HIP_TRY(hipLaunchKernel(first, dim3((unsigned)blocks), dim3((unsigned)block), d_stats ? args7 : args6, (size_t)lds_bytes, stream));
    }
    HIP_TRY(hipGetLastError());
    s->ev_next = next_slot;               /* (only now: a launch that did not happen has zeroed nothing) */
    HIP_TRY(hipEventRecord(s->ev[slot].stop, stream));
    s->ev[slot].pending = true;
    return RT_OK;
}

} // namespace

extern "C" {

/* used by rt_multi.hip to report through rt_last_error() */
int rt_internal_set_error(int code, const char *msg) { return fail(code, msg ? msg : ""); }

int rt_capi_version(void) { return RT_CAPI_VERSION; }

int rt_capi_tuning_version(void) { return RT_CAPI_TUNING_VERSION; }

const char *rt_last_error(void) { return g_last_error.c_str(); }

int rt_device_count(int *count) {
    if (!count) return fail(RT_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(RT_ERR_NO_DEVICE, hipGetErrorString(e)); }
    *count = n;
    return RT_OK;
}

int rt_scene_create(const rt_scene_desc *desc, int device, rt_scene **out) {
    if (!desc || !out) return fail(RT_ERR_INVALID, "desc/out is NULL");
    *out = nullptr;
    rt_scene *s = new (std::nothrow) rt_scene();
    if (!s) return fail(RT_ERR_INVALID, "out of memory");
    int rc = adopt_desc(desc, s);
    if (rc == RT_OK) rc = pack_scene(s);
    if (rc) { delete s; return rc; }
    s->device = device;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        delete s;
        return fail(RT_ERR_NO_DEVICE, "no HIP device (this library has no CPU path)");
    }
    if (device < 0 || device >= ndev) { delete s; return fail(RT_ERR_INVALID, "device index out of range"); }
    rc = upload_scene(s);
    if (rc) { rt_scene_destroy(s); return rc; }
    {
        /* one word of pinned host memory for what a kernel has to tell the host (a HELP wait that timed out) */
        hipError_t e = hipHostMalloc(reinterpret_cast<void **>(&s->h_error), sizeof(unsigned int), hipHostMallocDefault);
        if (e != hipSuccess) { rt_scene_destroy(s); return fail(RT_ERR_HIP, std::string("hipHostMalloc: ") + hipGetErrorString(e)); }
        *s->h_error = 0u;
    }
    *out = s;
    return RT_OK;
}

int rt_scene_destroy(rt_scene *s) {
    if (!s) return RT_OK;
    if (s->d_image || s->d_fb || s->d_counters || s->ev_ready) (void)hipSetDevice(s->device);
    if (s->ev_ready)
        for (int i = 0; i < kEventRing; ++i) { (void)hipEventDestroy(s->ev[i].start); (void)hipEventDestroy(s->ev[i].stop); }
    if (s->d_image) (void)hipFree(s->d_image);
    if (s->d_fb) (void)hipFree(s->d_fb);
    if (s->d_counters) (void)hipFree(s->d_counters);
    if (s->d_help) (void)hipFree(s->d_help);
    if (s->d_stack) (void)hipFree(s->d_stack);
    if (s->h_error) (void)hipHostFree(s->h_error);
    if (s->d_timeline) (void)hipFree(s->d_timeline);
    delete s;
    return RT_OK;
}

int rt_render_device(rt_scene *s, const rt_camera_desc *cam, int W, int H, int x0, int x1, int max_depth,
                     void *d_out_rgb, void *hip_stream) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    return launch(s, cam, W, H, x0, x1, max_depth, static_cast<float *>(d_out_rgb),
                  static_cast<hipStream_t>(hip_stream));
}

int rt_render(rt_scene *s, const rt_camera_desc *cam, int W, int H, int x0, int x1, int max_depth,
              float *out_rgb) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    if (W <= 0 || H <= 0 || x0 < 0 || x1 > W || x0 > x1) return fail(RT_ERR_INVALID, "need 0 <= x0 <= x1 <= W, W,H > 0");
    const size_t bytes = (size_t)(x1 - x0) * (size_t)H * 3 * sizeof(float);
    if (bytes && !out_rgb) return fail(RT_ERR_INVALID, "out_rgb is NULL");
    HIP_TRY(hipSetDevice(s->device));
    if (bytes > s->d_fb_bytes) {
        if (s->d_fb) { HIP_TRY(hipFree(s->d_fb)); s->d_fb = nullptr; s->d_fb_bytes = 0; }
        HIP_TRY(hipMalloc(&s->d_fb, bytes));
        s->d_fb_bytes = bytes;
    }
    int rc = launch(s, cam, W, H, x0, x1, max_depth, static_cast<float *>(s->d_fb), nullptr);
    if (rc) return rc;
    s->timing.last_download_ms = 0.0;
    if (bytes) {
        hipEvent_t t0, t1;
        HIP_TRY(hipEventCreate(&t0));
        HIP_TRY(hipEventCreate(&t1));
        HIP_TRY(hipEventRecord(t0, nullptr));
        HIP_TRY(hipMemcpy(out_rgb, s->d_fb, bytes, hipMemcpyDeviceToHost));
        HIP_TRY(hipEventRecord(t1, nullptr));
        HIP_TRY(hipEventSynchronize(t1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
        s->timing.last_download_ms = ms;
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
    }
    HIP_TRY(hipDeviceSynchronize());
    return device_report(s);
}

int rt_render_stats(rt_scene *s, const rt_camera_desc *cam, int W, int H, int x0, int x1, int max_depth,
                    float *out_rgb, uint64_t *stats, int n_stats, uint64_t *wave_cycles, int n_wave_cycles) {
    if (!s || !stats || n_stats < 0) return fail(RT_ERR_INVALID, "scene/stats is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    if (W <= 0 || H <= 0 || x0 < 0 || x1 > W || x0 > x1) return fail(RT_ERR_INVALID, "need 0 <= x0 <= x1 <= W, W,H > 0");
    const size_t bytes = (size_t)(x1 - x0) * (size_t)H * 3 * sizeof(float);
    HIP_TRY(hipSetDevice(s->device));
    if (bytes > s->d_fb_bytes) {
        if (s->d_fb) { HIP_TRY(hipFree(s->d_fb)); s->d_fb = nullptr; s->d_fb_bytes = 0; }
        HIP_TRY(hipMalloc(&s->d_fb, bytes));
        s->d_fb_bytes = bytes;
    }
    /* counters, then one cycle count per wavefront tile */
    const int tzl = s->tile_z_log2 >= 0 ? s->tile_z_log2 : (s->objects.size() <= 128 ? 4 : 2);
    const int tile_z = 1 << tzl, tile_x = 64 >> tzl;
    const size_t n_tiles = (size_t)((H + tile_z - 1) / tile_z) * (size_t)((x1 - x0 + tile_x - 1) / tile_x);
    const size_t words = RT_STATS_COUNT + n_tiles * RT_TILE_STATS;
    unsigned long long *d_stats = nullptr;
    HIP_TRY(hipMalloc(&d_stats, words * sizeof(unsigned long long)));
    hipError_t e = hipMemset(d_stats, 0, words * sizeof(unsigned long long));
    int rc = e == hipSuccess ? launch(s, cam, W, H, x0, x1, max_depth, static_cast<float *>(s->d_fb), nullptr, d_stats)
                             : fail(RT_ERR_HIP, hipGetErrorString(e));
    unsigned long long host[RT_STATS_COUNT] = {0};
    if (rc == RT_OK) {
        e = hipMemcpy(host, d_stats, sizeof(host), hipMemcpyDeviceToHost);
        if (e == hipSuccess && out_rgb && bytes) e = hipMemcpy(out_rgb, s->d_fb, bytes, hipMemcpyDeviceToHost);
        if (e == hipSuccess && wave_cycles && n_wave_cycles > 0)
            e = hipMemcpy(wave_cycles, d_stats + RT_STATS_COUNT,
                          std::min((size_t)n_wave_cycles, n_tiles * RT_TILE_STATS) * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(RT_ERR_HIP, hipGetErrorString(e));
    }
    (void)hipFree(d_stats);
    if (rc) return rc;
    for (int k = 0; k < n_stats; ++k) stats[k] = k < RT_STATS_COUNT ? host[k] : 0;
    return RT_OK;
}

/* One frame of the counting build on this launch shape; per macro row its longest tile and its sum: later launches of the SAME
 * shape (W, H, x0, x1, max_depth, tile shape) start their queues at the row of the longest tile (launch(), LEARNED START ROW). */
int rt_learn_tile_order(rt_scene *s, const rt_camera_desc *cam, int W, int H, int x0, int x1, int max_depth) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    if (W <= 0 || H <= 0 || x0 < 0 || x1 > W || x0 >= x1) return fail(RT_ERR_INVALID, "need 0 <= x0 < x1 <= W, W,H > 0");
    int tzl;
    {
        std::lock_guard<std::mutex> lock(s->mu);
        s->row_peak.clear(); s->row_sum.clear();
        tzl = s->tile_z_log2 >= 0 ? s->tile_z_log2 : (s->objects.size() <= 128 ? 4 : 2);
    }
    const int tile_z = 1 << tzl, tile_x = 64 >> tzl;
    const size_t tiles_z = (size_t)((H + tile_z - 1) / tile_z), tiles_x = (size_t)((x1 - x0 + tile_x - 1) / tile_x);
    std::vector<uint64_t> stats(RT_STATS_COUNT), tiles(tiles_z * tiles_x * RT_TILE_STATS);
    int rc = rt_render_stats(s, cam, W, H, x0, x1, max_depth, nullptr, stats.data(), RT_STATS_COUNT, tiles.data(), (int)tiles.size());
    if (rc) return rc;
    const size_t macro_rows = (tiles_z + RT_MACRO_ROWS - 1) / RT_MACRO_ROWS;
    std::vector<double> peak(macro_rows, 0.0), sum(macro_rows, 0.0);
    for (size_t row = 0; row < tiles_z; ++row)
        for (size_t col = 0; col < tiles_x; ++col) {
            const double c = (double)tiles[(row * tiles_x + col) * RT_TILE_STATS];
            peak[row / RT_MACRO_ROWS] = std::max(peak[row / RT_MACRO_ROWS], c);
            sum[row / RT_MACRO_ROWS] += c;
        }
    std::lock_guard<std::mutex> lock(s->mu);
    const int key[6] = {W, H, x0, x1, max_depth, tzl};
    std::copy(key, key + 6, s->order_key);
    s->row_peak.swap(peak);
    s->row_sum.swap(sum);
    /* What is fastest for this shape is MEASURED (seven frames per candidate into the handle's own buffer, the first warms up,
     * the shortest of the others counts; a candidate replaces the best so far only if it beats it by 3 %): the rule's sweep, or
     * from the longest tile's row upwards or downwards.  (Trying the HEAVY band and the tile priorities the other way round per
     * shape as well gave nothing beyond the strip model's +-4 %: profiles/r03_experiments.txt 23-24) */
    HIP_TRY(hipSetDevice(s->device));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    {
        hipError_t e = hipEventCreate(&e0);
        if (e == hipSuccess) e = hipEventCreate(&e1);
        if (e != hipSuccess) {
            if (e0) (void)hipEventDestroy(e0);
            s->row_peak.clear(); s->row_sum.clear();
            return fail(RT_ERR_HIP, std::string("hipEventCreate: ") + hipGetErrorString(e));
        }
    }
    rc = RT_OK;
    auto frame_ms = [&]() {
        float shortest = 1e30f;
        for (int rep = 0; rep < 7 && rc == RT_OK; ++rep) {
            hipError_t e = hipEventRecord(e0, nullptr);
            if (e == hipSuccess) rc = launch(s, cam, W, H, x0, x1, max_depth, static_cast<float *>(s->d_fb), nullptr);
            if (rc != RT_OK) break;
            e = hipEventRecord(e1, nullptr);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            float ms = 0.0f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
            if (e != hipSuccess) { rc = fail(RT_ERR_HIP, hipGetErrorString(e)); break; }
            if (rep > 0) shortest = std::min(shortest, ms);
        }
        return shortest;
    };
    s->learned_sweep = -1;
    float best_ms = frame_ms();
    for (int sweep = 0; sweep <= 1 && rc == RT_OK; ++sweep) {
        const int keep = s->learned_sweep;
        s->learned_sweep = sweep;
        const float ms = frame_ms();
        if (rc == RT_OK && ms < 0.97f * best_ms) best_ms = ms; else s->learned_sweep = keep;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != RT_OK) { s->learned_sweep = -1; s->row_peak.clear(); s->row_sum.clear(); }
    return rc;
}

int rt_get_timing(const rt_scene *cs, rt_timing *out) {
    if (!cs || !out) return fail(RT_ERR_INVALID, "scene/out is NULL");
    rt_scene *s = const_cast<rt_scene *>(cs);
    std::lock_guard<std::mutex> lock(s->mu);
    if (s->ev_ready) {
        HIP_TRY(hipSetDevice(s->device));
        /* drain in launch order so last_kernel_ms is the newest launch */
        for (int k = 0; k < kEventRing; ++k) {
            int rc = drain_event(s, (s->ev_next + k) % kEventRing);
            if (rc) return rc;
        }
    }
    *out = s->timing;
    return device_report(s);
}

int rt_reset_timing(rt_scene *s) {
    if (!s) return fail(RT_ERR_INVALID, "scene is NULL");
    rt_timing t;
    int rc = rt_get_timing(s, &t);        /* drains pending events */
    if (rc) return rc;
    std::lock_guard<std::mutex> lock(s->mu);
    s->timing.sum_kernel_ms = 0.0;
    s->timing.launches = 0;
    return RT_OK;
}

int rt_get_launch_info(const rt_scene *s, rt_launch_info *out) {
    if (!s || !out) return fail(RT_ERR_INVALID, "scene/out is NULL");
    *out = s->launch;
    return RT_OK;
}

int rt_get_timeline(rt_scene *s, uint64_t *out, int n_words) {
    if (!s || !out || n_words < 0) return fail(RT_ERR_INVALID, "scene/out is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!s->d_timeline || s->timeline_valid == 0) return fail(RT_ERR_INVALID, "no timeline recorded: set option \"timeline\" to 1 and render");
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, s->d_timeline, std::min((size_t)n_words, s->timeline_valid) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_set_option(rt_scene *s, const char *key, int value) {
    if (!s || !key) return fail(RT_ERR_INVALID, "scene/key is NULL");
    std::lock_guard<std::mutex> lock(s->mu);
    if (!std::strcmp(key, "tile_z")) {
        int lg = -1;
        for (int i = 0; i <= 6; ++i) if ((1 << i) == value) lg = i;
        if (lg < 0) return fail(RT_ERR_INVALID, "tile_z must be a power of two in [1, 64]");
        s->tile_z_log2 = lg;
        return RT_OK;
    }
    if (!std::strcmp(key, "block_threads")) {
        if (value != 0 && (value < 64 || value > 1024 || (value % 64) != 0))       /* launch() refuses what exceeds the chosen kernel's launch bounds */
            return fail(RT_ERR_INVALID, "block_threads must be 0 (auto) or a multiple of 64 up to 1024 (a launch refuses more than its kernel's launch bounds)");
        s->block_threads_opt = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "stack")) {
        if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "stack must be 0 (auto), 1 (LDS) or 2 (HBM)");
        s->stack_opt = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "first_row")) {
        if (value < -1 || value > 999) return fail(RT_ERR_INVALID, "first_row is in thousandths of the image height, [0, 999], or -1 (automatic)");
        s->first_row_permille = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "help")) {
        if (value < -1 || value > 64) return fail(RT_ERR_INVALID, "help must be -1 (automatic), 0 (off), 1 (on) or a number of candidate leaves, [2, 64]");
        s->help_opt = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "timeline")) {
#ifdef RT_TIMELINE
        s->timeline_opt = value != 0;
        return RT_OK;
#else
        return value == 0 ? (int)RT_OK
                          : fail(RT_ERR_INVALID, "the timeline is recorded by diagnostic builds only: make -C tilecoderaytracer_amd/csrc variant NAME=timeline DEFS=-DRT_TIMELINE=1");
#endif
    }
    if (!std::strcmp(key, "primary")) { s->primary_opt = value != 0; return RT_OK; }
    if (!std::strcmp(key, "heavy")) {
        if (value < -1 || value > 4096) return fail(RT_ERR_INVALID, "heavy must be -1 (automatic), 0 (off) or 1 + the band's half-width in tile rows");
        s->heavy_opt = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "learned_order")) {
        if (value != 0) return fail(RT_ERR_INVALID, "learned_order accepts 0 only (forget the order of rt_learn_tile_order)");
        s->row_peak.clear(); s->row_sum.clear();
        s->learned_sweep = -1;
        return RT_OK;
    }
    if (!std::strcmp(key, "tile_prio")) {
        if (value < -1 || value > 1) return fail(RT_ERR_INVALID, "tile_prio must be -1 (automatic), 0 (off) or 1 (on)");
        s->tile_prio_opt = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "help_spin_limit")) {
        if (value < -1) return fail(RT_ERR_INVALID, "help_spin_limit must be >= -1");
        s->help_spin_opt = value;
        return RT_OK;
    }
    if (!std::strcmp(key, "wide")) { s->wide_opt = value < 0 ? -1 : (value != 0); return RT_OK; }
    if (!std::strcmp(key, "pairs")) {
        s->pairs_opt = value != 0;
        return RT_OK;
    }

    if (!std::strcmp(key, "grid_mult")) {
        if (value < 0 || value > 64) return fail(RT_ERR_INVALID, "grid_mult must be in [0, 64]");
        s->grid_mult = value;
        return RT_OK;
    }
    /* options that change the tables: pack_scene() commits only on success, so a value the
     * scene cannot be packed with leaves the handle exactly as it was */
    auto repack_with = [&](int &field, int v) {
        const int old = field;
        if (old == v) return (int)RT_OK;
        field = v;
        int rc = pack_scene(s);
        if (rc) { field = old; return rc; }
        rc = upload_scene(s);
        if (rc) {                                   /* device trouble: go back to the tables that were there */
            const std::string why = g_last_error;
            field = old;
            if (pack_scene(s) == RT_OK) (void)upload_scene(s);
            return fail(rc, why);
        }
        return (int)RT_OK;
    };
    if (!std::strcmp(key, "aa_planes")) return repack_with(s->aa_planes, value != 0);
    if (!std::strcmp(key, "tight_planes")) return repack_with(s->tight_planes, value != 0);
    if (!std::strcmp(key, "fast")) return repack_with(s->fast_opt, value != 0);
    if (!std::strcmp(key, "tables")) {             /* decides the table format too (FAST tables live in LDS) */
        if (value < 0 || value > 2) return fail(RT_ERR_INVALID, "tables must be 0 (auto), 1 (LDS) or 2 (global memory)");
        return repack_with(s->tables_opt, value);
    }
    if (!std::strcmp(key, "cull")) return repack_with(s->cull_opt, value != 0);
    if (!std::strcmp(key, "svox")) {
        if (value < -1 || value > (1 << 20)) return fail(RT_ERR_INVALID, "svox must be -1 (automatic), 0 (no SHADOW VOXELS) or a number of voxels up to 2^20");
        return repack_with(s->svox_opt, value);
    }
    if (!std::strcmp(key, "cluster_leaf")) {
        if (value < -1 || value > 255) return fail(RT_ERR_INVALID, "cluster_leaf must be in [0, 255], or -1 (automatic)");
        return repack_with(s->cluster_leaf, value);
    }
    return fail(RT_ERR_INVALID, std::string("unknown option: ") + key);
}

} // extern "C"
