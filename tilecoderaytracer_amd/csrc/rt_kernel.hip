/*
 * rt_kernel.hip -- the render kernel for gfx950 (MI355X).
 *
 * One wavefront lane per pixel.  Restates, for the GPU, the hot path of
 * ccelio/TileCodeRayTracer: raytrace_main's pixel loop
 * (src/RayTracer.cpp:904-923) -> Camera::createEyeRay (src/Camera.cpp:71-84)
 * -> calculatePixel (src/RayTracer.cpp:448-638) -> getCollision (:50-89),
 * inShade (:709-771), cosineShade (:654-701), the three collision() routines
 * and the CollisionObject constructor (src/SceneObject.h:47-105).
 *
 * What is different from the reference, and why it is still bit-identical:
 *  - The reference builds a full hit record for EVERY candidate object; only
 *    `distance` takes part in choosing the nearest hit / the shadow verdict.
 *    Here every candidate yields a distance only, and the record (point,
 *    normal, colour, reflected ray) is built once, for the winner, with the
 *    same operations in the same order.
 *  - The recursion `final_k = local_k + (rf_k * C_{k+1}) * oc_k`
 *    (src/RayTracer.cpp:601) is flattened: a forward loop over bounce levels
 *    pushes {local_k, object} on a per-lane stack (16 B per level), a backward loop
 *    combines inside-out, so the association order is the reference's.
 *  - The object list is walked as a table of ITEMS (rt_tables.h) in Scene
 *    index order; the item counter is wave-uniform and the records are read
 *    from LDS with the same address in every lane (a broadcast).  Before that,
 *    the wavefront culls the table cooperatively: lane i tests item i's box
 *    against a bound of all 64 rays, one ballot yields the candidates.
 *  - Scenes with clustered sphere runs: the (ray, leaf) pairs that the per-lane
 *    box tests leave are compacted into full wavefront rounds (PAIRS, NEAREST
 *    PAIRS); long shadow scans are shared with the workgroup's wavefronts that have
 *    nothing else to do (HELP), and the tiles known to be long are rendered one per
 *    workgroup from the start (HEAVY tiles).  The same tests on the same operands;
 *    nearest = minimum of (distance, Scene index), shadow = OR.
 *  - One body, several __global__ entry points (bottom of the file); the host
 *    picks by scene (rt_capi.hip, launch()).
 *
 * Arithmetic contract: IEEE-754 binary32, no FMA contraction
 * (-ffp-contract=off and the pragma below), correctly rounded '/' and sqrtf
 * (hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt), denormals kept,
 * exact fmodf.  Never build this file with -ffast-math.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_capi_tuning.h"      /* RT_STATS_COUNT: the counting build's counters are part of the (tuning) ABI */
#include "rt_tables.h"

#pragma clang fp contract(off)

/* RT_KIND_SPHERE / _INFINITE_PLANE / _FINITE_PLANE: include/rt_capi.h */

namespace {

struct V3 { float x, y, z; };

__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 xyz(const float4 q) { return mk(q.x, q.y, q.z); }
/* vector3d operators, src/vector3d.h:97-124 -- association order is part of the contract */
__device__ __forceinline__ float dot3(const V3 a, const V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 sub3(const V3 a, const V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 add3(const V3 a, const V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 scale3(const V3 v, const float f) { return mk(v.x * f, v.y * f, v.z * f); }
/* sqrtf(x) for 2^-96 <= x <= 2^120 (and for a NaN or negative x, whose result -- NaN -- nobody uses): the compiler's own
 * expansion of the correctly rounded square root -- v_sqrt_f32 (1 ulp), then the two neighbours tried with an exact fma
 * residual each -- without the steps that scale a tiny x up and the result back down and pass 0 and infinity through
 * (seven of its sixteen instructions).  Same instructions on the same operands as sqrtf() in that range, hence the same bits. */
__device__ __forceinline__ float sqrt_in_range(const float x) {
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float below = __int_as_float(__float_as_int(s0) - 1), above = __int_as_float(__float_as_int(s0) + 1);
    const float r_below = __builtin_fmaf(-below, s0, x);
    const float r_above = __builtin_fmaf(-above, s0, x);
    const float s1 = (0.0f >= r_below) ? below : s0;
    return (0.0f < r_above) ? above : s1;
}
__device__ __forceinline__ bool sqrt_range_ok(const float x) { return (x >= 0x1p-96f) && (x <= 0x1p+120f); }

/* n / l from r = the refined reciprocal of l: the tail of the compiler's own expansion of a correctly rounded divide
 * (normalize3() below says when it may be used) */
__device__ __forceinline__ float quotient_by_refined_reciprocal(const float n, const float l, const float r) {
    const float q0 = n * r;
    const float q1 = __builtin_fmaf(__builtin_fmaf(-l, q0, n), r, q0);
    return __builtin_fmaf(__builtin_fmaf(-l, q1, n), r, q1);
}

/* vector3d::normalize, src/vector3d.h:55-73: sqrtf then three true (correctly rounded) divides by the same length.
 *
 * The compiler expands every IEEE divide n / l into v_div_scale (x2), v_rcp, two fma that refine the reciprocal,
 * q0 = n r, two residual/correction fma pairs (the last as v_div_fmas) and v_div_fixup: twelve instructions, thirty-six
 * per normalisation -- a quarter of them, the refined reciprocal of l, three times over.  When no scaling is needed the
 * scale and fixup steps are the identity (v_div_scale leaves its operand alone unless an operand is zero or denormal,
 * the exponents differ by 96 or more, the quotient or 1/l would be denormal, or |n| < 2^-103; v_div_fixup passes the
 * quotient through unless an operand is zero, infinite or NaN), so the SAME fma sequence with the reciprocal refined
 * once gives the same bits: eighteen instructions.  The guard -- 2^-60 <= every |component| and 2^-96 <= the squared
 * length <= 2^120, in every active lane -- puts all of those cases out of reach (|n| <= l, so the quotient is in
 * [2^-120, 1]) and lets the square root take its short form too (sqrt_in_range()); a wavefront with a lane outside it (a
 * zero component: rays along an axis) takes the plain square root and divides.  *length_out = the length, which callers
 * that need it (the distance to the light) would otherwise compute a second time. */
__device__ __forceinline__ V3 normalize3(const V3 v, float *length_out = nullptr) {
    const float squared = v.x * v.x + v.y * v.y + v.z * v.z;
    const float smallest = fminf(fminf(fabsf(v.x), fabsf(v.y)), fabsf(v.z));
    /* in range for the short square root and the short divides (then the length is in [2^-48, 2^60]); a NaN anywhere: plain */
    const bool plain = !(smallest >= 0x1p-60f) || !sqrt_range_ok(squared);
    float length, x, y, z;
    if (__builtin_amdgcn_ballot_w64(plain) != 0ull) {
        length = sqrtf(squared);
        x = v.x / length; y = v.y / length; z = v.z / length;
    } else {
        length = sqrt_in_range(squared);
        const float r0 = __builtin_amdgcn_rcpf(length);
        const float r = __builtin_fmaf(__builtin_fmaf(-length, r0, 1.0f), r0, r0);
        x = quotient_by_refined_reciprocal(v.x, length, r);
        y = quotient_by_refined_reciprocal(v.y, length, r);
        z = quotient_by_refined_reciprocal(v.z, length, r);
    }
    if (length_out) *length_out = length;
    return mk(x, y, z);
}

/* vector3d::normalize of a vector that is usually already of unit length (the
 * reference re-normalises normals it has just normalised, src/SceneObject.h:62,
 * src/RayTracer.cpp:566-567).  When x*x + y*y + z*z rounds to exactly 1.0f the
 * reference computes sqrtf(1.0f) = 1.0f and v / 1.0f = v, so if that holds in
 * every active lane the wavefront skips the square root and the divides; the
 * result is the reference's either way. */
__device__ __forceinline__ V3 renormalize3(const V3 v);

/* Slack of an item's box in the nearest-hit culls, per axis, from the distances `far_k` between the origin box
 * and the item box's far side on axis k.  Sphere-like items: RT_SPHERE_SLACK of the L1 distance on every axis
 * (box_needed()).  Plane items (RT_ITEM_TIGHT): RT_PLANE_SLACK of the distance on THAT axis -- component k of the
 * hit point is t d_k + o_k, rounded twice -- which is also what lets an item be unbounded on some axes (an
 * axis-aligned infinite plane is a slab: far_k = inf there gives an infinite slack, i.e. no constraint). */
#define RT_CULL_SLACK(bits, fx, fy, fz, ex, ey, ez)                                                  \
    do {                                                                                             \
        const bool tight_ = ((bits) & RT_ITEM_TIGHT) != 0u;                                          \
        const float sum_ = ((fx) + (fy)) + (fz);                                                     \
        const float k_ = tight_ ? RT_PLANE_SLACK : RT_SPHERE_SLACK;                                  \
        ex = k_ * (tight_ ? (fx) : sum_) + 1.0e-4f;                                                  \
        ey = k_ * (tight_ ? (fy) : sum_) + 1.0e-4f;                                                  \
        ez = k_ * (tight_ ? (fz) : sum_) + 1.0e-4f;                                                  \
    } while (0)

/* Wave-level "does any active lane need this": a uniform (scalar) branch, so
 * the guarded block costs no exec-mask bookkeeping; lanes that do not need it
 * run it anyway and discard the result. */
__device__ __forceinline__ bool wave_any(const bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }

__device__ __forceinline__ V3 renormalize3(const V3 v) {
    const float s = v.x * v.x + v.y * v.y + v.z * v.z;
    if (!wave_any(s != 1.0f)) return v;
    const float length = sqrtf(s);
    return mk(v.x / length, v.y / length, v.z / length);
}

/* Work counters of the diagnostic ("counting") build, rt_render_stats().  In
 * the production kernel kStats is false and every use folds away. */
enum {
    ST_NEAREST_RAYS = 0,    /* lanes: nearest-hit rays traced                              */
    ST_SHADOW_RAYS,         /* lanes: shadow rays traced                                   */
    ST_WAVE_NEAREST,        /* wavefronts: nearest_hit() calls                             */
    ST_WAVE_SHADOW,         /* wavefronts: in_shade() calls                                */
    ST_WAVE_SPHERE_TESTS,   /* wavefronts: sphere tests issued                             */
    ST_WAVE_PLANE_TESTS,    /* wavefronts: plane tests issued                              */
    ST_WAVE_BOX_TESTS,      /* wavefronts: cluster box tests issued                        */
    ST_LANE_SPHERE_TESTS,   /* lanes: sphere tests the lane itself needed                  */
    ST_CYCLES_NEAREST,      /* shader cycles wavefronts spent inside nearest-hit scans     */
    ST_CYCLES_SHADOW,       /* ... inside shadow scans                                     */
    ST_CYCLES_TILE,         /* ... on whole tiles (camera ray to framebuffer store)        */
    ST_CYCLES_WINNER,       /* ... on the winner's CollisionObject (point, normal, texture) */
    ST_CYCLES_LIGHTS,       /* ... in the light loop, shadow scans included                 */
    ST_CYCLES_REFLECT,      /* ... reflecting (phase 3)                                     */
    ST_SHADOW_CANDIDATES,   /* wavefronts: items left by the shadow scans' bundle cull      */
    ST_SHADOW_LEAVES_UNION, /* wavefronts: leaves some lane's segment needed                */
    ST_SHADOW_LEAVES_MAXLANE, /* wavefronts: per scan, the most leaves one lane needed      */
    ST_NEAREST_1_16,        /* wavefronts: nearest-hit scans with 1-16 lanes tracing a ray, ... */
    ST_NEAREST_17_32,
    ST_NEAREST_33_48,
    ST_NEAREST_49_64,       /* ... with 49-64 */
    ST_NEAREST_UNCULLED,    /* wavefronts: nearest-hit scans whose bundle cull was skipped (directions all over) */
    ST_NEAREST_UNCULLED_BOX,/* wavefronts: cluster box tests issued in those scans */
    ST_NEAREST_UNCULLED_SPHERE, /* wavefronts: sphere tests issued in those scans */
    ST_NEAREST_SPHERE,      /* wavefronts: sphere tests issued in all nearest-hit scans */
    ST_COUNT
};
template <bool kStats> struct Stats { };
template <> struct Stats<true> { unsigned int c[ST_COUNT]; };
template <bool kStats> __device__ __forceinline__ void st_lane(Stats<kStats> &, int, bool) {}
template <> __device__ __forceinline__ void st_lane<true>(Stats<true> &st, int k, bool cond) { st.c[k] += cond ? 1u : 0u; }
template <bool kStats> __device__ __forceinline__ void st_wave(Stats<kStats> &, int) {}
template <> __device__ __forceinline__ void st_wave<true>(Stats<true> &st, int k) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(true);
    const int lane = (int)(threadIdx.x & 63u);
    st.c[k] += (lane == __ffsll((long long)m) - 1) ? 1u : 0u;       /* first active lane counts for the wave */
}

template <bool kStats> __device__ __forceinline__ unsigned long long st_clock() {
    if constexpr (kStats) return __builtin_amdgcn_s_memtime();
    return 0ull;
}
template <bool kStats> __device__ __forceinline__ void st_cycles(Stats<kStats> &, int, unsigned long long) {}
template <> __device__ __forceinline__ void st_cycles<true>(Stats<true> &st, int k, unsigned long long since) {
    const unsigned int dt = (unsigned int)(__builtin_amdgcn_s_memtime() - since);
    st.c[k] += ((threadIdx.x & 63u) == 0u) ? dt : 0u;
}

template <bool kStats> __device__ __forceinline__ void st_maxlane(Stats<kStats> &, int, int) {}
template <> __device__ __forceinline__ void st_maxlane<true>(Stats<true> &st, int k, int mine) {
    int most = 0;
    for (int l = 0; l < 64; ++l) most = max(most, __builtin_amdgcn_readlane(mine, l));
    st.c[k] += ((threadIdx.x & 63u) == 0u) ? (unsigned int)most : 0u;
}

/* SPHERE TESTS WITHOUT SCALAR INSTRUCTIONS.  A CU of gfx950 issues ONE scalar instruction per cycle for its four SIMDs
 * (scripts/ubench/issue_rate.hip: 0.87-0.97 s_add or s_and_b64 per CU and cycle with one to eight wavefronts per SIMD, next to
 * 1.75 vector ones; a 2 : 1 mix of vector and scalar instructions tops out at 1.56 + 0.78), and the sphere-grid frames ran 0.67
 * scalar instructions + 0.19 branches per CU and cycle next to 1.34 vector ones: the scalar unit was as busy as the vector
 * pipes.  Most of that came from the reference's accept / reject ladder (src/SceneSphere.cpp:60-116) written with `&&`: every
 * per-lane condition is a v_cmp into a scalar register pair, every `&&` an s_and_b64, and the nested conditions became
 * exec-mask branches -- 16 scalar instructions and 4 branches per sphere test.  Here the ladder is arithmetic.  For a finite v
 * and a finite d^2 (q):
 *     candidate  <=>  !(v < 0) && !(q < 1e-9)          <=>  min(v, q - 1e-9f) >= 0
 *                     (a float difference of two floats has the sign of the exact one -- denormals are kept, so it is never
 *                     rounded to zero --; min skips a NaN operand, which the `!(x < c)` form also accepts)
 *     root2 > 0:      holds for every candidate (v >= 0, sqrt(q) >= 3e-5)
 *     the chosen root (root1, or root2 when root1 < 0) < 65535  <=>  65534.99609375f - root >= 0   (the float below 65535)
 * so a hit is ONE comparison of the minimum of the two margins, and the result -- the reported distance root1, or +infinity
 * for a miss -- one select.  Lanes that are no candidates may carry a NaN root (q < 0): min skips it and their own margin is
 * negative.  A wavefront with a lane whose q is not finite or beyond the short square root's range (v not finite implies q not
 * finite) takes the ladder as written instead (the *_exact routines): overflowing scenes, never the benchmarks'. */
__device__ __forceinline__ float sphere_margin(const float v, const float q) { return __builtin_fminf(v, q - (float)1E-9); }
__device__ __forceinline__ float sphere_hit_or_inf(const float v, const float root, const float margin) {
    const float root1 = v - root, root2 = v + root;
    const float chosen = (root1 < 0.0f) ? root2 : root1;
    const float ok = __builtin_fminf(margin, 65534.99609375f - chosen);
    return (ok >= 0.0f) ? root1 : __builtin_huge_valf();
}
/* the ladder as the reference writes it; +infinity for a miss */
__device__ __forceinline__ float sphere_hit_or_inf_exact(const float v, const float q) {
    const bool candidate = !(v < (float)0) && !(q < (float)1E-9);
    const float root = sqrtf(q);
    const float root1 = v - root, root2 = v + root;
    const bool ok = (root2 > (float)0) && ((root1 < (float)0) ? (root2 < 65535.0f) : (root1 < 65535.0f));
    return (candidate && ok) ? root1 : __builtin_huge_valf();
}
__device__ __forceinline__ bool sphere_operands_plain(const float q_magnitudes) { return q_magnitudes <= 0x1p+120f; }   /* false for a NaN */

/* SceneSphere::collision reduced to its distance, src/SceneSphere.cpp:50-116: the distance the reference reports (v - sqrt(d^2),
 * negative for inside hits), or +infinity when it returns NULL */
__device__ __forceinline__ float sphere_hit_distance(const float4 s, const V3 o, const V3 d) {
    const V3 OE = mk(s.x - o.x, s.y - o.y, s.z - o.z);
    const float v = dot3(OE, d);
    const float q = s.w - (dot3(OE, OE) - v * v);
    const float margin = sphere_margin(v, q);
    float t = __builtin_huge_valf();
    if (wave_any(margin >= 0.0f)) {
        if (wave_any(!sphere_operands_plain(fabsf(q)))) t = sphere_hit_or_inf_exact(v, q);
        else t = sphere_hit_or_inf(v, sqrt_in_range(q), margin);
    }
    return t;
}

/* The members of a leaf of a clustered sphere run against a shadow segment: does any of them block?  The tests
 * are sphere_hit_distance()'s, RT_MEMBERS_ABREAST of them side by side: they depend on nothing but the ray, so their
 * dependency chains (LDS read, ~20 dependent operations, the square root's sequence) overlap instead of queueing
 * up -- a wavefront that scans the whole field for all of its rays (the horizon tiles) is bound by exactly that
 * chain.  Blocking is an OR -- here: the minimum of the reported distances against the distance to the light --, so the
 * grouping cannot change the result, and testing a sphere twice cannot either (callers pass any valid sphere for a member
 * that does not exist). */
#ifndef RT_MEMBERS_ABREAST
#define RT_MEMBERS_ABREAST 4
#endif
/* (both used in the nearest-hit scans further down: defined here, ahead of their first use) */
#ifndef RT_NEAR_FLUSH_TWO
#define RT_NEAR_FLUSH_TWO 1     /* the nearest-hit pair flush tests two members side by side (grid-32 3.550 -> 3.530 ms, without shadows 1.828 -> 1.810, grid-16 d8 3.813 -> 3.800; r04_experiments 19) */
#endif
#ifndef RT_NEAR_DIRECT_TWO
#define RT_NEAR_DIRECT_TWO 1    /* a leaf that most lanes need: its members two side by side in the nearest-hit scan too (-0.2 ... -0.7 %) */
#endif
/* four sphere tests side by side: the smallest distance any of them reports (+infinity: none hit), from the tests' v, q (= d^2)
 * and margins */
__device__ __forceinline__ float four_spheres_nearest_vq(const float v0, const float v1, const float v2, const float v3,
                                                         const float q0, const float q1, const float q2, const float q3,
                                                         const float m0, const float m1, const float m2, const float m3) {
    float t0, t1, t2, t3;
    if (wave_any(!sphere_operands_plain((fabsf(q0) + fabsf(q1)) + (fabsf(q2) + fabsf(q3))))) {
        t0 = sphere_hit_or_inf_exact(v0, q0); t1 = sphere_hit_or_inf_exact(v1, q1);
        t2 = sphere_hit_or_inf_exact(v2, q2); t3 = sphere_hit_or_inf_exact(v3, q3);
    } else {
        const float r0 = sqrt_in_range(q0), r1 = sqrt_in_range(q1), r2 = sqrt_in_range(q2), r3 = sqrt_in_range(q3);
        t0 = sphere_hit_or_inf(v0, r0, m0); t1 = sphere_hit_or_inf(v1, r1, m1);
        t2 = sphere_hit_or_inf(v2, r2, m2); t3 = sphere_hit_or_inf(v3, r3, m3);
    }
    return __builtin_fminf(__builtin_fminf(t0, t1), __builtin_fminf(t2, t3));
}

#define RT_FOUR_SPHERES_VQ(s0, s1, s2, s3, o, d)                                                                            \
    const V3 e0 = mk(s0.x - o.x, s0.y - o.y, s0.z - o.z), e1 = mk(s1.x - o.x, s1.y - o.y, s1.z - o.z);                     \
    const V3 e2 = mk(s2.x - o.x, s2.y - o.y, s2.z - o.z), e3 = mk(s3.x - o.x, s3.y - o.y, s3.z - o.z);                     \
    const float v0 = dot3(e0, d), v1 = dot3(e1, d), v2 = dot3(e2, d), v3 = dot3(e3, d);                                    \
    const float q0 = s0.w - (dot3(e0, e0) - v0 * v0), q1 = s1.w - (dot3(e1, e1) - v1 * v1);                                \
    const float q2 = s2.w - (dot3(e2, e2) - v2 * v2), q3 = s3.w - (dot3(e3, e3) - v3 * v3);                                \
    const float m0 = sphere_margin(v0, q0), m1 = sphere_margin(v1, q1), m2 = sphere_margin(v2, q2), m3 = sphere_margin(v3, q3)

/* two sphere tests side by side, for the nearest-hit scans: the distances the two report (+infinity: no hit) */
__device__ __forceinline__ void two_spheres_distances(const float4 s0, const float4 s1, const V3 o, const V3 d, float *t0_out, float *t1_out) {
    const V3 e0 = mk(s0.x - o.x, s0.y - o.y, s0.z - o.z), e1 = mk(s1.x - o.x, s1.y - o.y, s1.z - o.z);
    const float v0 = dot3(e0, d), v1 = dot3(e1, d);
    const float q0 = s0.w - (dot3(e0, e0) - v0 * v0), q1 = s1.w - (dot3(e1, e1) - v1 * v1);
    const float m0 = sphere_margin(v0, q0), m1 = sphere_margin(v1, q1);
    float t0 = __builtin_huge_valf(), t1 = t0;
    if (wave_any(__builtin_fmaxf(m0, m1) >= 0.0f)) {
        if (wave_any(!sphere_operands_plain(fabsf(q0) + fabsf(q1)))) { t0 = sphere_hit_or_inf_exact(v0, q0); t1 = sphere_hit_or_inf_exact(v1, q1); }
        else { t0 = sphere_hit_or_inf(v0, sqrt_in_range(q0), m0); t1 = sphere_hit_or_inf(v1, sqrt_in_range(q1), m1); }
    }
    *t0_out = t0;
    *t1_out = t1;
}

/* BLOCKED WITHOUT THE SQUARE ROOT.  A shadow scan does not want the distance a sphere reports, only whether it is below the
 * distance to the light (src/RayTracer.cpp:727-729), and for nearly every sphere that a shadow ray meets that is plain from v:
 * the reported distance is root1 = fl(v - sqrt(q)) <= v, so a candidate (margin >= 0) with
 *     v <= limit,   limit = min(fl(dist * (1 - 2^-20)), 3e4) < dist   (-infinity when dist is a NaN: nothing is below a NaN)
 *     q <= 1e9      (then root2 = fl(v + sqrt(q)) <= 3e4 + 31 623 < 65535: the reference's "< 65535" holds for either root)
 * is a hit with root1 < dist whatever the square root's digits are.  `blocked` is kept as a number: >= 0 <=> blocked, and a test
 * adds to it by a maximum -- the least of the three margins.  Only a candidate that is not plainly blocking (its nearest point
 * to the centre lies beyond the light) on a ray that nothing has blocked yet needs the square root: then, and for operands that
 * are not finite, the four are evaluated as four_spheres_nearest() does.  14 vector instructions per sphere fewer (the square
 * root's nine and the ladder's) on all but a few per cent of the tests. */
struct ShadowRay { float dist, limit; };
__device__ __forceinline__ ShadowRay shadow_ray(const float dist_to_light) {
    ShadowRay r;
    r.dist = dist_to_light;
    r.limit = (dist_to_light == dist_to_light) ? __builtin_fminf(dist_to_light * 0.99999905f, 3.0e4f) : -__builtin_huge_valf();
    return r;
}
__device__ __forceinline__ float blocked_number(const bool blocked) { return blocked ? 1.0f : -1.0f; }     /* (not 0: -0 >= 0 would read as "not yet") */

__device__ __forceinline__ float four_spheres_block(const float4 s0, const float4 s1, const float4 s2, const float4 s3,
                                                    const V3 o, const V3 d, const ShadowRay ray, float blocked) {
    RT_FOUR_SPHERES_VQ(s0, s1, s2, s3, o, d);
    const float candidate = __builtin_fmaxf(__builtin_fmaxf(m0, m1), __builtin_fmaxf(m2, m3));
    /* nothing to do unless some ray that is not blocked yet has a candidate among the four (a NaN margin -- v and q both NaN:
     * no hit -- is skipped by the maximum, or passes and is sorted out by the exact tests) */
    if (wave_any(__builtin_fminf(candidate, -blocked) >= 0.0f)) {
        const float u0 = __builtin_fminf(m0, __builtin_fminf(ray.limit - v0, 1.0e9f - q0));
        const float u1 = __builtin_fminf(m1, __builtin_fminf(ray.limit - v1, 1.0e9f - q1));
        const float u2 = __builtin_fminf(m2, __builtin_fminf(ray.limit - v2, 1.0e9f - q2));
        const float u3 = __builtin_fminf(m3, __builtin_fminf(ray.limit - v3, 1.0e9f - q3));
        /* (with operands that are not finite the margins mean nothing: the lane's verdict then comes from the exact tests) */
        const bool plain = sphere_operands_plain((fabsf(q0) + fabsf(q1)) + (fabsf(q2) + fabsf(q3)));
        const float sure = __builtin_fmaxf(__builtin_fmaxf(u0, u1), __builtin_fmaxf(u2, u3));
        blocked = plain ? __builtin_fmaxf(blocked, sure) : blocked;
        /* still not blocked, with a candidate that did not settle it -- or operands that are not plain numbers */
        if (wave_any(!plain || __builtin_fminf(candidate, -blocked) >= 0.0f)) {
            const float nearest = four_spheres_nearest_vq(v0, v1, v2, v3, q0, q1, q2, q3, m0, m1, m2, m3);
            blocked = (nearest < ray.dist) ? 1.0f : blocked;
        }
    }
    return blocked;
}

/* one sphere, the same way */
__device__ __forceinline__ float sphere_blocks(const float4 s, const V3 o, const V3 d, const ShadowRay ray, float blocked) {
    const V3 OE = mk(s.x - o.x, s.y - o.y, s.z - o.z);
    const float v = dot3(OE, d);
    const float q = s.w - (dot3(OE, OE) - v * v);
    const float m = sphere_margin(v, q);
    if (wave_any(__builtin_fminf(m, -blocked) >= 0.0f)) {
        const float u = __builtin_fminf(m, __builtin_fminf(ray.limit - v, 1.0e9f - q));
        const bool plain = sphere_operands_plain(fabsf(q));
        blocked = plain ? __builtin_fmaxf(blocked, u) : blocked;
        if (wave_any(!plain || __builtin_fminf(m, -blocked) >= 0.0f)) {
            float t;
            if (wave_any(!plain)) t = sphere_hit_or_inf_exact(v, q);
            else t = sphere_hit_or_inf(v, sqrt_in_range(q), m);
            blocked = (t < ray.dist) ? 1.0f : blocked;
        }
    }
    return blocked;
}

/* two, the same way (the pair flush of the 80-register kernel, which does not hold four) */
__device__ __forceinline__ float two_spheres_block(const float4 s0, const float4 s1, const V3 o, const V3 d, const ShadowRay ray, float blocked) {
    const V3 e0 = mk(s0.x - o.x, s0.y - o.y, s0.z - o.z), e1 = mk(s1.x - o.x, s1.y - o.y, s1.z - o.z);
    const float v0 = dot3(e0, d), v1 = dot3(e1, d);
    const float q0 = s0.w - (dot3(e0, e0) - v0 * v0), q1 = s1.w - (dot3(e1, e1) - v1 * v1);
    const float m0 = sphere_margin(v0, q0), m1 = sphere_margin(v1, q1);
    const float candidate = __builtin_fmaxf(m0, m1);
    if (wave_any(__builtin_fminf(candidate, -blocked) >= 0.0f)) {
        const float u0 = __builtin_fminf(m0, __builtin_fminf(ray.limit - v0, 1.0e9f - q0));
        const float u1 = __builtin_fminf(m1, __builtin_fminf(ray.limit - v1, 1.0e9f - q1));
        const bool plain = sphere_operands_plain(fabsf(q0) + fabsf(q1));
        blocked = plain ? __builtin_fmaxf(blocked, __builtin_fmaxf(u0, u1)) : blocked;
        if (wave_any(!plain || __builtin_fminf(candidate, -blocked) >= 0.0f)) {
            float t0, t1;
            if (wave_any(!plain)) { t0 = sphere_hit_or_inf_exact(v0, q0); t1 = sphere_hit_or_inf_exact(v1, q1); }
            else { t0 = sphere_hit_or_inf(v0, sqrt_in_range(q0), m0); t1 = sphere_hit_or_inf(v1, sqrt_in_range(q1), m1); }
            blocked = (__builtin_fminf(t0, t1) < ray.dist) ? 1.0f : blocked;
        }
    }
    return blocked;
}

template <bool kStats, bool kAbreast>
__device__ __forceinline__ bool leaf_members_block(const float4 *g, const int n, const V3 o, const V3 d, const float dist_to_light,
                                                   const bool lane_needs, const bool blocked_in, Stats<kStats> &st) {
    int i = 0;
    const ShadowRay ray = shadow_ray(dist_to_light);
    float blocked = blocked_number(blocked_in);
    if constexpr (kAbreast && RT_MEMBERS_ABREAST == 4) {      /* not in the plain kernel: its 72 registers do not hold four tests */
        for (; i + 4 <= n; i += 4) {
            if constexpr (kStats) { for (int k = 0; k < 4; ++k) { st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, lane_needs); } }
            blocked = four_spheres_block(g[i], g[i + 1], g[i + 2], g[i + 3], o, d, ray, blocked);
        }
    }
#pragma unroll 2
    for (; i < n; ++i) {
        st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, lane_needs);
        blocked = sphere_blocks(g[i], o, d, ray, blocked);
    }
    return blocked >= 0.0f;
}

/* PLANE TESTS WITHOUT SCALAR INSTRUCTIONS (see SPHERE TESTS WITHOUT SCALAR INSTRUCTIONS above for why).  Every accept / reject
 * condition of the reference's plane routines is written as a MARGIN (>= 0: passes) and the conditions are combined by minima:
 * one comparison and one select per test where the `&&` / `||` forms take a v_cmp per condition and an s_and / s_or per
 * operator.  The result is the distance the reference reports, or +infinity for a miss.
 *   x > 0   <=>  x - 0x1p-149f >= 0   (denormals are kept: the smallest one is representable)
 *   x <= c  <=>  c - x >= 0, x < c <=> pred(c) - x >= 0: a float difference of floats has the sign of the exact one
 *   numerator and denominator non-zero with equal signs  <=>  min(numerator * copysign(1, denom), |denom|) > 0
 * min / max skip a NaN operand where a comparison with it is false: the two forms then differ only in cases whose distance is
 * a NaN itself (a NaN numerator or denominator, 0 * infinity in a bound) -- a "hit" at a NaN distance, which neither scan can
 * tell from a miss: getCollision's `distance < closest` and the shadow scan's `distance < dist` are false for it
 * (src/RayTracer.cpp:75-78, 727-729), and take_nearer() / the running minimum skip it likewise. */

/* Plane prefilter shared by both plane kinds (>= 0: the quotient is worth computing).  t = numerator / denom is only
 * worth computing when it can matter; both skips are exact:
 *  (1) unless numerator and denom are non-zero with equal signs, t <= 0 (or
 *      NaN, which loses every later comparison): the reference's `t < 1E-10` /
 *      `t < 1E-5` rejects it;
 *  (2) if |numerator| > |denom| * bound (with a 1e-6 margin for the rounding of
 *      the product, and only when the product is a normal number), the true
 *      quotient exceeds `bound`, so the correctly rounded t is >= bound and the
 *      caller's `t < bound` (nearest so far / distance to the light) is false. */
__device__ __forceinline__ float plane_candidate_margin(const float numerator, const float denom, const float bound) {
    const float same_sign = __builtin_fminf(numerator * __builtin_copysignf(1.0f, denom), fabsf(denom)) - 0x1p-149f;
    const float prod = fabsf(denom) * bound;
    /* !too_far: prod < 1e-30f (the float below it: 0x1.4484bep-100f) or |numerator| <= prod * 1.000001f */
    const float near_enough = __builtin_fmaxf(0x1.4484bep-100f - prod, prod * 1.000001f - fabsf(numerator));
    return __builtin_fminf(same_sign, near_enough);
}

/* SceneInfinitePlane::collision reduced to t, src/SceneInfinitePlane.cpp:29-51 */
__device__ __forceinline__ float infinite_plane_hit_distance(const float4 q0, const V3 o, const V3 d, const float bound) {
    const V3 n = xyz(q0);
    const float numerator = -q0.w - dot3(o, n);
    const float denom = dot3(d, n);
    const float candidate = plane_candidate_margin(numerator, denom, bound);
    float result = __builtin_huge_valf();
    if (wave_any(candidate >= 0.0f)) {
        const float t = numerator / denom;
        /* hit <=> candidate && !(t < 1E-10) */
        result = (__builtin_fminf(candidate, t - (float)1E-10) >= 0.0f) ? t : result;
    }
    return result;
}

/* SceneFinitePlane::collision reduced to t, src/SceneFinitePlane.cpp:86-124 */
__device__ __forceinline__ float finite_plane_hit_distance(const float4 *g, const V3 o, const V3 d, const float bound) {
    const float4 q0 = g[0];
    const V3 n = xyz(q0);
    const float numerator = -q0.w - dot3(o, n);
    const float denom = dot3(d, n);
    const float candidate = plane_candidate_margin(numerator, denom, bound);
    float result = __builtin_huge_valf();
    if (wave_any(candidate >= 0.0f)) {
        const float4 q1 = g[1], q2 = g[2], q3 = g[3];
        const float t = numerator / denom;
        const V3 p = add3(scale3(d, t), o);
        const V3 PO = sub3(p, xyz(q1));
        const float x = dot3(PO, xyz(q2));
        const float y = dot3(PO, xyz(q3));
        /* `t < 1E-5` is a double comparison in the reference (:102): (double)t < 1e-5 <=> t <= 9.99999974737875e-06f, the float just
         * below 1e-5; the float above that one: 0x1.4f8b5ap-17f.  miss <=> t <= ... || x < 0 || x > h_dist || y < 0 || y > v_dist */
        const float inside = __builtin_fminf(__builtin_fminf(__builtin_fminf(x, q1.w - x), __builtin_fminf(y, q2.w - y)), t - 0x1.4f8b5ap-17f);
        result = (__builtin_fminf(candidate, inside) >= 0.0f) ? t : result;
    }
    return result;
}

/* Axis-aligned rectangles (everything Scene::makeSceneBox builds,
 * src/Scene.cpp:392-416: normal, horizontal and vertical are +-unit axes).
 * With n = sn*e_n, h = sh*e_a, v = sv*e_b the reference's dot products reduce to
 * one multiplication by +-1 plus additions of +-0:
 *     o.n = (o_n*sn + (+-0)) + (+-0),   x = PO.h = ((+-0) + PO_a*sh) + (+-0), ...
 * For a ray whose origin and direction are all FINITE those zeros can only
 * change the sign of a zero result, and a zero numerator, denominator, x or y
 * takes the same branch whatever its sign (t = +-0 is rejected by `t < 1E-5`,
 * `denom == 0` and `x < 0` do not see the sign).  So, in coordinates permuted
 * to (n, a, b), SceneFinitePlane::collision (src/SceneFinitePlane.cpp:86-124)
 * is evaluated with a third of the arithmetic and an identical outcome.  Rays
 * with a non-finite component take the general routine instead (inf*0 = NaN
 * would differ).  Record: r0 = {dto, sn, sh, sv}, r1 = {po_a, po_b, h_dist, v_dist}
 * (h_dist, v_dist >= 0). */
__device__ __forceinline__ float aa_rectangle_hit_distance(const float4 r0, const float4 r1, const V3 op, const V3 dp, const float bound) {
    const float numerator = -r0.x - op.x * r0.y;
    const float denom = dp.x * r0.y;
    const float candidate = plane_candidate_margin(numerator, denom, bound);
    float result = __builtin_huge_valf();
    if (wave_any(candidate >= 0.0f)) {
        const float t = numerator / denom;
        const float pa = dp.y * t + op.y;
        const float pb = dp.z * t + op.z;
        const float x = (pa - r1.x) * r0.z;
        const float y = (pb - r1.y) * r0.w;
        const float inside = __builtin_fminf(__builtin_fminf(__builtin_fminf(x, r1.z - x), __builtin_fminf(y, r1.w - y)), t - 0x1.4f8b5ap-17f);
        result = (__builtin_fminf(candidate, inside) >= 0.0f) ? t : result;
    }
    return result;
}

/* relative growth of a box that must contain every sphere the reference's float
 * arithmetic can report from a given origin; derived at box_needed() */
#ifndef RT_SPHERE_SLACK
#define RT_SPHERE_SLACK 1.5e-3f
#endif
/* the same for items that are finite planes (RT_ITEM_TIGHT, rt_tables.h).  The reference's hit point is
 * p = t d + o in floats (src/SceneFinitePlane.cpp:106-116): off the exact ray by at most 2^-23 (|t d_k| + |p_k|)
 * per axis -- the second term is covered by the host's padding of the box (1e-4 of its magnitude), the first is
 * 1.2e-7 of the distance travelled; the ray's direction is a unit vector to 2e-7.  1e-5 of the L1 distance
 * leaves a factor of 30 to both. */
#ifndef RT_PLANE_SLACK
#define RT_PLANE_SLACK 1.0e-5f
#endif

__device__ __forceinline__ bool ray_is_finite(const V3 o, const V3 d) {
    /* a NaN or infinity in any component makes the sum non-finite */
    return isfinite((fabsf(o.x) + fabsf(o.y) + fabsf(o.z)) + (fabsf(d.x) + fabsf(d.y) + fabsf(d.z)));
}

/* Conservative box test for clustered sphere runs (rt_tables.h).  The box
 * [lo, hi] (already inflated on the host by 1 % of its largest extent + 1e-4)
 * contains every member sphere.  A member can only be a CANDIDATE of
 * sphere_hit_distance() -- computed v >= 0 and computed d^2 >= 1e-9 -- if, with
 * D = |c_i - o|:
 *   the ray LINE passes within  r_eff = sqrt(r_i^2 + 1.4e-6 D^2) <= r_i + 1.2e-3 D
 *   of c_i  (the float evaluation of r^2 - (OE.OE - v*v) is off by at most
 *   ~17 * 2^-24 * (D^2 + r^2), and |d|^2 = 1 +- 4e-7), and
 *   the parameter of closest approach v_i >= -2.4e-7 D;
 * and its reported distance v - sqrt(d^2) is >= the parameter at which the ray
 * enters the ball B(c_i, r_eff), minus 2.4e-7 D.
 * Every such ball lies inside the box grown by `ex` = RT_SPHERE_SLACK * far,
 * where far >= D is the L1 distance from the origin to the box's farthest
 * corner.  (The slack matters: shadow rays that start thousands of units away,
 * on the ground plane near the horizon, pass high over a field of spheres, and
 * a slack of 4e-3 made every leaf below them a candidate.)
 * So a slab test of the ray against the grown box decides: no intersection, or
 * exit behind the origin, or entry beyond `max_dist` (nearest distance so far /
 * distance to the light) => no member can matter.  The slab arithmetic itself
 * carries relative slack 1e-5; 1/d may be approximate (v_rcp).  A NaN in the
 * ray makes every member test fail in the reference too, and every comparison
 * below is written so that a NaN bound means "needed".
 * Lanes that do not need a box may still run its member tests (the guard is
 * wave-level); by the argument above those tests find nothing. */
__device__ __forceinline__ bool box_needed(const float4 centre, const float4 half, const V3 o, const V3 inv,
                                           const float max_dist) {
    /* The box comes as centre and half-extent (rt_tables.h): the slab of axis k is t in c_k (1/d_k) -+ (h_k + slack) |1/d_k| with
     * c = centre - o -- no minimum or maximum per axis (those issue at half rate: Roofline in DESIGN.md), and the L1 distance to
     * the farthest corner is the sum of |c_k| + h_k.  (A direction component of exactly 0 gives +-infinity -+ infinity = NaN, which
     * the minima and maxima below skip: that axis then does not constrain -- conservative.) */
    const float cx = centre.x - o.x, cy = centre.y - o.y, cz = centre.z - o.z;
    const float far = ((fabsf(cx) + half.x) + (fabsf(cy) + half.y)) + (fabsf(cz) + half.z);
    const float ex = RT_SPHERE_SLACK * far;
    const float tcx = cx * inv.x, tcy = cy * inv.y, tcz = cz * inv.z;
    const float tgx = (half.x + ex) * fabsf(inv.x), tgy = (half.y + ex) * fabsf(inv.y), tgz = (half.z + ex) * fabsf(inv.z);
    const float t_enter = fmaxf(fmaxf(tcx - tgx, tcy - tgy), tcz - tgz);
    const float t_exit = fminf(fminf(tcx + tgx, tcy + tgy), tcz + tgz);
    /* no intersection (t_exit < t_enter), exit behind the origin (t_exit < 0) or entry beyond max_dist, each with the slab
     * arithmetic's tolerance -- as ONE margin: the three conditions share the tolerance, so it is added to their minimum (a
     * v_min3), and a NaN anywhere leaves the margin a NaN or positive: "needed" */
    const float tolerance = __builtin_fmaf(1.0e-5f, fabsf(t_enter) + fabsf(t_exit), 1.0e-6f);
    const float reach = __builtin_fmaf(1.0e-3f, fabsf(max_dist), max_dist);
    const float margin = fminf(fminf(t_exit - t_enter, t_exit), reach - t_enter) + tolerance;
    return !(margin < 0.0f);
}

/* A wave-uniform value, made opaque at the point of use: whatever is derived
 * from it (a per-lane address, an int -> float conversion) is then computed
 * there, instead of being hoisted out of the tile loop into a vector register
 * that lives -- or is spilled -- across the scans. */
__device__ __forceinline__ int here(int uniform_value) {
    asm volatile("" : "+v"(uniform_value));          /* opaque, wherever the compiler kept it ... */
    return __builtin_amdgcn_readfirstlane(uniform_value);   /* ... and scalar again */
}

/* a value the compiler may not compute with before this point (an empty volatile asm is never hoisted or speculated) */
__device__ __forceinline__ V3 not_speculated(V3 v) {
    asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z));
    return v;
}

__device__ __forceinline__ float uniform_f(const float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

__device__ __forceinline__ V3 approx_inverse(const V3 d) {
    return mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
}

/* Wavefront-wide reductions with DPP (no LDS traffic).
 * Box of a per-lane point over the lanes with `use` set: three minima and three
 * maxima reduced side by side (the six chains fill each other's DPP wait
 * states): four butterfly steps inside each row of 16 lanes, then row_bcast
 * 15 / 31 fold the rows so that lane 63 holds the result.  All 64 lanes must be
 * active.  v_min/v_max ignore a NaN operand, like fminf/fmaxf. */
__device__ __forceinline__ void wave_bounds3(const V3 v, const bool use, V3 *lo, V3 *hi) {
    const float inf = __builtin_huge_valf();
    float a = use ? v.x : inf, b = use ? v.y : inf, c = use ? v.z : inf;
    float d = use ? v.x : -inf, e = use ? v.y : -inf, f = use ? v.z : -inf;
#define RT_DPP_STEP(ctrl)                                                                              \
    "v_min_f32_dpp %0, %0, %0 " ctrl "\n v_min_f32_dpp %1, %1, %1 " ctrl "\n v_min_f32_dpp %2, %2, %2 " ctrl "\n" \
    "v_max_f32_dpp %3, %3, %3 " ctrl "\n v_max_f32_dpp %4, %4, %4 " ctrl "\n v_max_f32_dpp %5, %5, %5 " ctrl "\n"
    asm volatile("s_nop 1\n"
                 RT_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RT_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef RT_DPP_STEP
    *lo = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63)),
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63)),
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c), 63)));
    *hi = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), 63)),
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e), 63)),
             __int_as_float(__builtin_amdgcn_readlane(__float_as_int(f), 63)));
}

/* Wavefront-wide minimum of an unsigned key (all 64 lanes active), same DPP
 * steps; a single chain, so the wait states are spelled out. */
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t k) {
#define RT_DPP_STEP(ctrl) "s_nop 1\n v_min_u32_dpp %0, %0, %0 " ctrl "\n"
    asm volatile(RT_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RT_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(k));
#undef RT_DPP_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)k, 63);
}

/* Wavefront-wide OR of two 64-bit per-lane values given as four words (all 64 lanes active): four chains side by side, like
 * wave_bounds3()'s six */
__device__ __forceinline__ void wave_or_u64x2(uint32_t a_lo, uint32_t a_hi, uint32_t b_lo, uint32_t b_hi,
                                              unsigned long long *a, unsigned long long *b) {
#define RT_DPP_STEP(ctrl) "v_or_b32_dpp %0, %0, %0 " ctrl "\n v_or_b32_dpp %1, %1, %1 " ctrl "\n v_or_b32_dpp %2, %2, %2 " ctrl "\n v_or_b32_dpp %3, %3, %3 " ctrl "\n"
    asm volatile("s_nop 1\n"
                 RT_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_half_mirror row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_mirror row_mask:0xf bank_mask:0xf")
                 RT_DPP_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 RT_DPP_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1\n"
                 : "+v"(a_lo), "+v"(a_hi), "+v"(b_lo), "+v"(b_hi));
#undef RT_DPP_STEP
    *a = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)a_lo, 63) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)a_hi, 63) << 32);
    *b = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)b_lo, 63) | ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)b_hi, 63) << 32);
}

/* SHADOW VOXELS (rt_tables.h): the cell of u = (x - lo) * scale on one axis of the grid -- RT_SVOX_TAIL cells below the core,
 * n core cells, RT_SVOX_TAIL above; -1: beyond the last cell, or a NaN.  Tail cell j holds the points whose distance d beyond
 * the core, in core cells, has 16^j <= 1 + 15 d < 16^(j+1): j = floor(log16) from the float's exponent.  svox_axis_bounds()
 * (rt_capi.hip) widens every cell by more than these roundings can move a point. */
__device__ __forceinline__ int svox_axis_cell(const float u, const int n) {
    const float fn = (float)n;
    const bool core = u >= 0.0f && u < fn;
    const float d = u < 0.0f ? -u : u - fn;
    const float w = d * 15.0f + 1.0f;
    const int j = ((int)(__float_as_uint(w) >> 23) - 127) >> 2;
    const bool tail = w >= 1.0f && j < RT_SVOX_TAIL;           /* (a NaN fails the comparison; infinity has j = 32) */
    return core ? RT_SVOX_TAIL + (int)u : (tail ? (u < 0.0f ? RT_SVOX_TAIL - 1 - j : RT_SVOX_TAIL + n + j) : -1);
}

/* getCollision (src/RayTracer.cpp:50-89) over the ITEM table with a
 * wave-cooperative cull -- the nearest-hit counterpart of in_shade() below.
 *
 * Must be called by the whole (converged) wavefront; `active` says whether
 * this lane has a ray.  The wavefront's rays are bounded by a box of origins
 * [omin, omax] and a box of directions [dmin, dmax] (DPP min/max reductions).
 * Any point a ray can reach is o + t*d with o in the origin box, d in the
 * direction box, 0 <= t <= the largest `nearest so far` -- a superset of the
 * real rays.  LANE i tests ITEM base+i: with [a, b] = (item box) - (origin box),
 * grown by the same distance-proportional slack as box_needed(), the item is
 * reachable only if some t satisfies, on every axis k,
 *      t * dmin_k <= b_k   and   t * dmax_k >= a_k,
 * which is an interval intersection with wave-uniform coefficients.  One
 * ballot gives the candidates; only they get the exact per-lane tests.  The
 * object a ray hits contains the hit point, and so does its (inflated) box, so
 * no object that can win is ever culled.  A ray with a NaN component hits
 * nothing in the reference either and may be ignored by the reductions.
 * Plain items are visited in Scene index order (strict `<` keeps the first of
 * equal distances, as the reference does); clustered groups come last and
 * break ties on the Scene index. */
/* (t, idx) before (best, best_idx) in the order the reference's scan implies:
 * nearer, or as near with a smaller Scene index (getCollision keeps the first
 * of equals, src/RayTracer.cpp:71-80). */
__device__ __forceinline__ bool nearer(const float t, const int idx, const float best, const int best_idx) {
    return t < best || (t == best && idx < best_idx);
}
/* (best, best_idx) := the earlier of it and (t, idx) in that order, without a branch or a scalar instruction (SPHERE TESTS
 * WITHOUT SCALAR INSTRUCTIONS, above): t is a reported distance or +infinity -- or, from a plane test, a NaN, which loses both
 * comparisons and is skipped by the minimum: nothing changes, as in the reference's `distance < closest` --; best_idx = -1
 * (nothing yet) goes with best = 65535, which no reported distance reaches */
__device__ __forceinline__ void take_nearer(const float t, const int idx, float *best, int *best_idx) {
    const int on_tie = min(idx, *best_idx);
    const int kept = (t == *best) ? on_tie : *best_idx;
    *best_idx = (t < *best) ? idx : kept;
    *best = __builtin_fminf(*best, t);
}

/* The bundle cull of nearest_hit_items() asks, per axis, for the t >= 0 with
 * t dmin <= b and t dmax >= a (dmin, dmax: the bundle's direction range; a, b:
 * the item box relative to the origin box).  Each inequality bounds t from
 * below or from above depending on the sign of its direction bound; this
 * returns, for one axis, the factors that turn b (first pair) and a (second
 * pair) into that lower / upper bound.  The factor of the bound an inequality
 * does not give is NaN, which min/max skip.  A zero direction bound gives an
 * infinite factor: "0 <= b" becomes t <= +-inf (or no bound when b == 0). */
__device__ __forceinline__ void bound_multipliers(const float dmin, const float dmax, float *lower_b, float *upper_b,
                                                  float *lower_a, float *upper_a) {
    const float nan = __builtin_nanf("");
    const float rmin = __builtin_amdgcn_rcpf(fabsf(dmin)), rmax = __builtin_amdgcn_rcpf(fabsf(dmax));
    const bool min_negative = dmin < 0.0f, max_positive = dmax > 0.0f;
    *lower_b = uniform_f(min_negative ? -rmin : nan);
    *upper_b = uniform_f(min_negative ? nan : rmin);
    *lower_a = uniform_f(max_positive ? rmax : nan);
    *upper_a = uniform_f(max_positive ? nan : -rmax);
}

#define RT_COOP_IDLE 0x80000000u     /* state word of a published ray whose lane has no shadow ray (or is blocked already) */

/* HELP (clustered scenes, whole frames and wide strips: rt_render_kernel_clusters*).  A wavefront that has run
 * out of tiles does not leave: it waits at its workgroup's DESK (RT_DESK_WORDS words of LDS) until all the workgroup's
 * wavefronts are out of tiles, and meanwhile serves the others.  A wavefront whose shadow scan is left with
 * p.help_leaves or more candidate leaves, and that sees a colleague waiting, publishes its 64 rays (global
 * memory, 2 KB per workgroup) and the candidate mask at the desk and opens it; everybody -- the owner included --
 * then takes candidates from a shared cursor, four bits of the mask at a time, and ORs the rays it found blocked
 * into the desk's verdict.  The owner closes the desk when the cursor is through, waits until the helpers that
 * are still inside have left, and reads the verdict.  Blocking is an OR over the candidates (src/RayTracer.cpp:
 * 727-729), so who tests which leaf does not matter.  This is what shortens the END of a frame -- or of a GPU's
 * strip of it --, when a few tiles with scans over the whole scene are all that is left and most wavefronts
 * would idle.  Nobody ever waits for a helper to ARRIVE; the owner's wait for helpers to LEAVE is bounded by
 * one leaf's tests (and by RT_HELP_SPIN_LIMIT, after which the kernel gives up helping for good). */
enum { RT_DESK_FREE = 0, RT_DESK_FILLING = 1, RT_DESK_OPEN = 2, RT_DESK_CLOSING = 3 };
static_assert(ST_COUNT == RT_STATS_COUNT, "RT_STATS_COUNT in include/rt_capi_tuning.h must equal ST_COUNT");

/* the desk's words are read and written with workgroup-scope atomics on the LDS pointer itself (ds_read / ds_write that
 * the compiler may neither cache nor move); a volatile generic pointer turned them into flat loads */
__device__ __forceinline__ uint32_t desk_read(uint32_t *desk, const int word) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(desk + word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
}
__device__ __forceinline__ void desk_write(uint32_t *desk, const int word, const uint32_t value) {
    __hip_atomic_store(desk + word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

/* A share of the candidate leaves of one round of a shadow scan (HELP): candidates number share, share + n_shares, ...
 * of leaf_mask (bit i = item base + i), tested for this wavefront's 64 rays; returns blocked */

template <bool kStats>
__device__ __forceinline__ bool shadow_leaf_share(const float4 *lds, const float4 *items, const int base,
                                                  unsigned long long leaf_mask, const int share, const int n_shares,
                                                  bool blocked, const V3 o, const V3 d, const V3 inv,
                                                  const float dist_to_light, Stats<kStats> &st) {
    /* this share's leaves: every n_shares-th candidate, from candidate number `share` on */
    unsigned long long mine = 0ull;
    if (n_shares == 1) {
        mine = leaf_mask;
    } else {
        int turn = 0;
        for (unsigned long long rest = leaf_mask; rest != 0ull; rest &= rest - 1ull) {
            if (turn == share) mine |= rest & (0ull - rest);
            turn = turn + 1 == n_shares ? 0 : turn + 1;
        }
    }
    /* two at a time: the box tests of two leaves side by side, like the owner's (RT_LEAVES_ABREAST) */
    while (mine != 0ull) {
        if (!wave_any(!blocked)) return true;
        const int item = base + (__ffsll((long long)mine) - 1);
        mine &= mine - 1ull;
        const bool two = mine != 0ull;
        const int item2 = two ? base + (__ffsll((long long)mine) - 1) : item;
        mine &= mine - 1ull;                            /* 0 stays 0 */
        const float4 i0 = items[2 * item], i1 = items[2 * item + 1];
        const float4 j0 = items[2 * item2], j1 = items[2 * item2 + 1];
        st_wave(st, ST_WAVE_BOX_TESTS);
        if (two) st_wave(st, ST_WAVE_BOX_TESTS);
        const bool needs_a = !blocked && box_needed(i0, i1, o, inv, dist_to_light);
        const bool needs_b = two && !blocked && box_needed(j0, j1, o, inv, dist_to_light);
        if (wave_any(needs_a)) {
            const uint32_t bits = __float_as_uint(i0.w);
            st_wave(st, ST_SHADOW_LEAVES_UNION);
            blocked = leaf_members_block<kStats, true>(lds + (bits >> 16), (int)((bits >> 8) & 255u), o, d, dist_to_light, needs_a, blocked, st);
        }
        const bool still_b = needs_b && !blocked;
        if (wave_any(still_b)) {
            const uint32_t bits = __float_as_uint(j0.w);
            st_wave(st, ST_SHADOW_LEAVES_UNION);
            blocked = leaf_members_block<kStats, true>(lds + (bits >> 16), (int)((bits >> 8) & 255u), o, d, dist_to_light, still_b, blocked, st);
        }
    }
    return blocked;
}

/* HELP: take candidates off the desk's cursor, four mask bits at a time, until it is through */
template <bool kStats>
__device__ __forceinline__ bool help_shadow_candidates(const float4 *lds, const float4 *items, uint32_t *desk,
                                                       const int base, const unsigned long long leaf_mask, bool blocked,
                                                       const V3 o, const V3 d, const V3 inv, const float dist_to_light,
                                                       Stats<kStats> &st) {
    const int lane = (int)(threadIdx.x & 63u);
    for (;;) {
        int chunk = 0;
        if (lane == 0) chunk = (int)atomicAdd(desk + RT_DESK_CURSOR, 1u);
        chunk = __builtin_amdgcn_readfirstlane(chunk);
        if (chunk >= 16) break;
        const unsigned long long part = leaf_mask & (0xFull << (4 * chunk));
        if (part != 0ull) blocked = shadow_leaf_share<kStats>(lds, items, base, part, 0, 1, blocked, o, d, inv, dist_to_light, st);
    }
    return blocked;
}

/* HELP: one visit to the open desk: take candidates of the owner's scan until the cursor is through */
template <bool kStats>
__device__ __forceinline__ void serve_desk(const RtParams &p, const float4 *lds, uint32_t *desk, const float4 *help_rays, Stats<kStats> &st) {
    const int lane = (int)(threadIdx.x & 63u);
    if (lane == 0) atomicAdd(desk + RT_DESK_INSIDE, 1u);
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");                      /* INSIDE is out before the state is read again */
    if (desk_read(desk, RT_DESK_STATE) == (uint32_t)RT_DESK_OPEN) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const unsigned long long leaf_mask = (unsigned long long)desk_read(desk, RT_DESK_MASK_LO) |
                                             ((unsigned long long)desk_read(desk, RT_DESK_MASK_HI) << 32);
        const int base = (int)desk_read(desk, RT_DESK_BASE);
        const volatile float4 *rays = help_rays + (size_t)blockIdx.x * 128;
        const float ox = rays[lane].x, oy = rays[lane].y, oz = rays[lane].z, dist = rays[lane].w;
        const float dx = rays[64 + lane].x, dy = rays[64 + lane].y, dz = rays[64 + lane].z;
        const uint32_t state = __float_as_uint(rays[64 + lane].w);
        const V3 o = mk(ox, oy, oz), d = mk(dx, dy, dz);
        const bool blocked = help_shadow_candidates<kStats>(lds, lds + p.shadow_items_off, desk, base, leaf_mask,
                                                            state == RT_COOP_IDLE, o, d, approx_inverse(d), dist, st);
        const unsigned long long verdict = __builtin_amdgcn_ballot_w64(blocked && state != RT_COOP_IDLE);
        if (lane == 0 && verdict != 0ull) {
            atomicOr(desk + RT_DESK_VERDICT_LO, (uint32_t)verdict);
            atomicOr(desk + RT_DESK_VERDICT_HI, (uint32_t)(verdict >> 32));
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicSub(desk + RT_DESK_INSIDE, 1u);
}

/* NEAREST PAIRS -- the pair compaction of the shadow scans (PAIRS, above in_shade()) for the nearest-hit
 * scan.  A leaf that fewer than RT_PAIR_DIRECT_LANES lanes need is not tested for the whole wavefront:
 * the needing lanes push their lane number to the next free slots, the slots note the leaf, and a
 * flush lets every slot find the nearest member of its leaf for its ray (minimum of (distance, Scene
 * index)); the results then go back PUSH by push -- within one push every ray has at most one slot, so
 * the rays' lanes pull theirs (slot = first slot of the push + rank among the needing lanes) and take
 * the minimum with what they hold.  Same tests, and the minimum does not depend on their order
 * (src/RayTracer.cpp:71-80).  While pairs wait, `best` is stale (too large): culls by it are weaker,
 * never wrong; the buffer is flushed from RT_NEAR_FLUSH_PAIRS pairs on to keep them effective. */
#ifndef RT_NEAR_FLUSH_PAIRS
#define RT_NEAR_FLUSH_PAIRS 32
#endif
#ifndef RT_PAIR_DIRECT_LANES
#define RT_PAIR_DIRECT_LANES 32      /* a leaf this many lanes need is tested for the whole wavefront at once (16 / 24 / 32 / 40: 5.24 / 5.26 / 5.24 / 5.31 ms on grid-32 with four members abreast) */
#endif

/* lane `lane_select` of `vector` := value (both wave-uniform) */
__device__ __forceinline__ int write_lane(const int value, const int lane_select, const int vector) {
    return (int)(threadIdx.x & 63u) == lane_select ? value : vector;
}
struct NearPairs {
    int slot;               /* per slot (= lane): the ray's lane | member count << 6 | the leaf's first member quad << 11 */
    int ids;                /* per slot: u32 index of the leaf's members' Scene indices */
    int push_lo, push_hi;   /* per push (lane r = push r of this buffer): the lanes that needed its leaf */
    int fill, pushes;       /* wave-uniform: slots in use, pushes recorded */
};

__device__ __forceinline__ float lane_pull_f(const int byte_addr, const float v) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(byte_addr, __float_as_int(v)));
}

template <bool kStats>
__device__ __forceinline__ void flush_near_pairs(const float4 *lds, NearPairs &pb, const V3 o, const V3 d,
                                                 float *best_io, int *best_idx_io, Stats<kStats> &st) {
    if (pb.fill == 0) return;
    const uint32_t *lds_u32 = reinterpret_cast<const uint32_t *>(lds);
    const int lane = (int)(threadIdx.x & 63u);
    const bool has = lane < pb.fill;
    const int src = pb.slot & 63, count = (pb.slot >> 6) & 31, geom = pb.slot >> 11;
    const int from = src << 2;
    const V3 po = mk(lane_pull_f(from, o.x), lane_pull_f(from, o.y), lane_pull_f(from, o.z));
    const V3 pd = mk(lane_pull_f(from, d.x), lane_pull_f(from, d.y), lane_pull_f(from, d.z));
    const int rot = count == 16 ? ((geom >> 4) & 15) : 0;
    const int members = has ? count : 0;                  /* of this slot's leaf; none without a pair */
    float pair_t = 65535.0f;
    int pair_idx = -1;
#if RT_NEAR_FLUSH_TWO
    /* two members side by side: one gate for both square roots, one for both updates */
    for (int i = 0; wave_any(i < members); i += 2) {
        int j0 = i + rot, j1 = i + 1 + rot;
        j0 = j0 >= count ? j0 - count : j0; j1 = j1 >= count ? j1 - count : j1;
        j1 = (i + 1 < count) ? j1 : j0;                                /* (an odd leaf's last member twice: the same (distance, index)) */
        if constexpr (kStats) { for (int k = 0; k < 2; ++k) { st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, i + k < members); st_wave(st, ST_NEAREST_SPHERE); } }
        float t0, t1;
        two_spheres_distances(lds[geom + j0], lds[geom + j1], po, pd, &t0, &t1);
        t0 = (i < members) ? t0 : __builtin_huge_valf();
        t1 = (i + 1 < members) ? t1 : __builtin_huge_valf();
        if (wave_any(__builtin_fminf(t0, t1) <= pair_t)) {
            take_nearer(t0, (int)lds_u32[pb.ids + j0], &pair_t, &pair_idx);
            take_nearer(t1, (int)lds_u32[pb.ids + j1], &pair_t, &pair_idx);
        }
    }
#else
    for (int i = 0; wave_any(i < members); ++i) {
        int j = i + rot;
        j = j >= count ? j - count : j;
        st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, i < members);
        st_wave(st, ST_NEAREST_SPHERE);
        float t = sphere_hit_distance(lds[geom + j], po, pd);
        t = (i < members) ? t : __builtin_huge_valf();
        if (wave_any(t <= pair_t)) take_nearer(t, (int)lds_u32[pb.ids + j], &pair_t, &pair_idx);
    }
#endif
    /* back to the rays' lanes, push by push */
    float best = *best_io;
    int best_idx = *best_idx_io;
    int first = 0;
    for (int r = 0; r < pb.pushes; ++r) {
        const unsigned long long needers = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane(pb.push_lo, r) |
                                           ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane(pb.push_hi, r) << 32);
        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(needers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needers, 0u));
        const int theirs = (first + rank) << 2;
        const float ot = lane_pull_f(theirs, pair_t);
        const int oi = __builtin_amdgcn_ds_bpermute(theirs, pair_idx);
        if (((needers >> lane) & 1ull) != 0ull && oi >= 0 && (best_idx < 0 || nearer(ot, oi, best, best_idx))) { best = ot; best_idx = oi; }
        first += __popcll(needers);
    }
    *best_io = best;
    *best_idx_io = best_idx;
    pb.fill = 0;
    pb.pushes = 0;
}

/* kMode: 0 a tile of a scene without clustered runs (item tables); 4 of a scene with them (PAIRS, HELP), 5 the same in the
 * kernel with the larger register budget; 6 FAST tables (nearest_hit_fast / in_shade_fast) */
template <bool kStats, int kMode>
__device__ __forceinline__ void nearest_hit_items(const RtParams &p, const float4 *lds, float4 *wlds, const bool active,
                                                  const V3 o, const V3 d, const bool have_origin_box,
                                                  const V3 origins_lo, const V3 origins_hi,
                                                  float *best_out, int *best_idx_out, Stats<kStats> &st) {
    constexpr bool kPairs = kMode != 0;
    NearPairs pairs = {0, 0, 0, 0, 0, 0};
    float best = 65535.0f;
    int best_idx = -1;
    st_lane(st, ST_NEAREST_RAYS, active);
    st_wave(st, ST_WAVE_NEAREST);
    const uint32_t *lds_u32 = reinterpret_cast<const uint32_t *>(lds);
    const int lane = (int)(threadIdx.x & 63u);
    const float inf = __builtin_huge_valf();

    /* the bundle (a handful of items is not worth bounding it for) */
    float dminx = -inf, dmaxx = inf, dminy = -inf, dmaxy = inf, dminz = -inf, dmaxz = inf;
    bool cull = p.cull != 0 && p.n_near_items >= RT_NEAR_CULL_MIN_ITEMS;
    if (cull) {
        V3 dlo, dhi;
        wave_bounds3(d, active, &dlo, &dhi);
        dminx = dlo.x; dminy = dlo.y; dminz = dlo.z; dmaxx = dhi.x; dmaxy = dhi.y; dmaxz = dhi.z;
        /* directions all over the place: the cone is everything, skip the cull */
        cull = !((dminx < 0.0f && dmaxx > 0.0f) && (dminy < 0.0f && dmaxy > 0.0f) && (dminz < 0.0f && dmaxz > 0.0f));
        if (!cull) st_wave(st, ST_NEAREST_UNCULLED);
    }
    const bool stat_unculled = !cull && p.cull != 0 && p.n_near_items >= RT_NEAR_CULL_MIN_ITEMS;
    (void)stat_unculled;
    float ominx = 0, omaxx = 0, ominy = 0, omaxy = 0, ominz = 0, omaxz = 0;
    float lax = 0, hax = 0, lbx = 0, hbx = 0, lay = 0, hay = 0, lby = 0, hby = 0, laz = 0, haz = 0, lbz = 0, hbz = 0;
    if (cull) {
        if (have_origin_box) {                 /* the eye for primary rays, else the previous level's shading points */
            ominx = origins_lo.x; ominy = origins_lo.y; ominz = origins_lo.z;
            omaxx = origins_hi.x; omaxy = origins_hi.y; omaxz = origins_hi.z;
        } else {
            V3 olo, ohi;
            wave_bounds3(o, active, &olo, &ohi);
            ominx = olo.x; ominy = olo.y; ominz = olo.z; omaxx = ohi.x; omaxy = ohi.y; omaxz = ohi.z;
        }
        bound_multipliers(dminx, dmaxx, &lax, &hax, &lbx, &hbx);
        bound_multipliers(dminy, dmaxy, &lay, &hay, &lby, &hby);
        bound_multipliers(dminz, dmaxz, &laz, &haz, &lbz, &hbz);
    }
    const bool finite_rays = !wave_any(active && !ray_is_finite(o, d));
    const float4 *items = lds + p.near_items_off;
    V3 inv = mk(0.0f, 0.0f, 0.0f);           /* 1 / d for the leaf box tests, if the scene has any */
    if (p.n_clusters > 0) inv = approx_inverse(d);

    for (int base = 0; base < p.n_near_items; base += 64) {
        unsigned long long mask;
        uint32_t key = 0xFFFFFFFFu;        /* this lane's item: bundle entry distance (high bits) | lane, see below */
        if (cull) {
            const int mine = min(base + lane, p.n_near_items - 1);
            const float4 b0 = items[2 * mine], b1 = items[2 * mine + 1];
            /* (b0: the box's centre, b1: its half-extent) */
            float ax = (b0.x - b1.x) - omaxx, bx = (b0.x + b1.x) - ominx;
            float ay = (b0.y - b1.y) - omaxy, by = (b0.y + b1.y) - ominy;
            float az = (b0.z - b1.z) - omaxz, bz = (b0.z + b1.z) - ominz;
            /* a plane's hit point is off its ray by rounding only; a sphere's box has to hold the coarse float test */
            float ex, ey, ez;
            RT_CULL_SLACK(__float_as_uint(b0.w), fmaxf(fabsf(ax), fabsf(bx)), fmaxf(fabsf(ay), fabsf(by)), fmaxf(fabsf(az), fabsf(bz)), ex, ey, ez);
            ax -= ex; ay -= ey; az -= ez; bx += ex; by += ey; bz += ez;
            /* feasible t: [t_lo, t_hi], starting from [0, 65535 (the reference's infinity) + slack] */
            /* t dmin <= b and t dmax >= a per axis, as lower / upper bounds of t through
             * the multipliers of bound_multipliers(); a NaN product is no bound */
            const float t_lo = fmaxf(fmaxf(fmaxf(0.0f, fmaxf(bx * lax, ax * lbx)), fmaxf(by * lay, ay * lby)), fmaxf(bz * laz, az * lbz));
            const float t_hi = fminf(fminf(fminf(65600.0f, fminf(bx * hax, ax * hbx)), fminf(by * hay, ay * hby)), fminf(bz * haz, az * hbz));
            const bool empty = (t_lo - 1.0e-4f * fabsf(t_lo) - 1.0e-6f > t_hi + 1.0e-4f * fabsf(t_hi)) || (t_hi < -1.0e-6f);
            const bool candidate = base + lane < p.n_near_items && !empty;
            mask = __builtin_amdgcn_ballot_w64(candidate);
            if (candidate) key = (__float_as_uint(t_lo) & ~63u) | (uint32_t)lane;      /* t_lo >= 0: its bits order like its value */
        } else {
            const int left = p.n_near_items - base;
            mask = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        }
        /* Many candidates: take them nearest first (by the bundle's entry distance into
         * the item's box, a lower bound for every ray of it) and stop as soon as every
         * lane already has a hit nearer than that -- rays grazing a field of spheres
         * would otherwise test all of it.  The order of the tests does not change the
         * result: the winner is the minimum of (distance, Scene index), which is what
         * the reference's in-order scan with a strict `<` finds. */
        const bool ordered = cull && __popcll(mask) >= RT_ORDER_MIN_CANDIDATES;
        while (mask != 0ull) {
            int src;
            if (ordered) {
                const uint32_t nearest_key = wave_min_u32(key);
                if (nearest_key == 0xFFFFFFFFu) break;                 /* cannot happen while mask != 0; keeps the loop finite regardless */
                src = (int)(nearest_key & 63u);
                const float entry = __uint_as_float(nearest_key & ~63u);
                /* entry == 0: the origin box meets the item's box, and a ray that starts inside a
                 * sphere reports a NEGATIVE distance (root1, src/SceneSphere.cpp:136-140): such
                 * items are always tested, whatever the lanes hold already */
                if (entry > 0.0f && !wave_any(active && !(entry - 1.0e-4f * entry - 1.0e-6f > best))) break;
                if (lane == src) key = 0xFFFFFFFFu;
                mask &= ~(1ull << src);
            } else {
                src = __ffsll((long long)mask) - 1;
                mask &= mask - 1ull;
            }
            const int item = base + src;
            const float4 i0 = items[2 * item], i1 = items[2 * item + 1];
            const uint32_t bits = __float_as_uint(i0.w), bits1 = __float_as_uint(i1.w);
            const int kind = (int)(bits & 15u);
            const float4 *g = lds + (bits >> 16);
            const int idx = (int)(bits1 & 4095u);
            float t;
            if (kind == RT_KIND_SPHERE) {
                st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, active);
                take_nearer(sphere_hit_distance(g[0], o, d), idx, &best, &best_idx);
            } else if (kind == RT_KIND_SPHERE_LEAF) {               /* a leaf of a clustered run; bits1 = its members' Scene indices */
                const int n = (int)((bits >> 8) & 255u);
                st_wave(st, ST_WAVE_BOX_TESTS);
                if (stat_unculled) st_wave(st, ST_NEAREST_UNCULLED_BOX);
                const bool lane_needs = active && box_needed(i0, i1, o, inv, best);
                const unsigned long long needers = __builtin_amdgcn_ballot_w64(lane_needs);
                if (needers == 0ull) continue;
                if (kPairs && __popcll(needers) < RT_PAIR_DIRECT_LANES && n < 32) {      /* NEAREST PAIRS, above */
                    const int wanted = __popcll(needers);
                    if (pairs.fill + wanted > 63 || pairs.pushes == 64) flush_near_pairs<kStats>(lds, pairs, o, d, &best, &best_idx, st);
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(needers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needers, 0u));
                    const int who = __builtin_amdgcn_ds_permute((lane_needs ? pairs.fill + rank : 63) << 2, lane);
                    const bool fresh = lane >= pairs.fill && lane < pairs.fill + wanted;
                    pairs.slot = fresh ? (who | (n << 6) | (int)((bits >> 16) << 11)) : pairs.slot;
                    pairs.ids = fresh ? (int)bits1 : pairs.ids;
                    pairs.push_lo = write_lane((int)(uint32_t)needers, pairs.pushes, pairs.push_lo);
                    pairs.push_hi = write_lane((int)(uint32_t)(needers >> 32), pairs.pushes, pairs.push_hi);
                    pairs.fill += wanted;
                    pairs.pushes += 1;
                    if (pairs.fill >= RT_NEAR_FLUSH_PAIRS) flush_near_pairs<kStats>(lds, pairs, o, d, &best, &best_idx, st);
                    continue;
                }
                const uint32_t *ids = lds_u32 + bits1;
#if RT_NEAR_DIRECT_TWO
                int i = 0;
                for (; i + 2 <= n; i += 2) {
                    if constexpr (kStats) { for (int k = 0; k < 2; ++k) { st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, lane_needs); st_wave(st, ST_NEAREST_SPHERE); if (stat_unculled) st_wave(st, ST_NEAREST_UNCULLED_SPHERE); } }
                    float t0, t1;
                    two_spheres_distances(g[i], g[i + 1], o, d, &t0, &t1);
                    if (wave_any(__builtin_fminf(t0, t1) <= best)) {
                        take_nearer(t0, (int)ids[i], &best, &best_idx);
                        take_nearer(t1, (int)ids[i + 1], &best, &best_idx);
                    }
                }
                for (; i < n; ++i) {
#else
#pragma unroll 2
                for (int i = 0; i < n; ++i) {
#endif
                    st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, lane_needs);
                    st_wave(st, ST_NEAREST_SPHERE); if (stat_unculled) st_wave(st, ST_NEAREST_UNCULLED_SPHERE);
                    t = sphere_hit_distance(g[i], o, d);
                    if (wave_any(t <= best)) take_nearer(t, (int)ids[i], &best, &best_idx);
                }
            } else if (kind == RT_KIND_INFINITE_PLANE) {
                st_wave(st, ST_WAVE_PLANE_TESTS);
                take_nearer(infinite_plane_hit_distance(g[0], o, d, best), idx, &best, &best_idx);
            } else if (kind >= RT_KIND_FINITE_AA && kind < RT_KIND_FINITE_AA + 3 && finite_rays) {
                st_wave(st, ST_WAVE_PLANE_TESTS);
                const int axis = kind - RT_KIND_FINITE_AA;             /* of the normal; the record is in cyclic order from it */
                /* one copy of the test per axis: the rotation costs nothing then (six moves otherwise) */
                if (axis == 0)      t = aa_rectangle_hit_distance(g[0], g[1], mk(o.x, o.y, o.z), mk(d.x, d.y, d.z), best);
                else if (axis == 1) t = aa_rectangle_hit_distance(g[0], g[1], mk(o.y, o.z, o.x), mk(d.y, d.z, d.x), best);
                else                t = aa_rectangle_hit_distance(g[0], g[1], mk(o.z, o.x, o.y), mk(d.z, d.x, d.y), best);
                take_nearer(t, idx, &best, &best_idx);
            } else {                                             /* finite plane, general routine on the full record */
                st_wave(st, ST_WAVE_PLANE_TESTS);
                take_nearer(finite_plane_hit_distance(lds + (bits1 >> 12), o, d, best), idx, &best, &best_idx);
            }
        }
    }
    if constexpr (kPairs) flush_near_pairs<kStats>(lds, pairs, o, d, &best, &best_idx, st);
    *best_out = best;
    *best_idx_out = active ? best_idx : -1;
}

/* inShade + inShadeCollisionDetection, src/RayTracer.cpp:709-771: any non-light
 * object of the scan range with distance < dist_to_light blocks the light.
 *
 * Must be called by the whole (converged) wavefront; `active` says whether
 * this lane has a shadow ray at all.  The wavefront first culls the shadow
 * ITEMS (rt_tables.h: one per object, or per leaf of a clustered sphere run)
 * cooperatively: the 64 shadow segments all end at the same light, so they lie
 * inside the bounding box B of {their origins} + {light}; LANE i tests ITEM
 * base+i's box against B and one ballot yields the candidate mask.  Only
 * candidates get the exact per-lane tests.  An object can only block if the
 * reference finds a hit with 0 < distance < dist_to_light; that hit point is
 * within rounding of the segment and of the object, so both boxes contain it
 * once grown by `fuzz` (the item boxes are inflated on the host, B here: RT_SPHERE_SLACK of
 * its size covers the sphere routine's distance-dependent slack, see
 * box_needed()).  A NaN bound compares "overlapping".  Blocking is a boolean OR,
 * so order does not matter. */
/* Centre and half-extent of the box [lo, hi] of a bounce level's shading points
 * (the half-extent rounded up so that [c - e, c + e] really covers the box), as
 * scalars: what every light's shadow scan of that level starts from. */
__device__ __forceinline__ void shading_point_bundle(const V3 lo, const V3 hi, V3 *centre, V3 *half) {
    const V3 c = mk(0.5f * (lo.x + hi.x), 0.5f * (lo.y + hi.y), 0.5f * (lo.z + hi.z));
    *half = mk(uniform_f(fmaxf(hi.x - c.x, c.x - lo.x) * 1.000001f), uniform_f(fmaxf(hi.y - c.y, c.y - lo.y) * 1.000001f),
               uniform_f(fmaxf(hi.z - c.z, c.z - lo.z) * 1.000001f));
    *centre = mk(uniform_f(c.x), uniform_f(c.y), uniform_f(c.z));
}

/* BOTH LIGHTS' SHADOW CULLS IN ONE PASS (FAST tables, two lights and at most 32 shadow items: the built-in scene).  A scan's bundle cull gives every item a lane; with 18-30 items half the wavefront idles, and the scan towards the
 * other light repeats the pass.  Here lane i tests item i against the segments towards light 0 and lane 32 + i the same item
 * against those towards light 1: the light's position is a per-lane select, and what the scans derive from it as wavefront-wide
 * scalars (the centre segment, its reciprocals, the slack of its reach: in_shade()) is per-lane arithmetic done once for both
 * -- the same test on the same numbers.  Bits 0-31 of the result: the candidates towards light 0, bits 32-63: towards light 1. */
__device__ __forceinline__ unsigned long long shadow_cull_two_lights(const float4 *boxes, const int n_items, const V3 c, const V3 half,
                                                                    const V3 light0, const V3 light1) {
    const int lane = (int)(threadIdx.x & 63u);
    const bool second = lane >= 32;
    const V3 light = mk(second ? light1.x : light0.x, second ? light1.y : light0.y, second ? light1.z : light0.z);
    const V3 seg = sub3(light, c);
    const V3 sinv = approx_inverse(seg);
    const float reach = (fabsf(seg.x) + fabsf(seg.y) + fabsf(seg.z)) + (half.x + half.y + half.z);
    const float grow_more = (RT_SPHERE_SLACK - RT_PLANE_SLACK) * reach;
    const float grow = RT_PLANE_SLACK * reach + 1.0e-4f;
    const int mine = min(lane & 31, n_items - 1);
    const float4 b0 = boxes[2 * mine], b1 = boxes[2 * mine + 1];
    const float more = (__float_as_uint(b0.w) & RT_ITEM_TIGHT) != 0u ? 0.0f : grow_more;
    const float gx = (half.x + grow) + more, gy = (half.y + grow) + more, gz = (half.z + grow) + more;
    /* b0: the item box's centre, b1: its half-extent: the slab of axis k is (centre - c) / seg -+ (half + grown) / |seg| */
    const float tcx = (b0.x - c.x) * sinv.x, tcy = (b0.y - c.y) * sinv.y, tcz = (b0.z - c.z) * sinv.z;
    const float tgx = (b1.x + gx) * fabsf(sinv.x), tgy = (b1.y + gy) * fabsf(sinv.y), tgz = (b1.z + gz) * fabsf(sinv.z);
    const float s_enter = fmaxf(fmaxf(tcx - tgx, tcy - tgy), tcz - tgz);
    const float s_exit = fminf(fminf(tcx + tgx, tcy + tgy), tcz + tgz);
    /* every comparison is false on a NaN, which then means "candidate" */
    const bool apart = (s_exit < s_enter - 1.0e-4f * (fabsf(s_enter) + fabsf(s_exit)) - 1.0e-6f) ||
                       (s_exit < -1.0e-4f) || (s_enter > 1.0001f);
    return __builtin_amdgcn_ballot_w64((lane & 31) < n_items && !apart);
}

/* PAIRS (shadow scans of scenes with clustered sphere runs, kMode != 0).  The per-lane box
 * tests leave, for every candidate leaf, the lanes whose ray needs it; testing the leaf's
 * members for the whole wavefront then wastes the other lanes -- two thirds of them on the
 * sphere-grid frames (profiles/: 0.32 of the issued sphere tests were needed by their lane),
 * all but one or two when a single ray grazes the whole field.  So a leaf that fewer than
 * RT_PAIR_DIRECT_LANES lanes need is not tested at once: its (ray, leaf) pairs are appended to a
 * buffer of up to 63 SLOTS, one per lane -- the needing lanes push their lane number to the
 * next free slots (ds_permute), the slots note the leaf (one packed word per slot) -- and a full buffer is FLUSHED: every
 * slot pulls its ray (ds_bpermute), walks its own leaf's members (per-lane LDS addresses,
 * rotated by the leaf's position so that the slots do not meet in one LDS bank), and the
 * verdicts go back to the rays' lanes.  The same sphere test on the same operands as the direct
 * route, and blocking is an OR, so the result is the reference's (src/RayTracer.cpp:727-729). */
#ifndef RT_PAIR_DIRECT_LANES
#define RT_PAIR_DIRECT_LANES 32
#endif
#ifndef RT_BLOCK_BOUND
#define RT_BLOCK_BOUND 256
#endif
/* OLD TILES FIRST.  Tiles take from 10 us to a millisecond, and a SIMD's five to seven resident wavefronts share its issue
 * slots evenly: a long tile advances at a fifth of the speed it would have alone, and a launch short of tiles (one GPU's
 * strip of a multi-GPU frame: 4-6 tiles per wavefront slot) ends with its long tiles running on while the slots around
 * them have nothing left to start.  The long tiles are the ones whose rays go on bouncing, so a wavefront's priority on
 * its SIMD (s_setprio) follows its tile's bounce level -- 1, 2, 3 from the first reflection on, back to 0 for the next
 * tile: the old tile runs ahead of the young ones beside it, which finish later but are not what the launch waits for.
 * No state (a clock-based age in a scalar register cost the 96-register kernel six more spilled VGPRs and whole frames
 * 1-4 %).  RtParams::tile_prio; automatic for strips of up to three fifths of the image's width (whole frames lose about
 * 1 % to it).  profiles/r03_experiments.txt, 14. */
#ifndef RT_NT_STORES
#define RT_NT_STORES 1           /* the image leaves through streaming stores (HBM bytes per built-in frame 272 -> 241 MB) */
#endif
#ifndef RT_FLUSH_TWO_ABREAST
#define RT_FLUSH_TWO_ABREAST 1     /* the 80-register kernel's pair flush tests two members abreast (four: 23 spilled registers, grid-32 3.68 -> 3.73 ms; two: 7, 3.68 -> 3.64; r04_experiments 14) */
#endif
#ifndef RT_LEAVES_ABREAST
#define RT_LEAVES_ABREAST 1           /* shadow scans: the box tests of two consecutive candidate leaves side by side (grid-32 5.14 -> 4.99 ms) */
#endif
struct ShadowPairs {
    int slot;               /* per slot (= lane): the ray's lane | member count << 6 | the leaf's first member quad << 11 */
    int fill;               /* wave-uniform: slots in use */
};

template <bool kStats, bool kAbreast>
__device__ __forceinline__ bool flush_shadow_pairs(const float4 *lds, ShadowPairs &pb, const V3 o, const V3 d,
                                                   const float dist_to_light, bool blocked, Stats<kStats> &st) {
    if (pb.fill == 0) return blocked;
    const int lane = (int)(threadIdx.x & 63u);
    const bool has = lane < pb.fill;
    const int src = pb.slot & 63, count = (pb.slot >> 6) & 31, geom = pb.slot >> 11;
    const int from = src << 2;
    const V3 po = mk(lane_pull_f(from, o.x), lane_pull_f(from, o.y), lane_pull_f(from, o.z));
    const V3 pd = mk(lane_pull_f(from, d.x), lane_pull_f(from, d.y), lane_pull_f(from, d.z));
    const float pdist = lane_pull_f(from, dist_to_light);
    const int rot = count == 16 ? ((geom >> 4) & 15) : 0;
    /* (a member number past the leaf's end reads the leaf's first member instead: testing a sphere twice changes nothing) */
    const int members = has ? count : 0;
    const ShadowRay ray = shadow_ray(pdist);
    float pair_verdict = has ? -1.0f : 1.0f;                 /* >= 0: this pair blocks (four_spheres_block()); a slot without a pair asks for nothing */
    if constexpr (kAbreast && RT_MEMBERS_ABREAST == 4) {     /* four members side by side (leaf_members_block()), where the registers allow */
        for (int i = 0; wave_any(i < members); i += 4) {
            int j0 = i + rot, j1 = i + 1 + rot, j2 = i + 2 + rot, j3 = i + 3 + rot;
            j0 = j0 >= count ? j0 - count : j0; j1 = j1 >= count ? j1 - count : j1;
            j2 = j2 >= count ? j2 - count : j2; j3 = j3 >= count ? j3 - count : j3;
            if constexpr (kStats) { for (int k = 0; k < 4; ++k) { st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, i + k < members); } }
            pair_verdict = four_spheres_block(lds[geom + (i < count ? j0 : 0)], lds[geom + (i + 1 < count ? j1 : 0)],
                                              lds[geom + (i + 2 < count ? j2 : 0)], lds[geom + (i + 3 < count ? j3 : 0)], po, pd, ray, pair_verdict);
        }
    } else {
#if RT_FLUSH_TWO_ABREAST
        for (int i = 0; wave_any(i < members); i += 2) {
            int j0 = i + rot, j1 = i + 1 + rot;
            j0 = j0 >= count ? j0 - count : j0; j1 = j1 >= count ? j1 - count : j1;
            if constexpr (kStats) { for (int k = 0; k < 2; ++k) { st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, i + k < members); } }
            pair_verdict = two_spheres_block(lds[geom + (i < count ? j0 : 0)], lds[geom + (i + 1 < count ? j1 : 0)], po, pd, ray, pair_verdict);
        }
#else
        for (int i = 0; wave_any(i < members); ++i) {
            int j = i + rot;
            j = j >= count ? j - count : j;
            st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, i < members);
            pair_verdict = sphere_blocks(lds[geom + (i < count ? j : 0)], po, pd, ray, pair_verdict);
        }
#endif
    }
    const bool pair_blocked = has && pair_verdict >= 0.0f;
    /* the verdicts, back to the rays' lanes (few pairs block) */
    unsigned long long verdicts = __builtin_amdgcn_ballot_w64(pair_blocked);
    while (verdicts != 0ull) {
        const int slot = __ffsll((long long)verdicts) - 1;
        verdicts &= verdicts - 1ull;
        if (lane == (__builtin_amdgcn_readlane(pb.slot, slot) & 63)) blocked = true;
    }
    pb.fill = 0;
    return blocked;
}

template <bool kStats, int kMode>
__device__ __forceinline__ bool in_shade(const RtParams &p, const float4 *lds, float4 *wlds, float4 *help_rays, const bool active,
                                         const V3 o, const V3 d, const float dist_to_light, const V3 light,
                                         const V3 origins_centre, const V3 origins_half,
                                         const unsigned long long voxels_say, Stats<kStats> &st) {
    constexpr bool kPairs = kMode != 0;
    constexpr bool kHelped = kMode == 4 || kMode == 5;      /* a first-pass tile of a clustered scene: HELP */
    constexpr bool kRoomy = kMode == 5;                     /* the kernel with registers to spare: the pair flush tests four members abreast (the other: two) */
    ShadowPairs pairs = {0, 0};
    bool blocked = !active;
    int stat_my_leaves = 0;
    if (p.n_shadow_items == 0) return false;
    st_lane(st, ST_SHADOW_RAYS, active);
    st_wave(st, ST_WAVE_SHADOW);

    /* The bundle of shadow segments: every origin lies within half-extent e of the
     * centre c of [origins_lo, origins_hi], and all segments end at the light, so
     * the point at parameter s of any of them is within (1-s) e of c + s (light - c).
     * An item can matter only if its box, grown by e (and the rounding slack),
     * meets that centre segment for some s in [0, 1]: a slab test per item-lane. */
    const bool cull = p.cull != 0 && p.n_shadow_items >= RT_SHADOW_CULL_MIN_ITEMS;
    const V3 c = origins_centre;             /* both from shading_point_bundle(), once per bounce level */
    V3 e = origins_half, sinv = mk(0, 0, 0); /* e: half-extent, plus slack below */
    float grow_more = 0.0f;
    if (cull) {
        const V3 seg = sub3(light, c);
        sinv = approx_inverse(seg);
        const float reach = (fabsf(seg.x) + fabsf(seg.y) + fabsf(seg.z)) + (e.x + e.y + e.z);
        /* the same in every lane: keep them in scalar registers.  `e` carries the slack of a plane item
         * (RT_ITEM_TIGHT); sphere-like items add `grow_more` */
        grow_more = uniform_f((RT_SPHERE_SLACK - RT_PLANE_SLACK) * reach);
        const float grow = RT_PLANE_SLACK * reach + 1.0e-4f;
        e = mk(uniform_f(e.x + grow), uniform_f(e.y + grow), uniform_f(e.z + grow));
        sinv = mk(uniform_f(sinv.x), uniform_f(sinv.y), uniform_f(sinv.z));
    }
    const int lane = (int)(threadIdx.x & 63u);
    const bool finite_rays = !wave_any(active && !ray_is_finite(o, d));
    const float4 *items = lds + p.shadow_items_off;
    V3 inv = mk(0.0f, 0.0f, 0.0f);
    if (p.n_clusters > 0) inv = approx_inverse(d);
    for (int base = 0; base < p.n_shadow_items; base += 64) {
        unsigned long long mask;
        if (cull) {
            const int mine = min(base + lane, p.n_shadow_items - 1);
            const float4 b0 = items[2 * mine], b1 = items[2 * mine + 1];
            const float more = (__float_as_uint(b0.w) & RT_ITEM_TIGHT) != 0u ? 0.0f : grow_more;
            const float gx = e.x + more, gy = e.y + more, gz = e.z + more;
            /* b0: the item box's centre, b1: its half-extent: the slab of axis k is (centre - c) / seg -+ (half + grown) / |seg| */
            const float tcx = (b0.x - c.x) * sinv.x, tcy = (b0.y - c.y) * sinv.y, tcz = (b0.z - c.z) * sinv.z;
            const float tgx = (b1.x + gx) * fabsf(sinv.x), tgy = (b1.y + gy) * fabsf(sinv.y), tgz = (b1.z + gz) * fabsf(sinv.z);
            const float s_enter = fmaxf(fmaxf(tcx - tgx, tcy - tgy), tcz - tgz);
            const float s_exit = fminf(fminf(tcx + tgx, tcy + tgy), tcz + tgz);
            /* every comparison is false on a NaN, which then means "candidate" */
            const bool apart = (s_exit < s_enter - 1.0e-4f * (fabsf(s_enter) + fabsf(s_exit)) - 1.0e-6f) ||
                               (s_exit < -1.0e-4f) || (s_enter > 1.0001f);
            mask = __builtin_amdgcn_ballot_w64(base + lane < p.n_shadow_items && !apart);
        } else {
            const int left = p.n_shadow_items - base;
            mask = left >= 64 ? ~0ull : ((1ull << left) - 1ull);
        }
        /* SHADOW VOXELS (rt_tables.h): what the lanes' voxels say can block a segment towards this light at all -- the bundle
         * is one box around all 64 shading points, the voxels follow each of them (one round: at most 64 shadow items) */
        if constexpr (kHelped) {
            mask &= voxels_say;
        }
        if constexpr (kStats) { for (int k = __popcll(mask); k > 0; --k) st_wave(st, ST_SHADOW_CANDIDATES); }
        if (kHelped && p.n_clusters > 0) {
            const int plain = min(max(p.shadow_first_leaf - base, 0), 64);
            const unsigned long long leaf_mask = plain >= 64 ? 0ull : (mask & ~((1ull << plain) - 1ull));
            if constexpr (kHelped) {                                 /* HELP, above shadow_leaf_share() */
                uint32_t *desk = reinterpret_cast<uint32_t *>(wlds + p.desk_off);
                if (p.help_rays_quads != 0 && __popcll(leaf_mask) >= p.help_leaves &&
                    (desk_read(desk, RT_DESK_FINISHED) != 0u || desk_read(desk, RT_DESK_DEDICATED) != 0u) &&
                    desk_read(desk, RT_DESK_BROKEN) == 0u) {
                    int mine = 0;
                    if (lane == 0) mine = atomicCAS(desk + RT_DESK_STATE, (uint32_t)RT_DESK_FREE, (uint32_t)RT_DESK_FILLING) == (uint32_t)RT_DESK_FREE;
                    if (__builtin_amdgcn_readfirstlane(mine)) {
                        float4 *rays = help_rays + (size_t)blockIdx.x * 128;
                        rays[lane] = make_float4(o.x, o.y, o.z, dist_to_light);
                        rays[64 + lane] = make_float4(d.x, d.y, d.z, __uint_as_float(blocked ? RT_COOP_IDLE : 1u));
                        /* the helpers are wavefronts of this workgroup, on this CU and behind the same vector L1: workgroup
                         * scope orders the rays before the OPEN below at the cost of a wait (an agent-scope fence writes the
                         * XCD's L2 back and invalidates it, once per opened desk) */
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) {
                            desk_write(desk, RT_DESK_CURSOR, 0u);
                            desk_write(desk, RT_DESK_MASK_LO, (uint32_t)leaf_mask);
                            desk_write(desk, RT_DESK_MASK_HI, (uint32_t)(leaf_mask >> 32));
                            desk_write(desk, RT_DESK_BASE, (uint32_t)base);
                            desk_write(desk, RT_DESK_VERDICT_LO, 0u);
                            desk_write(desk, RT_DESK_VERDICT_HI, 0u);
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                            desk_write(desk, RT_DESK_STATE, (uint32_t)RT_DESK_OPEN);
                        }
                        blocked = help_shadow_candidates<kStats>(lds, items, desk, base, leaf_mask, blocked, o, d, inv, dist_to_light, st);
                        if (lane == 0) desk_write(desk, RT_DESK_STATE, (uint32_t)RT_DESK_CLOSING);
                        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");      /* CLOSING is out before INSIDE is read */
                        /* (a helper that arrives now finds the desk CLOSING and leaves at once without touching the verdict) */
                        bool timed_out = p.help_spin_limit < 0;
                        for (int spins = 0; !timed_out && desk_read(desk, RT_DESK_INSIDE) != 0u; ++spins) {
                            if (spins >= p.help_spin_limit) timed_out = true;
                            else __builtin_amdgcn_s_sleep(2);
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");      /* the helpers' verdicts are out before they left */
                        if (timed_out) {
                            /* A helper is still inside (cannot happen: it tests one chunk of leaves and leaves) and may hold
                             * leaves it has not reported: the desk's verdict is incomplete.  Never hang on it and never trust
                             * it: the owner tests every leaf of the mask itself (an OR: testing a leaf twice changes nothing),
                             * this workgroup stops helping for good, and the host gets to know (RT_ERR_HIP from the next
                             * rt_render / rt_get_timing). */
                            if (lane == 0) {
                                desk_write(desk, RT_DESK_BROKEN, 1u);
                                __hip_atomic_store(reinterpret_cast<unsigned int *>(p.error_word), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            }
                            blocked = shadow_leaf_share<kStats>(lds, items, base, leaf_mask, 0, 1, blocked, o, d, inv, dist_to_light, st);
                        } else {
                            const unsigned long long verdict = (unsigned long long)desk_read(desk, RT_DESK_VERDICT_LO) |
                                                               ((unsigned long long)desk_read(desk, RT_DESK_VERDICT_HI) << 32);
                            blocked = blocked || ((verdict >> lane) & 1ull) != 0ull;
                            if (lane == 0) desk_write(desk, RT_DESK_STATE, (uint32_t)RT_DESK_FREE);
                        }
                        mask &= ~leaf_mask;
                    }
                }
            }
        }
        while (mask != 0ull) {
            const int item = base + (__ffsll((long long)mask) - 1);
            mask &= mask - 1ull;
            if (!wave_any(!blocked)) { st_maxlane(st, ST_SHADOW_LEAVES_MAXLANE, stat_my_leaves); return true; }
            const float4 i0 = items[2 * item], i1 = items[2 * item + 1];
            const uint32_t bits = __float_as_uint(i0.w), bits1 = __float_as_uint(i1.w);
            const int kind = (int)(bits & 15u);
            const float4 *g = lds + (bits >> 16);
            if (kind == RT_KIND_SPHERE) {
                st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, !blocked);
                blocked = sphere_blocks(g[0], o, d, shadow_ray(dist_to_light), blocked_number(blocked)) >= 0.0f;
            } else if (kind == RT_KIND_SPHERE_LEAF) {               /* a leaf of a clustered run */
                /* one leaf whose box test is done: its (ray, leaf) pairs into the buffer, or its members for the whole wavefront */
                auto take_leaf = [&](const uint32_t leaf_bits, const bool lane_needs, const unsigned long long needers) {
                    const int n = (int)((leaf_bits >> 8) & 255u);
                    st_wave(st, ST_SHADOW_LEAVES_UNION);
                    if constexpr (kStats) stat_my_leaves += lane_needs ? 1 : 0;
                    if (kPairs && __popcll(needers) < RT_PAIR_DIRECT_LANES && n < 32) {      /* PAIRS, above */
                        const int wanted = __popcll(needers);
                        if (pairs.fill + wanted > 63) blocked = flush_shadow_pairs<kStats, kRoomy>(lds, pairs, o, d, dist_to_light, blocked, st);
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(needers >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)needers, 0u));
                        /* every lane sends; the ones that do not need the leaf send to lane 63, which is never a slot */
                        const int who = __builtin_amdgcn_ds_permute((lane_needs ? pairs.fill + rank : 63) << 2, lane);
                        const bool fresh = lane >= pairs.fill && lane < pairs.fill + wanted;
                        pairs.slot = fresh ? (who | (n << 6) | (int)((leaf_bits >> 16) << 11)) : pairs.slot;
                        pairs.fill += wanted;
                        return;
                    }
                    blocked = leaf_members_block<kStats, kPairs>(lds + (leaf_bits >> 16), n, o, d, dist_to_light, lane_needs, blocked, st);
                };
#if RT_LEAVES_ABREAST
                const int item2 = mask != 0ull ? base + (__ffsll((long long)mask) - 1) : item;
                const float4 j0 = items[2 * item2], j1 = items[2 * item2 + 1];
                if (kPairs && mask != 0ull && (__builtin_amdgcn_readfirstlane((int)__float_as_uint(j0.w)) & 15) == RT_KIND_SPHERE_LEAF) {
                    /* the next candidate is a leaf too (the leaves are the last items of the table): both box tests side by side */
                    mask &= mask - 1ull;
                    st_wave(st, ST_WAVE_BOX_TESTS); st_wave(st, ST_WAVE_BOX_TESTS);
                    const bool needs_a = !blocked && box_needed(i0, i1, o, inv, dist_to_light);
                    const bool needs_b = !blocked && box_needed(j0, j1, o, inv, dist_to_light);
                    const unsigned long long needers_a = __builtin_amdgcn_ballot_w64(needs_a);
                    if (needers_a != 0ull) take_leaf(bits, needs_a, needers_a);
                    /* rays that the first leaf has blocked meanwhile no longer need the second */
                    const bool still_b = needs_b && !blocked;
                    const unsigned long long needers_b = __builtin_amdgcn_ballot_w64(still_b);
                    if (needers_b != 0ull) take_leaf(__float_as_uint(j0.w), still_b, needers_b);
                    continue;
                }
#endif
                st_wave(st, ST_WAVE_BOX_TESTS);
                const bool lane_needs = !blocked && box_needed(i0, i1, o, inv, dist_to_light);
                const unsigned long long needers = __builtin_amdgcn_ballot_w64(lane_needs);
                if (needers == 0ull) continue;
                take_leaf(bits, lane_needs, needers);
            } else if (kind == RT_KIND_INFINITE_PLANE) {
                st_wave(st, ST_WAVE_PLANE_TESTS);
                blocked = blocked || infinite_plane_hit_distance(g[0], o, d, dist_to_light) < dist_to_light;
            } else if (kind >= RT_KIND_FINITE_AA && kind < RT_KIND_FINITE_AA + 3 && finite_rays) {
                st_wave(st, ST_WAVE_PLANE_TESTS);
                const int axis = kind - RT_KIND_FINITE_AA;
                /* one copy of the test per axis: the rotation costs nothing then (six moves otherwise) */
                float t;
                if (axis == 0)      t = aa_rectangle_hit_distance(g[0], g[1], mk(o.x, o.y, o.z), mk(d.x, d.y, d.z), dist_to_light);
                else if (axis == 1) t = aa_rectangle_hit_distance(g[0], g[1], mk(o.y, o.z, o.x), mk(d.y, d.z, d.x), dist_to_light);
                else                t = aa_rectangle_hit_distance(g[0], g[1], mk(o.z, o.x, o.y), mk(d.z, d.x, d.y), dist_to_light);
                blocked = blocked || t < dist_to_light;
            } else {                                             /* finite plane, general routine on the full record */
                st_wave(st, ST_WAVE_PLANE_TESTS);
                blocked = blocked || finite_plane_hit_distance(lds + (bits1 >> 12), o, d, dist_to_light) < dist_to_light;
            }
        }
    }
    st_maxlane(st, ST_SHADOW_LEAVES_MAXLANE, stat_my_leaves);
    if constexpr (kPairs) blocked = flush_shadow_pairs<kStats, kRoomy>(lds, pairs, o, d, dist_to_light, blocked, st);
    return blocked;
}

/* ---- FAST tables (rt_tables.h): the scans of scenes without clustered sphere runs ------------------------
 * One kind-sorted item list serves both scans; a candidate's exact test reads its two record quads from LDS at
 * an address computed from the (scalar) item number, and its kind and Scene index arrive in a scalar register
 * (s_load from the image in global memory): one LDS round trip per candidate where the item tables need two
 * dependent ones, and no vector instruction spent on dispatch.  The culls are the ones of nearest_hit_items()
 * and in_shade(); the tests are the same routines on the same operands. */

/* one candidate's exact distance test, dispatched on the (scalar) kind: the distance the reference reports, +infinity for a miss */
template <bool kStats>
__device__ __forceinline__ float fast_item_distance(const RtParams &p, const float4 *lds, const uint32_t ctl,
                                                    const float4 r0, const float4 r1, const V3 o, const V3 d,
                                                    const float bound, const bool finite_rays, const bool counts, Stats<kStats> &st) {
    const int kind = (int)(ctl & 15u);
    if (kind >= RT_KIND_FINITE_AA && finite_rays) {
        st_wave(st, ST_WAVE_PLANE_TESTS);
        /* one copy of the test per axis of the normal: the rotation of the ray costs nothing then */
        if (kind == RT_KIND_FINITE_AA)          return aa_rectangle_hit_distance(r0, r1, mk(o.x, o.y, o.z), mk(d.x, d.y, d.z), bound);
        else if (kind == RT_KIND_FINITE_AA + 1) return aa_rectangle_hit_distance(r0, r1, mk(o.y, o.z, o.x), mk(d.y, d.z, d.x), bound);
        else                                    return aa_rectangle_hit_distance(r0, r1, mk(o.z, o.x, o.y), mk(d.z, d.x, d.y), bound);
    } else if (kind == RT_KIND_SPHERE) {
        st_wave(st, ST_WAVE_SPHERE_TESTS); st_lane(st, ST_LANE_SPHERE_TESTS, counts);
        return sphere_hit_distance(r0, o, d);
    } else if (kind == RT_KIND_INFINITE_PLANE) {
        st_wave(st, ST_WAVE_PLANE_TESTS);
        return infinite_plane_hit_distance(r0, o, d, bound);
    } else {
        /* a finite plane that is not axis-aligned -- or is, but some ray has a non-finite component: the general
         * routine on the full record (an AA item finds it through its Scene index) */
        st_wave(st, ST_WAVE_PLANE_TESTS);
        const uint32_t *lds_u32 = reinterpret_cast<const uint32_t *>(lds);
        const uint32_t full = kind == RT_KIND_FINITE_PLANE ? __float_as_uint(r1.x) : (lds_u32[p.objinfo_off * 4 + (ctl >> 8)] & 0xFFFFu);
        return finite_plane_hit_distance(lds + full, o, d, bound);
    }
}

/* getCollision (src/RayTracer.cpp:50-89) over the FAST item list; the cull and the nearest-first order are
 * nearest_hit_items()'s (see there for why they are exact), written for few instructions per scan: the cull always
 * runs (a bundle whose directions point everywhere gets no bound from it: every item a candidate with entry
 * distance 0, taken in lane order), candidates are always taken by their entry distance, whose tolerance is already
 * in the key, and what is the same for the whole wavefront is kept in 64-bit lane masks (scalar registers). */
template <bool kStats>
__device__ __forceinline__ void nearest_hit_fast(const RtParams &p, const float4 *lds, const uint32_t *__restrict__ ctl_words,
                                                 const bool active, const V3 o, const V3 d, const bool have_origin_box,
                                                 const V3 origins_lo, const V3 origins_hi, const bool camera_rays,
                                                 const int tile_x0, const int tile_z0,
                                                 float *best_out, int *best_idx_out, Stats<kStats> &st) {
    float best = 65535.0f;
    int best_idx = -1;
    st_lane(st, ST_NEAREST_RAYS, active);
    st_wave(st, ST_WAVE_NEAREST);
    const int lane = (int)(threadIdx.x & 63u);
    const int n_items = p.n_fast_items;
    const float4 *boxes = lds + p.fast_box_off;
    const float4 *recs = lds + p.fast_rec_off;
    const unsigned long long active_mask = __builtin_amdgcn_ballot_w64(active);

    /* The camera rays of a tile (level 0) with a PRIMARY table (rt_tables.h): the host has projected every item's box to the
     * rectangle of pixels whose ray can reach it, with the distance it is at least away; lane i compares item i's rectangle
     * with the tile's -- no bundle, no reciprocals, no box arithmetic.  Every other scan: the bundle cull. */
    const bool by_pixels = camera_rays && p.n_primary > 0;
    V3 dlo = d, dhi = d, olo = origins_lo, ohi = origins_hi;
    float lax = 0, hax = 0, lbx = 0, hbx = 0, lay = 0, hay = 0, lby = 0, hby = 0, laz = 0, haz = 0, lbz = 0, hbz = 0;
    if (!by_pixels) {
        wave_bounds3(d, active, &dlo, &dhi);
        if (!have_origin_box) wave_bounds3(o, active, &olo, &ohi);      /* the eye for primary rays, else the previous level's shading points */
        bound_multipliers(dlo.x, dhi.x, &lax, &hax, &lbx, &hbx);
        bound_multipliers(dlo.y, dhi.y, &lay, &hay, &lby, &hby);
        bound_multipliers(dlo.z, dhi.z, &laz, &haz, &lbz, &hbz);
    }
    const bool finite_rays = (__builtin_amdgcn_ballot_w64(!ray_is_finite(o, d)) & active_mask) == 0ull;

    for (int base = 0; base < n_items; base += 64) {
        uint32_t key;                      /* this lane's item: tolerant bundle entry distance (high bits) | lane */
        if (by_pixels) {                   /* n_items <= 64: one round */
            const uint4 rect = reinterpret_cast<const uint4 *>(lds)[p.primary_off + min(lane, n_items - 1)];
            const int x_lo = (int)(short)(rect.x & 0xFFFFu), x_hi = (int)rect.x >> 16;
            const int z_lo = (int)(short)(rect.y & 0xFFFFu), z_hi = (int)rect.y >> 16;
            const int tile_x1 = tile_x0 + (64 >> p.tile_z_log2) - 1, tile_z1 = tile_z0 + (1 << p.tile_z_log2) - 1;
            const bool candidate = lane < n_items && x_lo <= tile_x1 && x_hi >= tile_x0 && z_lo <= tile_z1 && z_hi >= tile_z0;
            key = candidate ? ((rect.z & ~63u) | (uint32_t)lane) : 0xFFFFFFFFu;
        } else {
            const int mine = min(base + lane, n_items - 1);
            const float4 b0 = boxes[2 * mine], b1 = boxes[2 * mine + 1];
            /* (b0: the box's centre, b1: its half-extent) */
            float ax = (b0.x - b1.x) - ohi.x, bx = (b0.x + b1.x) - olo.x;
            float ay = (b0.y - b1.y) - ohi.y, by = (b0.y + b1.y) - olo.y;
            float az = (b0.z - b1.z) - ohi.z, bz = (b0.z + b1.z) - olo.z;
            float ex, ey, ez;
            RT_CULL_SLACK(__float_as_uint(b0.w), fmaxf(fabsf(ax), fabsf(bx)), fmaxf(fabsf(ay), fabsf(by)), fmaxf(fabsf(az), fabsf(bz)), ex, ey, ez);
            ax -= ex; ay -= ey; az -= ez; bx += ex; by += ey; bz += ez;
            const float t_lo = fmaxf(fmaxf(fmaxf(0.0f, fmaxf(bx * lax, ax * lbx)), fmaxf(by * lay, ay * lby)), fmaxf(bz * laz, az * lbz));
            const float t_hi = fminf(fminf(fminf(65600.0f, fminf(bx * hax, ax * hbx)), fminf(by * hay, ay * hby)), fminf(bz * haz, az * hbz));
            /* the entry distance less the tolerance of this arithmetic: what `best` is compared with below.
             * 0 (also: anything that rounds to it) = the origin box meets the item's box: always tested, see nearest_hit_items() */
            const float entry = fmaxf(t_lo - 1.0e-4f * t_lo - 1.0e-6f, 0.0f);
            const bool empty = (entry > t_hi + 1.0e-4f * fabsf(t_hi)) || (t_hi < -1.0e-6f);
            const bool candidate = base + lane < n_items && !empty;
            key = candidate ? ((__float_as_uint(entry) & ~63u) | (uint32_t)lane) : 0xFFFFFFFFu;
        }
        for (;;) {
            const uint32_t nearest_key = wave_min_u32(key);
            if (nearest_key == 0xFFFFFFFFu) break;                       /* no candidate left */
            const uint32_t entry_bits = nearest_key & ~63u;
            /* every ray already has a hit nearer than anything in that box (and in all the remaining ones) */
            if (entry_bits != 0u && (active_mask & ~__builtin_amdgcn_ballot_w64(__uint_as_float(entry_bits) > best)) == 0ull) break;
            const int src = (int)(nearest_key & 63u);
            if (lane == src) key = 0xFFFFFFFFu;
            const int item = base + src;
            const uint32_t ctl = ctl_words[item];
            const float4 r0 = recs[2 * item], r1 = recs[2 * item + 1];
            const float t = fast_item_distance<kStats>(p, lds, ctl, r0, r1, o, d, best, finite_rays, active, st);
            take_nearer(t, (int)(ctl >> 8), &best, &best_idx);
        }
    }
    *best_out = best;
    *best_idx_out = active ? best_idx : -1;
}

/* inShade (src/RayTracer.cpp:709-771) over the first n_fast_shadow items of the FAST list; in_shade()'s cull.
 * The verdict is kept as the smallest blocking distance found so far (-inf for a lane without a shadow ray):
 * blocked <=> that is below the distance to the light, which is src/RayTracer.cpp:727-729's OR. */
template <bool kStats>
__device__ __forceinline__ bool in_shade_fast(const RtParams &p, const float4 *lds, const uint32_t *__restrict__ ctl_words,
                                              const bool active, const V3 o, const V3 d, const float dist_to_light,
                                              const V3 light, const V3 origins_centre, const V3 origins_half,
                                              const bool culled_already, const unsigned long long culled, Stats<kStats> &st) {
    const int n_items = p.n_fast_shadow;
    if (n_items == 0) return false;
    const float4 *boxes = lds + p.fast_box_off;
    const float4 *recs = lds + p.fast_rec_off;
    st_lane(st, ST_SHADOW_RAYS, active);
    st_wave(st, ST_WAVE_SHADOW);
    const float inf = __builtin_huge_valf();
    float nearest_block = active ? inf : -inf;
    const int lane = (int)(threadIdx.x & 63u);
    const bool finite_rays = (__builtin_amdgcn_ballot_w64(!ray_is_finite(o, d)) & __builtin_amdgcn_ballot_w64(active)) == 0ull;
    /* the candidates of one round of 64 items, in table order; true: every ray is blocked */
    auto test_candidates = [&](unsigned long long mask, const int base) {
        if constexpr (kStats) { for (int k = __popcll(mask); k > 0; --k) st_wave(st, ST_SHADOW_CANDIDATES); }
        while (mask != 0ull) {
            if (__builtin_amdgcn_ballot_w64(nearest_block < dist_to_light) == ~0ull) return true;
            const int item = base + (__ffsll((long long)mask) - 1);
            mask &= mask - 1ull;
            const uint32_t ctl = ctl_words[item];
            const float4 r0 = recs[2 * item], r1 = recs[2 * item + 1];
            nearest_block = __builtin_fminf(nearest_block, fast_item_distance<kStats>(p, lds, ctl, r0, r1, o, d, dist_to_light, finite_rays,
                                                                                         !(nearest_block < dist_to_light), st));
        }
        return false;
    };
    if (culled_already) {                    /* BOTH LIGHTS' SHADOW CULLS IN ONE PASS: at most 32 items */
        if (test_candidates(culled, 0)) return true;
        return nearest_block < dist_to_light;
    }
    const V3 c = origins_centre;
    const V3 seg = sub3(light, c);
    V3 sinv = approx_inverse(seg);
    const float reach = (fabsf(seg.x) + fabsf(seg.y) + fabsf(seg.z)) + (origins_half.x + origins_half.y + origins_half.z);
    /* the same in every lane: keep them in scalar registers.  `e` carries the slack of a plane item
     * (RT_ITEM_TIGHT); sphere-like items add `grow_more` */
    const float grow_more = uniform_f((RT_SPHERE_SLACK - RT_PLANE_SLACK) * reach);
    const float grow = RT_PLANE_SLACK * reach + 1.0e-4f;
    const V3 e = mk(uniform_f(origins_half.x + grow), uniform_f(origins_half.y + grow), uniform_f(origins_half.z + grow));
    sinv = mk(uniform_f(sinv.x), uniform_f(sinv.y), uniform_f(sinv.z));
    for (int base = 0; base < n_items; base += 64) {
        const int mine = min(base + lane, n_items - 1);
        const float4 b0 = boxes[2 * mine], b1 = boxes[2 * mine + 1];
        const float more = (__float_as_uint(b0.w) & RT_ITEM_TIGHT) != 0u ? 0.0f : grow_more;
        const float gx = e.x + more, gy = e.y + more, gz = e.z + more;
        /* b0: the item box's centre, b1: its half-extent: the slab of axis k is (centre - c) / seg -+ (half + grown) / |seg| */
        const float tcx = (b0.x - c.x) * sinv.x, tcy = (b0.y - c.y) * sinv.y, tcz = (b0.z - c.z) * sinv.z;
        const float tgx = (b1.x + gx) * fabsf(sinv.x), tgy = (b1.y + gy) * fabsf(sinv.y), tgz = (b1.z + gz) * fabsf(sinv.z);
        const float s_enter = fmaxf(fmaxf(tcx - tgx, tcy - tgy), tcz - tgz);
        const float s_exit = fminf(fminf(tcx + tgx, tcy + tgy), tcz + tgz);
        /* every comparison is false on a NaN, which then means "candidate" */
        const bool apart = (s_exit < s_enter - 1.0e-4f * (fabsf(s_enter) + fabsf(s_exit)) - 1.0e-6f) ||
                           (s_exit < -1.0e-4f) || (s_enter > 1.0001f);
        if (test_candidates(__builtin_amdgcn_ballot_w64(base + lane < n_items && !apart), base)) return true;
    }
    return nearest_block < dist_to_light;
}

/* Texture_CheckerBoard::getTexturePixel, src/Texture_CheckerBoard.h:31-65.
 * Returns 1 for the light colour, 2 for the dark colour. */
/* fmodf(x, y) for 0 <= x < 2^20 y, y a normal number well inside the exponent
 * range; ry = v_rcp_f32(y) (1 ulp).  The result is the library's, bit for bit:
 * q = rint(x * ry) is within 3/4 of x / y, so r = x - q y (one fma: exact
 * product, one rounding) has |r| <= 3/4 y; for x >= y both x and q y are
 * multiples of ulp(y) and |r| < 2^(ey+1), so r is representable and the fma
 * returned it exactly, and so is r + y when r < 0.  The unique value in [0, y)
 * congruent to x is fmodf's.  For x < y fmodf returns x itself. */
__device__ __forceinline__ float fmod_small_quotient(const float x, const float y, const float ry) {
    const float q = __builtin_rintf(x * ry);
    float r = __builtin_fmaf(-q, y, x);
    r = (r < 0.0f) ? r + y : r;
    return (x < y) ? x : r;
}

__device__ __forceinline__ bool fmod_small_quotient_ok(const float x, const float y) {
    return (y >= 0x1p-100f) && (y <= 0x1p+100f) && (fabsf(x) < 0x1p+20f * y);
}

__device__ __forceinline__ int checkerboard_select(const float width, const float height, float x, float y) {
    /* every lane that is here takes the short route, or none does */
    if (!wave_any(!(fmod_small_quotient_ok(x, width) && fmod_small_quotient_ok(y, height)))) {
        const float rw = __builtin_amdgcn_rcpf(width), rh = __builtin_amdgcn_rcpf(height);
        if (x >= 0) x = fmod_small_quotient(x, width, rw);
        else        x = fmod_small_quotient(fmod_small_quotient(-x, width, rw) + width / 2.0f, width, rw);
        if (y >= 0) y = fmod_small_quotient(y, height, rh);
        else        y = fmod_small_quotient(fmod_small_quotient(-y, height, rh) + height / 2.0f, height, rh);
    } else
    {
        if (x >= 0) x = fmodf(x, width);
        else        x = fmodf((fmodf((-x), width) + width / 2.0f), width);
        if (y >= 0) y = fmodf(y, height);
        else        y = fmodf((fmodf((-y), height) + height / 2.0f), height);
    }
    if (x < width / 2) return (y < height / 2) ? 1 : 2;
    return (y < height / 2) ? 2 : 1;
}

/* colour of a stack entry / hit: material colour, or one of the texture's two */
__device__ __forceinline__ V3 entry_colour(const RtParams &p, const float4 *lds, const float4 m0,
                                           const uint32_t mbits, const int texsel) {
    if (texsel == 0) return xyz(m0);
    const int tex = (int)(mbits >> 1) - 1;
    return xyz(lds[p.tex_off + tex * RT_TEX_QUADS + (texsel - 1)]);
}

} // namespace

/* Entry [level][threadIdx.x] of this workgroup's slice of the HBM bounce stack.
 * Computed where it is used, from scalar pieces (the host keeps the whole
 * buffer below 2^32 entries), so no per-lane address lives across the scans. */
__device__ __forceinline__ size_t hbm_stack_entry(const RtParams &p, const int level) {
    const unsigned int row = (unsigned int)here((int)blockIdx.x) * (unsigned int)(p.max_depth + 1) + (unsigned int)level;
    return (size_t)(row * (unsigned int)here(p.stack_stride) + threadIdx.x);
}

/* One wavefront tile: camera rays, the bounce loop, the unwind, the store. */
template <bool kStats, int kMode>
__device__ __forceinline__ void render_tile(const RtParams &p, const float4 *lds, float4 *wlds, float4 *help_rays,
                                            const uint32_t *__restrict__ ctl_words, float *__restrict__ out,
                                            float4 *__restrict__ bounce_stack, unsigned long long *__restrict__ stats_out,
                                            Stats<kStats> &st, const int wave_in, const int my_xcc, const int steal,
                                            int &next_pop, unsigned int *const ask_head) {
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);      /* the tile number is the same in all lanes: a scalar register's worth */
    const uint32_t *lds_u32 = reinterpret_cast<const uint32_t *>(lds);
    const int lane = (int)(threadIdx.x & 63u);
    unsigned long long t_start = 0ull, t_start_real = 0ull;
    const int tile_row = wave / p.tiles_x;                  /* tile number, row-major */
    const int tile_col = wave - tile_row * p.tiles_x;
    unsigned int tile_sphere0 = 0u, tile_box0 = 0u;
    if constexpr (kStats) {
        tile_sphere0 = st.c[ST_WAVE_SPHERE_TESTS]; tile_box0 = st.c[ST_WAVE_BOX_TESTS];
        t_start = __builtin_amdgcn_s_memtime();
        t_start_real = __builtin_amdgcn_s_memrealtime();
    }

    /* pixel of this lane: wavefront tiles are tile_x columns by tile_z rows;
     * consecutive lanes walk z, the contiguous axis of pixels[x][z] */
    const int tzl_a = here(p.tile_z_log2);
    const int x = p.x0 + tile_col * (64 >> tzl_a) + (lane >> tzl_a);
    const int z = (tile_row << tzl_a) + (lane & ((1 << tzl_a) - 1));
    const bool inside = (x < p.x1) && (z < p.H);

    /* Camera::createEyeRay, src/Camera.cpp:71-84, with dx = (float)x / W,
     * dz = (float)z / H from the pixel loop, src/RayTracer.cpp:916-918 */
    V3 o = mk(p.eye[0], p.eye[1], p.eye[2]);
    V3 d;
    {
        const float dx_percent = ((float)x) / (float)here(p.W);
        const float dy_percent = ((float)z) / (float)here(p.H);
        const float scalar_x = dx_percent * p.sw - p.shw;
        const float scalar_y = dy_percent * p.sh - p.shh;
        V3 pixel = add3(mk(p.so[0], p.so[1], p.so[2]), scale3(mk(p.ch[0], p.ch[1], p.ch[2]), scalar_x));
        pixel = add3(pixel, scale3(mk(p.cv[0], p.cv[1], p.cv[2]), scalar_y));
        d = normalize3(sub3(pixel, o));
    }

    const V3 null_color = mk(p.null_color[0], p.null_color[1], p.null_color[2]);
    V3 C = null_color;        /* value returned by the deepest calculatePixel call of this lane */
    int top = 0;              /* reflective levels pushed by this lane */
    int levels = 0;           /* wave-uniform: levels any lane entered */
    bool alive = inside;

    /* calculatePixel, src/RayTracer.cpp:448-638, levels 0..max_depth */
    /* a box around the origins of the rays about to be traced: the eye at level 0,
     * afterwards the shading points of the level before (reflected rays start there) */
    V3 box_lo = o, box_hi = o;
    bool have_box = true;
    for (int level = 0; level <= p.max_depth; ++level) {
        if (__ballot(alive) == 0ull) break;
        levels = level + 1;
        if (level >= 1 && level <= 3 && p.tile_prio != 0) {       /* OLD TILES FIRST */
            if (level == 1) __builtin_amdgcn_s_setprio(1);
            else if (level == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
        /* ---- phase 1 (per lane): nearest hit and the winner's CollisionObject ---- */
        bool shade = false;          /* this lane hit a non-light object and shades it */
        V3 P = o, N = d;
        int idx = 0, texsel = 0;     /* winner: Scene index and texture selector (material is re-read when needed) */
        float t = 0.0f;
        const unsigned long long t_scan = st_clock<kStats>();
        if constexpr (kStats) {
            const int n_alive = __popcll(__builtin_amdgcn_ballot_w64(alive));
            st_wave(st, n_alive <= 16 ? ST_NEAREST_1_16 : n_alive <= 32 ? ST_NEAREST_17_32 : n_alive <= 48 ? ST_NEAREST_33_48 : ST_NEAREST_49_64);
        }
        if constexpr (kMode == 6) {
            const int tzl_n = here(p.tile_z_log2);
            nearest_hit_fast<kStats>(p, lds, ctl_words, alive, o, d, have_box, box_lo, box_hi, level == 0,
                                     p.x0 + here(tile_col) * (64 >> tzl_n), here(tile_row) << tzl_n, &t, &idx, st);
        }
        else nearest_hit_items<kStats, kMode>(p, lds, wlds, alive, o, d, have_box, box_lo, box_hi, &t, &idx, st);   /* whole wavefront, converged */
        st_cycles(st, ST_CYCLES_NEAREST, t_scan);
        const unsigned long long t_winner = st_clock<kStats>();
        if (alive) {
            if (idx < 0) {                                   /* :507-509 */
                C = null_color;
                alive = false;
                idx = 0;
            } else {
                const uint32_t info = lds_u32[p.objinfo_off * 4 + idx];
                const float4 *g = lds + (info & 0xFFFFu);
                const int kind = (int)((info >> 16) & 3u);
                const int mat = (int)(info >> 20);
                const float4 m1 = lds[p.mat_off + mat * RT_MAT_QUADS + 1];
                const uint32_t mbits = __float_as_uint(m1.w);
                if (kind == RT_KIND_SPHERE) {                /* src/SceneSphere.cpp:118-149 */
                    const float4 s = g[0];
                    P = add3(scale3(d, t), o);
                    N = normalize3(sub3(P, xyz(s)));
                } else {                                     /* src/SceneInfinitePlane.cpp:53-95, src/SceneFinitePlane.cpp:106-150 */
                    const float4 q0 = g[0], q1 = g[1], q2 = g[2], q3 = g[3], q4 = g[4];
                    const V3 ip = add3(scale3(d, t), o);
                    if ((mbits >> 1) != 0u) {
                        const V3 PO = sub3(ip, xyz(q1));
                        const float tx = dot3(PO, xyz(q2));
                        const float ty = dot3(PO, xyz(q3));
                        const int tex = (int)(mbits >> 1) - 1;
                        const float4 t0 = lds[p.tex_off + tex * RT_TEX_QUADS];
                        const float4 t1 = lds[p.tex_off + tex * RT_TEX_QUADS + 1];
                        texsel = checkerboard_select(t0.w, t1.w, tx, ty);
                    }
                    N = (dot3(xyz(q0), d) < 0) ? xyz(q0) : xyz(q4);
                    P = add3(ip, scale3(N, (float)1E-3));
                }
                if (mbits & 1u) {                            /* hit a light: :520-527 */
                    const float4 m0 = lds[p.mat_off + mat * RT_MAT_QUADS];
                    C = scale3(entry_colour(p, lds, m0, mbits, texsel), m1.z);
                    alive = false;
                } else {
                    shade = true;
                }
            }
        }

        /* ---- phase 2 (whole wavefront, converged): lights in Scene index order, :540-591.
         * Every lane walks the light loop so that the shadow scan can cull scene
         * items for the wavefront as a whole; lanes with nothing to shade carry
         * shade == false through it. ---- */
        /* a shading lane accumulates its colour in C (its previous C is dead: it is
         * overwritten at the end of every level a lane is alive in) */
        st_cycles(st, ST_CYCLES_WINNER, t_winner);
        const unsigned long long t_lights = st_clock<kStats>();
        if (shade) C = mk(0.0f, 0.0f, 0.0f);
        if (wave_any(shade)) {
            /* SHADOW VOXELS (rt_tables.h): which shadow items can matter to this lane's segments towards the (at most two) lights,
             * from the voxel its shading point lies in.  A lane outside the grid (or with a NaN: it fails every comparison) says
             * "any", a lane without a shadow ray "none".  Asked for here, before the bundle's reductions below: the answers are
             * ORed over the wavefront after them, into two scalars the scans AND into their candidates. */
            uint4 voxel_masks = make_uint4(0u, 0u, 0u, 0u);
            if constexpr (kMode == 4 || kMode == 5) {
                if (p.svox_off != 0) {
                    const float ux = (P.x - p.svox_lo[0]) * p.svox_scale[0];
                    const float uy = (P.y - p.svox_lo[1]) * p.svox_scale[1];
                    const float uz = (P.z - p.svox_lo[2]) * p.svox_scale[2];
                    const bool core = ux >= 0.0f && ux < (float)p.svox_n[0] && uy >= 0.0f && uy < (float)p.svox_n[1] &&
                                      uz >= 0.0f && uz < (float)p.svox_n[2];
                    int cx = RT_SVOX_TAIL + (int)ux, cy = RT_SVOX_TAIL + (int)uy, cz = RT_SVOX_TAIL + (int)uz;
                    bool in_grid = core;
                    if (wave_any(shade && !core)) {                 /* the cells beyond the core: only where somebody needs them */
                        cx = svox_axis_cell(ux, p.svox_n[0]); cy = svox_axis_cell(uy, p.svox_n[1]); cz = svox_axis_cell(uz, p.svox_n[2]);
                        in_grid = (cx | cy | cz) >= 0;
                    }
                    if (shade) voxel_masks = make_uint4(~0u, ~0u, ~0u, ~0u);     /* (outside the grid: any item may matter) */
                    if (shade && in_grid) {
                        const int voxel = (cz * (p.svox_n[1] + 2 * RT_SVOX_TAIL) + cy) * (p.svox_n[0] + 2 * RT_SVOX_TAIL) + cx;
                        voxel_masks = (reinterpret_cast<const uint4 *>(ctl_words) + (size_t)p.svox_off)[voxel];
                    }
                }
            }
            /* box of the shading points, shared by every light's shadow scan */
            V3 bundle_centre = mk(0, 0, 0), bundle_half = bundle_centre;
            if constexpr (kMode == 6) {          /* FAST tables: the culls always run */
                have_box = true;
                wave_bounds3(P, shade, &box_lo, &box_hi);
                shading_point_bundle(box_lo, box_hi, &bundle_centre, &bundle_half);
            } else {
                have_box = p.cull != 0 && (p.n_shadow_items >= RT_SHADOW_CULL_MIN_ITEMS || p.n_near_items >= RT_NEAR_CULL_MIN_ITEMS);
                if (have_box) {
                    wave_bounds3(P, shade, &box_lo, &box_hi);
                }
                if (p.cull != 0 && p.n_shadow_items >= RT_SHADOW_CULL_MIN_ITEMS) shading_point_bundle(box_lo, box_hi, &bundle_centre, &bundle_half);
            }
            /* The reference re-normalises the hit's normal twice per light (renormalize3(), below); whether that is the
             * identity -- N.N rounds to exactly 1 in every shading lane, the usual case -- does not depend on the light:
             * asked once per level here instead of twice per light there. */
            const bool normals_are_unit = !wave_any(shade && (N.x * N.x + N.y * N.y + N.z * N.z) != 1.0f);
            unsigned long long voxels_say0 = ~0ull, voxels_say1 = ~0ull;       /* bit i: shadow item i can matter to some lane's segment towards light 0 / 1 */
            if constexpr (kMode == 4 || kMode == 5) {
                if (p.svox_off != 0) wave_or_u64x2(voxel_masks.x, voxel_masks.y, voxel_masks.z, voxel_masks.w, &voxels_say0, &voxels_say1);
            }
            /* BOTH LIGHTS' SHADOW CULLS IN ONE PASS, where the table allows (above in_shade()) */
            /* (the FAST scans only: the clustered-scene kernels paid for the extra code with 18 more spilled scalars -- 256-sphere
             * grid, 18 items: 3.900 -> 3.892 ms with it, the 1 024-sphere grid, 45 items and no use for it, 3.668 -> 3.709) */
            bool both_culls = false;
            unsigned long long culled_both = 0ull;
            if constexpr (kMode == 6) {
                both_culls = p.n_lights == 2 && p.n_fast_shadow > 0 && p.n_fast_shadow <= 32;
                if (both_culls)
                    culled_both = shadow_cull_two_lights(lds + p.fast_box_off, p.n_fast_shadow, bundle_centre, bundle_half,
                                                         xyz(lds[p.lights_off]), xyz(lds[p.lights_off + RT_LIGHT_QUADS]));
            }
            for (int l = 0; l < p.n_lights; ++l) {
                const float4 l0 = lds[p.lights_off + l * RT_LIGHT_QUADS];
                const float4 l1 = lds[p.lights_off + l * RT_LIGHT_QUADS + 1];
                const unsigned long long voxels_say = l == 0 ? voxels_say0 : (l == 1 ? voxels_say1 : ~0ull);
                const unsigned long long culled = l == 0 ? (culled_both & 0xFFFFFFFFull) : (culled_both >> 32);
                /* inShade, :743-771 */
                const V3 dir = sub3(xyz(l0), P);
                float dist_to_light;                         /* |dir|, :748 -- the length normalize3() takes the root of anyway */
                const V3 light_ray = normalize3(dir, &dist_to_light);        /* == Ray(P, dir).direction == cosineShade's light_ray == specular L */
                const unsigned long long t_shadow = st_clock<kStats>();
                bool blocked;
                if constexpr (kMode == 6) blocked = in_shade_fast<kStats>(p, lds, ctl_words, shade, P, light_ray, dist_to_light, xyz(l0), bundle_centre, bundle_half,
                                                                          both_culls, culled, st);
                else blocked = in_shade<kStats, kMode>(p, lds, wlds, help_rays, shade, P, light_ray, dist_to_light, xyz(l0), bundle_centre, bundle_half,
                                                       voxels_say, st);
                st_cycles(st, ST_CYCLES_SHADOW, t_shadow);
                if (shade && !blocked) {
                    /* the winner's material, re-read here rather than kept in registers across the shadow scan */
                    const int mat = (int)(lds_u32[p.objinfo_off * 4 + idx] >> 20);
                    const float4 m0 = lds[p.mat_off + mat * RT_MAT_QUADS];
                    const float4 m1 = lds[p.mat_off + mat * RT_MAT_QUADS + 1];
                    const V3 object_color = entry_colour(p, lds, m0, __float_as_uint(m1.w), texsel);
                    const float diffuse_factor = m0.w, specular_factor = m1.x;
                    /* CollisionObject ctor: Ray(point, normal) re-normalises, src/SceneObject.h:62.  (The rare path goes through an
                     * opaque copy: left to itself the compiler computes the square root and the three divides before the light
                     * loop, on every bounce level of every tile, speculatively -- 55 instructions and two spilled registers.) */
                    V3 normal_dir = N;
                    if (!normals_are_unit) normal_dir = renormalize3(not_speculated(N));
                    const V3 light_color = xyz(l1);
                    /* cosineShade, :654-701 */
                    if (diffuse_factor > (float)0) {
                        float cosine_dot_factor = dot3(normal_dir, light_ray);
                        if (cosine_dot_factor > (float)0) {
                            const float factor = cosine_dot_factor * diffuse_factor * l0.w;
                            C.x += factor * object_color.x * light_color.x;
                            C.y += factor * object_color.y * light_color.y;
                            C.z += factor * object_color.z * light_color.z;
                        }
                        C.x = (C.x > 1.0f) ? 1.0f : C.x;
                        C.y = (C.y > 1.0f) ? 1.0f : C.y;
                        C.z = (C.z > 1.0f) ? 1.0f : C.z;
                    }
                    /* specular, :561-588 */
                    V3 Nn = normal_dir;                                                       /* third normalisation, :566-567 */
                    if (!normals_are_unit) Nn = renormalize3(not_speculated(normal_dir));
                    const V3 R = sub3(light_ray, scale3(Nn, 2.0f * dot3(light_ray, Nn)));
                    const float dot = dot3(d, R);
                    if (dot > (float)0) {
                        float pow_factor = dot;
#pragma unroll
                        for (int j = 0; j < 19; ++j) pow_factor *= dot;
                        const float spec_factor = pow_factor * specular_factor;
                        C = add3(C, scale3(light_color, spec_factor));
                    }
                }
            }
        }

        /* ---- phase 3 (per lane): reflect or finish, :595-604 ---- */
        st_cycles(st, ST_CYCLES_LIGHTS, t_lights);
        const unsigned long long t_reflect = st_clock<kStats>();
        if (shade) {
            const int mat = (int)(lds_u32[p.objinfo_off * 4 + idx] >> 20);
            const float reflective_factor = lds[p.mat_off + mat * RT_MAT_QUADS + 1].y;
            const float n_dot_incoming = dot3(N, d);         /* src/SceneObject.h:65 */
            if (reflective_factor > (float)0 && level == p.max_depth) {
                /* The reflected ray of the LAST level is never traced: the call at max_depth + 1 returns NULL_COLOR at once
                 * (:454-455), so this level's sum can be formed here, with the operations the unwind below would apply to its
                 * stack entry -- final = local + (rf * NULL_COLOR) * colour, :601 -- and the entry (for depth 4, the one level
                 * that does not fit LDS: 26 MB of HBM writes per built-in frame) need not exist, nor the reflected direction. */
                const uint32_t info = lds_u32[p.objinfo_off * 4 + idx];
                const int mat_last = (int)(info >> 20);
                const float4 m0 = lds[p.mat_off + mat_last * RT_MAT_QUADS];
                const float4 m1 = lds[p.mat_off + mat_last * RT_MAT_QUADS + 1];
                const V3 oc = entry_colour(p, lds, m0, __float_as_uint(m1.w), texsel);
                const V3 refl = mk(null_color.x * m1.y * oc.x, null_color.y * m1.y * oc.y, null_color.z * m1.y * oc.z);
                C = add3(C, refl);
                alive = false;
            } else if (reflective_factor > (float)0) {
                const V3 reflected = mk(-2 * N.x * n_dot_incoming + d.x,
                                        -2 * N.y * n_dot_incoming + d.y,
                                        -2 * N.z * n_dot_incoming + d.z);
                float4 e;
                e.x = C.x; e.y = C.y; e.z = C.z;
                e.w = __uint_as_float((uint32_t)idx | ((uint32_t)texsel << 16));
                if (level < p.stack_lds_levels) wlds[here(p.stack_off) + level * here(p.stack_stride) + threadIdx.x] = e;
                else                            bounce_stack[hbm_stack_entry(p, level)] = e;
                top = level + 1;
                o = P;
                d = normalize3(reflected);                   /* Ray(point, reflected) */
                /* if the loop ends now the call at max_depth+1 returns NULL_COLOR, :454-455 */
                C = null_color;
            } else {
                alive = false;                               /* C already holds final_color */
            }
        }
        st_cycles(st, ST_CYCLES_REFLECT, t_reflect);
    }

    if (p.tile_prio != 0) __builtin_amdgcn_s_setprio(0);
    /* The kernels that ask for the next tile late (scenes with clustered runs: render_body) ask HERE, when the rays are through:
     * the answer is back by the time the unwind is done and is taken into a scalar before the pixels are stored (below). */
    if (ask_head != nullptr && lane == 0) next_pop = (int)atomicAdd(ask_head, 1u);
    /* unwind: final_k = local_k + (rf_k * C_{k+1}) * oc_k, inside-out (:601) */
    for (int k = levels - 1; k >= 0; --k) {
        if (k < top) {
            float4 e;
            if (k < p.stack_lds_levels) e = wlds[here(p.stack_off) + k * here(p.stack_stride) + threadIdx.x];
            else                        e = bounce_stack[hbm_stack_entry(p, k)];
            const uint32_t bits = __float_as_uint(e.w);
            const uint32_t info = lds_u32[p.objinfo_off * 4 + (bits & 0xFFFFu)];
            const int mat = (int)(info >> 20);
            const float4 m0 = lds[p.mat_off + mat * RT_MAT_QUADS];
            const float4 m1 = lds[p.mat_off + mat * RT_MAT_QUADS + 1];
            const V3 oc = entry_colour(p, lds, m0, __float_as_uint(m1.w), (int)(bits >> 16));
            const V3 refl = mk(C.x * m1.y * oc.x, C.y * m1.y * oc.y, C.z * m1.y * oc.z);
            C = add3(mk(e.x, e.y, e.z), refl);
        }
    }

    /* The next tile's queue entry, asked for earlier, becomes a scalar BEFORE this tile's pixels are stored.  Waiting for a
     * vector-memory result means waiting for every vector-memory operation issued before it -- one counter, in order, and the
     * stores count too: a wavefront that reads the entry after its stores waits until the pixels have reached the L2, 3.5 us
     * between two 13 us tiles of the built-in scene (timeline of round 3: "gap between consecutive tiles of a slot"). */
    next_pop = __builtin_amdgcn_readfirstlane(next_pop);
    if (inside) {
        const int tzl_b = here(p.tile_z_log2);
        const int sx = here(tile_col) * (64 >> tzl_b) + (lane >> tzl_b);   /* x - x0 */
        const int sz = (here(tile_row) << tzl_b) + (lane & ((1 << tzl_b) - 1));
        float *dst = out + ((size_t)sx * (size_t)p.H + (size_t)sz) * 3;
#if RT_NT_STORES
        /* written once, never read here: streaming stores leave the L2 to the bounce stack and the scratch lines */
        __builtin_nontemporal_store(C.x, dst); __builtin_nontemporal_store(C.y, dst + 1); __builtin_nontemporal_store(C.z, dst + 2);
#else
        dst[0] = C.x; dst[1] = C.y; dst[2] = C.z;
#endif
    }
    if constexpr (kStats) {
        st_cycles(st, ST_CYCLES_TILE, t_start);
        /* per wavefront tile: shader cycles spent on it, then its wave-level counters
         * (the totals are added up once per wavefront, after its last tile) */
        {
            unsigned long long *rec = stats_out + ST_COUNT + (size_t)wave * RT_TILE_STATS;
            if (lane == 0) {
                rec[0] = __builtin_amdgcn_s_memtime() - t_start;
                rec[4] = t_start_real;                               /* 100 MHz constant clock */
                rec[5] = __builtin_amdgcn_s_memrealtime();
                rec[3] = (unsigned long long)my_xcc * 1000ull + (unsigned long long)steal;   /* diagnostic: XCD and steal distance */
            }
            if (st.c[ST_WAVE_SPHERE_TESTS] != tile_sphere0) atomicAdd(&rec[1], (unsigned long long)(st.c[ST_WAVE_SPHERE_TESTS] - tile_sphere0));
            if (st.c[ST_WAVE_BOX_TESTS] != tile_box0) atomicAdd(&rec[2], (unsigned long long)(st.c[ST_WAVE_BOX_TESTS] - tile_box0));

        }
    }
}

/* which of the tile queues still have tiles to hand out (bit q: queue q).  Its own function, called once per exhausted
 * queue */
#ifndef RT_SCAN_INLINE
#define RT_SCAN_INLINE __forceinline__
#endif
/* FIRST TILES WITHOUT THE QUEUES.  Every wavefront of the grid gets its first tile by arithmetic: wavefront w of workgroup b
 * takes entry (b / 8) * (wavefronts per workgroup) + w of queue b mod 8, and a queue's head counts the entries handed out
 * BEYOND those (entry = first_entries(queue) + head).  A launch used to begin with all of its wavefronts -- 7 168 in the plain
 * kernels -- asking the eight heads at once: 900 atomics per word at 88 per microsecond, 10-12 us until the median wavefront
 * had a tile (a built-in strip chunk is 100 us of work).  Every workgroup of the grid runs, so every one of these entries is
 * rendered exactly once whatever XCD the workgroup landed on (the queue is b mod 8, not the XCC_ID: placement is speed only). */
/* (The first `heavy_blocks` workgroups of a launch with HEAVY tiles start with one of those and take their ordinary tiles from
 * the queues afterwards: the entries handed out by arithmetic are the first of the order -- the most expensive rows -- and must
 * not wait a quarter of a millisecond behind a HEAVY tile.  The others count from 0 behind them.) */
__device__ __forceinline__ int first_entries(const int queue, const int heavy_blocks) {
    const int blocks = (int)gridDim.x - heavy_blocks;
    return queue < blocks ? ((blocks - queue + RT_TILE_QUEUES - 1) / RT_TILE_QUEUES) * (int)(blockDim.x >> 6) : 0;
}

__device__ RT_SCAN_INLINE unsigned int queues_with_tiles(const unsigned int *tile_counter, const int n_macros, const int heavy_blocks) {
    const int lane = (int)(threadIdx.x & 63u);
    int left = 0;
    if (lane < RT_TILE_QUEUES) {
        const int len_k = lane < n_macros ? ((n_macros - lane + RT_TILE_QUEUES - 1) / RT_TILE_QUEUES) * RT_MACRO_ROWS : 0;
        const unsigned int taken = __hip_atomic_load(tile_counter + lane * RT_QUEUE_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        left = len_k - first_entries(lane, heavy_blocks) - (int)min(taken, 0x3fffffffu);
    }
    return (unsigned int)__builtin_amdgcn_ballot_w64(left > 0) & 0xFFu;
}

template <bool kStats, bool kGlobalTables = false, bool kClusters = false, bool kRoomy = false, bool kFast = false>
__device__ __forceinline__ void render_body(const RtParams &p, const float4 *__restrict__ image,
                                            float *__restrict__ out, unsigned int *__restrict__ tile_counter,
                                            float4 *__restrict__ bounce_stack,
                                            unsigned long long *__restrict__ stats_out,
                                            unsigned int *__restrict__ help_area) {
    extern __shared__ float4 wlds[];                          /* LDS: the tables, the low levels of the bounce stack, the HELP desk */
    Stats<kStats> st;
    if constexpr (kStats) {
#pragma unroll
        for (int k = 0; k < ST_COUNT; ++k) st.c[k] = 0u;
    }

    /* stage the scene tables: global -> LDS, once per workgroup.  The large-scene kernel
     * (kGlobalTables) leaves them where they are: every table read of the scans has the same
     * address in all lanes of a wavefront, so it is one 16-byte request to the XCD's L2, which
     * holds a scene of any size the ABI admits -- no capacity limit, and LDS (hence occupancy)
     * is spent on the bounce stack only. */
    const float4 *lds = kGlobalTables ? image : wlds;
    /* The queue heads of the NEXT launch of this scene (RtParams::next_counters: another block of counters) are zeroed here,
     * by nine threads of workgroup 0: launches of one scene are stream-ordered (rt_capi.hip, launch()), so nobody uses that
     * block while this kernel runs, and the next launch finds it at zero -- no fill kernel and no dependency on one in front
     * of every launch (3-4 us and a gap each, which a strip rendered in eight 100 us chunks pays eight times).  (Counting the
     * wavefronts out and letting the last one re-zero this launch's own heads was measured first: 7 168 atomics on one word
     * at the end of a 0.75 ms frame cost 0.1 ms.) */
    if (blockIdx.x == 0 && threadIdx.x <= RT_TILE_QUEUES)
        reinterpret_cast<unsigned int *>(p.next_counters)[threadIdx.x * RT_QUEUE_STRIDE] = 0u;
    /* FAST tables: the items' control words are read from the image in global memory (scalar loads) */
    const uint32_t *__restrict__ ctl_words = reinterpret_cast<const uint32_t *>(image) + (kFast ? p.fast_ctl_off : 0);
    /* HELP: the clustered-scene kernels get the workgroups' ray areas (128 quads each) */
    constexpr bool kHelp = kClusters;
    float4 *help_rays = kHelp ? reinterpret_cast<float4 *>(help_area) : nullptr;
    if constexpr (kHelp) {
        if (p.help_rays_quads != 0 && threadIdx.x < RT_DESK_WORDS) reinterpret_cast<uint32_t *>(wlds + p.desk_off)[threadIdx.x] = 0u;
    }
    if constexpr (!kGlobalTables) {
        for (int q = threadIdx.x; q < p.image_quads; q += blockDim.x) wlds[q] = image[q];
        if constexpr (kFast) {               /* this launch's PRIMARY table comes with the kernel arguments */
            if ((int)threadIdx.x < p.n_primary) wlds[p.primary_off + threadIdx.x] = reinterpret_cast<const float4 *>(p.primary)[threadIdx.x];
        }
        __syncthreads();
    }

    /* Bounce stack, [level][threadIdx.x], one 16-byte entry per reflective level
     * per lane.  The lowest levels -- the ones nearly every chain uses -- live in
     * LDS behind the scene tables, as many as fit while seven workgroups per CU
     * still do; deeper levels go to this workgroup's slice of an HBM buffer,
     * written and read coalesced. */

    /* Self-scheduling (the reference's strategy 2, src/RayTracer.cpp:956-992:
     * a shared queue of pixels; here queues of wavefront tiles).  The grid is
     * only as large as the chip can hold and every WAVEFRONT pulls its next
     * tile until the tiles run out, so expensive tiles (the horizon, mirror
     * balls) cannot pile up the way a static block -> tile map lets them.
     *
     * XCD-aware: there is one queue per XCD (8 counters on separate cache
     * lines, 8x less contention than one word).  The image is cut into MACRO
     * tiles of RT_MACRO_ROWS vertically adjacent wavefront tiles; macro tile m
     * belongs to queue m mod 8 -- horizontally adjacent macro tiles go to
     * different XCDs, so costly image regions are dealt evenly -- and a queue
     * hands its macro tiles out tile by tile.  The vertically adjacent tiles of
     * a macro tile are therefore rendered at about the same time by wavefronts
     * of ONE XCD, and their 48-byte column segments merge into whole 64-byte
     * sectors in that XCD's L2 before they leave for HBM.  A wavefront whose
     * own queue is empty steals from the other XCDs' queues (placement is a
     * speed matter only; any XCC_ID value gives the same image).  All lanes are
     * active here and every wavefront walks all 8 queues to their end, so the
     * grid always drains.
     */
    const int lane = (int)(threadIdx.x & 63u);
    const int my_xcc = (int)(__builtin_amdgcn_s_getreg(RT_GETREG_XCC_ID) & 7u);
  {
    /* HEAVY tiles first, one per workgroup: the tiles on the horizon line of a scene with clustered sphere runs keep
     * ONE wavefront busy for a millisecond or more (their shadow rays start tens of thousands of units away, where
     * the reference's float sphere test is so coarse that every sphere is a legitimate candidate: 10 scans x 64
     * leaves x 16 members for all 64 rays), which is what a GPU's strip of a multi-GPU frame then waits for.  The
     * host names them (a band of tile rows along the horizon line, RtParams::heavy_*); wavefront 0 of a workgroup
     * renders one at a time while the workgroup's other wavefronts stand at the desk and share every long shadow
     * scan from the first one on -- the HELP protocol with helpers that are there from the start.  The ordinary
     * tile queues skip the band.  (One loop hands out both kinds of tile, so that render_tile() is inlined once.) */
    int heavy_phase = 0;               /* 0: ordinary tiles; 1: this workgroup's first HEAVY tile (number blockIdx.x, no atomic); 2: further ones */
    int heavy_blocks = 0;              /* the launch's first workgroups, which start with a HEAVY tile each (first_entries()) */
    if constexpr (kHelp) {
        if (p.help_rays_quads != 0 && p.heavy_half >= 0) heavy_blocks = min((int)gridDim.x, (2 * p.heavy_half + 1) * p.tiles_x);
        if ((int)blockIdx.x < heavy_blocks) {
            uint32_t *desk = reinterpret_cast<uint32_t *>(wlds + p.desk_off);
            /* (the wavefront's number as a scalar: a condition on threadIdx.x counts as divergent, and with it everything the tile
             * loop carries -- queue, pop, phase -- would live in vector registers, spilled across every tile) */
            if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0) {
                heavy_phase = 1;
                if (lane == 0) desk_write(desk, RT_DESK_DEDICATED, 1u);
            } else {
                for (int spins = 0; spins < RT_HELP_SPIN_LIMIT; ++spins) {
                    if (desk_read(desk, RT_DESK_PHASE) != 0u || desk_read(desk, RT_DESK_BROKEN) != 0u) break;
                    if (desk_read(desk, RT_DESK_STATE) != (uint32_t)RT_DESK_OPEN) { __builtin_amdgcn_s_sleep(8); continue; }
                    serve_desk<kStats>(p, lds, desk, help_rays, st);
                }
            }
        }
    }
    const int macro_rows = (p.tiles_z + RT_MACRO_ROWS - 1) / RT_MACRO_ROWS;
    const int n_macros = macro_rows * p.tiles_x;
    /* ask for the following tile while this one is rendered; the answer is only needed afterwards.  Not so in scenes
     * with clustered sphere runs, whose tiles take from tens of microseconds to milliseconds: a tile asked for
     * ahead of a long one waits for it while other wavefronts idle (a strip's timeline showed tiles STARTING a
     * millisecond after the queues had run dry); there the next tile is asked for when this one is done */
    const bool ask_ahead = !(kClusters || kStats) || p.n_clusters == 0;     /* the plain kernels: always */
    /* FIRST TILES WITHOUT THE QUEUES (above first_entries()): the wavefront starts on queue b mod 8 -- its own XCD's when
     * workgroups are dealt to the XCDs in turn, which nothing here relies on -- as if it had just been handed the entry
     * (b / 8) * wavefronts + w; from there on the loop is the one it always was */
    const unsigned int ordinary_block = blockIdx.x - (unsigned int)heavy_blocks;             /* (wraps for the HEAVY workgroups, which ask the queues) */
    const int home = (int)((ordinary_block - (unsigned int)my_xcc) & (RT_TILE_QUEUES - 1));     /* queue b mod 8, counted from this XCD's */
    int steal = home;
    int next_pop = (int)(ordinary_block >> 3) * (int)(blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) -
                   first_entries((int)(ordinary_block & (RT_TILE_QUEUES - 1)), heavy_blocks);
    unsigned int candidates = ~0u;     /* the other queues that had tiles when this wavefront's own ran dry (~0: not looked yet) */
    bool fresh = (int)blockIdx.x < heavy_blocks;       /* the current queue has not been asked yet */
    for (;;) {
        int wave;                      /* tile number, row-major */
        if (kHelp && heavy_phase != 0) {
            unsigned int *const heavy_head = tile_counter + RT_TILE_QUEUES * RT_QUEUE_STRIDE;
            int h = (int)blockIdx.x;
            if (heavy_phase == 2) {
                if (lane == 0) h = (int)atomicAdd(heavy_head, 1u);
                h = __builtin_amdgcn_readfirstlane(h) + heavy_blocks;
            }
            heavy_phase = 2;
            if (h >= (2 * p.heavy_half + 1) * p.tiles_x) {              /* the band is done: on to the ordinary tiles */
                heavy_phase = 0;
                uint32_t *desk = reinterpret_cast<uint32_t *>(wlds + p.desk_off);
                if (lane == 0) {
                    desk_write(desk, RT_DESK_DEDICATED, 0u);
                    desk_write(desk, RT_DESK_PHASE, 1u);
                }
                continue;
            }
            /* the rows of the band from the horizon line outwards: 0, +1, -1, +2, -2, ... */
            const int k = h / p.tiles_x, tile_col = h - k * p.tiles_x;
            const int offset = (k & 1) ? (k + 1) / 2 : -(k / 2);
            const int tile_row = ((p.heavy_row0_q16 + tile_col * p.heavy_slope_q16) >> 16) + offset;
            if (tile_row < 0 || tile_row >= p.tiles_z) continue;
            wave = tile_row * p.tiles_x + tile_col;
        } else {
            /* (a full circle from the queue it started on) */
            if (steal >= RT_TILE_QUEUES + home) break;
            const int queue = (my_xcc + steal) & (RT_TILE_QUEUES - 1);
            unsigned int *const head = tile_counter + queue * RT_QUEUE_STRIDE;
            /* macro tiles queue, queue + 8, queue + 16, ... */
            const int queue_len = queue < n_macros ? ((n_macros - queue + RT_TILE_QUEUES - 1) / RT_TILE_QUEUES) * RT_MACRO_ROWS : 0;
            if (fresh) {
                if (lane == 0) next_pop = (int)atomicAdd(head, 1u);
                fresh = false;
            }
            /* a head counts the entries handed out beyond the wavefronts' first ones */
            const int pop = __builtin_amdgcn_readfirstlane(next_pop) + first_entries(queue, heavy_blocks);
            if (pop >= queue_len) {
                /* This queue is through.  Which of the others still have tiles is found by ONE look at all the heads, when this
                 * wavefront's own queue runs dry: an atomic on each exhausted queue in turn cost 3-4 us per queue -- 25 us
                 * between a wavefront's last tile and its exit, or before the tile it finally found (a built-in strip's
                 * timeline: the tiles that started last had waited that long for their wavefront to come by).  Heads only
                 * grow: a queue seen empty stays empty, and the ones seen with tiles are then tried in ring order. */
                if constexpr (kClusters || kStats || kGlobalTables) {
                    /* (the clustered-scene kernels: tiles of 100 us, the walk is nothing next to them, and the few registers
                     * of the look cost their frames 2.5 %; the large-scene kernel's 2 %) */
                    ++steal;
                } else {
                    if (candidates == ~0u) {
                        const unsigned int nonempty = queues_with_tiles(tile_counter, n_macros, heavy_blocks);                   /* bit q: queue q */
                        candidates = ((nonempty >> my_xcc) | (nonempty << (RT_TILE_QUEUES - my_xcc))) & 0xFFu & ~(1u << (steal & 7));     /* bit k: queue my_xcc + k; not the one just found empty */
                    }
                    if (candidates == 0u) break;
                    steal = __builtin_ctz(candidates);
                    candidates &= candidates - 1u;
                }
                fresh = true;
                continue;
            }
            if (ask_ahead && lane == 0) next_pop = (int)atomicAdd(head, 1u);
            const int macro = (pop / RT_MACRO_ROWS) * RT_TILE_QUEUES + queue;
            const int queued_row = macro / p.tiles_x;
            const int tile_col = macro - queued_row * p.tiles_x;
            /* from first_macro_row (< macro_rows) upwards, or (rows_downwards) downwards; both wrap around */
            const int shifted_row = p.rows_downwards ? p.first_macro_row - queued_row : p.first_macro_row + queued_row;
            const int macro_row = shifted_row >= macro_rows ? shifted_row - macro_rows : (shifted_row < 0 ? shifted_row + macro_rows : shifted_row);
            const int tile_row = macro_row * RT_MACRO_ROWS + (pop % RT_MACRO_ROWS);
            bool skip = tile_row >= p.tiles_z;                      /* ragged top macro row */
            if constexpr (kHelp) {                                  /* a HEAVY tile: rendered by a workgroup, above */
                if (p.help_rays_quads != 0 && p.heavy_half >= 0) {
                    const int off_line = tile_row - ((p.heavy_row0_q16 + tile_col * p.heavy_slope_q16) >> 16);
                    skip = skip || (off_line >= -p.heavy_half && off_line <= p.heavy_half);
                }
            }
            if (skip) {
                if (!ask_ahead && lane == 0) next_pop = (int)atomicAdd(head, 1u);
                continue;
            }
            wave = tile_row * p.tiles_x + tile_col;
        }
#ifdef RT_TIMELINE
        /* diagnostic builds (make variant DEFS=-DRT_TIMELINE): when this tile was started ... (in the product kernels even these
         * few lines cost registers: six more spilled in the plain kernel, twenty in the clustered-scene one) */
        if (p.timeline != 0ull && lane == 0)
            reinterpret_cast<unsigned long long *>(p.timeline)[(size_t)wave * RT_TIMELINE_WORDS] = __builtin_amdgcn_s_memrealtime();
        const int tile_number = here(wave);
#endif
        /* (the kernels that do not ask ahead ask inside, when the tile's rays are through -- not for a HEAVY tile, whose
         * successor comes from the HEAVY tiles' own head) */
        unsigned int *const ask_head = (ask_ahead || (kHelp && heavy_phase != 0)) ? nullptr
                                     : tile_counter + ((my_xcc + steal) & (RT_TILE_QUEUES - 1)) * RT_QUEUE_STRIDE;
        render_tile<kStats, kFast ? 6 : (kClusters ? (kRoomy ? 5 : 4) : 0)>(p, lds, wlds, help_rays, ctl_words, out, bounce_stack, stats_out, st, wave, my_xcc, steal,
                                                                           next_pop, ask_head);
#ifdef RT_TIMELINE
        if (p.timeline != 0ull && lane == 0) {                   /* ... when it was done, and by whom */
            unsigned long long *rec = reinterpret_cast<unsigned long long *>(p.timeline) + (size_t)tile_number * RT_TIMELINE_WORDS;
            rec[1] = __builtin_amdgcn_s_memrealtime();
            rec[2] = (unsigned long long)blockIdx.x * 16ull + (threadIdx.x >> 6);
            rec[3] = (kHelp && heavy_phase != 0) ? 1ull : 0ull;
        }
#endif
    }
    if constexpr (kHelp) {
        /* HELP: out of tiles -- serve the colleagues until they are, too */
        if (p.help_rays_quads != 0) {
            uint32_t *desk = reinterpret_cast<uint32_t *>(wlds + p.desk_off);
            const uint32_t n_waves = blockDim.x >> 6;
            if (lane == 0) atomicAdd(desk + RT_DESK_FINISHED, 1u);
            for (int spins = 0; spins < RT_HELP_SPIN_LIMIT; ++spins) {
                if (desk_read(desk, RT_DESK_FINISHED) >= n_waves || desk_read(desk, RT_DESK_BROKEN) != 0u) break;
                if (desk_read(desk, RT_DESK_STATE) != (uint32_t)RT_DESK_OPEN) { __builtin_amdgcn_s_sleep(8); continue; }
                serve_desk<kStats>(p, lds, desk, help_rays, st);
            }
        }
    }
  }
    if constexpr (kStats) {
#pragma unroll
        for (int k = 0; k < ST_COUNT; ++k)
            if (st.c[k]) atomicAdd(&stats_out[k], (unsigned long long)st.c[k]);
    }
}

/* The kernels read RtParams where they use it, through the kernarg segment pointer (the struct
 * is the first kernel argument): handed to the body as a by-value argument, its ~70 dwords are
 * all loaded at kernel entry and stay live in scalar registers, which left the scans to spill 85
 * SGPRs to VGPR lanes (v_writelane / v_readlane plus their wait states in the loops); read on
 * demand (s_load through the scalar cache) the kernel spills a handful and needs next to no scratch. */
#define RT_PARAMS_FROM_KERNARG(name, by_value)                                                      \
    (void)by_value;                                                                                 \
    const RtParams &name = *(const RtParams *)(__builtin_amdgcn_kernarg_segment_ptr())

/* All render kernels share one signature.  `help_area`: the clustered-scene kernels' HELP areas (128 quads of
 * global memory per workgroup for the rays a wavefront publishes at its workgroup's desk); unused by the others. */
#define RT_KERNEL_ARGS                                                                                            \
    const RtParams p_in_kernarg, const float4 *__restrict__ image, float *__restrict__ out,                      \
    unsigned int *__restrict__ tile_counter, float4 *__restrict__ bounce_stack, unsigned int *__restrict__ help_area
#define RT_KERNEL_ARGS_STATS                                                                                      \
    const RtParams p_in_kernarg, const float4 *__restrict__ image, float *__restrict__ out,                      \
    unsigned int *__restrict__ tile_counter, float4 *__restrict__ bounce_stack,                                   \
    unsigned long long *__restrict__ stats_out, unsigned int *__restrict__ help_area

/* Scenes without clustered sphere runs, FAST tables (the built-in scene: the bench headline).  72 VGPRs: seven
 * wavefronts per SIMD where LDS allows (the bounce stack keeps its LDS place up to seven workgroups per CU,
 * RT_STACK_LDS_SHARE) */
#ifndef RT_WAVES_PER_SIMD
#define RT_WAVES_PER_SIMD 7
#endif
extern "C" __global__ void __launch_bounds__(RT_BLOCK_BOUND, RT_WAVES_PER_SIMD)
rt_render_kernel(RT_KERNEL_ARGS) {
#ifdef RT_FAST_PARAMS_BY_VALUE
    const RtParams &p = p_in_kernarg;
#else
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
#endif
    render_body<false, false, false, false, true>(p, image, out, tile_counter, bounce_stack, nullptr, help_area);
}

/* the same over the two item tables: option "fast" = 0, and option "cull" = 0 (the plain in-order scans) */
extern "C" __global__ void __launch_bounds__(RT_BLOCK_BOUND, RT_WAVES_PER_SIMD)
rt_render_kernel_items(RT_KERNEL_ARGS) {
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
    render_body<false>(p, image, out, tile_counter, bounce_stack, nullptr, help_area);
}

/* scenes whose tables are large (or do not fit LDS at all): the tables stay in global memory */
extern "C" __global__ void __launch_bounds__(RT_BLOCK_BOUND, RT_WAVES_PER_SIMD)
rt_render_kernel_large(RT_KERNEL_ARGS) {
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
    render_body<false, true>(p, image, out, tile_counter, bounce_stack, nullptr, help_area);
}

/* scenes with clustered sphere runs (PAIRS, NEAREST PAIRS, HELP, HEAVY tiles): 80 registers, six wavefronts per SIMD */
#ifndef RT_WAVES_PER_SIMD_CLUSTERS
#define RT_WAVES_PER_SIMD_CLUSTERS 6
#endif
/* (workgroups of up to eight wavefronts: scenes whose tables are large share one LDS copy among more of them, launch() in rt_capi.hip) */
#ifndef RT_BLOCK_BOUND_CLUSTERS
#define RT_BLOCK_BOUND_CLUSTERS 512
#endif
extern "C" __global__ void __launch_bounds__(RT_BLOCK_BOUND_CLUSTERS, RT_WAVES_PER_SIMD_CLUSTERS)
rt_render_kernel_clusters(RT_KERNEL_ARGS) {
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
    render_body<false, false, true>(p, image, out, tile_counter, bounce_stack, nullptr, help_area);
}

/* the same with the registers of five wavefronts per SIMD, for scenes whose tables leave room for no more than
 * five workgroups per CU anyway (the 1 024-sphere grid: 31.5 KB); the pair flush tests four members abreast here */
#ifndef RT_WAVES_PER_SIMD_WIDE
#define RT_WAVES_PER_SIMD_WIDE 5
#endif
extern "C" __global__ void __launch_bounds__(RT_BLOCK_BOUND_CLUSTERS, RT_WAVES_PER_SIMD_WIDE)
rt_render_kernel_clusters_wide(RT_KERNEL_ARGS) {
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
    render_body<false, false, true, true>(p, image, out, tile_counter, bounce_stack, nullptr, help_area);
}

/* the counting builds (rt_render_stats): same arithmetic and control flow plus work counters.  One for the item
 * tables -- with and without clustered runs: the clustered-scene body, which is the plain one when a scene has no
 * leaves -- and one for the FAST tables */
extern "C" __global__ void __launch_bounds__(512)
rt_render_kernel_stats(RT_KERNEL_ARGS_STATS) {
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
    render_body<true, false, true>(p, image, out, tile_counter, bounce_stack, stats_out, help_area);
}

extern "C" __global__ void __launch_bounds__(512)
rt_render_kernel_fast_stats(RT_KERNEL_ARGS_STATS) {
    RT_PARAMS_FROM_KERNARG(p, p_in_kernarg);
    render_body<true, false, false, false, true>(p, image, out, tile_counter, bounce_stack, stats_out, help_area);
}
