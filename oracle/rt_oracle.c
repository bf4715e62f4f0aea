/*
 * rt_oracle.c -- plain-C, single-threaded, literal restatement of the render
 * path of ccelio/TileCodeRayTracer (float mode, x86, PARTIONING_STRATEGY 0).
 *
 * TEST INFRASTRUCTURE ONLY -- see rt_oracle.h for who may use this and for
 * the parity status ("parity unpinned" by the strict definition; checked
 * against the digests recorded in SURVEY.md Appendix D).
 *
 * Build:  gcc -O2 -ffp-contract=off -fno-fast-math (oracle/Makefile).
 * The arithmetic is IEEE-754 binary32 in exactly the reference's operation
 * order; do not "simplify" any expression in this file.  Like the reference,
 * it builds a complete hit record for EVERY candidate object and recurses.
 *
 * Defined semantics for the one undefined read in the reference:
 * CollisionObject::hitALightSource_Var is never initialised
 * (src/SceneObject.h:47-105) and only ever set to true
 * (src/SceneSphere.cpp:163-164, src/SceneInfinitePlane.cpp:104-105,
 * src/SceneFinitePlane.cpp:159-160); here it is false unless the hit object
 * is a light.
 */
#include "rt_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAX_OBJECT_COUNT 4000      /* src/Scene.h:8 */
#define ORC_FLOAT_MAX_VALUE  65535     /* src/rt_project_parameters.h:74 */

typedef orc_vec3 vec3;

/* ---------------------------------------------------------------- vector3d
 * src/vector3d.h:48-124 */
static inline vec3 v3(float x, float y, float z) { vec3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline vec3 v_add(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }   /* :115 */
static inline vec3 v_sub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }   /* :117 */
static inline vec3 v_scale(vec3 v, float f) { return v3(v.x * f, v.y * f, v.z * f); }       /* :119-122 */
static inline vec3 v_mul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }   /* :123 */
static inline vec3 v_neg(vec3 v) { return v3(0 - v.x, 0 - v.y, 0 - v.z); }                 /* :111 */
static inline float v_dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }    /* :97 */
static inline float v_length(vec3 v) { return sqrtf(v.x * v.x + v.y * v.y + v.z * v.z); }  /* :75-85 */
static inline vec3 v_normalize(vec3 v) {                                                   /* :55-73 */
    float length = sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return v3(v.x / length, v.y / length, v.z / length);
}
static inline vec3 v_cross(vec3 v1, vec3 v2) {                                             /* :101-104 */
    return v3(v1.y * v2.z - v1.z * v2.y,
              v1.z * v2.x - v1.x * v2.z,
              v1.x * v2.y - v1.y * v2.x);
}

/* --------------------------------------------------------------------- Ray
 * src/Ray.h:15-30 : both non-default constructors normalise the direction */
typedef struct ray { vec3 origin, direction; } ray;
static inline ray ray_make(vec3 o, vec3 d) { ray r; r.origin = o; r.direction = v_normalize(d); return r; }
static inline ray ray_between(vec3 o, vec3 pf, vec3 pi) {
    ray r; r.origin = o; r.direction = v_normalize(v_sub(pf, pi)); return r;
}

/* ------------------------------------------------------------ Color_Values
 * src/Color_Values.h:7-17 */
static const vec3 COLOR_WHITE      = {1.f, 1.f, 1.f};
static const vec3 COLOR_RED        = {1.f, 0.f, 0.f};
static const vec3 COLOR_YELLOW     = {1.f, 1.f, 0.f};
static const vec3 COLOR_GREEN      = {0.f, 1.f, 0.f};
static const vec3 COLOR_CYAN       = {0.f, 1.f, 1.f};
static const vec3 COLOR_BLUE       = {0.f, 0.f, 1.f};
static const vec3 COLOR_BLACK      = {0.f, 0.f, 0.f};
static const vec3 COLOR_DARK_GREY  = {0.33f, 0.33f, 0.33f};
static const vec3 COLOR_LIGHT_GREY = {2 / 3.f, 2 / 3.f, 2 / 3.f};
static const vec3 COLOR_BROWN      = {0.2f, 0.2f, 0.0f};

static const vec3 NULL_COLOR = {0.75f, 0.75f, 0.75f};   /* src/RayTracer.h:52 */

/* ------------------------------------------------------------------- Scene */
struct orc_scene {
    orc_object *objects;              /* src/Scene.cpp:16 */
    int object_count;
    int scene_object_start_index;     /* src/Scene.h:41-42; zero for a static Scene */
    int scene_object_final_index;
};

orc_scene *orc_scene_new(void) {
    orc_scene *s = (orc_scene *)calloc(1, sizeof(*s));
    if (!s) return NULL;
    s->objects = (orc_object *)calloc(ORC_MAX_OBJECT_COUNT, sizeof(orc_object));
    if (!s->objects) { free(s); return NULL; }
    return s;
}
void orc_scene_free(orc_scene *s) { if (s) { free(s->objects); free(s); } }
int orc_scene_object_count(const orc_scene *s) { return s->object_count; }
int orc_scene_get_object(const orc_scene *s, int i, orc_object *out) {
    if (i < 0 || i >= s->object_count) return 1;
    *out = s->objects[i];
    return 0;
}
int orc_scene_shadow_range(const orc_scene *s, int *begin, int *end) {
    *begin = s->scene_object_start_index; *end = s->scene_object_final_index; return 0;
}

/* Scene::addObject, src/Scene.cpp:470-479 (capacity quirk: max 3999) */
static int scene_add(orc_scene *s, const orc_object *o) {
    if (s->object_count + 1 >= ORC_MAX_OBJECT_COUNT) return -1;
    s->objects[s->object_count] = *o;
    return s->object_count++;
}

/* SceneObject(vector3d) + ObjMaterial(), src/SceneObject.cpp:21-27,
 * src/ObjMaterial.h:13-21 */
static void object_base(orc_object *o, int kind, vec3 origin) {
    memset(o, 0, sizeof(*o));
    o->kind = kind;
    o->origin = origin;
    o->color = v3(1.0f, 1.0f, 1.0f);
    o->diffuse = 1.0f;
    o->specular = 1.0f;
    o->reflective = 0;
    o->has_texture = 0;
    o->is_light = 0;
    o->intensity = 1.0f;
}

/* SceneSphere(vector3d, radius), src/SceneSphere.cpp:44-48 */
int orc_add_sphere(orc_scene *s, orc_vec3 origin, float radius) {
    orc_object o;
    object_base(&o, ORC_SPHERE, origin);
    o.radius = radius;
    o.radius_squared = radius * radius;
    return scene_add(s, &o);
}

/* SceneInfinitePlane(o, n, h), src/SceneInfinitePlane.cpp:11-26 */
int orc_add_infinite_plane(orc_scene *s, orc_vec3 _origin, orc_vec3 _normal, orc_vec3 _horizontal) {
    orc_object o;
    object_base(&o, ORC_INFINITE_PLANE, _origin);
    o.normal = v_normalize(_normal);
    o.horizontal = v_normalize(_horizontal);
    o.vertical = v_normalize(v_cross(o.normal, o.horizontal));
    o.reverse_normal = v_neg(o.normal);                 /* not re-normalised here */
    o.distance_to_origin = -v_dot(o.origin, o.normal);
    return scene_add(s, &o);
}

/* SceneFinitePlane(o, n, h, v_dist, h_dist), src/SceneFinitePlane.cpp:18-47 */
int orc_add_finite_plane_axes(orc_scene *s, orc_vec3 _origin, orc_vec3 _normal, orc_vec3 _horizontal,
                              float v_dist, float h_dist) {
    orc_object o;
    object_base(&o, ORC_FINITE_PLANE, _origin);
    o.plane_origin = _origin;
    o.vertical = v_cross(_normal, _horizontal);          /* cross of the un-normalised inputs */
    o.normal = v_normalize(_normal);
    o.horizontal = v_normalize(_horizontal);
    o.vertical = v_normalize(o.vertical);
    o.reverse_normal = v_normalize(v_neg(o.normal));
    o.v_distance = v_dist;
    o.h_distance = h_dist;
    o.distance_to_origin = -v_dot(_origin, o.normal);
    return scene_add(s, &o);
}

/* SceneFinitePlane(o, vertical_corner, horizontal_corner), src/SceneFinitePlane.cpp:49-80 */
int orc_add_finite_plane_corners(orc_scene *s, orc_vec3 _origin, orc_vec3 _vertical_corner,
                                 orc_vec3 _horizontal_corner) {
    orc_object o;
    vec3 horizontal, vertical, normal, h, new_super_origin;
    object_base(&o, ORC_FINITE_PLANE, _origin);
    o.plane_origin = _origin;
    horizontal = v_sub(_horizontal_corner, _origin);
    vertical = v_sub(_vertical_corner, _origin);
    normal = v_cross(horizontal, vertical);
    o.v_distance = v_length(vertical);
    o.h_distance = v_length(horizontal);
    o.vertical = v_normalize(vertical);
    o.horizontal = v_normalize(horizontal);
    o.normal = v_normalize(normal);
    o.reverse_normal = v_normalize(v_neg(o.normal));
    o.distance_to_origin = -v_dot(_origin, o.normal);
    /* the SceneObject origin becomes the far corner (:74-79) */
    h = v_scale(o.horizontal, o.h_distance);
    new_super_origin = v_add(v_add(v_scale(o.vertical, o.v_distance), o.plane_origin), h);
    o.origin = new_super_origin;
    return scene_add(s, &o);
}

#define CHECK_IDX(s, i) do { if ((i) < 0 || (i) >= (s)->object_count) return 1; } while (0)
int orc_set_color(orc_scene *s, int i, orc_vec3 c) { CHECK_IDX(s, i); s->objects[i].color = c; return 0; }
int orc_set_diffuse(orc_scene *s, int i, float f) { CHECK_IDX(s, i); s->objects[i].diffuse = f; return 0; }
int orc_set_specular(orc_scene *s, int i, float f) { CHECK_IDX(s, i); s->objects[i].specular = f; return 0; }
int orc_set_reflective(orc_scene *s, int i, float f) { CHECK_IDX(s, i); s->objects[i].reflective = f; return 0; }
int orc_set_light(orc_scene *s, int i) { CHECK_IDX(s, i); s->objects[i].is_light = 1; return 0; }
int orc_set_intensity(orc_scene *s, int i, float f) { CHECK_IDX(s, i); s->objects[i].intensity = f; return 0; }
/* Texture_CheckerBoard(l, d) + setWidth/setHeight, src/Texture_CheckerBoard.h:23-29,
 * src/ObjTexture.h:33-47 */
int orc_set_checkerboard(orc_scene *s, int i, orc_vec3 light, orc_vec3 dark, float w, float h) {
    CHECK_IDX(s, i);
    s->objects[i].has_texture = 1;
    s->objects[i].tex_light = light;
    s->objects[i].tex_dark = dark;
    s->objects[i].tex_width = w;
    s->objects[i].tex_height = h;
    return 0;
}

/* Scene::SetObjectIndices, src/Scene.cpp:486-504 */
void orc_set_object_indices(orc_scene *s, int my_rank, int group_size) {
    int new_object_count = s->object_count / group_size;
    int start_index = my_rank * new_object_count;
    if (my_rank == group_size - 1) new_object_count = s->object_count - start_index;
    s->scene_object_start_index = start_index;
    s->scene_object_final_index = start_index + new_object_count;
}

/* Scene::makeSceneBox, src/Scene.cpp:392-416 */
static void make_scene_box(orc_scene *s, vec3 _origin, vec3 _dims, int idx[6]) {
    vec3 c0 = _origin;
    vec3 c1 = _origin; c1.x += _dims.x;
    vec3 c2 = _origin; c2.y += _dims.y;
    vec3 c3 = _origin; c3.z += _dims.z;
    vec3 c4 = _origin; c4.x += _dims.x; c4.y += _dims.y;
    vec3 c5 = _origin; c5.x += _dims.x; c5.z += _dims.z;
    vec3 c6 = _origin; c6.y += _dims.y; c6.z += _dims.z;
    vec3 c7 = v3(_origin.x + _dims.x, _origin.y + _dims.y, _origin.z + _dims.z);
    idx[0] = orc_add_finite_plane_corners(s, c0, c3, c2);
    idx[1] = orc_add_finite_plane_corners(s, c0, c3, c1);
    idx[2] = orc_add_finite_plane_corners(s, c0, c1, c2);
    idx[3] = orc_add_finite_plane_corners(s, c7, c4, c6);
    idx[4] = orc_add_finite_plane_corners(s, c7, c4, c5);
    idx[5] = orc_add_finite_plane_corners(s, c7, c5, c6);
}

/* Scene::initialize, src/Scene.cpp:209-387.  In the reference the six planes
 * of a box are constructed first and appended one by one; since nothing else
 * is appended in between, appending at construction gives the same order. */
int orc_scene_initialize(orc_scene *s) {
    int i, k, b[6];
    /* lights :233-251 */
    i = orc_add_sphere(s, v3((float)6.99, (float)6.99, (float)5.5), (float).15);
    orc_set_light(s, i); orc_set_intensity(s, i, (float).75);
    i = orc_add_sphere(s, v3(0, 0, (float)4.8), (float).15);
    orc_set_light(s, i); orc_set_intensity(s, i, (float)1.0);
    /* balls :254-262 */
    i = orc_add_sphere(s, v3(0, 0, 2), 1);
    orc_set_color(s, i, COLOR_RED); orc_set_reflective(s, i, 1.00f);
    i = orc_add_sphere(s, v3(0, 0, 0), (float)0.01);
    /* :266-283 (the i==2 branch is dead) */
    for (k = 0; k < 2; k++) {
        i = orc_add_sphere(s, v3((float)(-2.5 + ((k + 0) * 2.5)), 3, 1), 1);
        orc_set_color(s, i, COLOR_RED);
        if (k == 1) {
            orc_set_color(s, i, COLOR_WHITE);
            orc_set_reflective(s, i, 1.00f);
            orc_set_diffuse(s, i, 0.00f);
        } else {
            orc_set_specular(s, i, (float).5);
        }
    }
    /* origin ball :287-291 */
    i = orc_add_sphere(s, v3(0, 0, 0), (float).10);
    orc_set_color(s, i, COLOR_CYAN);
    /* ground plane :295-309 */
    i = orc_add_infinite_plane(s, v3(0, 0, 0), v3(0, 0, 1), v3(1, 0, 0));
    orc_set_color(s, i, COLOR_GREEN);
    orc_set_reflective(s, i, (float).5);
    orc_set_diffuse(s, i, (float).5);
    orc_set_checkerboard(s, i, COLOR_WHITE, COLOR_BLACK, 3.0f, 3.0f);
    /* pedestal :313-344 */
    make_scene_box(s, v3(-.5f, -.5f, 0), v3(1.f, 1.f, 1.f), b);
    for (k = 0; k < 6; k++) {
        orc_set_color(s, b[k], COLOR_BROWN);
        orc_set_reflective(s, b[k], 0.00f);
        orc_set_diffuse(s, b[k], 1.00f);
    }
    make_scene_box(s, v3(-.70f, -.70f, 0), v3(1.4f, 1.4f, 0.25f), b);
    for (k = 0; k < 6; k++) {
        orc_set_color(s, b[k], COLOR_BROWN);
        orc_set_specular(s, b[k], 0.20f);
    }
    /* outside walls :348-363 */
    make_scene_box(s, v3(-7, -7, -1), v3(14, 14, 7), b);
    for (k = 0; k < 6; k++) {
        orc_set_color(s, b[k], COLOR_DARK_GREY);
        orc_set_reflective(s, b[k], 0.00f);
        orc_set_diffuse(s, b[k], 1.00f);
        orc_set_specular(s, b[k], 0.0f);
    }
    /* top part of the room :366-380 */
    make_scene_box(s, v3(-6, -6, 5), v3(12, 12, 1), b);
    for (k = 0; k < 6; k++) {
        orc_set_color(s, b[k], COLOR_LIGHT_GREY);
        orc_set_reflective(s, b[k], 0.5f);
        orc_set_specular(s, b[k], 0.5f);
    }
    s->scene_object_start_index = 0;                  /* :383-384 */
    s->scene_object_final_index = s->object_count;
    return 0;
}

/* one float-loop sphere pyramid of Scene::initializeTwoMirrors, src/Scene.cpp:132-203 */
static void pyramid(orc_scene *s, float BASE_X, float BASE_Y, float BASE_Z, float offset,
                    double add_x, int add_y, float radius, vec3 color) {
    float i_start = 0, j_start = 0, i, j, k;
    for (k = 0; k < BASE_Z; k += offset) {
        i_start += offset;
        j_start += offset;
        for (i = i_start; i < BASE_X - i_start; i += offset) {
            for (j = j_start; j < BASE_Y - j_start; j += offset) {
                /* i + <double literal> is evaluated in double, j + <int> in float */
                int idx = orc_add_sphere(s, v3((float)(i + add_x), j + add_y, k), radius);
                if (idx >= 0) orc_set_color(s, idx, color);
            }
        }
    }
}

/* Scene::initializeTwoMirrors, src/Scene.cpp:23-206 */
int orc_scene_initialize_two_mirrors(orc_scene *s, orc_camera *cam) {
    int i;
    i = orc_add_sphere(s, v3(5, 10, 10), (float).15);
    orc_set_light(s, i); orc_set_intensity(s, i, .75f);
    i = orc_add_sphere(s, v3(0, 0, 0), (float).10); orc_set_color(s, i, COLOR_CYAN);
    i = orc_add_sphere(s, v3(-40, 100, 40), 10);
    orc_set_color(s, i, COLOR_YELLOW); orc_set_specular(s, i, (float)0.25);
    i = orc_add_sphere(s, v3(0, 0, 0), (float).05); orc_set_color(s, i, COLOR_CYAN);
    i = orc_add_sphere(s, v3(0, 0, 0), (float).02); orc_set_color(s, i, COLOR_CYAN);
    /* ground :65-81 */
    i = orc_add_infinite_plane(s, v3(0, 0, 0), v3(0, 0, 1), v3(1, 0, 0));
    orc_set_checkerboard(s, i, COLOR_WHITE, COLOR_BLACK, 3.0f, 3.0f);
    orc_set_reflective(s, i, (float).05);
    orc_set_diffuse(s, i, (float).5);
    /* mirror 1 :85-107 */
    i = orc_add_finite_plane_axes(s, v3((float)-1.75, 7, 0), v3(0, -1, 0), v3(1, 0, 0), 5, 3.5f);
    orc_set_color(s, i, COLOR_WHITE); orc_set_reflective(s, i, (float)1.0); orc_set_diffuse(s, i, (float).0);
    i = orc_add_finite_plane_axes(s, v3(-2, 7, 0), v3(0, -1, 0), v3(1, 0, 0), (float)5.25, (float)4.0);
    orc_set_color(s, i, COLOR_BROWN); orc_set_diffuse(s, i, (float).5);
    /* mirror 2 :110-134 */
    i = orc_add_finite_plane_axes(s, v3((float)1.75, -7, 0), v3(0, 1, 0), v3(-1, 0, 0), 5, (float)3.5);
    orc_set_color(s, i, COLOR_WHITE); orc_set_diffuse(s, i, (float).0); orc_set_reflective(s, i, (float)1.0);
    i = orc_add_finite_plane_axes(s, v3(2, -7, 0), v3(0, 1, 0), v3(-1, 0, 0), (float)5.25, (float)4.0);
    orc_set_color(s, i, COLOR_GREEN); orc_set_color(s, i, COLOR_BROWN); orc_set_diffuse(s, i, (float).5);
    /* pyramids :137-203 */
    pyramid(s, 14.50f, 15.0f, 16.5f, 0.5f, 2.65, 15, (float)0.33, COLOR_GREEN);
    pyramid(s, 5.f, 5.0f, 5.f, 0.65f, -6.0, 10, (float).5, COLOR_RED);
    pyramid(s, 1.f, 1.0f, 1.f, 0.33f, 0.0, 20, (float)0.33, COLOR_RED);
    if (cam) orc_camera_two_mirrors(cam);
    s->scene_object_start_index = 0;
    s->scene_object_final_index = s->object_count;
    return 0;
}

/* Synthetic grid-n scene of SURVEY.md section 8(d) / Appendix E, built only
 * through the public constructors/setters above.  Not part of the reference;
 * it exists so that oracle, host model and GPU path build identical scenes. */
int orc_scene_grid(orc_scene *s, int n, int shadows) {
    static const vec3 *const pal[6] = {&COLOR_RED, &COLOR_YELLOW, &COLOR_GREEN,
                                       &COLOR_CYAN, &COLOR_BLUE, &COLOR_WHITE};
    int i, j, idx;
    if (n < 1 || n * n + 4 >= ORC_MAX_OBJECT_COUNT) return 1;
    idx = orc_add_sphere(s, v3(-20.0f, 10.0f, 10.0f), (float).15);
    orc_set_light(s, idx); orc_set_intensity(s, idx, (float).75);
    idx = orc_add_sphere(s, v3(0.0f, 40.0f, 11.0f), (float).15);
    orc_set_light(s, idx); orc_set_intensity(s, idx, (float)1.0);
    for (i = 0; i < n; i++) {
        for (j = 0; j < n; j++) {
            int k = i * n + j;
            float cx = ((float)i - (float)(n - 1) * 0.5f) * 2.5f;
            float cy = 6.0f + (float)j * 2.5f;
            idx = orc_add_sphere(s, v3(cx, cy, 1.0f), 1.0f);
            orc_set_color(s, idx, *pal[k % 6]);
            if (((i + j) & 1) == 0) {
                orc_set_reflective(s, idx, 1.0f);
                orc_set_diffuse(s, idx, 0.0f);
            } else {
                orc_set_specular(s, idx, 0.5f);
            }
        }
    }
    idx = orc_add_infinite_plane(s, v3(0, 0, 0), v3(0, 0, 1), v3(1, 0, 0));
    orc_set_color(s, idx, COLOR_GREEN);
    orc_set_reflective(s, idx, (float).5);
    orc_set_diffuse(s, idx, (float).5);
    orc_set_checkerboard(s, idx, COLOR_WHITE, COLOR_BLACK, 3.0f, 3.0f);
    idx = orc_add_infinite_plane(s, v3(0, 0, 12), v3(0, 0, -1), v3(1, 0, 0));
    orc_set_color(s, idx, COLOR_LIGHT_GREY);
    orc_set_reflective(s, idx, (float).5);
    orc_set_specular(s, idx, (float).5);
    /* Like any scene built with addObject() alone on the reference's static
     * my_scene, the shadow scan range stays [0, 0) (no shadows) unless the
     * caller asks for SetObjectIndices(0, 1), which makes it [0, count). */
    if (shadows) orc_set_object_indices(s, 0, 1);
    return 0;
}

/* ------------------------------------------------------------------ Camera */
static void camera_derive(orc_camera *c) {
    /* src/Camera.cpp:28-39 and :55-66 */
    c->vector_vertical = v_cross(c->vector_horizontal, c->vector_outwards);
    c->vector_outwards = v_normalize(c->vector_outwards);
    c->vector_horizontal = v_normalize(c->vector_horizontal);
    c->vector_vertical = v_normalize(c->vector_vertical);
    c->eye_distance = 1;
    c->eye_origin = v_add(v_scale(c->vector_outwards, -c->eye_distance), c->screen_origin);
}

/* Camera::Camera, src/Camera.cpp:9-40 */
void orc_camera_default(orc_camera *c) {
    memset(c, 0, sizeof(*c));
    c->screen_width = 1;
    c->screen_height = 1;
    c->screen_halfwidth = c->screen_width / (float)2.0;
    c->screen_halfheight = c->screen_height / (float)2.0;
    c->screen_origin = v3(-4.f, -4.f, 1.5f);
    c->vector_horizontal = v3(.1f, -.08f, 0.f);
    c->vector_outwards = v3(.08f, .1f, .01f);
    camera_derive(c);
}

/* Camera::setSceneTwoMirrors, src/Camera.cpp:42-69 (applied to a default Camera) */
void orc_camera_two_mirrors(orc_camera *c) {
    orc_camera_default(c);
    c->screen_origin = v3(0, 0, (float)2.5);
    c->vector_outwards = v3((float).00, 1, (float)-.00);
    c->vector_horizontal = v3(1, (float)-.00, 0);
    camera_derive(c);
}

/* Camera::createEyeRay, src/Camera.cpp:71-84 */
static ray create_eye_ray(const orc_camera *c, float dx_percent, float dy_percent) {
    float scalar_x = dx_percent * c->screen_width - c->screen_halfwidth;
    float scalar_y = dy_percent * c->screen_height - c->screen_halfheight;
    vec3 pixel = v_add(c->screen_origin, v_scale(c->vector_horizontal, scalar_x));
    pixel = v_add(pixel, v_scale(c->vector_vertical, scalar_y));
    return ray_between(c->eye_origin, pixel, c->eye_origin);
}
void orc_camera_eye_ray(const orc_camera *c, float dx, float dy, orc_vec3 *origin, orc_vec3 *dir) {
    ray r = create_eye_ray(c, dx, dy);
    *origin = r.origin; *dir = r.direction;
}

/* --------------------------------------------------------- CollisionObject
 * src/SceneObject.h:36-152 */
typedef struct hit {
    float distance;
    vec3  color;
    ray   normal_ray, reflected_ray;
    vec3  intersection_point;
    int   reflective_material;
    float reflective_factor, specular_factor, diffuse_factor, intensity_factor;
    int   hit_a_light_source;
    int   inside_hit;
} hit;

/* CollisionObject(point, normal, incoming, dist, c, inside, obj), src/SceneObject.h:47-105 */
static void hit_make(hit *h, vec3 point, vec3 normal, vec3 incoming_ray, float dist, vec3 c,
                     int inside, const orc_object *obj) {
    float n_dot_incoming;
    h->normal_ray = ray_make(point, normal);
    h->intersection_point = point;
    h->inside_hit = inside;
    n_dot_incoming = v_dot(normal, incoming_ray);
    h->distance = dist;
    h->color = c;
    h->diffuse_factor = obj->diffuse;
    h->specular_factor = obj->specular;
    h->intensity_factor = obj->intensity;
    h->reflective_factor = obj->reflective;
    h->reflective_material = (h->reflective_factor > (float)0) ? 1 : 0;
    if (h->reflective_material) {
        vec3 reflected = v3(-2 * normal.x * n_dot_incoming + incoming_ray.x,
                            -2 * normal.y * n_dot_incoming + incoming_ray.y,
                            -2 * normal.z * n_dot_incoming + incoming_ray.z);
        h->reflected_ray = ray_make(point, reflected);
    } else {
        h->reflected_ray.origin = v3(0.0f, 0.0f, 0.0f);      /* Ray(), src/Ray.h:15-18 */
        h->reflected_ray.direction = v3(1.f, 0.f, 0.f);
    }
    h->hit_a_light_source = obj->is_light ? 1 : 0;           /* defined semantics, see top */
}

/* Texture_CheckerBoard::getTexturePixel, src/Texture_CheckerBoard.h:31-65 */
static vec3 checkerboard(const orc_object *o, float x, float y) {
    float width = o->tex_width, height = o->tex_height;
    if (x >= 0) x = fmodf(x, width);
    else        x = fmodf((fmodf((-x), width) + width / 2.0f), width);
    if (y >= 0) y = fmodf(y, height);
    else        y = fmodf((fmodf((-y), height) + height / 2.0f), height);
    if (x < width / 2) {
        if (y < height / 2) return o->tex_light;
        else                return o->tex_dark;
    } else {
        if (y < height / 2) return o->tex_dark;
        else                return o->tex_light;
    }
}

static __thread orc_counters g_cnt;
void orc_get_counters(orc_counters *out) { *out = g_cnt; }

/* SceneSphere::collision, src/SceneSphere.cpp:50-168.  Returns 1 and fills *h on a hit. */
static int sphere_collision(const orc_object *o, const ray *eyeRay, hit *h) {
    vec3 OE = v_sub(o->origin, eyeRay->origin);
    float v = v_dot(OE, eyeRay->direction);
    float d_squared, root1, root2, distance;
    int insideHit;
    vec3 intersection_point, normal_ray;
    if (v < (float)0) return 0;
    d_squared = o->radius_squared - (v_dot(OE, OE) - v * v);
    if (d_squared < (float)1E-9) return 0;
    root1 = v - sqrtf(d_squared);
    root2 = v + sqrtf(d_squared);
    insideHit = 0;
    distance = 65535.0f;
    if (root2 > (float)0) {
        if (root1 < (float)0) {
            if (root2 < distance) { distance = root2; insideHit = 1; }
            else return 0;
        } else {
            if (root1 < distance) { distance = root1; insideHit = 0; }
            else return 0;
        }
    } else {
        return 0;
    }
    (void)distance;   /* the reference reports v - sqrt(d_squared), not `distance` (:136-140) */
    intersection_point = v_add(v_scale(eyeRay->direction, (v - sqrtf(d_squared))), eyeRay->origin);
    normal_ray = v_normalize(v_sub(intersection_point, o->origin));
    hit_make(h, intersection_point, normal_ray, eyeRay->direction, (v - sqrtf(d_squared)),
             o->color, insideHit, o);
    return 1;
}

/* SceneInfinitePlane::computeNormal / SceneFinitePlane::computeNormal,
 * src/SceneInfinitePlane.cpp:108-115, src/SceneFinitePlane.cpp:165-172 */
static vec3 plane_compute_normal(const orc_object *o, vec3 eyeDir) {
    if (v_dot(o->normal, eyeDir) < 0) return o->normal;
    return o->reverse_normal;
}

/* SceneInfinitePlane::collision, src/SceneInfinitePlane.cpp:29-106 */
static int infinite_plane_collision(const orc_object *o, const ray *eyeRay, hit *h) {
    float numerator = -o->distance_to_origin - v_dot(eyeRay->origin, o->normal);
    float denom = v_dot(eyeRay->direction, o->normal);
    float t;
    vec3 intersection_point, c, temp_normal, new_intersection_point;
    if (denom == (float)0) return 0;
    t = numerator / denom;
    if (t < (float)1E-10) return 0;
    intersection_point = v_add(v_scale(eyeRay->direction, t), eyeRay->origin);
    c = o->color;
    if (o->has_texture) {
        vec3 PO = v_sub(intersection_point, o->origin);
        float x = v_dot(PO, o->horizontal);
        float y = v_dot(PO, o->vertical);
        c = checkerboard(o, x, y);
    }
    temp_normal = plane_compute_normal(o, eyeRay->direction);
    new_intersection_point = v_add(intersection_point, v_scale(temp_normal, (float)1E-3));
    hit_make(h, new_intersection_point, temp_normal, eyeRay->direction, t, c, 0, o);
    return 1;
}

/* SceneFinitePlane::collision, src/SceneFinitePlane.cpp:86-162 */
static int finite_plane_collision(const orc_object *o, const ray *eyeRay, hit *h) {
    float numerator = -o->distance_to_origin - v_dot(eyeRay->origin, o->normal);
    float denom = v_dot(eyeRay->direction, o->normal);
    float t, x, y;
    vec3 intersection_point, PO, c, temp_normal, new_intersection_point;
    if (denom == 0) return 0;
    t = numerator / denom;
    if (t < 1E-5) return 0;                     /* float promoted to double, :102 */
    intersection_point = v_add(v_scale(eyeRay->direction, t), eyeRay->origin);
    PO = v_sub(intersection_point, o->plane_origin);
    x = v_dot(PO, o->horizontal);
    y = v_dot(PO, o->vertical);
    if (x < 0 || x > o->h_distance || y < 0 || y > o->v_distance) return 0;
    c = o->color;
    if (o->has_texture) c = checkerboard(o, x, y);
    temp_normal = plane_compute_normal(o, eyeRay->direction);
    new_intersection_point = v_add(intersection_point, v_scale(temp_normal, (float)1E-3));
    hit_make(h, new_intersection_point, temp_normal, eyeRay->direction, t, c, 0, o);
    return 1;
}

static int object_collision(const orc_object *o, const ray *r, hit *h) {
    g_cnt.collision_tests++;
    switch (o->kind) {
    case ORC_SPHERE:         return sphere_collision(o, r, h);
    case ORC_INFINITE_PLANE: return infinite_plane_collision(o, r, h);
    default:                 return finite_plane_collision(o, r, h);
    }
}

/* getCollision, src/RayTracer.cpp:50-89 (x86: x_init = 0, x_final = count) */
static int get_collision(const orc_scene *s, const ray *r, hit *nearest) {
    float nearestDist = ORC_FLOAT_MAX_VALUE;
    int found = 0, x;
    hit temp;
    g_cnt.nearest_rays++;
    for (x = 0; x < s->object_count; x++) {
        if (object_collision(&s->objects[x], r, &temp) && temp.distance < nearestDist) {
            nearestDist = temp.distance;
            *nearest = temp;
            found = 1;
        }
    }
    return found;
}

/* inShadeCollisionDetection, src/RayTracer.cpp:709-739.  On x86 my_local_rank
 * is a zero-initialised global equal to FIEFDOM_MASTER_RANK, so the scan range
 * is [scene_object_start_index, scene_object_final_index). */
static int in_shade_collision_detection(const orc_scene *s, const ray *light_ray, float dist_to_light) {
    int inShade = 0, x;
    hit temp;
    g_cnt.shadow_rays++;
    for (x = s->scene_object_start_index; x < s->scene_object_final_index && !inShade; x++) {
        const orc_object *o = &s->objects[x];
        if (!o->is_light && object_collision(o, light_ray, &temp) && temp.distance < dist_to_light)
            inShade = 1;
    }
    return inShade;
}

/* inShade, src/RayTracer.cpp:743-771 */
static int in_shade(const orc_scene *s, const orc_object *light, const hit *collision) {
    vec3 dir = v_sub(light->origin, collision->intersection_point);
    float dist_to_light = v_length(dir);
    ray lightRay = ray_make(collision->intersection_point, dir);
    return in_shade_collision_detection(s, &lightRay, dist_to_light);
}

/* cosineShade, src/RayTracer.cpp:654-701 */
static void cosine_shade(vec3 *final_color, const vec3 *object_color, const orc_object *light,
                         const hit *collision) {
    vec3 light_color = light->color;
    float diffuse_coefficient = collision->diffuse_factor;
    float light_intensity = light->intensity;
    vec3 light_ray = v_normalize(v_sub(light->origin, collision->intersection_point));
    if (diffuse_coefficient > (float)0) {
        float cosine_dot_factor = v_dot(collision->normal_ray.direction, light_ray);
        if (cosine_dot_factor > (float)0) {
            float factor = cosine_dot_factor * diffuse_coefficient * light_intensity;
            final_color->x += factor * object_color->x * light_color.x;
            final_color->y += factor * object_color->y * light_color.y;
            final_color->z += factor * object_color->z * light_color.z;
        }
        final_color->x = (final_color->x > 1.0f) ? 1.0f : final_color->x;
        final_color->y = (final_color->y > 1.0f) ? 1.0f : final_color->y;
        final_color->z = (final_color->z > 1.0f) ? 1.0f : final_color->z;
    }
}

/* calculatePixel, src/RayTracer.cpp:448-638 */
static vec3 calculate_pixel(const orc_scene *s, const ray *r, int recursion_level, int max_recursion_level) {
    hit nearest;
    vec3 final_color = v3(0.0f, 0.0f, 0.0f), object_color;
    int i;
    memset(&nearest, 0, sizeof(nearest));
    if (recursion_level > max_recursion_level) return NULL_COLOR;
    if (!get_collision(s, r, &nearest)) return NULL_COLOR;
    if (nearest.hit_a_light_source) {                      /* :520-527 */
        float intensity = nearest.intensity_factor;
        return v_scale(nearest.color, intensity);
    }
    object_color = nearest.color;
    for (i = 0; i < s->object_count; i++) {                /* :540-591 */
        const orc_object *light = &s->objects[i];
        if (!light->is_light) continue;
        if (!in_shade(s, light, &nearest)) {
            vec3 L, N, R, V, light_color;
            float dot;
            cosine_shade(&final_color, &object_color, light, &nearest);
            /* specular :561-588 */
            L = v_normalize(v_sub(light->origin, nearest.intersection_point));
            light_color = light->color;
            N = v_normalize(nearest.normal_ray.direction);
            R = v_sub(L, v_scale(N, 2.0f * v_dot(L, N)));
            V = r->direction;
            dot = v_dot(V, R);
            if (dot > (float)0) {
                float pow_factor = dot, spec_factor;
                int j;
                for (j = 0; j < 19; j++) pow_factor *= dot;
                spec_factor = pow_factor * nearest.specular_factor;
                final_color = v_add(final_color, v_scale(light_color, spec_factor));
            }
        }
    }
    if (nearest.reflective_material) {                     /* :595-604 */
        ray temp_ray = nearest.reflected_ray;
        vec3 reflective_color = calculate_pixel(s, &temp_ray, recursion_level + 1, max_recursion_level);
        final_color = v_add(final_color,
                            v_mul(v_scale(reflective_color, nearest.reflective_factor), object_color));
    }
    return final_color;                                    /* no clamp: :619-631 is commented out */
}

/* raytrace_main pixel loop, src/RayTracer.cpp:904-923 (x outer, z inner) */
int orc_render(const orc_scene *s, const orc_camera *c, int W, int H, int x0, int x1,
               int max_depth, float *out) {
    int x, z;
    if (!s || !c || !out || W <= 0 || H <= 0 || x0 < 0 || x1 > W || x0 > x1 || max_depth < 0) return 1;
    memset(&g_cnt, 0, sizeof(g_cnt));
    for (x = x0; x < x1; x++) {
        for (z = 0; z < H; z++) {
            ray r = create_eye_ray(c, ((float)x) / W, ((float)z) / H);
            vec3 p = calculate_pixel(s, &r, 0, max_depth);
            float *o = out + ((size_t)(x - x0) * (size_t)H + (size_t)z) * 3;
            o[0] = p.x; o[1] = p.y; o[2] = p.z;
        }
    }
    return 0;
}

/* init_log + printPixelsToLog, src/RayTracer.cpp:2022-2061, 1574-1626, 2070-2110 */
int orc_write_screen_txt(const char *path, int W, int H, const float *rgb,
                         double run_time_s, double us_per_pixel) {
    FILE *f = fopen(path, "w");
    int i, j;
    if (!f) return 1;
    fputs("OSX Awesome Picture\n", f);
    fprintf(f, "Horizontal_Resolution:%i.\n", W);
    fprintf(f, "Vertical_Resolution:%i.\n", H);
    fprintf(f, "Hardware_Target:%s.\n", "OSX C++");
    fprintf(f, "Number_of_Cores:%i.\n", 1);
    fputs("IS_FOR_HARDWARE\n", f);
    fputs("NO_PARTIONING\n", f);
    fprintf(f, "Run_Time:%f.\n", run_time_s);
    fprintf(f, "us/pixel:%f.\n", us_per_pixel);
    fprintf(f, "filename:%s.\n", "raytracer_screen.txt");
    for (i = 0; i < W; i++) {
        for (j = 0; j < H; j++) {
            const float *p = rgb + ((size_t)i * (size_t)H + (size_t)j) * 3;
            fprintf(f, "(%f, %f, %f)\n", p[0], p[1], p[2]);
        }
    }
    return fclose(f) ? 1 : 0;
}
