/*
 * rt_oracle.h -- CPU restatement of the ccelio/TileCodeRayTracer render path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (tilecoderaytracer_amd/,
 * include/, the C-ABI library, the host executable) may include, link, load
 * or call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the timed CPU baseline.
 *
 * PARITY STATUS (read DESIGN.md "Oracle"): the reference ships no tests, no
 * golden vectors and no fixtures, and it does not compile as shipped (it
 * includes fixed_class.h / fixed_func.h, which are not in the tree).  Writing
 * stand-ins for those headers is not allowed in this build, so no oracle/_ref
 * binary exists and, by the strict definition, this oracle is
 * "parity unpinned".  What it IS checked against: the SHA-256 digests and
 * spot pixel values that SURVEY.md Appendix D recorded from the survey's own
 * scratch build of the reference (tests/test_oracle_pins.py).
 *
 * Every function cites the reference file:line it restates
 * (paths relative to /root/reference/).
 */
#ifndef RT_ORACLE_H_
#define RT_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_vec3 { float x, y, z; } orc_vec3;

enum { ORC_SPHERE = 0, ORC_INFINITE_PLANE = 1, ORC_FINITE_PLANE = 2 };

/* One scene object with every derived member the reference constructors
 * compute (src/SceneObject.h:189-199, src/SceneSphere.h:18-19,
 * src/SceneInfinitePlane.h:26-32, src/SceneFinitePlane.h:33-45). */
typedef struct orc_object {
    int       kind;
    /* SceneObject */
    orc_vec3  origin;
    int       is_light;
    float     intensity;
    /* ObjMaterial (src/ObjMaterial.h:13-21) */
    orc_vec3  color;
    float     diffuse, specular, reflective;
    int       has_texture;
    orc_vec3  tex_light, tex_dark;
    float     tex_width, tex_height;
    /* sphere */
    float     radius, radius_squared;
    /* planes */
    orc_vec3  plane_origin;       /* finite plane only */
    orc_vec3  normal, vertical, horizontal, reverse_normal;
    float     v_distance, h_distance, distance_to_origin;
} orc_object;

typedef struct orc_camera {
    float    screen_width, screen_height, screen_halfwidth, screen_halfheight;
    orc_vec3 screen_origin, vector_outwards, vector_vertical, vector_horizontal;
    float    eye_distance;
    orc_vec3 eye_origin;
} orc_camera;

typedef struct orc_scene orc_scene;

orc_scene *orc_scene_new(void);
void       orc_scene_free(orc_scene *s);
int        orc_scene_object_count(const orc_scene *s);
int        orc_scene_get_object(const orc_scene *s, int i, orc_object *out);
int        orc_scene_shadow_range(const orc_scene *s, int *begin, int *end);

/* primitive constructors; return the new object's index or -1 */
int orc_add_sphere(orc_scene *s, orc_vec3 origin, float radius);
int orc_add_infinite_plane(orc_scene *s, orc_vec3 o, orc_vec3 n, orc_vec3 h);
int orc_add_finite_plane_corners(orc_scene *s, orc_vec3 o, orc_vec3 vcorner, orc_vec3 hcorner);
int orc_add_finite_plane_axes(orc_scene *s, orc_vec3 o, orc_vec3 n, orc_vec3 h, float v_dist, float h_dist);

/* material / light setters on object idx */
int orc_set_color(orc_scene *s, int idx, orc_vec3 c);
int orc_set_diffuse(orc_scene *s, int idx, float f);
int orc_set_specular(orc_scene *s, int idx, float f);
int orc_set_reflective(orc_scene *s, int idx, float f);
int orc_set_checkerboard(orc_scene *s, int idx, orc_vec3 light, orc_vec3 dark, float w, float h);
int orc_set_light(orc_scene *s, int idx);
int orc_set_intensity(orc_scene *s, int idx, float f);
void orc_set_object_indices(orc_scene *s, int my_rank, int group_size);

/* whole scenes */
int orc_scene_initialize(orc_scene *s);                          /* museum, SCENE 1 */
int orc_scene_initialize_two_mirrors(orc_scene *s, orc_camera *cam); /* SCENE 2 */
int orc_scene_grid(orc_scene *s, int n, int shadows);            /* SURVEY.md App. E */

void orc_camera_default(orc_camera *c);
void orc_camera_two_mirrors(orc_camera *c);
void orc_camera_eye_ray(const orc_camera *c, float dx, float dy, orc_vec3 *origin, orc_vec3 *dir);

/* Render columns [x0,x1) x all z of a W x H image into
 * out[(x-x0)*H*3 + z*3 + c] (packed fp32, x-major: pixels[x][z]). */
int orc_render(const orc_scene *s, const orc_camera *c, int W, int H,
               int x0, int x1, int max_depth, float *out);

/* The same on n_threads host threads, the reference's static partitioning
 * (PARTIONING_STRATEGY 1, src/RayTracer.cpp:904-923 with CORE_NUM > 1): the image
 * sample is n_chunks chunks of chunk_cols columns starting at chunk_x0[k]
 * (ascending), dealt to the threads in contiguous shares; chunk k lands at
 * out + k * chunk_cols * H * 3.  cpus (may be NULL) names the logical CPU each
 * thread is pinned to.  *seconds = wall time from "all threads ready" to "all
 * done".  bench.py's cpu_baseline leg; rt_oracle_mt.c. */
int orc_render_static_partition(const orc_scene *s, const orc_camera *c, int W, int H, int max_depth,
                                const int *chunk_x0, int n_chunks, int chunk_cols,
                                int n_threads, const int *cpus, float *out, double *seconds);

/* work counters of the last orc_render on this thread (for DESIGN.md figures) */
typedef struct orc_counters {
    unsigned long long nearest_rays, shadow_rays, collision_tests;
} orc_counters;
void orc_get_counters(orc_counters *out);

/* byte-exact raytracer_screen.txt writer (src/RayTracer.cpp:1574-1626, 2022-2110) */
int orc_write_screen_txt(const char *path, int W, int H, const float *rgb,
                         double run_time_s, double us_per_pixel);

#ifdef __cplusplus
}
#endif
#endif /* RT_ORACLE_H_ */
