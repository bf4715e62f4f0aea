/*
 * rt_host_capi.h -- C glue over the C++ host model (celio_model.hpp) so that
 * Python (tests, bench.py) can build scenes with the reference's API calls and
 * obtain the flattened rt_scene_desc / rt_camera_desc.  This is NOT the
 * drop-in boundary (that is include/rt_capi.h); it is how non-C++ callers
 * reach the host side that sits above it.  All functions return 0 on success
 * unless stated otherwise.
 */
#ifndef RT_HOST_CAPI_H_
#define RT_HOST_CAPI_H_

#include "../../../include/rt_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_scene rth_scene;   /* a Scene + a Camera + their flattened form */

int  rth_scene_new(rth_scene **out);                     /* empty Scene(), default Camera() */
int  rth_scene_builtin(rth_scene **out);                 /* Scene::initialize()             */
int  rth_scene_two_mirrors(rth_scene **out);             /* Scene::initializeTwoMirrors()   */
int  rth_scene_grid(int n, int shadows, rth_scene **out);/* build_grid_scene()              */
void rth_scene_free(rth_scene *s);

/* primitive constructors + addObject(); return the object's index, or -1 */
int rth_add_sphere(rth_scene *s, const float origin[3], float radius);
int rth_add_infinite_plane(rth_scene *s, const float o[3], const float n[3], const float h[3]);
int rth_add_finite_plane_corners(rth_scene *s, const float o[3], const float vcorner[3], const float hcorner[3]);
int rth_add_finite_plane_axes(rth_scene *s, const float o[3], const float n[3], const float h[3],
                              float v_dist, float h_dist);
int rth_object_count(const rth_scene *s);

/* getObject(idx)->... setters */
int rth_set_color(rth_scene *s, int idx, const float rgb[3]);
int rth_set_diffuse(rth_scene *s, int idx, float f);
int rth_set_specular(rth_scene *s, int idx, float f);
int rth_set_reflective(rth_scene *s, int idx, float f);
int rth_set_checkerboard(rth_scene *s, int idx, const float light[3], const float dark[3], float w, float h);
int rth_set_light(rth_scene *s, int idx);
int rth_set_intensity(rth_scene *s, int idx, float f);
int rth_set_object_indices(rth_scene *s, int my_rank, int group_size);   /* Scene::SetObjectIndices */
int rth_camera_two_mirrors(rth_scene *s);                                /* Camera::setSceneTwoMirrors */
int rth_camera_eye_ray(const rth_scene *s, float dx, float dy, float origin[3], float dir[3]);

/* flattened views; valid until the scene is modified or freed */
const rt_scene_desc  *rth_scene_desc(rth_scene *s);
const rt_camera_desc *rth_camera_desc(rth_scene *s);

/* byte-exact raytracer_screen.txt (src/RayTracer.cpp:2022-2061, 1574-1626) */
int rth_write_screen_txt(const char *path, int W, int H, const float *rgb,
                         double run_time_s, double us_per_pixel);
/* the same with CORE_NUM = n_cores in the header (one "core" per GPU; > 1 writes the static
 * partition's label, src/RayTracer.cpp:2037-2058) */
int rth_write_screen_txt_cores(const char *path, int W, int H, const float *rgb,
                               double run_time_s, double us_per_pixel, int n_cores);

#ifdef __cplusplus
}
#endif
#endif /* RT_HOST_CAPI_H_ */
