/*
 * rt_host_capi.cpp -- C glue over celio_model.hpp (see rt_host_capi.h) and
 * the raytracer_screen.txt writer.  Pure host code: builds and describes
 * scenes, renders nothing.
 */
#include "rt_host_capi.h"

#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "celio_model.hpp"
#include "screen_txt.hpp"

using namespace CelioRayTracer;

struct rth_scene {
    Scene scene;
    Camera camera;
    FlatScene flat;
    rt_camera_desc cam_desc;
    bool dirty = true;
};

namespace {

vector3d v3(const float p[3]) { return vector3d(p[0], p[1], p[2]); }

SceneObject *object_at(rth_scene *s, int idx) {
    if (!s || idx < 0 || idx >= s->scene.getObjectCount()) return nullptr;
    return s->scene.getObject(idx);
}

int add(rth_scene *s, SceneObject *obj) {
    const int before = s->scene.getObjectCount();
    obj->setIndex(before);
    s->scene.addObject(obj);
    s->dirty = true;
    if (s->scene.getObjectCount() == before) { delete obj; return -1; }
    return before;
}

void refresh(rth_scene *s) {
    if (!s->dirty) return;
    s->scene.flatten(s->flat);
    s->camera.describe(s->cam_desc);
    s->dirty = false;
}

} // namespace

extern "C" {

int rth_scene_new(rth_scene **out) {
    if (!out) return 1;
    *out = new (std::nothrow) rth_scene();
    return *out ? 0 : 1;
}

int rth_scene_builtin(rth_scene **out) {
    if (rth_scene_new(out)) return 1;
    return (*out)->scene.initialize();
}

int rth_scene_two_mirrors(rth_scene **out) {
    if (rth_scene_new(out)) return 1;
    return (*out)->scene.initializeTwoMirrors(&(*out)->camera);
}

int rth_scene_grid(int n, int shadows, rth_scene **out) {
    if (rth_scene_new(out)) return 1;
    if (build_grid_scene((*out)->scene, (*out)->camera, n, shadows != 0)) {
        delete *out;
        *out = nullptr;
        return 1;
    }
    return 0;
}

void rth_scene_free(rth_scene *s) { delete s; }

int rth_add_sphere(rth_scene *s, const float o[3], float radius) {
    if (!s || !o) return -1;
    return add(s, new SceneSphere(v3(o), radius));
}
int rth_add_infinite_plane(rth_scene *s, const float o[3], const float n[3], const float h[3]) {
    if (!s || !o || !n || !h) return -1;
    return add(s, new SceneInfinitePlane(v3(o), v3(n), v3(h)));
}
int rth_add_finite_plane_corners(rth_scene *s, const float o[3], const float vc[3], const float hc[3]) {
    if (!s || !o || !vc || !hc) return -1;
    return add(s, new SceneFinitePlane(v3(o), v3(vc), v3(hc)));
}
int rth_add_finite_plane_axes(rth_scene *s, const float o[3], const float n[3], const float h[3],
                              float v_dist, float h_dist) {
    if (!s || !o || !n || !h) return -1;
    return add(s, new SceneFinitePlane(v3(o), v3(n), v3(h), v_dist, h_dist));
}
int rth_object_count(const rth_scene *s) { return s ? s->scene.getObjectCount() : 0; }

int rth_set_color(rth_scene *s, int idx, const float rgb[3]) {
    SceneObject *o = object_at(s, idx);
    if (!o || !rgb) return 1;
    o->getMaterial()->setColor(v3(rgb)); s->dirty = true; return 0;
}
int rth_set_diffuse(rth_scene *s, int idx, float f) {
    SceneObject *o = object_at(s, idx);
    if (!o) return 1;
    o->getMaterial()->setDiffuseFactor(f); s->dirty = true; return 0;
}
int rth_set_specular(rth_scene *s, int idx, float f) {
    SceneObject *o = object_at(s, idx);
    if (!o) return 1;
    o->getMaterial()->setSpecularFactor(f); s->dirty = true; return 0;
}
int rth_set_reflective(rth_scene *s, int idx, float f) {
    SceneObject *o = object_at(s, idx);
    if (!o) return 1;
    o->getMaterial()->setReflectiveFactor(f); s->dirty = true; return 0;
}
int rth_set_checkerboard(rth_scene *s, int idx, const float light[3], const float dark[3], float w, float h) {
    SceneObject *o = object_at(s, idx);
    if (!o || !light || !dark) return 1;
    ObjTexture *t = new Texture_CheckerBoard(v3(light), v3(dark));
    t->setHeight(h);
    t->setWidth(w);
    o->getMaterial()->setTexture(t);
    s->dirty = true;
    return 0;
}
int rth_set_light(rth_scene *s, int idx) {
    SceneObject *o = object_at(s, idx);
    if (!o) return 1;
    o->setAsLightSource(); s->dirty = true; return 0;
}
int rth_set_intensity(rth_scene *s, int idx, float f) {
    SceneObject *o = object_at(s, idx);
    if (!o) return 1;
    o->setIntensity(f); s->dirty = true; return 0;
}
int rth_set_object_indices(rth_scene *s, int my_rank, int group_size) {
    if (!s || group_size <= 0 || my_rank < 0 || my_rank >= group_size) return 1;
    s->scene.SetObjectIndices(my_rank, group_size); s->dirty = true; return 0;
}
int rth_camera_two_mirrors(rth_scene *s) {
    if (!s) return 1;
    s->camera.setSceneTwoMirrors(); s->dirty = true; return 0;
}
int rth_camera_eye_ray(const rth_scene *s, float dx, float dy, float origin[3], float dir[3]) {
    if (!s || !origin || !dir) return 1;
    Ray *r = s->camera.createEyeRay(dx, dy);
    r->getOrigin().store(origin);
    r->getDirection().store(dir);
    delete r;
    return 0;
}

const rt_scene_desc *rth_scene_desc(rth_scene *s) {
    if (!s) return nullptr;
    refresh(s);
    return &s->flat.desc;
}
const rt_camera_desc *rth_camera_desc(rth_scene *s) {
    if (!s) return nullptr;
    refresh(s);
    return &s->cam_desc;
}

int rth_write_screen_txt(const char *path, int W, int H, const float *rgb, double run_time_s,
                         double us_per_pixel) {
    return celio_write_screen_txt(path, W, H, rgb, run_time_s, us_per_pixel);
}

int rth_write_screen_txt_cores(const char *path, int W, int H, const float *rgb, double run_time_s,
                               double us_per_pixel, int n_cores) {
    return celio_write_screen_txt(path, W, H, rgb, run_time_s, us_per_pixel, n_cores);
}

} // extern "C"
