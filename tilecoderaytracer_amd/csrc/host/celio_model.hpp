/*
 * celio_model.hpp -- host-side object model with the API surface of
 * ccelio/TileCodeRayTracer (namespace CelioRayTracer): vector3d / Color, Ray,
 * ObjTexture / Texture_CheckerBoard, ObjMaterial, SceneObject and its three
 * primitives, Scene, Camera.  A user of the reference builds a scene with the
 * same calls; `Scene::flatten()` / `Camera::describe()` then lower the object
 * graph to the plain-old-data tables of include/rt_capi.h, which is all the
 * GPU path ever sees.
 *
 * Deliberately absent: SceneObject::collision() and everything that traces a
 * ray on the CPU (getCollision / calculatePixel / cosineShade / inShade).
 * Intersection and shading exist only as HIP kernels behind rt_capi.h; this
 * model describes, it does not render.  Each primitive instead implements
 * `describe(rt_object_desc&)`.
 *
 * The constructors reproduce the reference's derived geometry bit-for-bit
 * (order of normalisations, double literals narrowed to float at the call);
 * citations give the reference file:line each piece follows.
 */
#ifndef CELIO_MODEL_HPP_
#define CELIO_MODEL_HPP_

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../../include/rt_capi.h"

namespace CelioRayTracer {

typedef float sdecimal32;                       /* src/rt_project_parameters.h:41 */

#define CELIO_MAX_OBJECT_COUNT 4000             /* src/Scene.h:8 */

/* The reference prints progress/diagnostic lines to stdout (e.g.
 * "Changing Camera Scene.", src/Camera.cpp:44).  The executable turns them on
 * to reproduce its console output; libraries keep them off so that a caller's
 * stdout (bench.py's single JSON line) stays clean. */
inline bool &verbose() { static bool v = false; return v; }

/* src/vector3d.h:31-163 */
class vector3d {
public:
    union {
        struct { float x, y, z; };
        struct { float r, g, b; };
        struct { float red, grn, blue; };
    };
    vector3d() { x = 0.0f; y = 0.0f; z = 0.0f; }
    vector3d(sdecimal32 _x, sdecimal32 _y, sdecimal32 _z) { x = _x; y = _y; z = _z; }

    void normalize() {                          /* :55-73 */
        const float len = std::sqrt(x * x + y * y + z * z);
        x = x / len; y = y / len; z = z / len;
    }
    float length() const { return std::sqrt(x * x + y * y + z * z); }             /* :75-85 */
    float dot(const vector3d &v) const { return x * v.x + y * v.y + z * v.z; }    /* :93-99 */
    void cross(const vector3d &a, const vector3d &b) {                            /* :101-104 */
        const float cx = a.y * b.z - a.z * b.y;
        const float cy = a.z * b.x - a.x * b.z;
        const float cz = a.x * b.y - a.y * b.x;
        x = cx; y = cy; z = cz;
    }
    void operator+=(const vector3d &v) { x += v.x; y += v.y; z += v.z; }
    void operator-=(const vector3d &v) { x -= v.x; y -= v.y; z -= v.z; }
    void operator*=(float f) { x *= f; y *= f; z *= f; }
    vector3d operator-() const { return vector3d(0 - x, 0 - y, 0 - z); }         /* :111 (0 - x, not -x) */
    friend vector3d operator+(const vector3d &a, const vector3d &b) { return vector3d(a.x + b.x, a.y + b.y, a.z + b.z); }
    friend vector3d operator-(const vector3d &a, const vector3d &b) { return vector3d(a.x - b.x, a.y - b.y, a.z - b.z); }
    friend vector3d operator*(const vector3d &v, sdecimal32 f) { return vector3d(v.x * f, v.y * f, v.z * f); }
    friend vector3d operator*(sdecimal32 f, const vector3d &v) { return vector3d(v.x * f, v.y * f, v.z * f); }
    friend vector3d operator*(const vector3d &a, const vector3d &b) { return vector3d(a.x * b.x, a.y * b.y, a.z * b.z); }
    void store(float out[3]) const { out[0] = x; out[1] = y; out[2] = z; }
};
typedef vector3d Color;

/* src/Color_Values.h:7-17 */
static const Color COLOR_WHITE(1.f, 1.f, 1.f);
static const Color COLOR_RED(1.f, 0.f, 0.f);
static const Color COLOR_YELLOW(1.f, 1.f, 0.f);
static const Color COLOR_GREEN(0.f, 1.f, 0.f);
static const Color COLOR_CYAN(0.f, 1.f, 1.f);
static const Color COLOR_BLUE(0.f, 0.f, 1.f);
static const Color COLOR_BLACK(0.f, 0.f, 0.f);
static const Color COLOR_DARK_GREY(0.33f, 0.33f, 0.33f);
static const Color COLOR_LIGHT_GREY(2 / 3.f, 2 / 3.f, 2 / 3.f);
static const Color COLOR_BROWN(0.2f, 0.2f, 0.0f);

/* src/Ray.h:12-38: both non-default constructors normalise the direction */
class Ray {
public:
    Ray() : origin(0.0f, 0.0f, 0.0f), direction(1.f, 0.f, 0.f) {}
    Ray(vector3d _origin, vector3d _direction) : origin(_origin), direction(_direction) { direction.normalize(); }
    Ray(vector3d _origin, vector3d point_final, vector3d point_init)
        : origin(_origin), direction(point_final - point_init) { direction.normalize(); }
    vector3d getOrigin() const { return origin; }
    vector3d getDirection() const { return direction; }
private:
    vector3d origin, direction;
};

/* src/ObjTexture.h:14-55.  getTexturePixel() is evaluated on the device; the
 * host class only carries the parameters (describe()). */
class ObjTexture {
public:
    ObjTexture() : width(1), height(1) {}
    ObjTexture(sdecimal32 w, sdecimal32 h) : width(w), height(h) {}
    virtual ~ObjTexture() {}
    void setWidth(sdecimal32 w) { width = w; }
    void setHeight(sdecimal32 h) { height = h; }
    virtual void describe(rt_texture_desc &out) const = 0;
protected:
    float width, height;
};

/* src/Texture_CheckerBoard.h:13-71 (default size 2 x 2) */
class Texture_CheckerBoard : public ObjTexture {
public:
    Texture_CheckerBoard() : ObjTexture(2, 2), light_color(COLOR_WHITE), dark_color(COLOR_BLACK) {}
    Texture_CheckerBoard(Color l, Color d) : ObjTexture(2, 2), light_color(l), dark_color(d) {}
    void setLightColor(Color l) { light_color = l; }
    void setDarkColor(Color d) { dark_color = d; }
    void describe(rt_texture_desc &out) const override {
        light_color.store(out.light); dark_color.store(out.dark);
        out.width = width; out.height = height;
    }
private:
    Color light_color, dark_color;
};

/* src/ObjMaterial.h:10-81 */
class ObjMaterial {
public:
    ObjMaterial()
        : myColor(1.0f, 1.0f, 1.0f), myObjTexture_ptr(nullptr), absorption_factor(0.0f),
          diffuse_factor(1.0), specular_factor(1.0), reflective_factor(0), refractive_factor(0) {}
    void setColor(vector3d c) { myColor = c; }
    void setAbsorptionFactor(float a) { absorption_factor = a; warn("Absorption"); }
    void setDiffuseFactor(float d) { diffuse_factor = d; }
    void setSpecularFactor(float s) { specular_factor = s; }
    void setReflectiveFactor(float f) { reflective_factor = f; warn("Reflective"); }
    void setRefractiveFactor(float f) { refractive_factor = f; warn("Refractive"); }
    vector3d getColor() const { return myColor; }
    ObjTexture *getTexture() const { return myObjTexture_ptr; }
    void setTexture(ObjTexture *t) { myObjTexture_ptr = t; }
    float getDiffuseFactor() const { return diffuse_factor; }
    float getSpecularFactor() const { return specular_factor; }
    float getAbsorptionFactor() const { return absorption_factor; }
    float getReflectiveFactor() const { return reflective_factor; }
    float getRefractiveFactor() const { return refractive_factor; }
private:
    void warn(const char *which) const {        /* the reference's sanity print, :34,47,54 */
        if (verbose() && (absorption_factor + reflective_factor + refractive_factor) > 1)
            std::printf("***ERROR. Setting %s Factor.\n Ab: %f\n Refl: %f\n Refr: %f\n", which,
                        absorption_factor, reflective_factor, refractive_factor);
    }
    Color myColor;
    ObjTexture *myObjTexture_ptr;
    float absorption_factor, diffuse_factor, specular_factor, reflective_factor, refractive_factor;
};

/* src/SceneObject.h:26-200, src/SceneObject.cpp:9-27 */
class SceneObject {
public:
    SceneObject() : origin(), isaLightSource(false), intensity(1.0), my_object_index(0) {
        myMaterial.setDiffuseFactor(0.25f);     /* only the default ctor lowers diffuse */
    }
    explicit SceneObject(vector3d _o) : origin(_o), isaLightSource(false), intensity(1.0), my_object_index(0) {}
    virtual ~SceneObject() {}

    /* replaces `virtual CollisionObject* collision(Ray*)` (src/SceneObject.h:166):
     * the primitive describes itself, the device intersects it */
    virtual void describe(rt_object_desc &out) const = 0;

    ObjMaterial *getMaterial() { return &myMaterial; }
    const ObjMaterial *getMaterial() const { return &myMaterial; }
    void moveOrigin(sdecimal32 dx, sdecimal32 dy, sdecimal32 dz) { origin.x += dx; origin.y += dy; origin.z += dz; }
    void changeOrigin(vector3d o) { origin = o; }
    vector3d getOrigin() const { return origin; }
    void setIndex(int i) { my_object_index = i; }
    int getIndex() const { return my_object_index; }
    void setAsLightSource() { isaLightSource = true; }
    bool checkIsaLightSource() const { return isaLightSource; }
    void setIntensity(sdecimal32 d) { intensity = d; }
    sdecimal32 getIntensity() const { return intensity; }

protected:
    void describe_base(rt_object_desc &out, int kind) const {
        std::memset(&out, 0, sizeof(out));
        out.kind = kind;
        out.is_light = isaLightSource ? 1 : 0;
        out.texture = -1;                       /* Scene::flatten() assigns texture indices */
        out.intensity = intensity;
        origin.store(out.origin);
        myMaterial.getColor().store(out.color);
        out.diffuse = myMaterial.getDiffuseFactor();
        out.specular = myMaterial.getSpecularFactor();
        out.reflective = myMaterial.getReflectiveFactor();
    }
    ObjMaterial myMaterial;
    vector3d origin;
    bool isaLightSource;
    sdecimal32 intensity;
    int my_object_index;
};

/* src/SceneSphere.h:10-20, src/SceneSphere.cpp:38-48 */
class SceneSphere : public SceneObject {
public:
    SceneSphere() : SceneObject(), radius(1.0f), radius_squared(radius * radius) {}
    SceneSphere(vector3d _origin, sdecimal32 _radius)
        : SceneObject(_origin), radius(_radius), radius_squared(_radius * _radius) {}
    void describe(rt_object_desc &out) const override {
        describe_base(out, RT_KIND_SPHERE);
        out.radius = radius;
        out.radius_squared = radius_squared;
    }
private:
    sdecimal32 radius, radius_squared;
};

/* src/SceneInfinitePlane.h:16-32, src/SceneInfinitePlane.cpp:11-26 */
class SceneInfinitePlane : public SceneObject {
public:
    SceneInfinitePlane() : SceneObject(), distance_to_origin(0) {}
    SceneInfinitePlane(vector3d o, vector3d n, vector3d h) : SceneObject(o), normal(n), horizontal(h) {
        normal.normalize();
        horizontal.normalize();
        vertical.cross(normal, horizontal);
        vertical.normalize();
        reverseNormal = -normal;                /* not re-normalised, unlike the finite plane */
        distance_to_origin = -origin.dot(normal);
    }
    void describe(rt_object_desc &out) const override {
        describe_base(out, RT_KIND_INFINITE_PLANE);
        normal.store(out.normal); vertical.store(out.vertical); horizontal.store(out.horizontal);
        reverseNormal.store(out.reverse_normal);
        out.distance_to_origin = distance_to_origin;
    }
private:
    vector3d normal, vertical, horizontal, reverseNormal;
    sdecimal32 distance_to_origin;
};

/* src/SceneFinitePlane.h:16-47, src/SceneFinitePlane.cpp:18-80 */
class SceneFinitePlane : public SceneObject {
public:
    SceneFinitePlane() : SceneObject(), v_distance(0), h_distance(0), distance_to_origin(0) {}
    /* corner form, :49-80 -- `o` must be the corner shared by both edges */
    SceneFinitePlane(vector3d o, vector3d vertical_corner, vector3d horizontal_corner)
        : SceneObject(o), plane_origin(o) {
        horizontal = horizontal_corner - o;
        vertical = vertical_corner - o;
        normal.cross(horizontal, vertical);
        v_distance = vertical.length();
        h_distance = horizontal.length();
        vertical.normalize();
        horizontal.normalize();
        normal.normalize();
        reverseNormal = -normal;
        reverseNormal.normalize();
        distance_to_origin = -o.dot(normal);
        /* the light-source origin moves to the far corner, :74-79 */
        const vector3d h = h_distance * horizontal;
        changeOrigin(v_distance * vertical + plane_origin + h);
    }
    /* axis form, :18-47 */
    SceneFinitePlane(vector3d o, vector3d n, vector3d h, float v_dist, float h_dist)
        : SceneObject(o), plane_origin(o), normal(n), horizontal(h) {
        vertical.cross(normal, horizontal);     /* cross product of the inputs as given */
        normal.normalize();
        horizontal.normalize();
        vertical.normalize();
        reverseNormal = -normal;
        reverseNormal.normalize();
        v_distance = v_dist;
        h_distance = h_dist;
        distance_to_origin = -o.dot(normal);
    }
    void describe(rt_object_desc &out) const override {
        describe_base(out, RT_KIND_FINITE_PLANE);
        plane_origin.store(out.plane_origin);
        normal.store(out.normal); vertical.store(out.vertical); horizontal.store(out.horizontal);
        reverseNormal.store(out.reverse_normal);
        out.v_distance = v_distance; out.h_distance = h_distance;
        out.distance_to_origin = distance_to_origin;
    }
private:
    vector3d plane_origin, normal, vertical, horizontal, reverseNormal;
    sdecimal32 v_distance, h_distance, distance_to_origin;
};

/* src/Camera.h:11-41, src/Camera.cpp:9-84 */
class Camera {
public:
    Camera() {
        screen_width = 1;
        screen_height = 1;
        screen_halfwidth = screen_width / (sdecimal32)2.0;
        screen_halfheight = screen_height / (sdecimal32)2.0;
        screen_origin = vector3d(-4.f, -4.f, 1.5f);
        vector_horizontal = vector3d(.1f, -.08f, 0.f);
        vector_outwards = vector3d(.08f, .1f, .01f);
        derive();
    }
    void setSceneTwoMirrors() {                 /* :42-69 */
        if (verbose()) std::printf("Changing Camera Scene.\n");
        screen_origin = vector3d(0, 0, 2.5);
        vector_outwards = vector3d(.00, 1, -.00);
        vector_horizontal = vector3d(1, -.00, 0);
        derive();
    }
    Ray *createEyeRay(sdecimal32 dx_percent, sdecimal32 dy_percent) const {   /* :71-84; caller owns the Ray */
        const sdecimal32 sx = dx_percent * screen_width - screen_halfwidth;
        const sdecimal32 sy = dy_percent * screen_height - screen_halfheight;
        vector3d pixel = screen_origin + sx * vector_horizontal;
        pixel = pixel + sy * vector_vertical;
        return new Ray(eye_origin, pixel, eye_origin);
    }
    sdecimal32 getScreenWidth() const { return screen_width; }
    sdecimal32 getScreeHeight() const { return screen_height; }   /* (sic) src/Camera.h:17 */
    void describe(rt_camera_desc &out) const {
        out.screen_width = screen_width; out.screen_height = screen_height;
        out.screen_halfwidth = screen_halfwidth; out.screen_halfheight = screen_halfheight;
        screen_origin.store(out.screen_origin);
        vector_horizontal.store(out.vector_horizontal);
        vector_vertical.store(out.vector_vertical);
        eye_origin.store(out.eye_origin);
    }
private:
    void derive() {                             /* :28-39 == :55-66 */
        vector_vertical.cross(vector_horizontal, vector_outwards);   /* on the un-normalised inputs */
        vector_outwards.normalize();
        vector_horizontal.normalize();
        vector_vertical.normalize();
        eye_distance = 1;
        eye_origin = (-eye_distance) * vector_outwards + screen_origin;
    }
    sdecimal32 screen_width, screen_height, screen_halfwidth, screen_halfheight;
    vector3d screen_origin, vector_outwards, vector_vertical, vector_horizontal;
    sdecimal32 eye_distance;
    vector3d eye_origin;
};

/* The flattened scene: owns the arrays an rt_scene_desc points into. */
struct FlatScene {
    std::vector<rt_object_desc> objects;
    std::vector<rt_texture_desc> textures;
    rt_scene_desc desc;
};

/* src/Scene.h:15-43, src/Scene.cpp */
class Scene {
public:
    /* A reference Scene is a static object: its index members start at zero
     * and only initialize*()/SetObjectIndices() set them (src/Scene.cpp:14-20). */
    Scene() : object_count(0), scene_object_start_index(0), scene_object_final_index(0) {
        objects.reserve(64);
    }
    ~Scene() {}
    int initialize();                                   /* SCENE 1, "museum"        */
    int initializeTwoMirrors(Camera *myCamera);         /* SCENE 2, "two mirrors"   */
    int getObjectCount() const { return object_count; }
    void addObject(SceneObject *new_obj_ptr) {          /* src/Scene.cpp:470-479: max 3999 */
        if (object_count + 1 >= CELIO_MAX_OBJECT_COUNT) {
            if (verbose()) std::printf("***ERROR. Added too many objects to scene.\n");
        } else {
            objects.push_back(new_obj_ptr);
            ++object_count;
        }
    }
    SceneObject *getObject(int i) const { return objects[(size_t)i]; }
    SceneFinitePlane **makeSceneBox(vector3d _origin, vector3d _dims);
    void SetObjectIndices(int my_rank, int group_size) {   /* src/Scene.cpp:486-504 */
        int new_object_count = object_count / group_size;
        const int start_index = my_rank * new_object_count;
        if (my_rank == group_size - 1) new_object_count = object_count - start_index;
        scene_object_start_index = start_index;
        scene_object_final_index = start_index + new_object_count;
    }
    int getSceneObjectStartIndex() const { return scene_object_start_index; }
    int getSceneObjectFinalIndex() const { return scene_object_final_index; }

    /* lower the object graph to rt_capi.h tables (Scene index order) */
    void flatten(FlatScene &out) const {
        out.objects.resize((size_t)object_count);
        out.textures.clear();
        std::vector<const ObjTexture *> seen;
        for (int i = 0; i < object_count; ++i) {
            rt_object_desc &d = out.objects[(size_t)i];
            objects[(size_t)i]->describe(d);
            const ObjTexture *t = objects[(size_t)i]->getMaterial()->getTexture();
            if (t) {
                size_t k = 0;
                while (k < seen.size() && seen[k] != t) ++k;
                if (k == seen.size()) {
                    seen.push_back(t);
                    rt_texture_desc td;
                    t->describe(td);
                    out.textures.push_back(td);
                }
                d.texture = (int32_t)k;
            }
        }
        out.desc.n_objects = object_count;
        out.desc.objects = out.objects.empty() ? nullptr : out.objects.data();
        out.desc.n_textures = (int32_t)out.textures.size();
        out.desc.textures = out.textures.empty() ? nullptr : out.textures.data();
        out.desc.shadow_begin = scene_object_start_index;
        out.desc.shadow_end = scene_object_final_index;
        out.desc.null_color[0] = out.desc.null_color[1] = out.desc.null_color[2] = 0.75f;   /* src/RayTracer.h:52 */
    }

private:
    std::vector<SceneObject *> objects;     /* like the reference, objects are never freed */
    int object_count;
    int scene_object_start_index, scene_object_final_index;
};

/* src/Scene.cpp:392-416 */
inline SceneFinitePlane **Scene::makeSceneBox(vector3d o, vector3d dims) {
    vector3d c[8];
    for (int k = 0; k < 8; ++k) c[k] = o;
    c[1].x += dims.x;
    c[2].y += dims.y;
    c[3].z += dims.z;
    c[4].x += dims.x; c[4].y += dims.y;
    c[5].x += dims.x; c[5].z += dims.z;
    c[6].y += dims.y; c[6].z += dims.z;
    c[7] = vector3d(o.x + dims.x, o.y + dims.y, o.z + dims.z);
    SceneFinitePlane **planes = new SceneFinitePlane *[6];
    planes[0] = new SceneFinitePlane(c[0], c[3], c[2]);
    planes[1] = new SceneFinitePlane(c[0], c[3], c[1]);
    planes[2] = new SceneFinitePlane(c[0], c[1], c[2]);
    planes[3] = new SceneFinitePlane(c[7], c[4], c[6]);
    planes[4] = new SceneFinitePlane(c[7], c[4], c[5]);
    planes[5] = new SceneFinitePlane(c[7], c[5], c[6]);
    return planes;
}

/* src/Scene.cpp:209-387: the 32-object museum */
inline int Scene::initialize() {
    SceneObject *obj;
    ObjMaterial *mat;
    /* two light bulbs */
    obj = new SceneSphere(vector3d(6.99, 6.99, 5.5), .15);
    obj->setIndex(object_count); obj->setAsLightSource(); obj->setIntensity(.75);
    addObject(obj);
    obj = new SceneSphere(vector3d(0, 0, 4.8), .15);
    obj->setIndex(object_count); obj->setAsLightSource(); obj->setIntensity(1.0);
    addObject(obj);
    /* mirror ball on the pedestal and a speck at the origin */
    obj = new SceneSphere(vector3d(0, 0, 2), 1);
    obj->setIndex(object_count);
    obj->getMaterial()->setColor(COLOR_RED);
    obj->getMaterial()->setReflectiveFactor(1.00f);
    addObject(obj);
    obj = new SceneSphere(vector3d(0, 0, 0), 0.01);
    obj->setIndex(object_count);
    addObject(obj);
    /* two balls at the back: a glossy red one and a perfect white mirror */
    for (int i = 0; i < 2; i++) {
        obj = new SceneSphere(vector3d(-2.5 + ((i + 0) * 2.5), 3, 1), 1);
        obj->setIndex(object_count);
        mat = obj->getMaterial();
        mat->setColor(COLOR_RED);
        if (i == 1) {
            mat->setColor(COLOR_WHITE);
            mat->setReflectiveFactor(1.00f);
            mat->setDiffuseFactor(0.00f);
        } else {
            mat->setSpecularFactor(.5);
        }
        addObject(obj);
    }
    obj = new SceneSphere(vector3d(), .10);
    obj->setIndex(object_count);
    obj->getMaterial()->setColor(COLOR_CYAN);
    addObject(obj);
    /* checkerboard floor */
    obj = new SceneInfinitePlane(vector3d(0, 0, 0), vector3d(0, 0, 1), vector3d(1, 0, 0));
    obj->setIndex(object_count);
    mat = obj->getMaterial();
    mat->setColor(COLOR_GREEN);
    mat->setReflectiveFactor(.5);
    mat->setDiffuseFactor(.5);
    ObjTexture *tex = new Texture_CheckerBoard(COLOR_WHITE, COLOR_BLACK);
    tex->setHeight(3.0f);
    tex->setWidth(3.0f);
    mat->setTexture(tex);
    addObject(obj);

    struct BoxSpec { vector3d origin, dims; Color color; int refl, diff, spec; float r, d, s; };
    const BoxSpec boxes[4] = {
        {vector3d(-.5f, -.5f, 0), vector3d(1.f, 1.f, 1.f), COLOR_BROWN, 1, 1, 0, 0.00f, 1.00f, 0.f},        /* pedestal      */
        {vector3d(-.70f, -.70f, 0), vector3d(1.4f, 1.4f, 0.25f), COLOR_BROWN, 0, 0, 1, 0.f, 0.f, 0.20f},    /* pedestal foot */
        {vector3d(-7, -7, -1), vector3d(14, 14, 7), COLOR_DARK_GREY, 1, 1, 1, 0.00f, 1.00f, 0.0f},          /* room walls    */
        {vector3d(-6, -6, 5), vector3d(12, 12, 1), COLOR_LIGHT_GREY, 1, 0, 1, 0.5f, 0.f, 0.5f},             /* ceiling slab  */
    };
    for (const BoxSpec &b : boxes) {
        SceneFinitePlane **planes = makeSceneBox(b.origin, b.dims);
        for (int i = 0; i < 6; i++) {
            obj = planes[i];
            obj->setIndex(object_count);
            mat = obj->getMaterial();
            mat->setColor(b.color);
            if (b.refl) mat->setReflectiveFactor(b.r);
            if (b.diff) mat->setDiffuseFactor(b.d);
            if (b.spec) mat->setSpecularFactor(b.s);
            addObject(obj);
        }
        delete[] planes;
    }
    scene_object_start_index = 0;
    scene_object_final_index = object_count;
    return 0;
}

/* src/Scene.cpp:23-206 */
inline int Scene::initializeTwoMirrors(Camera *myCamera) {
    SceneObject *obj;
    obj = new SceneSphere(vector3d(5, 10, 10), .15);
    obj->setAsLightSource(); obj->setIntensity(.75f);
    addObject(obj);
    obj = new SceneSphere(vector3d(), .10);
    obj->getMaterial()->setColor(COLOR_CYAN);
    addObject(obj);
    obj = new SceneSphere(vector3d(-40, 100, 40), 10);      /* the sun */
    obj->getMaterial()->setColor(COLOR_YELLOW);
    obj->getMaterial()->setSpecularFactor(0.25);
    addObject(obj);
    obj = new SceneSphere(vector3d(), .05);
    obj->getMaterial()->setColor(COLOR_CYAN);
    addObject(obj);
    obj = new SceneSphere(vector3d(), .02);
    obj->getMaterial()->setColor(COLOR_CYAN);
    addObject(obj);
    obj = new SceneInfinitePlane(vector3d(0, 0, 0), vector3d(0, 0, 1), vector3d(1, 0, 0));
    ObjTexture *tex = new Texture_CheckerBoard(COLOR_WHITE, COLOR_BLACK);
    tex->setHeight(3.0f);
    tex->setWidth(3.0f);
    obj->getMaterial()->setTexture(tex);
    obj->getMaterial()->setReflectiveFactor(.05);
    obj->getMaterial()->setDiffuseFactor(.5);
    addObject(obj);
    /* mirror 1 and its frame */
    obj = new SceneFinitePlane(vector3d(-1.75, 7, 0), vector3d(0, -1, 0), vector3d(1, 0, 0), 5, 3.5f);
    obj->getMaterial()->setColor(COLOR_WHITE);
    obj->getMaterial()->setReflectiveFactor(1.0);
    obj->getMaterial()->setDiffuseFactor(.0);
    addObject(obj);
    obj = new SceneFinitePlane(vector3d(-2, 7, 0), vector3d(0, -1, 0), vector3d(1, 0, 0), 5.25, 4.0);
    obj->getMaterial()->setColor(COLOR_BROWN);
    obj->getMaterial()->setDiffuseFactor(.5);
    addObject(obj);
    /* mirror 2 and its frame */
    obj = new SceneFinitePlane(vector3d(1.75, -7, 0), vector3d(0, 1, 0), vector3d(-1, 0, 0), 5, 3.5);
    obj->getMaterial()->setColor(COLOR_WHITE);
    obj->getMaterial()->setDiffuseFactor(.0);
    obj->getMaterial()->setReflectiveFactor(1.0);
    addObject(obj);
    obj = new SceneFinitePlane(vector3d(2, -7, 0), vector3d(0, 1, 0), vector3d(-1, 0, 0), 5.25, 4.0);
    obj->getMaterial()->setColor(COLOR_BROWN);
    obj->getMaterial()->setDiffuseFactor(.5);
    addObject(obj);

    /* three sphere pyramids generated by float loops, :132-203.  The x
     * coordinate is `i + <double literal>` in the first pyramid, `i - 6`
     * (float) in the second and `i` in the third. */
    struct Pyramid { float bx, by, bz, offset; int form; float y_add; double radius; Color color; };
    const Pyramid pyr[3] = {
        {14.50f, 15.0f, 16.5f, 0.5f, 0, 15.f, 0.33, COLOR_GREEN},
        {5.f, 5.0f, 5.f, 0.65f, 1, 10.f, .5, COLOR_RED},
        {1.f, 1.0f, 1.f, 0.33f, 2, 20.f, 0.33, COLOR_RED},
    };
    for (const Pyramid &q : pyr) {
        float i_start = 0, j_start = 0;
        for (float k = 0; k < q.bz; k += q.offset) {
            i_start += q.offset;
            j_start += q.offset;
            for (float i = i_start; i < q.bx - i_start; i += q.offset) {
                for (float j = j_start; j < q.by - j_start; j += q.offset) {
                    const float cx = q.form == 0 ? (float)(i + 2.65) : q.form == 1 ? (i - 6) : i;
                    obj = new SceneSphere(vector3d(cx, j + q.y_add, k), (sdecimal32)q.radius);
                    obj->getMaterial()->setColor(q.color);
                    addObject(obj);
                }
            }
        }
    }
    if (verbose()) std::printf("ObjectCount: %d\n", object_count);
    myCamera->setSceneTwoMirrors();
    scene_object_start_index = 0;
    scene_object_final_index = object_count;
    return 0;
}

/* Synthetic benchmark scene "grid-n" (SURVEY.md section 8(d), Appendix E):
 * two lights, n*n unit spheres on a 2.5 pitch (checkerboard of mirrors and
 * glossy balls), a reflective checkerboard floor and a reflective ceiling;
 * viewed with Camera::setSceneTwoMirrors().  Built only through the public
 * API above.  With shadows == false the shadow scan range stays [0, 0), which
 * is what the reference does for a scene assembled with addObject() alone;
 * shadows == true calls SetObjectIndices(0, 1). */
inline int build_grid_scene(Scene &scene, Camera &camera, int n, bool shadows) {
    if (n < 1 || n * n + 4 >= CELIO_MAX_OBJECT_COUNT) return 1;
    const Color palette[6] = {COLOR_RED, COLOR_YELLOW, COLOR_GREEN, COLOR_CYAN, COLOR_BLUE, COLOR_WHITE};
    SceneObject *obj;
    obj = new SceneSphere(vector3d(-20.0f, 10.0f, 10.0f), .15);
    obj->setAsLightSource(); obj->setIntensity(.75);
    scene.addObject(obj);
    obj = new SceneSphere(vector3d(0.0f, 40.0f, 11.0f), .15);
    obj->setAsLightSource(); obj->setIntensity(1.0);
    scene.addObject(obj);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            const int k = i * n + j;
            const float cx = ((float)i - (float)(n - 1) * 0.5f) * 2.5f;
            const float cy = 6.0f + (float)j * 2.5f;
            obj = new SceneSphere(vector3d(cx, cy, 1.0f), 1.0f);
            obj->getMaterial()->setColor(palette[k % 6]);
            if (((i + j) & 1) == 0) {
                obj->getMaterial()->setReflectiveFactor(1.0f);
                obj->getMaterial()->setDiffuseFactor(0.0f);
            } else {
                obj->getMaterial()->setSpecularFactor(0.5f);
            }
            scene.addObject(obj);
        }
    }
    obj = new SceneInfinitePlane(vector3d(0, 0, 0), vector3d(0, 0, 1), vector3d(1, 0, 0));
    obj->getMaterial()->setColor(COLOR_GREEN);
    obj->getMaterial()->setReflectiveFactor(.5);
    obj->getMaterial()->setDiffuseFactor(.5);
    ObjTexture *tex = new Texture_CheckerBoard(COLOR_WHITE, COLOR_BLACK);
    tex->setHeight(3.0f);
    tex->setWidth(3.0f);
    obj->getMaterial()->setTexture(tex);
    scene.addObject(obj);
    obj = new SceneInfinitePlane(vector3d(0, 0, 12), vector3d(0, 0, -1), vector3d(1, 0, 0));
    obj->getMaterial()->setColor(COLOR_LIGHT_GREY);
    obj->getMaterial()->setReflectiveFactor(.5);
    obj->getMaterial()->setSpecularFactor(.5);
    scene.addObject(obj);
    if (shadows) scene.SetObjectIndices(0, 1);
    camera.setSceneTwoMirrors();
    return 0;
}

} // namespace CelioRayTracer

#endif /* CELIO_MODEL_HPP_ */
