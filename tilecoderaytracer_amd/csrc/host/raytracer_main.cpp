/*
 * raytracer_main.cpp -- the counterpart of the reference's main() /
 * raytrace_main() (src/RayTracer.cpp:1122-1321, 855-1114): open the log,
 * build the hard-coded Scene and Camera, render every pixel, print the timing
 * lines, write raytracer_screen.txt.  The pixel loop itself
 * (src/RayTracer.cpp:904-923) is one call through the C ABI (rt_render /
 * rt_render_multi) into the HIP kernels; there is no CPU renderer here.
 *
 * With no arguments it runs the shipped configuration: SCENE 1, 500 x 504,
 * MAX_RECURSION_LEVEL 50 (src/rt_project_parameters.h:27,65-66,73).  The
 * reference fixes those at compile time; here they are run-time options:
 *   --width W --height H --depth D
 *   --scene 1 | 2 | grid:N | grid:N:noshadow
 *   --gpus G            x-strips over G GPUs, cut by measured cost, sent to GPU 0 with RCCL
 *   --out FILE          (default raytracer_screen.txt)   --no-txt
 */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "../../../include/rt_capi.h"
#include "celio_model.hpp"
#include "screen_txt.hpp"

using namespace CelioRayTracer;

/* the reference's file-scope state (src/RayTracer.h:44-52) */
static Scene my_scene = Scene();
static Camera my_camera = Camera();
static std::vector<float> pixels;          /* pixels[x][z] as packed fp32 RGB */

static int usage(const char *argv0) {
    std::fprintf(stderr,
                 "usage: %s [--width W] [--height H] [--depth D] [--scene 1|2|grid:N[:noshadow]]\n"
                 "          [--gpus G] [--out FILE] [--no-txt]\n", argv0);
    return 1;
}

int main(int argc, char **argv) {
    int W = 500, H = 504, depth = 50, gpus = 1;
    bool write_txt = true;
    std::string scene_name = "1", out_path = "raytracer_screen.txt";
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto need = [&](int &dst) { if (i + 1 >= argc) return false; dst = std::atoi(argv[++i]); return true; };
        if (a == "--width") { if (!need(W)) return usage(argv[0]); }
        else if (a == "--height") { if (!need(H)) return usage(argv[0]); }
        else if (a == "--depth") { if (!need(depth)) return usage(argv[0]); }
        else if (a == "--gpus") { if (!need(gpus)) return usage(argv[0]); }
        else if (a == "--scene" && i + 1 < argc) scene_name = argv[++i];
        else if (a == "--out" && i + 1 < argc) out_path = argv[++i];
        else if (a == "--no-txt") write_txt = false;
        else return usage(argv[0]);
    }
    if (W <= 0 || H <= 0 || depth < 0 || gpus <= 0) return usage(argv[0]);
    verbose() = true;                          /* console output like the reference's */

    if (gpus == 1) std::cout << "Single-Core RayTracing!" << std::endl << std::endl;
    else std::printf("\nMulti-Core RayTracing!\n\n");
    std::printf("  %d NUMBER OF CORES\n", gpus);

    /* raytrace_main(), src/RayTracer.cpp:855-1114 */
    std::printf("Global Rank(%d) begins\n", 0);
    const auto t_process = std::chrono::steady_clock::now();

    if (scene_name == "1") {
        if (my_scene.initialize()) return 1;
    } else if (scene_name == "2") {
        if (my_scene.initializeTwoMirrors(&my_camera)) return 1;
    } else if (scene_name.rfind("grid:", 0) == 0) {
        const std::string rest = scene_name.substr(5);
        const int n = std::atoi(rest.c_str());
        const bool shadows = rest.find(":noshadow") == std::string::npos;
        if (build_grid_scene(my_scene, my_camera, n, shadows)) {
            std::fprintf(stderr, "bad grid size\n");
            return 1;
        }
    } else {
        return usage(argv[0]);
    }

    FlatScene flat;
    rt_camera_desc cam;
    my_scene.flatten(flat);
    my_camera.describe(cam);
    pixels.assign((size_t)W * (size_t)H * 3, 0.0f);

    std::printf("****** Start Ray Tracing. *******\n");
    const auto t0 = std::chrono::steady_clock::now();
    int rc;
    double kernel_ms = 0.0;
    if (gpus == 1) {
        rt_scene *scene = nullptr;
        rc = rt_scene_create(&flat.desc, 0, &scene);
        if (rc == RT_OK) rc = rt_render(scene, &cam, W, H, 0, W, depth, pixels.data());
        if (rc == RT_OK) {
            rt_timing tm;
            if (rt_get_timing(scene, &tm) == RT_OK) kernel_ms = tm.last_kernel_ms;
        }
        rt_scene_destroy(scene);
    } else {
        /* rt_render_multi() with the handle kept long enough to say how the image was cut: strips by measured cost, each
         * sent to GPU 0 in column chunks while the next chunk is rendered (rt_multi_render, chunks = 0) */
        rt_multi *multi = nullptr;
        rc = rt_multi_create(&flat.desc, gpus, &multi);
        if (rc == RT_OK) rc = rt_multi_render(multi, &cam, W, H, depth, 0, pixels.data());
        rt_multi_info info;
        if (rc == RT_OK && rt_multi_get_info(multi, &info) == RT_OK) {
            std::printf("Partition: %d x-strips of", info.ngpu);
            for (int g = 0; g < info.ngpu; ++g) {
                std::printf("%s%d", g ? "/" : " ", info.bounds[g + 1] - info.bounds[g]);
                if (info.kernel_ms[g] > kernel_ms) kernel_ms = info.kernel_ms[g];
            }
            if (info.transport == RT_MULTI_TRANSPORT_DIRECT)
                std::printf(" columns (%s), stored by the kernels straight into GPU 0's image; frame %f ms\n",
                            info.balanced ? "cut by measured cost" : "equal", info.frame_ms);
            else
                std::printf(" columns (%s), %d column chunk(s) per strip; frame %f ms\n",
                            info.balanced ? "cut by measured cost" : "equal", info.chunks, info.frame_ms);
        }
        rt_multi_destroy(multi);
    }
    if (rc != RT_OK) {
        std::fprintf(stderr, "render failed (%d): %s\n", rc, rt_last_error());
        return 1;
    }
    const auto t1 = std::chrono::steady_clock::now();
    /* The reference's Run_Time is CPU time since process start because its
     * start_time local shadows the global (src/RayTracer.cpp:893,1090); keep
     * "since process start" and also print the render-only figures. */
    const double run_time_s = std::chrono::duration<double>(t1 - t_process).count();
    const double render_s = std::chrono::duration<double>(t1 - t0).count();
    const double run_time_us = run_time_s * 1.0E6;
    std::printf("Finished.\n");
    std::printf("Total_Time (s) (Time.h)    : %f\n", run_time_s);
    std::printf("Render call (s)            : %f\n", render_s);
    if (kernel_ms > 0.0)
        std::printf("Render kernel (ms)         : %f  (%.1f Mrays/s)%s\n", kernel_ms,
                    (double)W * (double)H / (kernel_ms * 1e3), gpus > 1 ? "  [the GPU whose kernels took longest]" : "");
    std::printf("AverageRoundTime (us/pixel): %f\n", run_time_us / ((double)W * (double)H));

    if (write_txt) {
        std::printf("PrintScreen to Log.\n");
        std::printf("Greetings: %d, %d \n", W, H);
        if (celio_write_screen_txt(out_path.c_str(), W, H, pixels.data(), run_time_s,
                                   run_time_us / ((double)W * (double)H), gpus))
            return 1;
        std::printf("Closing log file.\n\n");
    }
    std::printf("Program Done.\n");
    return 0;
}
