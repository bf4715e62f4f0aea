/*
 * screen_txt.hpp -- writer for the reference's framebuffer log,
 * `raytracer_screen.txt`: init_log()'s header (src/RayTracer.cpp:2022-2061),
 * printPixelsToLog()'s three tag lines and one "(%f, %f, %f)\n" line per pixel,
 * x outer / z inner (src/RayTracer.cpp:1574-1626), tag lines as
 * "tag:value.\n" (src/RayTracer.cpp:2070-2110).
 *
 * The reference formats 3*W*H floats with sprintf("%f") + fputs, about 10 s
 * for a 4096 x 4096 image.  Here "%f" of a float is produced by exact integer
 * arithmetic (a float times 10^6 fits 64 bits once shifted), rounded half to
 * even exactly like glibc's printf, into a large buffer; values outside the
 * fast range (|v| >= 2^39, inf, nan) fall back to snprintf; blocks of pixels
 * are formatted by up to 16 threads and written in order.  The bytes are
 * identical to the reference's (tests/test_screen_txt.py).
 */
#ifndef SCREEN_TXT_HPP_
#define SCREEN_TXT_HPP_

#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

namespace celio_txt {

/* append printf("%f", (double)v) to dst, return the new end */
inline char *format_f(char *dst, float v) {
    uint32_t bits;
    std::memcpy(&bits, &v, 4);
    const uint32_t expo = (bits >> 23) & 0xFFu;
    uint32_t mant = bits & 0x7FFFFFu;
    int e;                                     /* |v| = mant * 2^e */
    if (expo == 0xFFu) return dst + std::snprintf(dst, 64, "%f", (double)v);
    if (expo == 0) { e = -149; } else { mant |= 0x800000u; e = (int)expo - 150; }
    if (e > 15) return dst + std::snprintf(dst, 64, "%f", (double)v);

    uint64_t scaled;                           /* round_half_even(|v| * 10^6) */
    if (e >= 0) {
        scaled = ((uint64_t)mant << e) * 1000000ull;           /* < 2^24 * 2^15 * 2^20 */
    } else {
        const int s = -e;
        const uint64_t prod = (uint64_t)mant * 1000000ull;     /* < 2^44 */
        if (s >= 64) {
            scaled = 0;
        } else {
            uint64_t q = prod >> s;
            const uint64_t rem = prod & ((1ull << s) - 1);
            const uint64_t half = 1ull << (s - 1);
            if (rem > half || (rem == half && (q & 1ull))) ++q;
            scaled = q;
        }
    }
    if (bits >> 31) *dst++ = '-';
    uint64_t ip = scaled / 1000000ull;
    uint32_t fp = (uint32_t)(scaled % 1000000ull);
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
    while (n) *dst++ = tmp[--n];
    *dst++ = '.';
    for (int k = 5; k >= 0; --k) { dst[k] = (char)('0' + fp % 10); fp /= 10; }
    return dst + 6;
}

} // namespace celio_txt

/* returns 0 ok / 1 error, like init_log() */
inline int celio_write_screen_txt(const char *path, int W, int H, const float *rgb,
                                  double run_time_s, double us_per_pixel, int n_cores = 1) {
    if (!path || W < 0 || H < 0 || (!rgb && (size_t)W * (size_t)H > 0)) return 1;
    std::FILE *f = std::fopen(path, "w");
    if (!f) {
        std::printf("Error Opening File %s\n", path);
        return 1;
    }
    std::fputs("OSX Awesome Picture\n", f);                    /* LOG_FILE_TITLE, src/rt_project_parameters.h:9 */
    std::fprintf(f, "Horizontal_Resolution:%i.\n", W);
    std::fprintf(f, "Vertical_Resolution:%i.\n", H);
    std::fprintf(f, "Hardware_Target:%s.\n", "OSX C++");        /* HARDWARE_TARGET, :21 */
    /* CORE_NUM and the PARTIONING_STRATEGY label, src/RayTracer.cpp:2037-2058: one "core" per GPU;
     * more than one means the static strip partition (strategy 1, DUMB_STATIC_PARTIONING) */
    std::fprintf(f, "Number_of_Cores:%i.\n", n_cores);
    std::fputs("IS_FOR_HARDWARE\n", f);
    std::fputs(n_cores > 1 ? "DUMB_STATIC_PARTIONING\n" : "NO_PARTIONING\n", f);
    std::fprintf(f, "Run_Time:%f.\n", run_time_s);
    std::fprintf(f, "us/pixel:%f.\n", us_per_pixel);
    std::fprintf(f, "filename:%s.\n", "raytracer_screen.txt");

    /* Pixel lines: blocks of pixels are formatted by a few threads at a time, each
     * into its own buffer (a line is at most 3 x 48 + 8 bytes), and written in
     * order; formatting, not the file system, is what takes the time. */
    const size_t n_pixels = (size_t)W * (size_t)H;
    const size_t kBlock = 1u << 17;                           /* pixels per block, ~4 MB of text */
    const size_t kLineMax = 3 * 48 + 8;
    unsigned n_threads = std::thread::hardware_concurrency();
    if (n_threads == 0) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    const size_t n_blocks = (n_pixels + kBlock - 1) / kBlock;
    if (n_blocks < n_threads) n_threads = n_blocks ? (unsigned)n_blocks : 1u;
    std::vector<std::unique_ptr<char[]>> bufs(n_threads);       /* uninitialised: only the bytes written get touched */
    for (auto &b : bufs) b.reset(new char[kBlock * kLineMax]);
    std::vector<size_t> used(n_threads, 0);
    int rc = 0;
    auto format_block = [&](unsigned slot, size_t first, size_t last) {
        char *const start = bufs[slot].get();
        char *p = start;
        for (size_t i = first; i < last; ++i) {
            const float *px = rgb + i * 3;
            *p++ = '(';
            p = celio_txt::format_f(p, px[0]); *p++ = ','; *p++ = ' ';
            p = celio_txt::format_f(p, px[1]); *p++ = ','; *p++ = ' ';
            p = celio_txt::format_f(p, px[2]); *p++ = ')'; *p++ = '\n';
        }
        used[slot] = (size_t)(p - start);
    };
    for (size_t b0 = 0; b0 < n_blocks; b0 += n_threads) {
        const unsigned in_round = (unsigned)std::min<size_t>(n_threads, n_blocks - b0);
        std::vector<std::thread> workers;
        for (unsigned t = 1; t < in_round; ++t)
            workers.emplace_back(format_block, t, (b0 + t) * kBlock, std::min(n_pixels, (b0 + t + 1) * kBlock));
        format_block(0, b0 * kBlock, std::min(n_pixels, (b0 + 1) * kBlock));
        for (std::thread &w : workers) w.join();
        for (unsigned t = 0; t < in_round; ++t)
            if (std::fwrite(bufs[t].get(), 1, used[t], f) != used[t]) rc = 1;
    }
    if (std::fclose(f)) rc = 1;
    return rc;
}

#endif /* SCREEN_TXT_HPP_ */
