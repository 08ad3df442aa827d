// passbench -- torch-free timing of the restoration passes through the C ABI of a libfdr build chosen at run time
// (dlopen), so that one GPU call can compare several builds and sizes without paying a Python / torch start-up each.
//
//   passbench <libfdr.so> <size> [batch=8] [steps=10] [streams=1] [group=1] [mode=1] [flags=0] [two_sweep=-1 (library default)] [graph=0] [ce_chunk_mb=-1]
//
// Prints: per-pass mean device time (hipEvent pairs, one stream, un-overlapped) with the fraction of the 8 TB/s HBM
// peak its algorithmic bytes give, then the batched throughput with the requested streams / group (median of 5).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); std::exit(1); } } while (0)

typedef struct fdr_plan fdr_plan;
struct Api {
    int (*plan_create)(int, int, int, int, unsigned, fdr_plan**);
    int (*plan_destroy)(fdr_plan*);
    int (*set_psf_motion)(fdr_plan*, int, double, float, void*);
    int (*synth)(int, uint64_t, uint64_t, size_t, float*, void*);
    int (*set_batching)(fdr_plan*, int, int);
    int (*profile)(fdr_plan*, int);
    int (*pass_times)(fdr_plan*, int*, float*, const char**, int*);
    int (*batch_dev)(fdr_plan*, const float*, size_t, int, int, int, int, float*, size_t, int, int, void*);
    int (*set_option)(fdr_plan*, int, long long);
    const char* (*last_error)(void);
};
#define FCK(x) do { int r_ = (x); if (r_ != 0) { std::printf("fdr error %d: %s at %s:%d\n", r_, api.last_error(), __FILE__, __LINE__); std::exit(1); } } while (0)

static double bytes_per_px(const std::string& name, bool half) {
    if (name.rfind("A ", 0) == 0) return half ? 8 : 12;
    if (name.rfind("B'", 0) == 0) return half ? 12 : 24;
    if (name.rfind("C'E", 0) == 0) return half ? 8 : 12;
    if (name.rfind("C1", 0) == 0) return 4;
    if (name.rfind("C2", 0) == 0) return 8;
    if (name.rfind("C'", 0) == 0) return half ? 8 : 12;
    if (name.rfind("E ", 0) == 0) return 8;
    if (name.rfind("B ", 0) == 0) return 24;
    if (name.rfind("C ", 0) == 0) return 16;
    if (name.rfind("D ", 0) == 0) return 12;
    return 0;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::printf("usage: passbench <libfdr.so> <size> [batch] [steps] [streams] [group] [mode] [flags]\n"); return 2; }
    const char* libpath = argv[1];
    const int S = std::atoi(argv[2]);
    const int B = argc > 3 ? std::atoi(argv[3]) : 8;
    const int steps = argc > 4 ? std::atoi(argv[4]) : 10;
    const int streams = argc > 5 ? std::atoi(argv[5]) : 1;
    const int group = argc > 6 ? std::atoi(argv[6]) : 1;
    const int mode = argc > 7 ? std::atoi(argv[7]) : 1;
    const unsigned flags = argc > 8 ? (unsigned)std::strtoul(argv[8], nullptr, 0) : 0u;
    const int two_sweep = argc > 9 ? std::atoi(argv[9]) : -1;
    const int graph = argc > 10 ? std::atoi(argv[10]) : 0;
    const int ce_chunk_mb = argc > 11 ? std::atoi(argv[11]) : -1;  // FDR_OPT_CE_CHUNK_MB (-1: library default)
    void* h = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
    if (!h) { std::printf("dlopen %s: %s\n", libpath, dlerror()); return 1; }
    Api api;
#define SYM(field, name) do { *(void**)(&api.field) = dlsym(h, name); if (!api.field) { std::printf("missing symbol %s\n", name); return 1; } } while (0)
    SYM(plan_create, "fdr_plan_create"); SYM(plan_destroy, "fdr_plan_destroy"); SYM(set_psf_motion, "fdr_set_psf_motion");
    SYM(synth, "fdr_synth_image_dev"); SYM(set_batching, "fdr_plan_set_batching"); SYM(profile, "fdr_plan_profile");
    SYM(pass_times, "fdr_plan_pass_times"); SYM(batch_dev, "fdr_wiener_batch_f32_dev"); SYM(last_error, "fdr_last_error");
    SYM(set_option, "fdr_plan_set_option");

    const size_t P = (size_t)S * S;
    float *d_in = nullptr, *d_out = nullptr;
    CK(hipMalloc((void**)&d_in, P * B * sizeof(float)));
    CK(hipMalloc((void**)&d_out, P * B * sizeof(float)));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    fdr_plan* plan = nullptr;
    FCK(api.plan_create(0, S, S, mode, flags, &plan));
    if (two_sweep >= 0) FCK(api.set_option(plan, 2 /* FDR_OPT_TWO_SWEEP_NORM */, two_sweep));
    if (ce_chunk_mb >= 0) FCK(api.set_option(plan, 4 /* FDR_OPT_CE_CHUNK_MB */, ce_chunk_mb));
    FCK(api.set_psf_motion(plan, 50, 30.0, 0.01f, st));
    FCK(api.synth(0, 0x5EED0003ull, 0, P * B, d_in, st));
    CK(hipStreamSynchronize(st));
    const bool half = mode == 1 && !(flags & 32u) && S >= 32;

    // un-overlapped per-pass times: one stream, one image per launch unless group > 1 was asked for
    FCK(api.set_batching(plan, 1, group));
    for (int k = 0; k < 2; ++k) FCK(api.batch_dev(plan, d_in, P, B, S, S, S, d_out, P, S, 1, st));
    CK(hipStreamSynchronize(st));
    FCK(api.profile(plan, 1));
    for (int k = 0; k < steps; ++k) FCK(api.batch_dev(plan, d_in, P, B, S, S, S, d_out, P, S, 1, st));
    CK(hipStreamSynchronize(st));
    int n = 0, launches[16];
    float ms[16];
    const char* names[16];
    FCK(api.pass_times(plan, &n, ms, names, launches));
    FCK(api.profile(plan, 0));
    double sum_us = 0;
    std::printf("== %s  size %d  batch %d  steps %d  mode %d flags %u two_sweep %d ce_chunk_mb %d\n", libpath, S, B, steps, mode, flags, two_sweep, ce_chunk_mb);
    for (int i = 0; i < n; ++i) {
        std::string nm = names[i];
        int nimg = 1;
        size_t pos = nm.rfind(" [");
        if (pos != std::string::npos && nm.size() > pos + 2 && std::isdigit((unsigned char)nm[pos + 2])) nimg = std::atoi(nm.c_str() + pos + 2);
        const double us = ms[i] * 1e3 / nimg;
        const double bpp = bytes_per_px(nm, half);
        std::printf("  %-44s %8.2f us/image  (%d launches)  %5.1f B/px  %6.0f GB/s  frac %.3f\n", nm.c_str(), us, launches[i], bpp, bpp * P / us / 1e3,
                    bpp * P / us / 1e3 / 8000.0);
        sum_us += us;
    }
    std::printf("  sum of passes %.2f us/image = %.0f Mpixels/s single stream\n", sum_us, P / sum_us);
    {   // -DFDR_DEBUG_STAMPS builds: phase stamps of pass B' (wave 0 of every workgroup of the LAST launch), shader-clock ticks
        int (*read_stamps)(unsigned long long*, size_t) = nullptr;
        *(void**)(&read_stamps) = dlsym(h, "fdr_debug_read_stamps");
        if (read_stamps) {
            const int ntiles = S / 8, wgs = std::min(8192, ntiles * std::max(1, group));
            std::vector<unsigned long long> stp((size_t)8192 * 32);
            if (read_stamps(stp.data(), stp.size()) == 0) {
                static const char* ph[6] = {"tile load (issue + landed)", "forward transforms", "filter W (loads + products)", "inverse transforms", "stores issued", "stores retired"};
                double acc[6] = {0}, life = 0, inner[2][8] = {{0}};
                unsigned long long t_first = ~0ull, t_last = 0;
                int cnt = 0;
                for (int b = 0; b < wgs; ++b) {
                    const unsigned long long* q = &stp[(size_t)b * 32];
                    if (q[0] == 0 || q[6] <= q[0]) continue;
                    for (int i = 0; i < 6; ++i) acc[i] += (double)(q[i + 1] - q[i]);
                    life += (double)(q[6] - q[0]);
                    // inside the transforms: slot 8 + 2 J = butterflies of step J done, 8 + 2 J + 1 = exchange J done (16.. inverse)
                    for (int d = 0; d < 2; ++d) {
                        unsigned long long prev = d == 0 ? q[1] : q[3];
                        for (int i = 0; i < 8; ++i) {
                            const unsigned long long t = q[8 + 8 * d + i];
                            if (t == 0 || t < prev) break;
                            inner[d][i] += (double)(t - prev);
                            prev = t;
                        }
                    }
                    t_first = std::min(t_first, q[0]); t_last = std::max(t_last, q[6]);
                    ++cnt;
                }
                if (cnt) {
                    std::printf("  pass B' phase stamps (wave 0 of %d workgroups of the last launch; counter ticks, mean per workgroup):\n", cnt);
                    for (int i = 0; i < 6; ++i) std::printf("    %-30s %9.0f  (%4.1f %%)\n", ph[i], acc[i] / cnt, 100.0 * acc[i] / life);
                    std::printf("    %-30s %9.0f\n", "workgroup lifetime", life / cnt);
                    (void)t_first; (void)t_last;
                    for (int d = 0; d < 2; ++d) {
                        std::printf("    inside the %s transforms (4 columns):", d == 0 ? "forward" : "inverse");
                        for (int i = 0; i < 8 && inner[d][i] > 0; ++i) std::printf("  %s%d %.0f", (i & 1) ? "exchange" : "butterflies", i / 2, inner[d][i] / cnt);
                        std::printf("\n");
                    }
                }
            }
        }
    }

    // throughput with the requested streams x group
    FCK(api.set_batching(plan, streams, group));
    if (graph) FCK(api.set_option(plan, 3 /* FDR_OPT_BATCH_GRAPH */, 1));
    for (int k = 0; k < 2; ++k) FCK(api.batch_dev(plan, d_in, P, B, S, S, S, d_out, P, S, 1, st));
    CK(hipStreamSynchronize(st));
    std::vector<double> t;
    for (int r = 0; r < 5; ++r) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < steps; ++k) FCK(api.batch_dev(plan, d_in, P, B, S, S, S, d_out, P, S, 1, st));
        CK(hipStreamSynchronize(st));
        t.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
    std::sort(t.begin(), t.end());
    const double per_img_us = t[2] / ((double)steps * B) * 1e6;
    std::printf("  batched%s %d streams x %d per launch: %.2f us/image = %.0f Mpixels/s (median of 5; min %.2f max %.2f us)\n", graph ? " (graph replay)" : "", streams, group, per_img_us,
                P / per_img_us, t[0] / ((double)steps * B) * 1e6, t[4] / ((double)steps * B) * 1e6);
    // checksum so that variants can be compared for equality of results
    std::vector<float> host(P);
    CK(hipMemcpy(host.data(), d_out, P * sizeof(float), hipMemcpyDeviceToHost));
    double cs = 0;
    for (size_t i = 0; i < P; ++i) cs += host[i];
    std::printf("  checksum image 0: %.6f\n", cs);
    FCK(api.plan_destroy(plan));
    return 0;
}
