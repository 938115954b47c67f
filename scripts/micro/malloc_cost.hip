// What hipMalloc / hipHostMalloc cost on this box, as a sequence of operations given on the command line (the pipeline's first
// call pays them): hipcc --offload-arch=gfx950 -O2 scripts/micro/malloc_cost.hip -o scripts/micro/malloc_cost
//   malloc_cost OP ...   OP = <GB> (hipMalloc) | p<MB> (hipHostMalloc) | s (hipStreamCreate x4, non-blocking) | e (4 events) |
//                             k (a trivial kernel + sync) | f (free everything so far) | h2d (H2D rates from pinned memory)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void nop(int* p) { if (p) p[0] = 1; }
int main(int argc, char** argv) {
    const double t00 = now();
    (void)hipFree(nullptr);
    printf("hipFree(nullptr) (runtime start-up): %.2f ms\n", now() - t00);
    std::vector<void*> dev, host;
    double live = 0;
    for (int i = 1; i < argc; ++i) {
        const char* a = argv[i];
        const double t0 = now();
        if (!strcmp(a, "s")) {
            hipStream_t st[4];
            for (auto& s : st) (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
            printf("4 streams: %.2f ms\n", now() - t0);
        } else if (!strcmp(a, "e")) {
            hipEvent_t ev[4];
            for (auto& e : ev) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
            printf("4 events: %.2f ms\n", now() - t0);
        } else if (!strcmp(a, "k")) {
            hipLaunchKernelGGL(nop, dim3(1), dim3(64), 0, nullptr, (int*)nullptr);
            (void)hipDeviceSynchronize();
            printf("kernel + sync: %.2f ms\n", now() - t0);
        } else if (!strcmp(a, "f")) {
            for (void* p : dev) (void)hipFree(p);
            for (void* p : host) (void)hipHostFree(p);
            dev.clear(); host.clear(); live = 0;
            printf("free all: %.2f ms\n", now() - t0);
        } else if (!strcmp(a, "h2d")) {
            void *h = nullptr, *d = nullptr;
            (void)hipHostMalloc(&h, 256 << 20, hipHostMallocDefault);
            (void)hipMalloc(&d, 256 << 20);
            for (int mb : {1, 4, 16, 64, 256}) {
                (void)hipMemcpy(d, h, (size_t)mb << 20, hipMemcpyHostToDevice);
                const double t1 = now();
                for (int k = 0; k < 4; ++k) (void)hipMemcpyAsync(d, h, (size_t)mb << 20, hipMemcpyHostToDevice, nullptr);
                (void)hipDeviceSynchronize();
                printf("H2D %4d MB pinned: %6.2f GB/s\n", mb, 4.0 * mb / 1024.0 / ((now() - t1) * 1e-3));
            }
        } else if (a[0] == 'p') {
            void* p = nullptr;
            const double m = atof(a + 1);
            (void)hipHostMalloc(&p, (size_t)(m * (1 << 20)), hipHostMallocDefault);
            printf("hipHostMalloc %6.1f MB: %8.2f ms\n", m, now() - t0);
            host.push_back(p);
        } else {
            void* p = nullptr;
            const double g = atof(a);
            const hipError_t e = hipMalloc(&p, (size_t)(g * (1ull << 30)));
            live += g;
            printf("hipMalloc %7.3f GB (live %6.1f GB): %8.2f ms %s\n", g, live, now() - t0, e == hipSuccess ? "" : "FAILED");
            dev.push_back(p);
        }
    }
    return 0;
}
