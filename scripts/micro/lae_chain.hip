// Micro-benchmark: ONE dependent logaddexp chain per lane (the shape of a pair-beam window scan), at a chosen
// number of waves per SIMD (dynamic LDS limits the occupancy).  Prints ns per chain step per wave.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I poreover_amd/csrc scripts/micro/lae_chain.hip -o /tmp/lae_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "po_device.h"

template <int MODE>
__global__ __launch_bounds__(64) void chain_kernel(int iters, const double* y, double* sink) {
    extern __shared__ double dyn[];
    __shared__ PoLaeTables tb;
    __shared__ double ring[32][28];
    po_lae_tables_load(&tb, threadIdx.x, 64);
    for (int i = threadIdx.x; i < 32 * 28; i += 64) (&ring[0][0])[i] = -1.0 - 1e-3 * i;
    __syncthreads();
    const PoLaeFast lae{&tb};
    const int lane = threadIdx.x;
    double self = -1.0 - 1e-3 * (lane + 1);
    const int row = lane % 28, prow = (lane * 7 + 3) % 28;
    double mx = -1e300;
    int mt = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {          // bare chain: register operands only
            self = lae(self - 2.3, self - 0.11);
        } else if (MODE == 1) {   // + parent value and result through an LDS ring, one time per step
            const double pp = ring[(i - 1) & 31][prow];
            const double out = lae(pp - 2.3, self - 0.11);
            ring[i & 31][row] = out;
            mt = (out >= mx) ? i : mt;
            mx = fmax(mx, out);
            self = out;
        } else {                  // + y values from memory (L1/L2-resident), as the real loop
            const double ya = y[(i & 1023) * 5 + (lane & 3)], yb = y[(i & 1023) * 5 + 4];
            const double pp = ring[(i - 1) & 31][prow];
            const double out = lae(pp + ya, self + yb);
            ring[i & 31][row] = out;
            mt = (out >= mx) ? i : mt;
            mx = fmax(mx, out);
            self = out;
        }
    }
    if (self + mx + mt == 12345.678) sink[0] = self + dyn[0];
}

int main(int argc, char** argv) {
    const int iters = 20000;
    double *sink, *y;
    hipMalloc(&sink, 8);
    hipMalloc(&y, 1024 * 5 * 8);
    double hy[5120];
    for (int i = 0; i < 5120; ++i) hy[i] = -0.1 - (i % 7) * 0.5;
    hipMemcpy(y, hy, sizeof(hy), hipMemcpyHostToDevice);
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    for (int mode = 0; mode < 3; ++mode)
        for (int wps = 1; wps <= 8; wps *= 2) {   // waves per SIMD
            const int per_cu = 4 * wps;
            size_t lds = (size_t)(160 * 1024) / per_cu - 12 * 1024;   // static LDS ~ 10 KB: dynamic fills the rest
            if ((long)lds < 0) lds = 0;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            auto launch = [&](int it) {
                if (mode == 0) hipLaunchKernelGGL(chain_kernel<0>, dim3(cus * per_cu), dim3(64), lds, 0, it, y, sink);
                else if (mode == 1) hipLaunchKernelGGL(chain_kernel<1>, dim3(cus * per_cu), dim3(64), lds, 0, it, y, sink);
                else hipLaunchKernelGGL(chain_kernel<2>, dim3(cus * per_cu), dim3(64), lds, 0, it, y, sink);
            };
            launch(100);
            hipEventRecord(e0, 0);
            launch(iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("mode %d  waves/SIMD %d: %.1f ns per chain step per wave; %.2e steps/s chip-wide (x64 lanes = %.2e lae/s)\n", mode, wps,
                   ms * 1e6 / iters, (double)iters * cus * per_cu / (ms * 1e-3), 64.0 * iters * cus * per_cu / (ms * 1e-3));
        }
    return 0;
}
