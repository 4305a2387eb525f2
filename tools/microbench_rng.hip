// Cost of the sampler alone at the solver's occupancy (512 threads per CU, 20 elements per thread).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../museinference.jl_amd/csrc/rng.hpp"
template <int MODE>
__global__ void __launch_bounds__(512) k(double* out, uint64_t* cyc, uint64_t seed) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    double acc = 0;
#pragma unroll 2
    for (int i0 = 2 * tid; i0 < 10000; i0 += 1024) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int i = i0 + v;
            if (MODE == 0) {
                muse::NormalPair np = muse::normal_pair(seed, blockIdx.x, i);
                sm[i] = np.n1 * 1.3 + np.n2;
                sm[10000 + i] = np.n1;
            } else if (MODE == 1) {  // philox only
                uint32_t w[4];
                muse::philox4x32_10(i, 0, blockIdx.x, 0, (uint32_t)seed, 0, w);
                sm[i] = (double)(w[0] ^ w[1]);
                sm[10000 + i] = (double)(w[2] ^ w[3]);
            } else if (MODE == 3) {  // log only
                const double u1 = (i + 0.5) * 1e-4;
                sm[i] = muse::log_unit(u1);
                sm[10000 + i] = u1;
            } else if (MODE == 4) {  // sincospi only
                const double u2 = (i + 0.25) * 0.9e-4;
                double sn, cs;
                muse::sincospi_02(2.0 * u2, sn, cs);
                sm[i] = cs;
                sm[10000 + i] = sn;
            } else if (MODE == 5) {  // sqrt only
                const double u1 = (i + 0.5) * 1e-4;
                sm[i] = __builtin_sqrt(u1);
                sm[10000 + i] = u1;
            } else if (MODE == 6) {  // nothing (loop + LDS stores)
                const double u1 = (i + 0.5) * 1e-4;
                sm[i] = u1;
                sm[10000 + i] = u1 + 1;
            } else {  // box-muller only
                const double u1 = (i + 0.5) * 1e-4, u2 = (i + 0.25) * 0.9e-4;
                const double r = __builtin_sqrt(-2.0 * muse::log_unit(u1));
                double sn, cs;
                muse::sincospi_02(2.0 * u2, sn, cs);
                sm[i] = r * cs;
                sm[10000 + i] = r * sn;
            }
        }
    }
    __syncthreads();
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    acc = sm[tid] + sm[10000 + tid];
    out[blockIdx.x * 512 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
void run(const char* name) {
    double* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160000);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 160000, 0, out, cyc, 7ull);
    hipDeviceSynchronize();
    uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("%-16s %8.0f cycles per 10000-element sim (%.0f per element-slot of a wave)\n", name, m, m / 20);
}
int main() { run<0>("normal_pair"); run<1>("philox only"); run<2>("box-muller only"); run<3>("log only"); run<4>("sincospi only"); run<5>("sqrt only"); run<6>("empty loop"); return 0; }
