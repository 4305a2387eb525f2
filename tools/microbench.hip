// Instruction-cost microbenchmark at the solver's occupancy (512-thread workgroup, 1 per CU).
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o gpurun_out/microbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define REP 256
template <int OP, int ILP>
__global__ void __launch_bounds__(512) k(double* out, uint64_t* cyc, double a0, uint32_t u0) {
    double a[ILP];
    uint32_t u[ILP];
    uint64_t w[ILP];
    for (int i = 0; i < ILP; ++i) { a[i] = a0 + i + threadIdx.x * 1e-3; u[i] = u0 + i + threadIdx.x; w[i] = u[i]; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (OP == 0) a[i] = __builtin_fma(a[i], 1.0000001, 0.5);
            if (OP == 1) a[i] = a[i] * 1.0000001;
            if (OP == 2) a[i] = a[i] + 0.5;
            if (OP == 3) { w[i] = (uint64_t)(uint32_t)w[i] * 0xD2511F53u + (w[i] >> 32); }
            if (OP == 4) u[i] = u[i] * 0xCD9E8D57u + 1;
            if (OP == 5) u[i] = __umulhi(u[i], 0xD2511F53u) + 1;
            if (OP == 6) a[i] = __builtin_amdgcn_rcp(a[i]);
            if (OP == 7) a[i] = __builtin_amdgcn_rsq(a[i]);
            if (OP == 8) u[i] = u[i] ^ (u[i] >> 3);
            if (OP == 9) a[i] = a[i] / 1.37;
            if (OP == 10) a[i] = __builtin_sqrt(a[i]);
            if (OP == 11) a[i] = (double)u[i] + a[i], u[i] += 3;
            if (OP == 12) a[i] = __builtin_ldexp(a[i], 1);
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    double s = 0; uint64_t q = 0;
    for (int i = 0; i < ILP; ++i) { s += a[i]; q += u[i] + w[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (double)q;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int OP, int ILP>
void run(const char* name) {
    double* out; uint64_t* cyc;
    hipMalloc(&out, 256 * 512 * 8); hipMalloc(&cyc, 256 * 8);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<OP, ILP>), dim3(256), dim3(512), 0, 0, out, cyc, 1.5, 12345u);
    hipDeviceSynchronize();
    uint64_t h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
    printf("%-22s ILP %d: %7.2f cycles per wave-instruction (2 waves/SIMD -> %.2f per SIMD-instruction)\n", name, ILP,
           m / (REP * ILP), m / (REP * ILP) / 2);
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0, 1>("v_fma_f64"); run<0, 4>("v_fma_f64"); run<0, 8>("v_fma_f64");
    run<1, 4>("v_mul_f64"); run<2, 4>("v_add_f64");
    run<3, 1>("v_mad_u64_u32"); run<3, 4>("v_mad_u64_u32");
    run<4, 4>("v_mul_lo_u32(+add)"); run<5, 4>("v_mul_hi_u32(+add)");
    run<6, 4>("v_rcp_f64"); run<7, 4>("v_rsq_f64"); run<8, 4>("xor+shift (2 ops)");
    run<9, 4>("f64 divide"); run<10, 4>("f64 sqrt"); run<11, 4>("cvt_f64_u32+add+iadd"); run<12, 4>("v_ldexp_f64");
    return 0;
}
