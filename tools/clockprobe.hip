// VALU issue/latency probe (development aid): every workgroup runs C independent chains of dependent fp64 FMAs (mode 0),
// 64-bit integer multiply-adds v_mad_u64_u32 (mode 1), fp64 multiplies (mode 2), fp64 adds (mode 3) or 32-bit integer
// xors (mode 4), and reads s_memtime (shader clocks; s_memrealtime, 100 MHz, gives the clock) around the loop.
// Prints shader cycles per instruction per wave at 1 and 2 waves per SIMD.  usage: clockprobe [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

template <int MODE, int C>
__global__ void __launch_bounds__(1024) probe(unsigned long long* out, int iters, double seed) {
    double a[C];
    unsigned long long m[C];
    unsigned x[C];
    for (int c = 0; c < C; ++c) {
        a[c] = seed + threadIdx.x + c;
        m[c] = threadIdx.x + 12345 + c;
        x[c] = threadIdx.x * 7 + c;
    }
    const double b = 1.0000001, k = 0.5;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < C; ++c) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(b), "v"(k));
                if (MODE == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(m[c]) : "v"(x[c]), "v"(0xD2511F53u) : "vcc");
                if (MODE == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[c]) : "v"(b));
                if (MODE == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[c]) : "v"(k));
                if (MODE == 4) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[c]) : "v"(0x9E3779B9u));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    double acc = 0;
    for (int c = 0; c < C; ++c) acc += a[c] + (double)m[c] + (double)x[c];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = t1 - t0;
        out[blockIdx.x * 4 + 1] = r1 - r0;
    }
    if (acc == 1.2345) out[3] = 1;
}

template <int MODE, int C>
void run(unsigned long long* d, int iters, const char* name) {
    const int grid = 256;
    std::vector<unsigned long long> h(grid * 4);
    for (int threads : {256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) probe<MODE, C><<<grid, threads>>>(d, iters, 1.0 + rep);
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        probe<MODE, C><<<grid, threads>>>(d, iters * 20, 3.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        const double long_cyc = (double)h[0] / ((double)iters * 20 * 8 * C), long_clk = (double)h[0] / (double)h[1] / 10.0,
                     long_frac = (double)h[1] * 1e-8 / (1e-3 * ms);
        const double inst_per_s = (double)grid * (threads / 64) * (double)iters * 20 * 8 * C / (1e-3 * ms);   // wave instructions / s
        probe<MODE, C><<<grid, threads>>>(d, iters, 2.0);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> cyc, clk;
        for (int b = 0; b < grid; ++b) {
            cyc.push_back((double)h[b * 4] / ((double)iters * 8 * C));
            clk.push_back((double)h[b * 4] / (double)h[b * 4 + 1] / 10.0);
        }
        std::sort(cyc.begin(), cyc.end());
        std::sort(clk.begin(), clk.end());
        printf("%-16s chains %d  waves/SIMD %d: %.2f cycles per instruction per wave (SIMD issue interval %.2f), clock %.2f GHz; wall clock: %.1f G lane-ops/s (%.1f TFLOP/s if FMA); long run: %.2f cycles, clock %.2f GHz, in-kernel span %.2f of wall\n", name, C,
               threads / 256, cyc[grid / 2], cyc[grid / 2] / (threads / 256), clk[grid / 2], inst_per_s * 64 / 1e9, inst_per_s * 128 / 1e12, long_cyc, long_clk, long_frac);
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long* d;
    hipMalloc(&d, 256 * 4 * sizeof(unsigned long long));
    run<0, 8>(d, iters, "v_fma_f64");
    run<0, 1>(d, iters, "v_fma_f64"); run<0, 4>(d, iters, "v_fma_f64"); run<0, 8>(d, iters, "v_fma_f64");
    run<1, 1>(d, iters, "v_mad_u64_u32"); run<1, 4>(d, iters, "v_mad_u64_u32");
    run<4, 1>(d, iters, "v_xor_b32"); run<4, 4>(d, iters, "v_xor_b32");
    return 0;
}
