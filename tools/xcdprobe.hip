// Where do workgroups land, and what does a flag hand-off between two of them cost (development aid)?
//  1. census: XCC_ID (and CU id) of every workgroup of a 512 x 256-thread launch -- is it blockIdx mod 8?
//  2. ping-pong between workgroup 0 and workgroup P (P = 8: same XCD if the census says so; P = 1: neighbouring XCD)
//     with workgroup-scope (sc0: the XCD's L2) and agent-scope (sc1) relaxed atomics: shader cycles per round trip.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void census(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        out[2 * blockIdx.x] = xcc;
        out[2 * blockIdx.x + 1] = hwid;
    }
}

template <int SCOPE>
__global__ void pingpong(unsigned long long* flag, int partner, int iters, unsigned long long* out) {
    typedef __attribute__((address_space(1))) unsigned long long gu64;
    gu64* f = (gu64*)flag;
    if (threadIdx.x != 0) return;
    const bool a = blockIdx.x == 0, b = (int)blockIdx.x == partner;
    if (!a && !b) return;
    unsigned long long t0 = 0, t1 = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    bool ok = true;
    for (int i = 1; i <= iters && ok; ++i) {
        // a writes 2i-1, b answers 2i
        const unsigned long long mine = a ? 2ull * i - 1 : 2ull * i, want = a ? 2ull * i : 2ull * i - 1;
        if (a) __hip_atomic_store(f, mine, __ATOMIC_RELAXED, SCOPE);
        unsigned spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_RELAXED, SCOPE) < want) {
            if (++spins > (1u << 22)) { ok = false; break; }
        }
        if (b && ok) __hip_atomic_store(f, mine, __ATOMIC_RELAXED, SCOPE);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (a) { out[0] = t1 - t0; out[1] = ok ? 1 : 0; }
}

int main() {
    unsigned* d;
    hipMalloc(&d, 2 * 512 * sizeof(unsigned));
    census<<<512, 256>>>(d);
    hipDeviceSynchronize();
    std::vector<unsigned> h(2 * 512);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int b = 0; b < 512; ++b) bad += (h[2 * b] & 0xf) != (h[2 * (b % 8)] & 0xf);
    printf("XCC_ID of workgroups 0..15:");
    for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b] & 0xf);
    printf("\nworkgroups whose XCC_ID differs from that of workgroup (b mod 8): %d of 512\n", bad);
    printf("CU/SE ids (HW_ID >> 8 & 0xf, >> 13 & 7) of workgroups 0, 8, 16, 256, 264:");
    for (int b : {0, 8, 16, 256, 264}) printf(" [cu %u se %u]", (h[2 * b + 1] >> 8) & 0xf, (h[2 * b + 1] >> 13) & 0x7);
    printf("\n");
    unsigned long long *flag, *out;
    hipMalloc(&flag, 256);
    hipMalloc(&out, 64);
    const int iters = 2000;
    for (int partner : {8, 1, 16, 9}) {
        for (int scope = 0; scope < 2; ++scope) {
            hipMemset(flag, 0, 256);
            hipMemset(out, 0, 64);
            if (scope == 0) pingpong<__HIP_MEMORY_SCOPE_WORKGROUP><<<32, 64>>>(flag, partner, iters, out);
            else pingpong<__HIP_MEMORY_SCOPE_AGENT><<<32, 64>>>(flag, partner, iters, out);
            hipDeviceSynchronize();
            unsigned long long r[2];
            hipMemcpy(r, out, 16, hipMemcpyDeviceToHost);
            printf("ping-pong 0 <-> %2d, %s: %s, %.0f shader cycles per round trip\n", partner, scope == 0 ? "workgroup scope (sc0)" : "agent scope (sc1)    ",
                   r[1] ? "ok" : "TIMED OUT (not coherent)", (double)r[0] / iters);
        }
    }
    return 0;
}
