// micro-benchmark: issue rate of the integer multiply forms on gfx950 (wave64): which of them are full rate.  Independent chains per lane,
// 8 wavefronts per SIMD; reports cycles per wave-instruction per SIMD at 2.4 GHz (4 = full rate, 8 = half, 16 = quarter).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(int *out, int iters, int s)
{
    unsigned a[8];
    unsigned long long q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 7 + i; q[i] = a[i]; }
    const unsigned m = 12u + (unsigned)s;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(m) : "vcc");
            if (MODE == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 2) asm volatile("v_mad_i32_i24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
            if (MODE == 3) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 4) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            if (MODE == 7) asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
            if (MODE == 8) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(a[i]) : "v"(m));
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r += a[i] + (unsigned)q[i];
    if (r == 0x12345u) out[0] = (int)r;
}

template <int MODE> void run(const char *name)
{
    int *out;
    hipMalloc(&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000, blocks = 256 * 8;   // 8 workgroups of 4 wavefronts per CU = 8 wavefronts per SIMD
    k<MODE><<<blocks, 256>>>(out, 10, 0);
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, iters, 0);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_simd = 8.0 * iters * 8;      // wave-instructions per SIMD
    printf("%-16s %.3f ms -> %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e6 / per_simd * 2.4);
    hipFree(out);
}

int main()
{
    run<6>("v_add_f32");
    run<4>("v_lshl_add_u32");
    run<2>("v_mad_i32_i24");
    run<8>("v_mad_u32_u24");
    run<3>("v_mul_u32_u24");
    run<1>("v_mul_lo_u32");
    run<5>("v_mul_hi_u32");
    run<0>("v_mad_u64_u32");
    run<7>("v_lshl_add_u64");
    return 0;
}
