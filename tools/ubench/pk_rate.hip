// micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_pk_mul_f32 on gfx950 (wave64), independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, long long *cyc, int iters)
{
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float r[16];
    v2f p[16];
    for (int i = 0; i < 16; ++i) { r[i] = i; p[i] = v2f{ (float)i, (float)i + 0.5f }; }
    v2f bb = { b, b }, aa = { a, a };
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
            if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(aa), "v"(bb));
            if (MODE == 2) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(p[i]) : "v"(bb));
            if (MODE == 3) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(b));
            if (MODE == 4) asm volatile("v_pk_add_f32 %0, %1, %0" : "+v"(p[i]) : "v"(bb));
            if (MODE == 5) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
            if (MODE == 6) { int sg; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(r[i])); asm volatile("" :: "s"(sg)); }
            if (MODE == 7) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
            if (MODE == 8) { int sg; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sg) : "v"(r[i])); asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "s"(sg), "v"(b)); }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MODE> void run(const char *name, int waves_per_simd)
{
    float *out; long long *cyc, h;
    hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
    const int iters = 4096;
    // one CU worth: blocks of 256 threads = 1 wave per SIMD; `waves_per_simd` blocks per CU need many blocks: launch 256 CUs * n
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, cyc, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, grid, block, 0, 0, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    // wave-instructions per SIMD per ns (1024 SIMDs): grid.x * 4 waves * iters * 16 instructions
    double winstr = (double)grid.x * 4 * iters * 16;
    printf("%-14s waves/SIMD %2d: %.2f counter ticks per instruction per wave; kernel %.3f ms -> %.3f ns per wave-instruction per SIMD\n", name, waves_per_simd,
           (double)h / (iters * 16.0), ms, ms * 1e6 / (winstr / 1024.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    for (int w : { 2, 8 }) {
        run<0>("v_fma_f32", w); run<3>("v_mul_f32", w); run<1>("v_pk_fma_f32", w); run<2>("v_pk_mul_f32", w); run<4>("v_pk_add_f32", w); run<5>("v_rcp_f32", w); run<6>("v_readlane", w); run<7>("v_mov_dpp", w); run<8>("readlane+fma", w);
    }
    return 0;
}
