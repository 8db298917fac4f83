// micro-benchmark: LDS instruction throughput on gfx950 (wave64), one CU's worth of wavefronts hammering the LDS
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(64) void k(float *out, int iters, int stride)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = 0.f;
    __syncthreads();
    float *p = lds + (threadIdx.x * stride) % 1792;
    const unsigned pa = (unsigned)(reinterpret_cast<size_t>(p) & 0xffffffffu); // LDS offset = low half of the generic address
    float v = threadIdx.x * 0.5f, acc = 0.f;
    int one = 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) asm volatile("ds_add_f32 %0, %1 offset:%2" :: "v"(pa), "v"(v), "n"(i * 4) : "memory");
            if (MODE == 1) { float t; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(t) : "v"(pa), "n"(i * 4) : "memory"); asm volatile("" :: "v"(t)); }
            if (MODE == 2) asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(pa), "v"(v), "n"(i * 4) : "memory");
            if (MODE == 3) asm volatile("ds_add_u32 %0, %1 offset:%2" :: "v"(pa), "v"(one), "n"(i * 4) : "memory");
            if (MODE == 5) { v4f t; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(t) : "v"(pa), "n"(i * 16) : "memory"); asm volatile("" :: "v"(t)); }
            if (MODE == 6) { v4f t = { v, v, v, v }; asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(pa), "v"(t), "n"(i * 16) : "memory"); }
            if (MODE == 7) { v2f t; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(t) : "v"(pa), "n"(i * 8) : "memory"); asm volatile("" :: "v"(t)); }
            if (MODE == 8) { v2f t = { v, v }; asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"(pa), "v"(t), "n"(i * 8) : "memory"); }
            if (MODE == 9) { v2f t; asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(t) : "v"(pa), "n"(i), "n"(i + 45) : "memory"); asm volatile("" :: "v"(t)); }
            if (MODE == 4) { float t; asm volatile("ds_add_rtn_f32 %0, %1, %2 offset:%3" : "=v"(t) : "v"(pa), "v"(v), "n"(i * 4) : "memory"); asm volatile("" :: "v"(t)); }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    acc = lds[threadIdx.x];
    if (acc == 123.456f) out[0] = acc;
}

template <int MODE> void run(const char *name, int waves_per_cu, int stride)
{
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * waves_per_cu;
    k<MODE><<<blocks, 64>>>(out, 10, stride);
    hipEventRecord(a);
    k<MODE><<<blocks, 64>>>(out, iters, stride);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double per_cu = (double)waves_per_cu * iters * 16;
    printf("%-16s stride %2d, %2d waves/CU: %.3f ms -> %.1f ns per wave-instruction per CU (%.1f cycles at 2.4 GHz)\n", name, stride, waves_per_cu, ms,
           ms * 1e6 / per_cu, ms * 1e6 / per_cu * 2.4);
    hipFree(out);
}

int main()
{
    for (int stride : { 1, 3 }) {
        run<0>("ds_add_f32", 8, stride);
        run<4>("ds_add_rtn_f32", 8, stride);
        run<3>("ds_add_u32", 8, stride);
        run<1>("ds_read_b32", 8, stride);
        run<2>("ds_write_b32", 8, stride);
    }
    // the batched Jacobi's pattern: lane r of each half owns a 28-float row (112-byte stride), 16-byte accesses; lanes 28..31 idle
    run<5>("ds_read_b128", 8, 28);
    run<6>("ds_write_b128", 8, 28);
    run<5>("ds_read_b128", 8, 4);
    run<6>("ds_write_b128", 8, 4);
    run<7>("ds_read_b64", 8, 2);
    run<8>("ds_write_b64", 8, 2);
    run<9>("ds_read2_b32", 8, 3);
    run<0>("ds_add_f32", 2, 3);
    run<0>("ds_add_f32", 16, 3);
    return 0;
}
