// Dev microbenchmark: VALU issue rate of plain vs packed f32 ops on gfx950, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 0.5f;
    const v2f pc = {c, c}, pd = {d, d};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { // 8 independent plain FMAs
                a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
                a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
            } else if (MODE == 1) { // 8 independent packed FMAs
                p0 = __builtin_elementwise_fma(p0, pc, pd); p1 = __builtin_elementwise_fma(p1, pc, pd);
                p2 = __builtin_elementwise_fma(p2, pc, pd); p3 = __builtin_elementwise_fma(p3, pc, pd);
                p4 = __builtin_elementwise_fma(p4, pc, pd); p5 = __builtin_elementwise_fma(p5, pc, pd);
                p6 = __builtin_elementwise_fma(p6, pc, pd); p7 = __builtin_elementwise_fma(p7, pc, pd);
            } else { // 8 independent plain adds
                a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c;
            }
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y +
              p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int blocks_per_cu)
{
    int iters = 20000;
    float *out;
    int cus = 256, blocks = cus * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)iters * 64 * blocks_per_cu; // 64 VALU instr per iter per wave, waves/SIMD = blocks_per_cu
    printf("%-10s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cycles @2.4GHz)\n", name, blocks_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
    hipFree(out);
}
int main()
{
    for (int w : {1, 2, 4, 8}) { run<0>("fma", w); run<1>("pk_fma", w); run<2>("add", w); }
    return 0;
}
