// Dev microbenchmark: do LDS instructions and VALU instructions overlap on a CU (gfx950)?
// MODE 0: VALU only, 1: LDS reads only (ds_read_b128, conflict free), 2: both in every wave (interleaved),
// 3: half of the waves VALU-only and half LDS-only.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    __shared__ float4 lds[1024];
    const int tid = threadIdx.x;
    for (int i = tid; i < 1024; i += 256) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    float a0 = tid, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float4 s0 = make_float4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0;
    const float c = 1.0001f, d = 0.5f;
    const bool do_valu = MODE == 0 || MODE == 2 || (MODE == 3 && ((tid >> 6) & 1) == 0);
    const bool do_lds = MODE == 1 || MODE == 2 || (MODE == 3 && ((tid >> 6) & 1) == 1);
    int idx = tid & 63;
    for (int i = 0; i < iters; ++i) {
        if (do_valu) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a0 = __builtin_fmaf(a0, c, d); a1 = __builtin_fmaf(a1, c, d); a2 = __builtin_fmaf(a2, c, d); a3 = __builtin_fmaf(a3, c, d);
                a4 = __builtin_fmaf(a4, c, d); a5 = __builtin_fmaf(a5, c, d); a6 = __builtin_fmaf(a6, c, d); a7 = __builtin_fmaf(a7, c, d);
            }
        }
        if (do_lds) {
            // 8 independent conflict-free b128 reads
            const float4 *p = lds + idx;
            float4 v0 = p[0], v1 = p[64], v2 = p[128], v3 = p[192], v4 = p[256], v5 = p[320], v6 = p[384], v7 = p[448];
            s0.x += v0.x + v4.x; s1.x += v1.x + v5.x; s2.x += v2.x + v6.x; s3.x += v3.x + v7.x;
            idx = (idx + 64) & 511;
        }
    }
    out[blockIdx.x * 256 + tid] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + s0.x + s1.x + s2.x + s3.x;
}

template <int MODE>
float run(int blocks_per_cu, int iters)
{
    float *out;
    int blocks = 256 * blocks_per_cu;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms; (void)hipEventElapsedTime(&ms, a, b);
    (void)hipFree(out);
    return ms;
}
int main()
{
    const int iters = 20000;
    for (int w : {2, 4}) {
        float v = run<0>(w, iters), l = run<1>(w, iters), both = run<2>(w, iters), split = run<3>(w, iters);
        printf("waves/SIMD=%d  VALU-only %.3f ms | LDS-only %.3f ms | both in each wave %.3f ms | half/half waves %.3f ms (VALU half does %d, LDS half does %d per pair)\n",
               w, v, l, both, split, 32, 8);
    }
    return 0;
}
