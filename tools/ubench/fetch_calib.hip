// Dev microbenchmark: calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access
// shapes the MFCC kernels use (4 B per lane reads of int16 pairs; 52-byte row pieces written at a
// 156-byte pitch), on known byte counts far larger than the 256 MiB Infinity Cache.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void k_read_dword(const uint32_t *in, uint32_t *out, size_t n_words)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) acc ^= in[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_read_dwordx4(const uint4 *in, uint32_t *out, size_t n_vec)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = in[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// rows of 39 floats; lanes 0..12 of each 16-lane group write the first 13 floats of one row
__global__ void k_write_rows13(float *out, size_t n_rows)
{
    const int l = threadIdx.x & 15;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4; r < n_rows; r += ((size_t)gridDim.x * blockDim.x) >> 4)
        if (l < 13) out[r * 39 + l] = (float)l;
}
// full rows: 39 floats written by 39 consecutive threads of a tile (the delta kernel's pattern)
__global__ void k_write_rows39(float *out, size_t n_rows)
{
    const size_t total = n_rows * 39;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) out[i] = 1.0f;
}
int main()
{
    const size_t bytes = (size_t)1 << 30; // 1 GiB
    uint32_t *in, *out;
    float *rows;
    const size_t n_rows = 4000000; // 4M rows x 156 B = 624 MB buffer; 13-float pieces = 208 MB of payload
    (void)hipMalloc(&in, bytes);
    (void)hipMalloc(&out, 64);
    (void)hipMalloc(&rows, n_rows * 39 * 4);
    (void)hipMemset(in, 1, bytes);
    (void)hipMemset(rows, 0, n_rows * 39 * 4);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k_read_dword, dim3(4096), dim3(256), 0, 0, in, out, bytes / 4);
    hipLaunchKernelGGL(k_read_dwordx4, dim3(4096), dim3(256), 0, 0, (const uint4 *)in, out, bytes / 16);
    hipLaunchKernelGGL(k_write_rows13, dim3(4096), dim3(256), 0, 0, rows, n_rows);
    hipLaunchKernelGGL(k_write_rows39, dim3(4096), dim3(256), 0, 0, rows, n_rows);
    (void)hipDeviceSynchronize();
    printf("known bytes: read_dword %zu, read_dwordx4 %zu, write_rows13 %zu (payload), write_rows39 %zu\n", bytes, bytes,
           n_rows * 52, n_rows * 156);
    return 0;
}
