// dev probe: does raw_buffer_load_b32 accept a byte offset that is only 2-byte aligned (sample pairs at odd sample offsets)?
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/unaligned_buf.hip -o tools/ubench/unaligned_buf && tools/ubench/unaligned_buf
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
__global__ void k(const uint16_t *src, uint32_t *dst, int n_bytes)
{
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, n_bytes, 0x00020000);
    const int lane = threadIdx.x;
    dst[lane] = __builtin_amdgcn_raw_buffer_load_b32(r, 2 * lane, 0, 0);           // odd lanes: 2-byte aligned only
    dst[64 + lane] = __builtin_amdgcn_raw_buffer_load_b32(r, n_bytes - 6 + 2 * (lane & 3), 0, 0); // around the end
}
int main()
{
    const int n = 512;
    std::vector<uint16_t> h(n);
    for (int i = 0; i < n; ++i) h[i] = (uint16_t)(1000 + i);
    uint16_t *d; uint32_t *o;
    hipMalloc(&d, n * 2 + 64); hipMalloc(&o, 128 * 4);
    hipMemset(d, 0xff, n * 2 + 64);
    hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, n * 2);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    std::vector<uint32_t> r(128);
    hipMemcpy(r.data(), o, 128 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const uint32_t want = (uint32_t)h[l] | ((uint32_t)h[l + 1] << 16);
        if (r[l] != want) { if (bad < 8) printf("lane %d got %08x want %08x\n", l, r[l], want); ++bad; }
    }
    printf("unaligned dword loads: %d of 64 wrong\n", bad);
    for (int l = 0; l < 4; ++l) printf("end probe off %d: %08x\n", n * 2 - 6 + 2 * l, r[64 + l]);
    return 0;
}
