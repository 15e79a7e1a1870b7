// mfx_launch.h -- launch-side helpers shared by the kernel translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

namespace mfx {

// compute units of the CURRENT device (cached per device: one thread may drive handles on several GPUs -- afet_hip --devs,
// multi-rank setups)
int num_cus();
// hipOccupancyMaxActiveBlocksPerMultiprocessor cached per (device, function, threads, LDS bytes); `fallback` when the query fails
int blocks_per_cu(const void *func, int threads, size_t lds_bytes, int fallback);

// a kernel launched with more than 64 KB of dynamic LDS must be told so once (hipFuncAttributeMaxDynamicSharedMemorySize); the
// largest size granted is remembered per (device, function), so that the per-launch cost is a table lookup
hipError_t allow_dynamic_lds(const void *func, size_t lds_bytes);

} // namespace mfx
