// mfx_delta_dev.h -- the delta stage's device code shared by k_delta16 (mfx_tail.hip) and the opt-in fused delta wave of
// k_front512 (mfx_front512.hip): one arithmetic, the same bits (deltacpu.cpp:16-29, mfcccpu.cpp:234-263).
#pragma once
#include <hip/hip_runtime.h>

#include "mfx_dev.h"
#include "mfx_kernels.h"

namespace mfx {
namespace {

constexpr int kDeltaRows = 64;   // output rows per tile of the delta stage (k_delta and the fused delta wave)

// Arithmetic of one regression coefficient, shared by k_delta and the fused delta wave so that both give
// the same bits: num = sum_l l*(x[t+l] - x[t-l]) accumulated in ascending l, each step one fma; the
// quotient num / (2 sum l^2) (deltacpu.cpp:28) as reciprocal multiply + one exact-remainder correction,
// which equals the correctly rounded quotient away from the denormal range.
__device__ __forceinline__ float delta_quot(float num, float d, float inv)
{
    const float q = num * inv;
    const float r = __builtin_fmaf(-q, d, num);
    return __builtin_fmaf(r, inv, q);
}

// Regression numerator for 4 columns at once, split into its LDS reads and its arithmetic so that a caller
// can put the reads of several work items in flight before the first use.  L = compile-time order.
// Ascending l, one fma per step and column -- the same arithmetic as k_delta.
template <int L>
struct DeltaTaps {
    float4 a[L], b[L];
    __device__ __forceinline__ void load(const float4 *c)
    {
#pragma unroll
        for (int l = 1; l <= L; ++l) {
            a[l - 1] = c[4 * l];
            b[l - 1] = c[-4 * l];
        }
    }
    __device__ __forceinline__ float4 quot(float d, float inv) const
    {
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int l = 1; l <= L; ++l) {
            const float fl = (float)l;
            num.x = __builtin_fmaf(fl, a[l - 1].x - b[l - 1].x, num.x);
            num.y = __builtin_fmaf(fl, a[l - 1].y - b[l - 1].y, num.y);
            num.z = __builtin_fmaf(fl, a[l - 1].z - b[l - 1].z, num.z);
            num.w = __builtin_fmaf(fl, a[l - 1].w - b[l - 1].w, num.w);
        }
        return make_float4(delta_quot(num.x, d, inv), delta_quot(num.y, d, inv), delta_quot(num.z, d, inv),
                           delta_quot(num.w, d, inv));
    }
};

// run-time order (any l): reads and arithmetic interleaved
__device__ __forceinline__ float4 delta_quot4_rt(const float4 *c, int l_rt, float d, float inv)
{
    float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 1; l <= l_rt; ++l) {
        const float4 a = c[4 * l], b = c[-4 * l];
        const float fl = (float)l;
        num.x = __builtin_fmaf(fl, a.x - b.x, num.x);
        num.y = __builtin_fmaf(fl, a.y - b.y, num.y);
        num.z = __builtin_fmaf(fl, a.z - b.z, num.z);
        num.w = __builtin_fmaf(fl, a.w - b.w, num.w);
    }
    return make_float4(delta_quot(num.x, d, inv), delta_quot(num.y, d, inv), delta_quot(num.z, d, inv),
                       delta_quot(num.w, d, inv));
}

// the first `nvalid` components of v to 4 consecutive LDS words
__device__ __forceinline__ void lds_put4(float *dst, float4 v, int nvalid)
{
    if (nvalid > 0) dst[0] = v.x;
    if (nvalid > 1) dst[1] = v.y;
    if (nvalid > 2) dst[2] = v.z;
    if (nvalid > 3) dst[3] = v.w;
}

// LDS floats the delta wave needs: staged statics + deltas (16-float rows) and the output tile
__host__ __device__ inline int delta_wave_lds_floats(int l1, int l2)
{
    return ((kDeltaRows + 2 * (l1 + l2)) + (kDeltaRows + 2 * l2)) * 16 + kDeltaRows * 48 + 8;
}

// One tile of the delta stage, rows of <= 16 columns, executed by NT threads (`lane` = thread index): rows
// [r0, r0 + rows) of segment
// sg from the compact statics `src` (pitch 16, zeros beyond cols) to whole [static | d | dd] output rows.
// A work item is a quad of 4 columns of one row.  Statics (with the clamped context rows) and deltas are
// staged in LDS as 16-float rows; the finished rows are assembled in LDS exactly as they lie in memory
// (same position modulo 16 bytes) and leave as aligned 16-byte stores of consecutive lanes.
// `out` must be 16-byte aligned and out_pitch == cols * (l2 > 0 ? 3 : 2).
// The statics of one tile on their way from memory: (64 + 32) staged rows x 4 quads over NT threads.
template <int NT>
struct DeltaFill {
    static constexpr int kFill = (96 * 4 + NT - 1) / NT;
    float4 v[kFill];
    // issue the loads of rows [r0 - D, r0 + rows + D) of the segment (clamped, mfcccpu.cpp:243-256)
    __device__ __forceinline__ void issue(const Segment &sg, int r0, int rows, int D, const float *__restrict__ src, int lane)
    {
        const float *sbase = src + sg.src_row0 * 16;
        const int n_pad4 = (rows + 2 * D) * 4;
#pragma unroll
        for (int j = 0; j < kFill; ++j) {
            const int i = lane + NT * j;
            const int rr = i >> 2, q = i & 3;
            int sr = r0 + rr + sg.shift;
            sr = max(sg.lo, min(sg.hi, sr));
            if (i < n_pad4) v[j] = *(const float4 *)(sbase + sr * 16 + 4 * q);
        }
    }
};

template <int L1, int L2, int NT>
__device__ __forceinline__ void delta_tile16(const Segment &sg, int r0, int rows, const float *__restrict__ src,
                                             float *__restrict__ out, int out_pitch, int cols, int l1, int l2,
                                             float *smem, int lane, DeltaFill<NT> &fill, int next_r0, int next_rows,
                                             unsigned long long *ph = nullptr)
{
    // `fill` holds this tile's loads (DeltaFill::issue); once they are staged in LDS the loads of the
    // tile at next_r0 (next_rows > 0) are issued into it and fly during the rest of this tile.
    // NT = 64: one wave (wave-level ordering is enough); NT = 256: a whole block
    auto sync = [] {
        if (NT == 64)
            wave_sync();
        else
            __syncthreads();
    };
    constexpr int kFill = DeltaFill<NT>::kFill;
    float4(&v)[kFill] = fill.v;
#ifdef MFX_DSTAMPS
    unsigned long long ph_last, ph_t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_last)::"memory");
#define PSTAMP(i)                                                                         \
    do {                                                                                  \
        __builtin_amdgcn_s_waitcnt(0);                                                    \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t)::"memory");  \
        ph[i] += ph_t - ph_last;                                                          \
        ph_last = ph_t;                                                                   \
    } while (0)
#else
#define PSTAMP(i)
#endif
    const int D = l1 + l2;
    float4 *s_pad4 = (float4 *)smem;                      // [rows + 2D][4]
    float4 *s_d4 = s_pad4 + (kDeltaRows + 2 * D) * 4;     // [rows + 2*l2][4]
    float *s_out = (float *)(s_d4 + (kDeltaRows + 2 * l2) * 4); // the output tile, phase-shifted (below)
    const int stat_row = sg.static_off - sg.shift;        // s_pad row that holds the static part of output row 0
    // first output dword of the tile, and its position inside a 16-byte group
    const int64_t g0 = (sg.out_row0 + r0) * (int64_t)out_pitch;
    const int phase = (int)(g0 & 3);
    float *so = s_out + phase;                            // so[rr * out_pitch + cc]
    const int n_pad4 = (rows + 2 * D) * 4;   // <= (64 + 32) * 4 quads
    PSTAMP(0);
#pragma unroll
    for (int j = 0; j < kFill; ++j) {
        const int i = lane + NT * j;
        const int rr = i >> 2, q = i & 3;
        if (i < n_pad4) {
            s_pad4[i] = v[j];
            const int orow = rr - stat_row;
            if (orow >= 0 && orow < rows) lds_put4(so + orow * out_pitch + 4 * q, v[j], cols - 4 * q);
        }
    }
    if (next_rows > 0) fill.issue(sg, next_r0, next_rows, D, src, lane);
    sync();
    PSTAMP(1);
    float den = 0.f;
    for (int l = 1; l <= l1; ++l) den += (float)(l * l);
    const float d1 = 2 * den, inv1 = 1.0f / d1;
    const int n_d4 = (rows + 2 * l2) * 4;
    auto put_d = [&](int i, float4 d) {
        const int rr = i >> 2, q = i & 3;
        s_d4[i] = d;
        const int orow = rr - l2;
        if (orow >= 0 && orow < rows) lds_put4(so + orow * out_pitch + cols + 4 * q, d, cols - 4 * q);
    };
    if (L1 > 0 && NT == 64) { // one wave: two work items per trip, their 4*L1 LDS reads in flight together
        for (int i = lane; i < n_d4; i += 2 * NT) {
            const int i2 = i + NT < n_d4 ? i + NT : i;
            DeltaTaps<(L1 > 0 ? L1 : 1)> t0, t1;
            t0.load(s_pad4 + ((i >> 2) + l1) * 4 + (i & 3));
            t1.load(s_pad4 + ((i2 >> 2) + l1) * 4 + (i2 & 3));
            put_d(i, t0.quot(d1, inv1));
            if (i2 != i) put_d(i2, t1.quot(d1, inv1));
        }
    } else if (L1 > 0) {      // a block: about one work item per thread
        for (int i = lane; i < n_d4; i += NT) {
            DeltaTaps<(L1 > 0 ? L1 : 1)> t0;
            t0.load(s_pad4 + ((i >> 2) + l1) * 4 + (i & 3));
            put_d(i, t0.quot(d1, inv1));
        }
    } else {
        for (int i = lane; i < n_d4; i += NT) put_d(i, delta_quot4_rt(s_pad4 + ((i >> 2) + l1) * 4 + (i & 3), l1, d1, inv1));
    }
    sync();
    PSTAMP(2);
    if (l2 > 0) {
        float den2 = 0.f;
        for (int l = 1; l <= l2; ++l) den2 += (float)(l * l);
        const float d2 = 2 * den2, inv2 = 1.0f / d2;
        const int n_dd4 = rows * 4;
        auto put_dd = [&](int i, float4 dd) {
            const int rr = i >> 2, q = i & 3;
            lds_put4(so + rr * out_pitch + 2 * cols + 4 * q, dd, cols - 4 * q);
        };
        if (L2 > 0 && NT == 64) {
            for (int i = lane; i < n_dd4; i += 2 * NT) {
                const int i2 = i + NT < n_dd4 ? i + NT : i;
                DeltaTaps<(L2 > 0 ? L2 : 1)> t0, t1;
                t0.load(s_d4 + ((i >> 2) + l2) * 4 + (i & 3));
                t1.load(s_d4 + ((i2 >> 2) + l2) * 4 + (i2 & 3));
                put_dd(i, t0.quot(d2, inv2));
                if (i2 != i) put_dd(i2, t1.quot(d2, inv2));
            }
        } else if (L2 > 0) {
            for (int i = lane; i < n_dd4; i += NT) {
                DeltaTaps<(L2 > 0 ? L2 : 1)> t0;
                t0.load(s_d4 + ((i >> 2) + l2) * 4 + (i & 3));
                put_dd(i, t0.quot(d2, inv2));
            }
        } else {
            for (int i = lane; i < n_dd4; i += NT)
                put_dd(i, delta_quot4_rt(s_d4 + ((i >> 2) + l2) * 4 + (i & 3), l2, d2, inv2));
        }
        sync();
    }
    // the tile leaves: dwords [phase, phase + n) of s_out map to memory at (g0 - phase), which is 16-byte
    // aligned; whole quads as one 16-byte store per lane, the ragged first and last quad word by word
    const int n = rows * out_pitch, end = phase + n;
    float *gal = out + (g0 - phase);
    const int q_first = phase ? 1 : 0, q_last = end >> 2; // full quads: [q_first, q_last)
    // <= 64 * 48 / 4 = 768 quads = 12 per lane, in rounds of 4: the round's LDS reads first, then its stores
    for (int j0 = q_first + lane; j0 < q_last; j0 += 4 * NT) {
        float4 w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) // (reads are unconditional, from a clamped index: no divergent definitions)
            w[u] = *(const float4 *)(s_out + 4 * min(j0 + NT * u, q_last - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (j0 + NT * u < q_last) { // written once, never read by this kernel: non-temporal (-6 % on k_delta16)
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t = {w[u].x, w[u].y, w[u].z, w[u].w};
                __builtin_nontemporal_store(t, (v4f *)(gal + 4 * (j0 + NT * u)));
            }
    }
    if (lane < 4) {
        if (phase && lane >= phase && lane < end) gal[lane] = s_out[lane];           // head of the first quad
        const int t = 4 * q_last + lane;                                             // tail beyond the last full quad
        if (t < end && (t >= 4 || !phase)) gal[t] = s_out[t];
    }
    sync();
    PSTAMP(3);
}

} // namespace
} // namespace mfx
