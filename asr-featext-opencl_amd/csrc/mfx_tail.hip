// mfx_tail.hip -- everything behind the front end, and their launchers:
//   k_melcep      stored magnitudes -> mel -> log -> DCT (streaming apply(), VTLN sweeps, the 4096-point batch path)
//   k_delta16 / k_delta4 / k_delta   delta + delta-delta (deltacpu.cpp:16-29, mfcccpu.cpp:234-263), whole output rows
//   k_norm_seg / k_norm_stats / k_norm_finalize / k_norm_apply   CMN / CVN / MINMAX (normalizercpu.cpp:22-89)
//   k_copy_small  small streaming blocks through pinned memory
// See DESIGN.md section 5.
#include "mfx_kernels.h"

#include <hip/hip_runtime.h>

#include "mfx_dev.h"
#include "mfx_launch.h"

#include <algorithm>
#include <cstdlib>
#include "mfx_delta_dev.h"

namespace mfx {

namespace {

// ------------------------------------------------------------------------------------------------
// melcep: stored magnitudes -> mel -> log -> DCT (streaming apply(), VTLN sweeps, the 4096-point batch path).  A wave
// takes 4 consecutive rows at a time: each row's magnitudes go to the wave's LDS buffer, its filters are walked on the
// wave's 64 lanes (MelWavePlan: whole filters in ascending bin order, mfcccpu.cpp:206-217), the log energies wait in
// lm[4][FS], and the DCT of the four rows runs on the matrix pipe (dct_mfma4) -- the same mel stage as the fused batch
// kernels (round 3: the round-1 piece plan and its vector-pipe DCT are gone).  blockIdx.y = filterbank of a VTLN sweep.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_melcep(MelcepParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int nb = p.num_banks, RS = p.mel64_row_stride, rounds = p.mel64_rounds;
    const int FS = lm_fs4(nb), MF = p.mag_floats;
    const int WR = mel64_rows(nb);                   // weight rows in LDS (lanes that carry a filter)
    float *s_mw = smem;                              // [WR][RS]
    int *s_mst = (int *)(s_mw + WR * RS);            // [rounds][64]
    int *s_mfid = s_mst + 64 * rounds;               // [rounds][64]
    int *s_L = s_mfid + 64 * rounds;                 // [8]
    float *s_wave = (float *)(s_L + 8) + wave * (MF + 4 * FS);
    float *mag = s_wave, *lm = s_wave + MF;

    const int table = blockIdx.y;
    const float *gw = p.mel64_w + (int64_t)table * 64 * RS;
    const int32_t *gst = p.mel64_start + (int64_t)table * 64 * rounds, *gfid = p.mel64_fid + (int64_t)table * 64 * rounds;
    float *feat = p.feat + (int64_t)table * p.feat_table_stride;
    for (int i = tid; i < WR * RS; i += blockDim.x) s_mw[i] = gw[i];
    for (int i = tid; i < 64 * rounds; i += blockDim.x) {
        s_mst[i] = gst[i];
        s_mfid[i] = gfid[i];
    }
    if (tid < 8) s_L[tid] = p.mel64_L[table * 8 + tid];
    for (int i = lane; i < MF + 4 * FS; i += 64) s_wave[i] = 0.f; // words past the last bin stay zero (finite) for good
    __syncthreads();

    const int dct_ks = p.dct_ksteps, dct_tiles64 = (p.dct_len + 63) >> 6;
    const int dct_bytes = p.dct_b4 ? dct_tiles64 * dct_ks * 1024 : 0;
    const __amdgpu_buffer_rsrc_t dct_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dct_b4, 0, dct_bytes, 0x00020000);
    const int q4 = p.spec_pitch >> 2; // rows are whole 16-byte words (spec_pitch is a multiple of 4, rows 16-byte aligned)
    const int nbins = (p.fft_size >> 1) + 1;
    for (int64_t grp = (int64_t)blockIdx.x * n_waves + wave; grp * 4 < p.n_rows; grp += (int64_t)gridDim.x * n_waves) {
        const int64_t row0 = grp * 4;
        const int count = (int)(p.n_rows - row0 < 4 ? p.n_rows - row0 : 4);
        for (int f = 0; f < count; ++f) {
            const float4 *src = (const float4 *)(p.spec + (row0 + f) * p.spec_pitch);
            for (int k = lane; k < q4; k += 64) ((float4 *)mag)[k] = src[k];
            // the row's padding words (bins > W2/2) are never written in memory: they meet zero weights in the walk and
            // must be finite (0 x NaN is NaN)
            if (nbins + lane < 4 * q4) mag[nbins + lane] = 0.f;
            wave_sync();
            mel64_walk_log(mag, lm + f * FS, FS - 1, s_mw, s_mst, s_mfid, s_L, rounds, RS, lane, WR);
            wave_sync();
        }
        dct4_store<3>(lm, FS, dct_rsrc, dct_bytes, dct_ks, dct_tiles64, p.dct_b4 != nullptr, lane, p.cols, feat, (int64_t)p.feat_pitch,
                   row0, count);
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// delta: regression coefficients over time (deltacpu.cpp:16-29) with the edge handling of
// MfccCpu::do_delta (mfcccpu.cpp:234-263) expressed as a clamped row accessor (Segment).
// grid = (tiles, segments); one tile = kDeltaRows output rows.
// ------------------------------------------------------------------------------------------------
// FAST16: cols <= 16 -> a row is 16 consecutive work items (no integer division by a run-time
// column count, 13..16 consecutive floats per row piece); otherwise the generic index split.
template <bool FAST16, int ROWS>
__global__ void __launch_bounds__(256) k_delta(DeltaParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.y];
    const int r0 = blockIdx.x * ROWS;
    if (r0 >= sg.n_out) return;
    const int rows = min(ROWS, sg.n_out - r0);
    const int cols = p.cols, l1 = p.l1, l2 = p.l2, D = l1 + l2;
    const int cw = FAST16 ? 16 : cols; // row width in work items and in LDS
    const int tid = threadIdx.x;
    float *s_pad = smem;                             // [rows + 2D][cw]
    float *s_d = smem + (ROWS + 2 * D) * cw;   // [rows + 2*l2][cw]
    // (tile indices stay below 2^16: floor(i / cols) as a multiply-high instead of an integer division per element)
    const uint32_t magic_c = 0xffffffffu / (uint32_t)(cols > 0 ? cols : 1) + 1;
    auto split = [&](int i, int &rr, int &c) {
        if (FAST16) {
            rr = i >> 4;
            c = i & 15;
        } else {
            rr = (int)__umulhi((uint32_t)i, magic_c);
            c = i - rr * cols;
        }
    };

    if (l1 > 0) {
        const int n_pad = (rows + 2 * D) * cw;
        for (int i = tid; i < n_pad; i += 256) {
            int rr, c;
            split(i, rr, c);
            int sr = r0 + rr + sg.shift;
            sr = max(sg.lo, min(sg.hi, sr));
            s_pad[i] = (c < cols) ? p.src[(sg.src_row0 + sr) * (int64_t)p.src_pitch + c] : 0.f;
        }
        __syncthreads();
        float den = 0.f;
        for (int l = 1; l <= l1; ++l) den += (float)(l * l);
        const float d1 = 2 * den, inv1 = 1.0f / d1;
        const int n_d = (rows + 2 * l2) * cw;
        for (int i = tid; i < n_d; i += 256) {
            float num = 0.f;
            for (int l = 1; l <= l1; ++l)
                num = __builtin_fmaf((float)l, s_pad[i + (l1 + l) * cw] - s_pad[i + (l1 - l) * cw], num);
            s_d[i] = delta_quot(num, d1, inv1);
        }
        __syncthreads();
    }
    float *s_dd = s_d + (ROWS + 2 * l2) * cw;  // [rows][cw]
    if (l2 > 0) {
        float den2 = 0.f;
        for (int l = 1; l <= l2; ++l) den2 += (float)(l * l);
        const float d2 = 2 * den2, inv2 = 1.0f / d2;
        const int n_dd = rows * cw;
        for (int i = tid; i < n_dd; i += 256) {
            float num = 0.f;
            for (int l = 1; l <= l2; ++l)
                num = __builtin_fmaf((float)l, s_d[i + (l2 + l) * cw] - s_d[i + (l2 - l) * cw], num);
            s_dd[i] = delta_quot(num, d2, inv2);
        }
        __syncthreads();
    }
    // The tile's output rows are one contiguous block of rows * width floats: write it with
    // consecutive threads on consecutive addresses (whole cache lines), statics included.
    const int width = p.out_pitch == cols * (l2 > 0 ? 3 : l1 > 0 ? 2 : 1) ? p.out_pitch : 0;
    const int stat_row = sg.static_off - sg.shift; // row of s_pad that holds the static part of output row 0
    if (width > 0 && l1 > 0) {
        float *obase = p.out + (sg.out_row0 + r0) * (int64_t)p.out_pitch;
        const uint32_t magic = 0xffffffffu / (uint32_t)width + 1; // floor(i / width) for i < 2^16
        const int n_o = rows * width;
        for (int i = tid; i < n_o; i += 256) {
            const int rr = (int)__umulhi((uint32_t)i, magic);
            const int cc = i - rr * width;
            float v;
            if (cc < cols)
                v = s_pad[(rr + stat_row) * cw + cc];
            else if (cc < 2 * cols)
                v = s_d[(rr + l2) * cw + cc - cols];
            else
                v = s_dd[rr * cw + cc - 2 * cols];
            obase[i] = v;
        }
        return;
    }
    // generic fallback (row pitch wider than the feature row, or statics only)
    const int n_o = rows * cw;
    for (int i = tid; i < n_o; i += 256) {
        int rr, c;
        split(i, rr, c);
        if (c >= cols) continue;
        float *orow = p.out + (sg.out_row0 + r0 + rr) * (int64_t)p.out_pitch;
        orow[c] = p.src[(sg.src_row0 + r0 + rr + sg.static_off) * (int64_t)p.src_pitch + c];
        if (l1 > 0) {
            orow[cols + c] = s_d[i + l2 * cw];
            if (l2 > 0) orow[2 * cols + c] = s_dd[i];
        }
    }
}

// The same stage for wide rows whose column count is a multiple of 4 (BASELINE configs[4]: 40 columns, 120-float rows): a
// work item is 4 consecutive columns, every load / LDS access / store a 16-byte word -- a quarter of k_delta's memory
// instructions and no per-element index arithmetic.  Same arithmetic per element as k_delta (delta_quot), same tile shape;
// requires cols % 4 == 0, src_pitch % 4 == 0, out_pitch == cols * groups, 16-byte aligned src / out.
template <int ROWS>
__global__ void __launch_bounds__(256) k_delta4(DeltaParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.y];
    const int r0 = blockIdx.x * ROWS;
    if (r0 >= sg.n_out) return;
    const int rows = min(ROWS, sg.n_out - r0);
    const int l1 = p.l1, l2 = p.l2, D = l1 + l2;
    const int q = p.cols >> 2;                       // 16-byte words per group of columns
    const int tid = threadIdx.x;
    float4 *s_pad = (float4 *)smem;                  // [rows + 2 D][q]
    float4 *s_d = s_pad + (ROWS + 2 * D) * q;        // [rows + 2 l2][q]
    float4 *s_dd = s_d + (ROWS + 2 * l2) * q;        // [rows][q]
    const uint32_t magic_q = 0xffffffffu / (uint32_t)q + 1; // floor(i / q) for i < 2^16
    {
        const int n_pad = (rows + 2 * D) * q;
        for (int i = tid; i < n_pad; i += 256) {
            const int rr = (int)__umulhi((uint32_t)i, magic_q), c = i - rr * q;
            int sr = r0 + rr + sg.shift;
            sr = max(sg.lo, min(sg.hi, sr));
            s_pad[i] = ((const float4 *)(p.src + (sg.src_row0 + sr) * (int64_t)p.src_pitch))[c];
        }
    }
    __syncthreads();
    {
        float den = 0.f;
        for (int l = 1; l <= l1; ++l) den += (float)(l * l);
        const float d1 = 2 * den, inv1 = 1.0f / d1;
        const int n_d = (rows + 2 * l2) * q;
        for (int i = tid; i < n_d; i += 256) {
            float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 1; l <= l1; ++l) {
                const float4 a = s_pad[i + (l1 + l) * q], b = s_pad[i + (l1 - l) * q];
                num.x = __builtin_fmaf((float)l, a.x - b.x, num.x);
                num.y = __builtin_fmaf((float)l, a.y - b.y, num.y);
                num.z = __builtin_fmaf((float)l, a.z - b.z, num.z);
                num.w = __builtin_fmaf((float)l, a.w - b.w, num.w);
            }
            s_d[i] = make_float4(delta_quot(num.x, d1, inv1), delta_quot(num.y, d1, inv1), delta_quot(num.z, d1, inv1),
                                 delta_quot(num.w, d1, inv1));
        }
    }
    __syncthreads();
    if (l2 > 0) {
        float den2 = 0.f;
        for (int l = 1; l <= l2; ++l) den2 += (float)(l * l);
        const float d2 = 2 * den2, inv2 = 1.0f / d2;
        const int n_dd = rows * q;
        for (int i = tid; i < n_dd; i += 256) {
            float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int l = 1; l <= l2; ++l) {
                const float4 a = s_d[i + (l2 + l) * q], b = s_d[i + (l2 - l) * q];
                num.x = __builtin_fmaf((float)l, a.x - b.x, num.x);
                num.y = __builtin_fmaf((float)l, a.y - b.y, num.y);
                num.z = __builtin_fmaf((float)l, a.z - b.z, num.z);
                num.w = __builtin_fmaf((float)l, a.w - b.w, num.w);
            }
            s_dd[i] = make_float4(delta_quot(num.x, d2, inv2), delta_quot(num.y, d2, inv2), delta_quot(num.z, d2, inv2),
                                  delta_quot(num.w, d2, inv2));
        }
        __syncthreads();
    }
    // the tile's output rows are one contiguous block: consecutive threads write consecutive 16-byte words, statics included
    const int wq = q * (l2 > 0 ? 3 : 2);
    const int stat_row = sg.static_off - sg.shift;
    float4 *obase = (float4 *)(p.out + (sg.out_row0 + r0) * (int64_t)p.out_pitch);
    const uint32_t magic_w = 0xffffffffu / (uint32_t)wq + 1;
    const int n_o = rows * wq;
    for (int i = tid; i < n_o; i += 256) {
        const int rr = (int)__umulhi((uint32_t)i, magic_w), cc = i - rr * wq;
        float4 v;
        if (cc < q)
            v = s_pad[(rr + stat_row) * q + cc];
        else if (cc < 2 * q)
            v = s_d[(rr + l2) * q + cc - q];
        else
            v = s_dd[rr * q + cc - 2 * q];
        obase[i] = v;
    }
}

// Delta stage from the compact statics (pitch 16) to whole output rows: the tile function of the fused
// delta wave run by a block.  grid = (tiles, segments) as k_delta; requires cols <= 16, l1 > 0,
// out_pitch == cols * (l2 > 0 ? 3 : 2), src_pitch == 16 and a 16-byte aligned `out`.
constexpr int kDelta16TilesPerBlock = 2;

template <int L1, int L2>
__global__ void __launch_bounds__(256, 7) k_delta16(DeltaParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.y];
    int r0 = blockIdx.x * (kDelta16TilesPerBlock * kDeltaRows);
    if (r0 >= sg.n_out) return;
    // a block walks 2 consecutive tiles (measured on C2: 1 -> 62 us, 2 -> 59 us, 4 -> 62 us, 8 -> 71 us per launch); the
    // statics of the second are in flight while the first is computed
    DeltaFill<256> fill;
    fill.issue(sg, r0, min(kDeltaRows, sg.n_out - r0), p.l1 + p.l2, p.src, threadIdx.x);
#pragma unroll 1
    for (int t = 0; t < kDelta16TilesPerBlock && r0 < sg.n_out; ++t, r0 += kDeltaRows) {
        const int rows = min(kDeltaRows, sg.n_out - r0);
        const int nr0 = r0 + kDeltaRows;
        const int nrows = (t + 1 < kDelta16TilesPerBlock && nr0 < sg.n_out) ? min(kDeltaRows, sg.n_out - nr0) : 0;
        delta_tile16<L1, L2, 256>(sg, r0, rows, p.src, p.out, p.out_pitch, p.cols, p.l1, p.l2, smem, threadIdx.x, fill, nr0,
                                  nrows);
    }
}

// ------------------------------------------------------------------------------------------------
// normalisation (normalizercpu.cpp:22-89): per segment, per column statistics in double.
// stats layout [seg][2][cols]: mean, multiplier.
//
// k_norm_stats: one block per (row chunk, segment).  The block reads its rows as they lie in memory: a thread is
// (row rr of the pass, column c), consecutive threads read consecutive floats of a row, a pass covers 256 / Cp
// whole rows (Cp = columns rounded up to a power of two).  Sums in double as the reference's (sum2 takes the
// float product v * v, normalizercpu.cpp:44); the partial results of the passes' rows are combined through LDS
// in a fixed order.  A segment longer than kNormChunkRows rows is cut into chunks whose partial results go to
// a scratch array and are combined, again in a fixed order, by k_norm_finalize -- results do not depend on timing.
// Statistics cover the first `stat_rows` rows of the segment (Segment::pad; 0 = all n_out rows): the reference
// computes them over the block it delivers and re-uses them for the flush rows (mfcccpu.cpp:377-388).
// ------------------------------------------------------------------------------------------------
constexpr int kNormChunkRows = 4096;
constexpr size_t kNormSegLdsBytes = 54 * 1024; // k_norm_seg: dynamic LDS per block (1024 rows of 13 columns: a 10 s utterance; two blocks per CU)

__device__ __forceinline__ void norm_finish_to(float *st, int cols, int norm_type, int c, int n, double S, double S2, float mn,
                                               float mx)
{
    const float mean = (float)(S / n);
    float mult = 1.f;
    if (norm_type == 2)
        mult = (float)sqrt((n - 1) / (S2 - S * (S / n)));
    else if (norm_type == 3)
        mult = 1.f / fmaxf(fabsf(mn - mean), fabsf(mx - mean));
    st[c] = mean;
    st[cols + c] = mult;
}

__device__ __forceinline__ void norm_finish(const NormParams &p, int seg, int c, int n, double S, double S2, float mn, float mx)
{
    norm_finish_to(p.stats + (int64_t)seg * 2 * p.cols, p.cols, p.norm_type, c, n, S, S2, mn, mx);
}

__global__ void __launch_bounds__(256) k_norm_stats(NormParams p)
{
    __shared__ double s_sum[256], s_sum2[256];
    __shared__ float s_min[256], s_max[256];
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.y];
    const int n = sg.pad > 0 ? sg.pad : sg.n_out;
    const int r0 = blockIdx.x * kNormChunkRows;
    if (r0 >= n && blockIdx.x > 0) return;
    const int r1 = min(n, r0 + kNormChunkRows);
    int lg = 0;
    while ((1 << lg) < p.cols) ++lg;
    const int Cp = 1 << lg, rpp = 256 >> lg;          // columns per row of threads, rows per pass (cols <= 256)
    const int tid = threadIdx.x, rr = tid >> lg, c = tid & (Cp - 1);
    const float *base = p.data + (sg.out_row0 + p.row_off) * (int64_t)p.pitch + p.col0;
    double sum = 0, sum2 = 0;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    if (c < p.cols)
        for (int r = r0 + rr; r < r1; r += rpp) {
            const float v = base[(int64_t)r * p.pitch + c];
            sum += v;
            sum2 += (double)(v * v);
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    s_sum[tid] = sum;
    s_sum2[tid] = sum2;
    s_min[tid] = mn;
    s_max[tid] = mx;
    __syncthreads();
    for (int s = rpp >> 1; s > 0; s >>= 1) {
        if (rr < s) {
            const int o = tid + (s << lg);
            s_sum[tid] += s_sum[o];
            s_sum2[tid] += s_sum2[o];
            s_min[tid] = fminf(s_min[tid], s_min[o]);
            s_max[tid] = fmaxf(s_max[tid], s_max[o]);
        }
        __syncthreads();
    }
    if (rr == 0 && c < p.cols) {
        if (p.chunks <= 1) {
            norm_finish(p, blockIdx.y, c, n, s_sum[tid], s_sum2[tid], s_min[tid], s_max[tid]);
        } else {
            double *q = p.partial + ((int64_t)blockIdx.y * p.chunks + blockIdx.x) * 4 * p.cols;
            q[c] = s_sum[tid];
            q[p.cols + c] = s_sum2[tid];
            q[2 * p.cols + c] = (double)s_min[tid];
            q[3 * p.cols + c] = (double)s_max[tid];
        }
    }
}

// chunks > 1: combine the chunk results of a segment in ascending chunk order
__global__ void __launch_bounds__(256) k_norm_finalize(NormParams p)
{
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.x];
    const int n = sg.pad > 0 ? sg.pad : sg.n_out;
    const int used = (n + kNormChunkRows - 1) / kNormChunkRows;
    for (int c = threadIdx.x; c < p.cols; c += 256) {
        double S = 0, S2 = 0;
        float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
        for (int k = 0; k < used; ++k) {
            const double *q = p.partial + ((int64_t)blockIdx.x * p.chunks + k) * 4 * p.cols;
            S += q[c];
            S2 += q[p.cols + c];
            mn = fminf(mn, (float)q[2 * p.cols + c]);
            mx = fmaxf(mx, (float)q[3 * p.cols + c]);
        }
        norm_finish(p, blockIdx.x, c, n, S, S2, mn, mx);
    }
}

// (x - mean) [* multiplier] in place over all n_out rows of the segment; grid.x is sized from the row count
__global__ void __launch_bounds__(256) k_norm_apply(NormParams p)
{
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.y];
    const int cols = p.cols;
    const int64_t total = (int64_t)sg.n_out * cols;
    const float *st = p.stats + (int64_t)blockIdx.y * 2 * cols;
    float *base = p.data + (sg.out_row0 + p.row_off) * (int64_t)p.pitch + p.col0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        float *q = base + r * p.pitch + c;
        const float v = *q;
        if (p.norm_type == 1)
            *q = v - st[c];
        else
            *q = (v - st[c]) * st[cols + c];
    }
}

// Statistics + apply of one SHORT segment in one block (an utterance of the batch entries, a small streaming block):
// the segment's rows are read once into LDS, the statistics are formed exactly as k_norm_stats forms them (same thread
// per (row class, column), same order of the double additions, same tree -- the results are the same bits), then every
// row is normalised from LDS and written back.  One read and one write of the data instead of two reads and one write,
// one launch instead of two: the reference's default configuration (CVN on 13 columns, ASR_OCL.cpp:560) spends
// 0.051 ms per 998 000 frames in the two-kernel form.  p.chunks = rows the block's LDS holds (a longer segment is
// processed from memory, correct but slow: the launcher does not choose this kernel for those).
#ifndef MFX_NORM_SEG_THREADS
#define MFX_NORM_SEG_THREADS 1024
#endif
constexpr int kNormSegThreads = MFX_NORM_SEG_THREADS; // the statistics keep k_norm_stats' 256-thread mapping; all threads move the rows
__global__ void __launch_bounds__(kNormSegThreads) k_norm_seg(NormParams p)
{
    extern __shared__ __attribute__((aligned(16))) float s_rows[];
    __shared__ double s_sum[256], s_sum2[256];
    __shared__ float s_min[256], s_max[256], s_st[512];
    const Segment sg = p.inline_seg ? p.seg0 : p.segs[blockIdx.x];
    const int cols = p.cols, n_out = sg.n_out;
    const int n = sg.pad > 0 ? sg.pad : sg.n_out;
    const int tid = threadIdx.x;
    // blockIdx.y = column group (static | delta | delta-delta blocks of the row, each with its own statistics)
    float *base = p.data + (sg.out_row0 + p.row_off) * (int64_t)p.pitch + p.col0 + blockIdx.y * cols;
    float *stats = p.stats + (int64_t)blockIdx.y * p.group_stats_stride + (int64_t)blockIdx.x * 2 * cols;
    const bool in_lds = n_out <= p.chunks;
    const int total = n_out * cols;
    // i / cols for i < 2^32 / cols (LDS-sized products).  cols == 1 would wrap the constant to 0: a one-column
    // configuration (ceps_len 1 without c0, one filter without a DCT) takes the shift form instead
    const uint32_t magic = cols > 1 ? 0xffffffffu / (uint32_t)cols + 1 : 0;
    const auto row_of = [&](int i) -> int { return cols > 1 ? (int)__umulhi((uint32_t)i, magic) : i; };
    if (in_lds) { // 16 reads in flight per thread (a plain loop waits for every read before the next)
        const bool contig = p.pitch == cols;
        for (int i0 = tid; i0 < total; i0 += kNormSegThreads * 16) {
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = min(i0 + kNormSegThreads * k, total - 1); // (clamped, not predicated: no branch, all reads issued at once)
                const int r = row_of(i), c = i - r * cols;
                v[k] = base[contig ? (int64_t)i : (int64_t)r * p.pitch + c];
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int i = i0 + kNormSegThreads * k;
                if (i < total) s_rows[i] = v[k];
            }
        }
        __syncthreads();
    }
    int lg = 0;
    while ((1 << lg) < cols) ++lg;
    const int Cp = 1 << lg, rpp = 256 >> lg;
    const int rr = tid >> lg, c = tid & (Cp - 1);
    double sum = 0, sum2 = 0;
    float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
    if (c < cols && tid < 256)
        for (int r = rr; r < n; r += rpp) {
            const float v = in_lds ? s_rows[r * cols + c] : base[(int64_t)r * p.pitch + c];
            sum += v;
            sum2 += (double)(v * v);
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
        }
    if (tid < 256) {
        s_sum[tid] = sum;
        s_sum2[tid] = sum2;
        s_min[tid] = mn;
        s_max[tid] = mx;
    }
    __syncthreads();
    for (int s = rpp >> 1; s > 0; s >>= 1) {
        if (rr < s) { // (rr < s <= rpp / 2: threads of the first 256 only)
            const int o = tid + (s << lg);
            s_sum[tid] += s_sum[o];
            s_sum2[tid] += s_sum2[o];
            s_min[tid] = fminf(s_min[tid], s_min[o]);
            s_max[tid] = fmaxf(s_max[tid], s_max[o]);
        }
        __syncthreads();
    }
    if (rr == 0 && c < cols) {
        norm_finish_to(stats, cols, p.norm_type, c, n, s_sum[tid], s_sum2[tid], s_min[tid], s_max[tid]);
        s_st[c] = stats[c]; // (this thread's own writes)
        s_st[256 + c] = stats[cols + c];
    }
    __syncthreads();
    if (in_lds) {
        for (int i = tid; i < total; i += kNormSegThreads) {
            const int r = row_of(i), cc = i - r * cols;
            const float v = s_rows[i];
            base[(int64_t)r * p.pitch + cc] = p.norm_type == 1 ? v - s_st[cc] : (v - s_st[cc]) * s_st[256 + cc];
        }
    } else {
        for (int64_t i = tid; i < (int64_t)n_out * cols; i += kNormSegThreads) {
            const int64_t r = i / cols;
            const int cc = (int)(i - r * cols);
            float *q = base + r * p.pitch + cc;
            const float v = *q;
            *q = p.norm_type == 1 ? v - s_st[cc] : (v - s_st[cc]) * s_st[256 + cc];
        }
    }
}

// Copy of a small block by a kernel instead of a DMA command (streaming interface, blocks under 1 MB: an SDMA copy of a few
// hundred KB costs more in command latency than in transfer time).  Either side may be page-locked host memory (mapped into
// the device's address space).  vec: dst and src are congruent modulo 16 -- 16-byte words between a head and a tail of
// 2-byte units; else 2-byte units throughout (byte counts are even: int16 samples or float rows).
__global__ void __launch_bounds__(256) k_copy_small(char *dst, const char *src, size_t bytes, int vec)
{
    const size_t gid = (size_t)blockIdx.x * 256 + threadIdx.x, stride = (size_t)gridDim.x * 256;
    if (vec) {
        size_t head = (16 - ((uintptr_t)dst & 15)) & 15;
        if (head > bytes) head = bytes;
        const size_t nvec = (bytes - head) >> 4, tail0 = head + (nvec << 4);
        const uint4 *s4 = (const uint4 *)(src + head);
        uint4 *d4 = (uint4 *)(dst + head);
        for (size_t v = gid; v < nvec; v += stride) d4[v] = s4[v];
        if (gid < (head >> 1)) ((short *)dst)[gid] = ((const short *)src)[gid];
        if (gid < ((bytes - tail0) >> 1)) ((short *)(dst + tail0))[gid] = ((const short *)(src + tail0))[gid];
    } else {
        for (size_t i = gid; i < (bytes >> 1); i += stride) ((short *)dst)[i] = ((const short *)src)[i];
    }
}

} // namespace

int num_cus()
{
    static thread_local int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cached[dev] = prop.multiProcessorCount;
        if (cached[dev] <= 0) cached[dev] = 256;
    }
    return cached[dev];
}

hipError_t allow_dynamic_lds(const void *func, size_t lds_bytes)
{
    if (lds_bytes <= 64 * 1024) return hipSuccess;
    struct Key {
        int dev;
        const void *func;
        size_t granted;
    };
    static thread_local Key cache[32];
    static thread_local int used = 0, next = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    Key *slot = nullptr;
    for (int i = 0; i < used; ++i)
        if (cache[i].dev == dev && cache[i].func == func) {
            if (cache[i].granted >= lds_bytes) return hipSuccess;
            slot = &cache[i];
            break;
        }
    const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if (!slot) {
        slot = &cache[next];
        next = (next + 1) % 32;
        if (used < 32) ++used;
    }
    *slot = Key{dev, func, lds_bytes};
    return hipSuccess;
}

int blocks_per_cu(const void *func, int threads, size_t lds_bytes, int fallback)
{
    struct Key {
        int dev, threads;
        const void *func;
        size_t lds;
        int value;
    };
    static thread_local Key cache[16];
    static thread_local int used = 0, next = 0;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    for (int i = 0; i < used; ++i)
        if (cache[i].dev == dev && cache[i].func == func && cache[i].threads == threads && cache[i].lds == lds_bytes) return cache[i].value;
    int v = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&v, func, threads, lds_bytes) != hipSuccess || v < 1) {
        (void)hipGetLastError();
        v = fallback;
    }
    cache[next] = Key{dev, threads, func, lds_bytes, v};
    next = (next + 1) % 16;
    if (used < 16) ++used;
    return v;
}

size_t melcep_lds_bytes(const MelcepParams &p, int n_waves)
{
    const size_t f = (size_t)mel64_rows(p.num_banks) * p.mel64_row_stride + (size_t)128 * p.mel64_rounds + 8 +
                     (size_t)n_waves * ((size_t)p.mag_floats + 4 * (size_t)lm_fs4(p.num_banks));
    return f * sizeof(float);
}

hipError_t launch_melcep(const MelcepParams &p, hipStream_t stream)
{
    if (p.n_rows <= 0) return hipSuccess;
    if (p.mag_floats < p.spec_pitch || (p.spec_pitch & 3) || (p.mag_floats & 3)) return hipErrorInvalidValue;
    int nw = 4; // waves per block: as many of 4 as the LDS holds
    while (nw > 1 && melcep_lds_bytes(p, nw) > 160 * 1024) nw >>= 1;
    const size_t lds = melcep_lds_bytes(p, nw);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (hipError_t e = allow_dynamic_lds((const void *)k_melcep, lds); e != hipSuccess) return e;
    int64_t blocks = ((p.n_rows + 3) / 4 + nw - 1) / nw;
    // persistent blocks: as many as are resident at once (registers allow 6 blocks of 4 waves; see launch_front_generic)
    int per_cu = blocks_per_cu((const void *)k_melcep, 64 * nw, lds, (int)std::min<size_t>(4, (160 * 1024) / lds)); // (per device and launch shape)
    if (per_cu > 8) per_cu = 8;
    const int cap = num_cus() * (per_cu < 1 ? 1 : per_cu);
    if (blocks > cap) blocks = cap;
    const int tables = p.n_tables > 1 ? p.n_tables : 1;
    if (tables > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_melcep, dim3((unsigned)blocks, (unsigned)tables), dim3(64 * nw), lds, stream, p);
    return hipGetLastError();
}

#ifndef MFX_DELTA_WIDE_ROWS_V
#define MFX_DELTA_WIDE_ROWS_V 32
#endif
hipError_t launch_delta(const DeltaParams &p, hipStream_t stream)
{
    if (p.n_segs <= 0 || p.tiles_per_seg_max <= 0) return hipSuccess;
    const int D = p.l1 + p.l2;
    const int groups = p.l2 > 0 ? 3 : 2;
    if (p.cols <= 16 && p.l1 > 0 && D <= 16 && p.src_pitch == 16 && p.out_pitch == p.cols * groups &&
        ((uintptr_t)p.out & 15) == 0 && ((uintptr_t)p.src & 15) == 0) {
        const size_t lds16 = (size_t)delta_wave_lds_floats(p.l1, p.l2) * sizeof(float);
        const bool u33 = p.l1 == 3 && p.l2 == 3;
        const int tiles_x = (p.tiles_per_seg_max + kDelta16TilesPerBlock - 1) / kDelta16TilesPerBlock;
        for (int s0 = 0; s0 < p.n_segs; s0 += 65535) {
            DeltaParams q = p;
            q.segs = p.segs + s0;
            q.n_segs = (p.n_segs - s0) < 65535 ? (p.n_segs - s0) : 65535;
            if (u33)
                hipLaunchKernelGGL((k_delta16<3, 3>), dim3(tiles_x, q.n_segs), dim3(256), lds16, stream, q);
            else
                hipLaunchKernelGGL((k_delta16<0, 0>), dim3(tiles_x, q.n_segs), dim3(256), lds16, stream, q);
        }
        return hipGetLastError();
    }
    // wide rows in whole 16-byte words: the vectorised form (C5: k_delta 0.052 ms -> see profiles/r03)
    if (p.cols > 16 && (p.cols & 3) == 0 && p.l1 > 0 && (p.src_pitch & 3) == 0 && p.out_pitch == p.cols * groups &&
        ((uintptr_t)p.out & 15) == 0 && ((uintptr_t)p.src & 15) == 0) {
        constexpr int R4 = MFX_DELTA_WIDE_ROWS_V;
        const size_t lds4 = (size_t)((R4 + 2 * D) + (R4 + 2 * p.l2) + R4) * p.cols * sizeof(float);
        if (lds4 <= 64 * 1024) {
            const int tiles_x4 = p.tiles_per_seg_max * (kDeltaRows / R4);
            for (int s0 = 0; s0 < p.n_segs; s0 += 65535) {
                DeltaParams qd = p;
                qd.segs = p.segs + s0;
                qd.n_segs = (p.n_segs - s0) < 65535 ? (p.n_segs - s0) : 65535;
                hipLaunchKernelGGL((k_delta4<R4>), dim3(tiles_x4, qd.n_segs), dim3(256), lds4, stream, qd);
            }
            return hipGetLastError();
        }
    }
    const bool fast16 = p.cols <= 16;
    const int cw = fast16 ? 16 : p.cols;
    // wide rows: tiles of MFX_DELTA_WIDE_ROWS output rows (less LDS per block: more blocks per CU in flight)
#ifndef MFX_DELTA_WIDE_ROWS
#define MFX_DELTA_WIDE_ROWS 32   // (C5: k_delta 0.072 -> 0.055 ms; 16 rows: 0.063)
#endif
    constexpr int RW = MFX_DELTA_WIDE_ROWS;
    const int rows_t = fast16 ? kDeltaRows : RW;
    const int tiles_x = p.tiles_per_seg_max * (kDeltaRows / rows_t);
    const size_t lds = (size_t)((rows_t + 2 * D) + (rows_t + 2 * p.l2) + rows_t) * cw * sizeof(float);
    if (hipError_t e = allow_dynamic_lds(fast16 ? (const void *)k_delta<true, kDeltaRows> : (const void *)k_delta<false, RW>, lds); e != hipSuccess)
        return e;
    // grid.y is limited to 65535: split the segment list
    for (int s0 = 0; s0 < p.n_segs; s0 += 65535) {
        DeltaParams q = p;
        q.segs = p.segs + s0;
        q.n_segs = (p.n_segs - s0) < 65535 ? (p.n_segs - s0) : 65535;
        if (fast16)
            hipLaunchKernelGGL((k_delta<true, kDeltaRows>), dim3(tiles_x, q.n_segs), dim3(256), lds, stream, q);
        else
            hipLaunchKernelGGL((k_delta<false, RW>), dim3(tiles_x, q.n_segs), dim3(256), lds, stream, q);
    }
    return hipGetLastError();
}

static int norm_chunks(int max_rows) { return max_rows <= kNormChunkRows ? 1 : (max_rows + kNormChunkRows - 1) / kNormChunkRows; }

hipError_t launch_norm_stats(const NormParams &p, hipStream_t stream)
{
    if (p.n_segs <= 0) return hipSuccess;
    if (p.cols > 256) return hipErrorInvalidValue;
    const int chunks = norm_chunks(p.max_rows);
    if (chunks > 1 && !p.partial) return hipErrorInvalidValue;
    for (int s0 = 0; s0 < p.n_segs; s0 += 65535) {
        NormParams q = p;
        q.segs = p.segs + s0;
        q.stats = p.stats + (int64_t)s0 * 2 * p.cols;
        q.n_segs = (p.n_segs - s0) < 65535 ? (p.n_segs - s0) : 65535;
        q.chunks = chunks;
        if (chunks > 1) q.partial = p.partial + (int64_t)s0 * chunks * 4 * p.cols;
        hipLaunchKernelGGL(k_norm_stats, dim3(chunks, q.n_segs), dim3(256), 0, stream, q);
        if (chunks > 1) hipLaunchKernelGGL(k_norm_finalize, dim3(q.n_segs), dim3(256), 0, stream, q);
    }
    return hipGetLastError();
}

// stats + apply in one launch when every segment's rows fit one block's LDS (see k_norm_seg)
bool norm_fused_fits(int max_rows, int cols)
{
    return max_rows > 0 && cols > 0 && cols <= 256 && (size_t)max_rows * cols * sizeof(float) <= kNormSegLdsBytes;
}

hipError_t launch_norm_fused(const NormParams &p, hipStream_t stream)
{
    if (p.n_segs <= 0) return hipSuccess;
    if (!norm_fused_fits(p.max_rows, p.cols)) return hipErrorInvalidValue;
    const size_t lds = (size_t)p.max_rows * p.cols * sizeof(float);
    for (int s0 = 0; s0 < p.n_segs; s0 += 65535) {
        NormParams q = p;
        q.segs = p.segs + s0;
        q.stats = p.stats + (int64_t)s0 * 2 * p.cols;
        q.n_segs = (p.n_segs - s0) < 65535 ? (p.n_segs - s0) : 65535;
        q.chunks = p.max_rows; // rows the block's LDS holds
        hipLaunchKernelGGL(k_norm_seg, dim3(q.n_segs, p.groups > 1 ? p.groups : 1), dim3(kNormSegThreads), lds, stream, q);
    }
    return hipGetLastError();
}

hipError_t launch_copy_small(void *dst, const void *src, size_t bytes, hipStream_t stream)
{
    if (bytes == 0) return hipSuccess;
    if (bytes & 1) return hipErrorInvalidValue;
    const int vec = (((uintptr_t)dst ^ (uintptr_t)src) & 15) == 0 ? 1 : 0;
    const size_t items = vec ? (bytes >> 4) + 16 : (bytes >> 1);
    size_t blocks = (items + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_copy_small, dim3((unsigned)blocks), dim3(256), 0, stream, (char *)dst, (const char *)src, bytes, vec);
    return hipGetLastError();
}

size_t norm_partial_doubles(int n_segs, int max_rows, int cols)
{
    const int ch = norm_chunks(max_rows);
    return ch <= 1 ? 0 : (size_t)n_segs * ch * 4 * cols;
}

hipError_t launch_norm_apply(const NormParams &p, hipStream_t stream)
{
    if (p.n_segs <= 0) return hipSuccess;
    // about 2048 elements per block, whatever the row count (a streaming block is one long segment)
    int64_t gx = ((int64_t)(p.max_rows > 0 ? p.max_rows : 1) * p.cols + 2047) / 2048;
    if (gx < 1) gx = 1;
    if (gx > 4096) gx = 4096;
    for (int s0 = 0; s0 < p.n_segs; s0 += 65535) {
        NormParams q = p;
        q.segs = p.segs + s0;
        q.stats = p.stats + (int64_t)s0 * 2 * p.cols;
        q.n_segs = (p.n_segs - s0) < 65535 ? (p.n_segs - s0) : 65535;
        hipLaunchKernelGGL(k_norm_apply, dim3((unsigned)gx, q.n_segs), dim3(256), 0, stream, q);
    }
    return hipGetLastError();
}

} // namespace mfx
