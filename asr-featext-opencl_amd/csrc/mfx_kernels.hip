// mfx_kernels.hip -- gfx950 (CDNA4, wave64) kernels of the MFCC front end.  See mfx_kernels.h
// for the inventory and DESIGN.md for the data layout and the roofline of each kernel.
//
// Numerics follow the reference CPU path (mfcccpu.cpp): frames = window * int16 sample (one
// rounding), unnormalised forward DFT, magnitude / W2, two-row triangular mel table walked in
// ascending bin order, log(max(., 1e-30)), DCT as a k-ordered dot product.
#include "mfx_kernels.h"

#include <hip/hip_runtime.h>

#include "mfx_dev.h"

#include <algorithm>
#include <cstdlib>

namespace mfx {

namespace {

// ------------------------------------------------------------------------------------------------
// 512-point front end.  One wave owns 4 frames per iteration, 16 lanes per frame; a 512-thread
// block is 16 such waves sharing one set of LDS tables and one work counter, one block per CU.
//
//   real 512-point DFT of a frame = complex 256-point DFT of z[n] = x[2n] + i x[2n+1] + real split
//   256 = 16 x 16:   lane l  : 16-point DFT over m of z[l + 16m]        (registers)
//                    twiddle : * W_256^(l*k1)                            (LDS table, [k1][l])
//                    LDS     : 16x16 transpose inside the frame's lane group
//                    lane q  : 16-point DFT over l -> Z[q + 16p], p = 0..15
//   split:  X[k] = 1/2 * ((Z[k] + conj Z[256-k]) + (-i W_512^k)(Z[k] - conj Z[256-k]))
//           the partner Z[256-k] lives in lane (16-q)%16 of the same 16-lane DPP row
//   |X[k]| / 512 -> LDS (or HBM when TO_SPEC), then mel/log/DCT on the 16 lanes of the frame.
//
// LDS per frame slot: 16 rows x 32 dwords, XOR-swizzled: (row r, column c) sits at column
// c ^ (r & 14).  The column writes (ds_write_b64, one row per instruction) stay 128 contiguous
// bytes; the row reads (ds_read_b128, lane q reads row q) then touch 16 distinct 16-byte bank
// groups per 16 lanes.  Magnitudes and the mel scratch reuse the slot once the transpose is done.
//
// PCM is fetched with buffer loads (hardware range check: reads past the end of the array return
// 0) one iteration ahead of its use.
// ------------------------------------------------------------------------------------------------
constexpr int kSlot = 512;    // dwords per frame slot
#ifndef MFX_WAVES512
#define MFX_WAVES512 16
#endif
constexpr int kWaves = MFX_WAVES512;    // waves per block of the 512-point kernel (16 waves per CU in all)
constexpr int kThreads = kWaves * 64;
constexpr int kMelOff = 304;  // mel scratch offset inside the slot (after 32 + 257 magnitudes)
constexpr int kTabStride = 36;   // dwords per lane row of the window / pass-twiddle tables in LDS (16 complex + pad)
constexpr int kSplitStride = 20; // dwords per lane row of the split-twiddle table (8 complex + pad)

// DCT on the matrix pipe (dct_mode 1): K steps of v_mfma_f32_16x16x4_f32 over the mel bands, 4 bands per step
constexpr int kDctSteps = 10;  // num_banks <= 40
constexpr int kDctRow = 12;    // dwords per lane row of the B operand table in LDS (16-byte words, disjoint bank quads)
// MFX_DCT_QUARTERS (default): the same DCT as 10 v_mfma_f32_4x4x1_16b_f32 -- 16 independent 4 x 4 outer products per
// instruction: block (kb = lane >> 4, cg = (lane >> 2) & 3) accumulates frames i = 0..3 x columns 4 cg + j over the bands
// 10 kb .. 10 kb + 9, one band per instruction; the four band quarters are then added across the 16-lane rows
// (v_permlane16_swap / v_permlane32_swap).  On gfx950 f32 matrix instructions run on the vector pipe's FP32 units
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0), so their cycles are the SIMD's cycles: 10 x 8 here against 10 x 32 for the
// 16 x 16 x 4 form, which used a quarter of its rows (4 frames).
#ifndef MFX_DCT_QUARTERS
#define MFX_DCT_QUARTERS 1
#endif
constexpr int kDctQ = 10;      // bands per quarter
constexpr int kDctQPad = 12;   // floats per quarter in the frame's log-energy row (16-byte aligned quarters)
__device__ __forceinline__ int dct_q_pos(int m) { return kDctQPad * (m / kDctQ) + (m % kDctQ); }

// Partner fetch of the real split: lanes l >= 1 get x[16 - l] (mirror, then shift right by one inside the row);
// lane 0, which the shift leaves without a source, keeps `own` -- its partner lives in its own registers.
__device__ __forceinline__ float row_partner_own0(float own, float x)
{
    int t = __builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true);   // row_mirror
    t = __builtin_amdgcn_update_dpp(__float_as_int(own), t, 0x111, 0xf, 0xf, false);    // row_shr:1, lane 0 keeps old
    return __int_as_float(t);
}

// x of lane 15 - l of the same 16-lane row
__device__ __forceinline__ float row_mirror(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x140, 0xf, 0xf, true));
}

template <bool ALIGNED, int NM>
struct PcmRegs {
    uint32_t d[ALIGNED ? NM : 2 * NM];
};

// issue the loads of one iteration (4 frames); `voff` = byte offset of this lane's first pair
template <bool ALIGNED, int NM>
__device__ __forceinline__ void pcm_issue(PcmRegs<ALIGNED, NM> &r, __amdgpu_buffer_rsrc_t rsrc, int voff)
{
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        if (ALIGNED) {
            r.d[m] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 64 * m, 0, 0);
        } else {
            r.d[2 * m] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 64 * m, 0, 0);
            r.d[2 * m + 1] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff + 64 * m + 4, 0, 0);
        }
    }
}

// The zero-stuffed form (256-point transforms on the 512-point core, k_front512<.., STUFF>): ONE sample per lane and row,
// 16 samples per row; `voff` = byte offset of this lane's first sample
template <int NM>
__device__ __forceinline__ void pcm_issue_stuffed(PcmRegs<true, NM> &r, __amdgpu_buffer_rsrc_t rsrc, int voff, int row_bytes)
{
#pragma unroll
    for (int m = 0; m < NM; ++m) r.d[m] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b16(rsrc, voff, row_bytes * m, 0);
}

// Dev-only in-kernel stamps (-DMFX_STAMPS): per-wave cycle sums per phase, written by lane 0 to
// p.spec (which is unused by the fused path).  Never part of a timed build.
#ifdef MFX_STAMPS
#define MFX_STAMP(i)                                                                   \
    do {                                                                               \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        st_acc[i] += t_ - st_last;                                                     \
        st_last = t_;                                                                  \
    } while (0)
#else
#define MFX_STAMP(i)
#endif

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

// STUFF: a 256-POINT transform on this core.  The 512-point real DFT of the frame with a zero after every sample,
// y[2n] = x[n], y[2n+1] = 0, is X_256[k mod 256]: the packed sequence is z[n] = x[n] + 0i, one sample per lane and row (16-bit
// loads, 16 samples per row, any alignment), and bins 0 .. 128 of the result are the 256-point spectrum -- the same arithmetic
// at the cost of a 512-point frame, where the one-wave-per-frame kernel took 2.6 x as long (8 kHz telephony, 200-tap windows).
// CH2: interleaved stereo input (one 32-bit word per sample, L | R << 16; mono = (L + R) >> 1 as everywhere, stereo_mean): a
// pair is one 8-byte load at any sample offset (STUFF: one word per lane and row)
template <bool ALIGNED, bool TO_SPEC, int NM, bool FUSE, bool STUFF = false, bool CH2 = false>
__global__ void __launch_bounds__(kThreads, 4) k_front512(FrontParams p) // (4 waves per SIMD whatever the block size)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int stuff_sh = STUFF ? (p.stuff >= 8 ? 2 : p.stuff >= 4 ? 1 : 0) : 0; // log2(lanes per sample) of the zero-stuffed forms
    const int slot = lane >> 4, l = lane & 15, odd_slot = slot & 1;

    // ---- LDS carve: shared tables, then one 4-slot region per wave
    const int cols = p.cols;
    const int rounds = p.mel_rounds, RS = p.mel_row_stride, DS = p.dct_stride, nb_pad = p.nb_pad;
    // Lane-major tables: lane l reads ITS window pairs / twiddles as 16-byte words (two complex values per
    // ds_read_b128, half the LDS instructions of the 8-byte form and whole batches in flight); the row strides
    // (36 and 20 dwords) put the 16 lanes of a b128 access group on disjoint bank quads.  The 4 frames of a
    // wave read the same words (broadcast).
    float *s_win = smem;                             // [16 l][kTabStride]: (w[2n], w[2n+1]), n = l + 16 m, m = 0..15
    float *s_tw = s_win + 16 * kTabStride;           // [16 l][kTabStride]: W_256^(l k), k = 0..15
    float *s_split = s_tw + 16 * kTabStride;         // [16 l][kSplitStride]: -i W_512^(l + 16 p), p = 0..7
    float *s_melw = s_split + 16 * kSplitStride;     // [16][RS]
    int *s_mmeta = (int *)(s_melw + 16 * RS);        // [rounds][16] (first bin, filter id): one 8-byte read per round
    float *s_dct = (float *)(s_mmeta + 32 * rounds); // [cols][DS]
    // dct_mode 0: transposed matrix [cols][DS]; dct_mode 1: matrix-pipe B operands per lane, [64][kDctRow]
    const int dct_floats = !p.dct ? 0 : p.dct_mode == 1 ? 64 * kDctRow : cols * DS;
    float *s_wave = s_dct + dct_floats + wave * (4 * kSlot);
    float *xb = s_wave + slot * kSlot;
    // FUSE: the last wave runs the delta stage; its region starts at its (unused) frame slots
    float *s_delta = s_dct + dct_floats + (kWaves - 1) * (4 * kSlot);
    const int delta_floats = FUSE ? delta_wave_lds_floats(p.dl1, p.dl2) : 0;
    const int tail_floats = delta_floats > 4 * kSlot ? delta_floats : 4 * kSlot;
    int *s_ctr = (int *)(s_dct + dct_floats + (kWaves - 1) * (4 * kSlot) + tail_floats); // block-local work counter
    unsigned *s_done = (unsigned *)(s_ctr + 4);   // FUSE: bit k = chunk k of this block's list has its statics in memory
    if (tid == 0) *s_ctr = 0;
    if (FUSE)
        for (int i = tid; i < p.done_words; i += kThreads) s_done[i] = 0u;

    for (int i = tid; i < 256; i += kThreads) { // HBM tables are [lane][m] as well
        ((float2 *)(s_win + (i >> 4) * kTabStride))[i & 15] = ((const float2 *)p.winpair)[i];
        ((float2 *)(s_tw + (i >> 4) * kTabStride))[i & 15] = ((const float2 *)p.twid_pass)[i];
    }
    for (int i = tid; i < 128; i += kThreads) // bins 0..127, natural order in HBM
        ((float2 *)(s_split + (i & 15) * kSplitStride))[i >> 4] = ((const float2 *)p.twid_split)[i];
    if (!TO_SPEC) {
        for (int i = tid; i < 16 * RS; i += kThreads) s_melw[i] = p.mel_lane_w[i];
        for (int i = tid; i < 16 * rounds; i += kThreads) {
            s_mmeta[2 * i] = p.mel_lane_start[i];
            const int fid = p.mel_lane_fid[i];
            // (matrix-pipe form: idle lanes park their value in a word nobody reads)
            if (MFX_DCT_QUARTERS)
                s_mmeta[2 * i + 1] = p.dct_mode != 1 ? fid : fid < 0 ? 4 * kDctQPad : dct_q_pos(fid);
            else
                s_mmeta[2 * i + 1] = (fid < 0 && p.dct_mode == 1) ? 4 * kDctSteps : fid;
        }
        if (p.dct_mode == 1) {
            // B operand of K step j on lane (k = lane >> 4, n = lane & 15) is dct[4 j + k][n]; zeros beyond the matrix
            for (int i = tid; i < 64 * kDctRow; i += kThreads) {
                // (MFX_DCT_QUARTERS: B operand of band 10 kb + j on lane (kb = lane >> 4, n = lane & 15))
                const int ln = i / kDctRow, j = i - ln * kDctRow, n = ln & 15;
                const int m = MFX_DCT_QUARTERS ? kDctQ * (ln >> 4) + j : 4 * j + (ln >> 4);
                s_dct[i] = (j < kDctSteps && m < p.num_banks && n < p.dct_len) ? p.dct[m * p.dct_len + n] : 0.f;
            }
        } else {
            for (int i = tid; i < dct_floats; i += kThreads) s_dct[i] = p.dct_t[i];
        }
    }
    // the slots are read (times zero weights) before every word has been written once: make them finite
    for (int i = lane; i < 4 * kSlot; i += 64) s_wave[i] = 0.f;
    __syncthreads();

    // ---- FUSE: the delta wave.  It consumes the block's tiles in order; a tile is ready once the
    // chunks it reads (its own rows and up to D rows either side) have their bits set in s_done.  The
    // front-end waves only ever produce, so the wait cannot deadlock; it is bounded all the same.
    const int chunk_base = FUSE ? p.blk_chunk_off[blockIdx.x] : 0;
    const int chunk_cnt = FUSE ? p.blk_chunk_off[blockIdx.x + 1] - chunk_base : 0;
    if (FUSE && wave == kWaves - 1) {
        __builtin_amdgcn_s_setprio(3); // little work, but everything it does is on the block's critical path
        const int t_end = p.blk_tile_off[blockIdx.x + 1];
        int t = p.blk_tile_off[blockIdx.x];
        if (t >= t_end) return;
        DeltaTile T = p.tiles[t];
#ifdef MFX_DSTAMPS
        unsigned long long ds_acc[4] = {0, 0, 0, 0}, ds_last, ds_t, ds_ph[4] = {0, 0, 0, 0};
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ds_last)::"memory");
        const unsigned long long ds_begin = ds_last;
#define DSTAMP(i)                                                                         \
    do {                                                                                  \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ds_t)::"memory");  \
        ds_acc[i] += ds_t - ds_last;                                                      \
        ds_last = ds_t;                                                                   \
    } while (0)
#else
#define DSTAMP(i)
#endif
        for (; t < t_end; ++t) {
            // the next descriptor is fetched while this tile is worked on (one past the end is a valid
            // address: the host pads the tile array by one entry)
            const DeltaTile nxt = p.tiles[t + 1];
            const int lo = __builtin_amdgcn_readfirstlane(T.dep_lo), hi = __builtin_amdgcn_readfirstlane(T.dep_hi);
            bool ok = false;
            for (int spins = 0; spins < (1 << 22); ++spins) {
                ok = true;
                for (int w = lo >> 5; w <= (hi >> 5); ++w) {
                    const int b0 = max(lo - 32 * w, 0), b1 = min(hi - 32 * w, 31);
                    const unsigned mask = (b1 == 31 ? 0xffffffffu : ((1u << (b1 + 1)) - 1u)) & ~((1u << b0) - 1u);
                    const unsigned v = __hip_atomic_load(&s_done[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if ((v & mask) != mask) ok = false;
                }
                if (ok) break;
                __builtin_amdgcn_s_sleep(8);
            }
            if (!ok) { // never expected: report instead of hanging
                if (lane == 0 && p.err_flag) atomicExch(p.err_flag, 1);
                return;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            DSTAMP(0);
            Segment sg;
            sg.src_row0 = sg.out_row0 = T.seg_row0;
            sg.n_out = 0;
            sg.shift = T.shift;
            sg.lo = T.lo;
            sg.hi = T.hi;
            sg.static_off = T.static_off;
            sg.pad = 0;
#ifdef MFX_DSTAMPS
            unsigned long long *php = ds_ph;
#else
            unsigned long long *php = nullptr;
#endif
            DeltaFill<64> fill;
            fill.issue(sg, T.r0, T.n_rows, p.dl1 + p.dl2, p.feat, lane);
            if (p.dl1 == 3 && p.dl2 == 3) // the reference driver's orders (ASR_OCL.cpp:560): reads unrolled
                delta_tile16<3, 3, 64>(sg, T.r0, T.n_rows, p.feat, p.out, p.out_pitch, cols, 3, 3, s_delta, lane, fill, 0, 0, php);
            else
                delta_tile16<0, 0, 64>(sg, T.r0, T.n_rows, p.feat, p.out, p.out_pitch, cols, p.dl1, p.dl2, s_delta, lane, fill,
                                       0, 0, php);
            DSTAMP(1);
            T = nxt;
        }
#ifdef MFX_DSTAMPS
        if (lane == 0 && p.spec) { // 100 MHz ticks: waiting, working, whole life, tiles
            unsigned long long *o = (unsigned long long *)p.spec + (size_t)blockIdx.x * 8;
            o[0] = ds_acc[0];
            o[1] = ds_acc[1];
            o[2] = ds_last - ds_begin;
            o[3] = (unsigned long long)(t_end - p.blk_tile_off[blockIdx.x]);
            for (int i = 0; i < 4; ++i) o[4 + i] = ds_ph[i];
        }
#endif
        return;
    }

#ifdef MFX_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    // Work distribution.  Chunks (<= 16 consecutive frames of one utterance) are dealt round-robin to
    // the waves of the grid (wave w takes chunks w, w + W, w + 2W, ...).  A shared atomic counter was
    // tried and lost: one word serves ~88 fetch-adds per microsecond, which bounds the kernel once
    // chunks are small enough to even out the tail.  The chunk walk is software pipelined so that a
    // chunk boundary costs no memory latency: the next chunk's descriptor is already in scalar
    // registers and the last iteration of a chunk prefetches the first frames of the next one.
    struct ChunkCtx {
        int64_t out_row;
        int n_live, odd0;
        __amdgpu_buffer_rsrc_t rsrc;
    };
    auto make_ctx = [&](int c) -> ChunkCtx {
        ChunkCtx x;
        const bool valid = c < p.n_chunks;
        const Chunk *chp = p.chunks + (valid ? c : 0);
        const int64_t pcm_off = chp->pcm_off;
        x.out_row = chp->out_row;
        const int n_frames = valid ? chp->n_frames : 0;
        const int64_t rows_left = p.row_limit - x.out_row;
        x.n_live = (int)(rows_left < n_frames ? (rows_left < 0 ? 0 : rows_left) : n_frames);
        // buffer descriptor over [chunk start, end of PCM): out-of-range lanes read 0
        // (CH2: p.pcm_total counts int16 elements, two per sample; every sample is a whole, aligned word)
        const int64_t base_s = CH2 ? pcm_off * 2 : (ALIGNED && !STUFF) ? pcm_off : (pcm_off & ~(int64_t)1);
        x.odd0 = (CH2 || (ALIGNED && !STUFF)) ? 0 : (int)(pcm_off & 1);
        // (rounded up to whole 32-bit words: with an odd sample count the array's last sample sits in a word whose
        // upper half lies past the end, and the range check would drop the whole word -- the base is 4-byte aligned,
        // so that word is inside the allocation, and the half past the end only ever meets a zero window tap)
        int64_t bytes_left = valid ? (((p.pcm_total - base_s) * 2 + 3) & ~(int64_t)3) : 0;
        if (bytes_left > 0xfffffff0ll) bytes_left = 0xfffffff0ll;
        if (bytes_left < 0) bytes_left = 0;
        const uintptr_t bp = (uintptr_t)(p.pcm + base_s);
        const uint32_t bp_lo = __builtin_amdgcn_readfirstlane((uint32_t)bp);
        const uint32_t bp_hi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
        const uint32_t nbytes = __builtin_amdgcn_readfirstlane((uint32_t)bytes_left);
        x.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)bp_hi << 32) | bp_lo), 0, nbytes, 0x00020000);
        return x;
    };
    // byte offset of (frame f, sample pair l) relative to the chunk's descriptor base
    auto lane_off = [&](const ChunkCtx &x, int f) -> int {
        // (STUFF: p.stuff = 512 / W2 = 2, 4 or 8 -- a zero after every sample once, twice or three times over: every
        // (p.stuff / 2)-th lane carries a sample, 16 / 8 / 4 samples per row; the other lanes meet zero window taps)
        if (CH2) return (f * p.shift + (STUFF ? (l >> stuff_sh) : 2 * l)) * 4;
        if (STUFF) return (x.odd0 + f * p.shift + (l >> stuff_sh)) * 2;
        const int s = x.odd0 + f * p.shift + 2 * l;
        return ALIGNED ? s * 2 : (s & ~1) * 2;
    };
    // Block b owns chunks b, b + B, b + 2B, ...; its waves draw from that list through a counter in
    // LDS (a ds_add_rtn costs ~100 cycles and contends with the block's other waves only), so waves that the
    // SIMD arbiter favours simply take more chunks instead of finishing early and idling the CU.
    auto next_index = [&]() -> int {
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return k;
    };
    auto chunk_of = [&](int k) -> int {
        if (FUSE) return k < chunk_cnt ? chunk_base + k : p.n_chunks; // the block's own contiguous list, in order
        const long long c = (long long)blockIdx.x + (long long)k * gridDim.x;
        return c < p.n_chunks ? (int)c : p.n_chunks;
    };
    int v_a = next_index(), v_b = next_index(), v_nn = next_index();
    int c_cur = chunk_of(__builtin_amdgcn_readfirstlane(v_a));
    int c_nxt = chunk_of(__builtin_amdgcn_readfirstlane(v_b));
    ChunkCtx ccur = make_ctx(c_cur);
    ChunkCtx cnxt = make_ctx(c_nxt);
    constexpr bool kOneWord = STUFF || (ALIGNED && !CH2); // registers per row: one word, or two (unaligned mono, stereo pairs)
    PcmRegs<kOneWord, NM> cur;
    auto issue = [&](__amdgpu_buffer_rsrc_t rsrc, int voff) {
        if constexpr (CH2 && STUFF) {
#pragma unroll
            for (int m = 0; m < NM; ++m) cur.d[m] = __builtin_amdgcn_raw_buffer_load_b32(rsrc, voff, (64 >> stuff_sh) * m, 0);
        } else if constexpr (CH2) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rsrc, voff, 128 * m, 0);
                cur.d[2 * m] = d[0];
                cur.d[2 * m + 1] = d[1];
            }
        } else if constexpr (STUFF) {
            pcm_issue_stuffed<NM>(cur, rsrc, voff, 32 >> stuff_sh);
        } else {
            pcm_issue<ALIGNED, NM>(cur, rsrc, voff);
        }
    };
    issue(ccur.rsrc, lane_off(ccur, slot));

    while (c_cur < p.n_chunks) {
        const int64_t out_row = ccur.out_row;
        const int n_live = ccur.n_live;
        const int odd0 = ccur.odd0;
        for (int f0 = 0; f0 < n_live; f0 += 4) {
            const int f = f0 + slot;
            const bool live = f < n_live;
            const bool last = f0 + 4 >= n_live;

            MFX_STAMP(0);
            // ---- framing + window: z[l + 16m] = (w[2n] x[2n], w[2n+1] x[2n+1])
            float2 a[16];
            const bool odd = !ALIGNED && !STUFF && !CH2 && ((odd0 + f * p.shift) & 1);
            float4 wq[(NM + 1) / 2];
#pragma unroll
            for (int m = 0; m < (NM + 1) / 2; ++m) wq[m] = ((const float4 *)(s_win + l * kTabStride))[m];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                if (m < NM) {
                    float x0, x1 = 0.f;
                    if constexpr (CH2 && STUFF) {
                        x0 = stereo_mean(cur.d[m]);
                    } else if constexpr (CH2) {
                        x0 = stereo_mean(cur.d[2 * m]);
                        x1 = stereo_mean(cur.d[2 * m + 1]);
                    } else {
                        uint32_t d;
                        if constexpr (ALIGNED || STUFF) {
                            d = cur.d[m];
                        } else {
                            const uint32_t d0 = cur.d[2 * m], d1 = cur.d[2 * m + 1];
                            d = odd ? ((d0 >> 16) | (d1 << 16)) : d0;
                        }
                        x0 = (float)(int)(short)(d & 0xffffu);
                        x1 = (float)((int)d >> 16);
                    }
                    const float2 w = (m & 1) ? make_float2(wq[m >> 1].z, wq[m >> 1].w) : make_float2(wq[m >> 1].x, wq[m >> 1].y);
                    a[m] = STUFF ? make_float2(w.x * x0, 0.f) : make_float2(w.x * x0, w.y * x1);
                } else {
                    a[m] = make_float2(0.f, 0.f);
                }
            }

            // Prefetch into the registers just consumed, unconditionally (a conditional issue would make
            // the number of loads in flight path-dependent and force a vmcnt(0) wait): the next 4 frames
            // of this chunk or, from the chunk's last iteration, the first 4 frames of the next chunk.
            issue(last ? cnxt.rsrc : ccur.rsrc, last ? lane_off(cnxt, slot) : lane_off(ccur, f + 4));

            MFX_STAMP(1);
            // ---- pass A + inter-pass twiddle
            fft16(a);
            {
                float4 tq[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) tq[k] = lds_read_b128((const float4 *)(s_tw + l * kTabStride) + k);
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const float4 t = tq[k >> 1];
                    if (k > 0) a[k] = cmul(a[k], make_float2(t.x, t.y));
                    a[k + 1] = cmul(a[k + 1], make_float2(t.z, t.w));
                }
            }

            MFX_STAMP(2);
            // ---- 16x16 transpose through the frame slot (XOR swizzle, see above).  Odd slots swap neighbouring
            // rows: the two slots of a 32-lane access group then write to complementary halves of the banks.
#pragma unroll
            for (int k = 0; k < 16; ++k) ((float2 *)(xb + (k ^ odd_slot) * 32))[l ^ (k & 14)] = a[k];
            wave_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 v = ((const float4 *)(xb + (l ^ odd_slot) * 32))[j ^ (l >> 1)];
                a[2 * j] = make_float2(v.x, v.y);
                a[2 * j + 1] = make_float2(v.z, v.w);
            }
            wave_sync();

            MFX_STAMP(3);
            // ---- pass B: a[pp] = Z[l + 16 pp]
#if !defined(MFX_ABLATE) || MFX_ABLATE < 3
            fft16(a);
#endif

            MFX_STAMP(4);
            // ---- real split + magnitude, one partner fetch per bin PAIR (k, 256 - k), k = l + 16 p, p < 8:
            //   S = Z[k] + conj Z[256-k], T = (-i W_512^k)(Z[k] - conj Z[256-k]):  X[k] = S + T,  X[256-k] = conj(S - T)
            // (the twiddle of bin 256 - k is the conjugate of bin k's).  The partner Z[256 - k] is register 15 - p of
            // lane (16 - l) % 16; lane 0 pairs with itself one register further (bin 16 p <-> bin 16 (16 - p)): the
            // second DPP move of the exchange (row_shr:1) has no source for lane 0 and leaves it that register.  Lane l ends with its own bins p < 8 and the bins of lane
            // (16 - l) % 16 for p >= 8 -- both go straight to their places (LDS or HBM), no second exchange.
            // Lane 0, p = 0 pairs bin 0 with the Nyquist bin 256 = Z[0] again: X[256] = Re Z[0] - Im Z[0] falls out
            // of the same formula; its self-paired bin 128 = conj-scaled Z[128] is done on the side.
            float mag_k[8], mag_p[8];
#if defined(MFX_ABLATE) && MFX_ABLATE >= 2
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                mag_k[pp] = a[pp].x + a[pp].y;
                mag_p[pp] = a[15 - pp].x + a[15 - pp].y;
            }
            const float mag128 = 0.f;
#else
            const float m128r = a[8].x + a[8].x, m128i = a[8].y + a[8].y;
            const float mag128 = __builtin_amdgcn_sqrtf(m128r * m128r + m128i * m128i);
            float4 csq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) csq[j] = ((const float4 *)(s_split + l * kSplitStride))[j];
#pragma unroll
            for (int pp = 0; pp < 8; ++pp) {
                const float zr = row_partner_own0(a[(16 - pp) & 15].x, a[15 - pp].x);
                const float zi = row_partner_own0(a[(16 - pp) & 15].y, a[15 - pp].y);
                const float2 cs = (pp & 1) ? make_float2(csq[pp >> 1].z, csq[pp >> 1].w) : make_float2(csq[pp >> 1].x, csq[pp >> 1].y);
                const float sr = a[pp].x + zr, si = a[pp].y - zi;
                const float dr = a[pp].x - zr, di = a[pp].y + zi;
                const float tr = cs.x * dr - cs.y * di;
                const float ti = cs.x * di + cs.y * dr;
                const float xr = sr + tr, xi = si + ti;
                const float yr = sr - tr, yi = si - ti;
                mag_k[pp] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi); // the window taps carry 0.5 / W2
                mag_p[pp] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
            }
#endif

            MFX_STAMP(5);
            if (TO_SPEC) {
                if (live) {
                    float *dst = p.spec + (out_row + f) * (int64_t)p.spec_pitch;
#pragma unroll
                    for (int pp = 0; pp < 8; ++pp) {
                        // (STUFF: a row holds bins 0 .. W2 / 2 of the short transform: 128, 64 or 32)
                        if (!STUFF || l + 16 * pp <= (128 >> stuff_sh)) dst[l + 16 * pp] = mag_k[pp];
                        if (!STUFF) dst[256 - l - 16 * pp] = mag_p[pp];
                    }
                    if (l == 0 && (!STUFF || stuff_sh == 0)) dst[128] = mag128;
                }
            } else {
#if defined(MFX_ABLATE) && MFX_ABLATE >= 1
                {   // dev-only: stop after the magnitudes, keep them live
                    float acc = mag128;
#pragma unroll
                    for (int pp = 0; pp < 8; ++pp) acc += mag_k[pp] + mag_p[pp];
                    float *dstx = p.feat + (out_row + (live ? f : 0)) * (int64_t)p.feat_pitch;
                    if (live && l < cols) dstx[l] = acc;
                    continue;
                }
#endif
                // odd slots keep their magnitudes 32 dwords further in: the two slots of a 32-lane
                // LDS access group then sit on complementary bank pairs for the b64 mel reads
                float *mg0 = xb + 32 * (slot & 1);
                {
                    float *mlo = mg0 + l, *mhi = mg0 + (144 - l); // bins l + 16 p and 256 - l - 16 p = (144 - l) + 16 (7 - p)
#pragma unroll
                    for (int pp = 0; pp < 8; ++pp) mlo[16 * pp] = mag_k[pp];
#pragma unroll
                    for (int pp = 0; pp < 8; ++pp) mhi[16 * pp] = mag_p[7 - pp];
                    if (l == 0) mg0[128] = mag128;
                }
                wave_sync();

                // ---- mel filterbank: per round every lane walks one filter's bins in ascending
                // order (mfcccpu.cpp:192-220).  Rounds are padded to a common even length with zero
                // weights; starts are even so that two bins come per ds_read_b64, and the host picks
                // them so that the 16 lanes of a slot (and the neighbouring slot, skewed by 32 dwords)
                // fall on distinct bank pairs.
                const float *wrow = s_melw + l * RS;
                // (uniform base of the iteration's first row + this lane's row inside the iteration: the compiler keeps the
                // 64-bit part in scalar registers; rows past the chunk are never stored, so their address needs no clamp)
                float *dst = p.feat + (out_row + f0) * (int64_t)p.feat_pitch + slot * p.feat_pitch;
                if (p.dct_mode == 1) {
                    // log mel energies to the frame's LDS row (8 dwords of skew per slot: the operand reads below
                    // then fall on distinct banks), then the DCT-II + lifter (mfcccpu.cpp:222-232) as ONE chain of
                    // f32 matrix instructions per 4 frames on the otherwise idle matrix pipe:
                    //   D[row][c] = sum_m A[row][m] B[m][c],  v_mfma_f32_16x16x4_f32, K step j covers m = 4j .. 4j+3,
                    // bit for bit an fmaf chain in ascending m.  Frame `slot` sits in rows 4 slot .. 4 slot + 3 (four
                    // copies: all 64 lanes read valid energies), so register 0 of the result is out[slot][c] on
                    // lane (slot, c) -- exactly the lane that stores it.
                    float *lm = xb + kMelOff + 8 * slot;
                    const int2 *mmeta = (const int2 *)s_mmeta + l;
                    for (int r = 0; r < rounds; ++r) {
                        const int L = p.mel_L[r];
                        const int2 mt = *mmeta;
                        mmeta += 16;
                        const float *mg = mg0 + mt.x;
                        const int fid = mt.y;
                        float acc = 0.f;
                        for (int s = 0; s < L; s += 8) {
                            float4 w[2];
                            float2 mm[4];
#pragma unroll
                            for (int q = 0; q < 2; ++q) w[q] = *(const float4 *)(wrow + s + 4 * q);
#pragma unroll
                            for (int q = 0; q < 4; ++q) mm[q] = lds_read_b64((const float2 *)(mg + s + 2 * q));
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                acc += w[q].x * mm[2 * q].x;
                                acc += w[q].y * mm[2 * q].y;
                                acc += w[q].z * mm[2 * q + 1].x;
                                acc += w[q].w * mm[2 * q + 1].y;
                            }
                        }
                        wrow += L;
                        lm[fid] = MFX_LOG(fmaxf(acc, 1e-30f)); // (idle lanes: fid names a word nobody reads)
                    }
                    wave_sync();
#if MFX_DCT_QUARTERS
                    // A operand of lane (kb = lane >> 4, i = lane & 3): frame i's log energies of bands 10 kb + t (the row of
                    // slot i, quarter kb: 12 floats, 16-byte aligned); B operand: this lane's 10 coefficients.  Two accumulator
                    // chains (even / odd bands of the quarter), each ascending in m.
                    const float4 *aq = (const float4 *)(s_wave + (lane & 3) * (kSlot + 8) + kMelOff + kDctQPad * slot);
                    const float4 *bq = (const float4 *)(s_dct + lane * kDctRow);
                    const float4 a0 = aq[0], a1 = aq[1], a2 = aq[2];
                    const float4 dq0 = bq[0], dq1 = bq[1], dq2 = bq[2];
                    f32x4 dacc = {0.f, 0.f, 0.f, 0.f}, dacc2 = {0.f, 0.f, 0.f, 0.f};
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.x, dq0.x, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.y, dq0.y, dacc2, 0, 0, 0);
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.z, dq0.z, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a0.w, dq0.w, dacc2, 0, 0, 0);
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.x, dq1.x, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.y, dq1.y, dacc2, 0, 0, 0);
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.z, dq1.z, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a1.w, dq1.w, dacc2, 0, 0, 0);
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.x, dq2.x, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a2.y, dq2.y, dacc2, 0, 0, 0);
                    // register i of lane (kb, c) now holds frame i's partial sum of column c over quarter kb; the lane that
                    // stores out[slot][l] is (row slot, column l): add the quarters across the four 16-lane rows while moving
                    // frame i's sums to row i (two butterfly steps: rows 1 <-> 0 / 3 <-> 2, then the two halves of the wave)
                    const auto r01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dacc[0] + dacc2[0]), __float_as_uint(dacc[1] + dacc2[1]), false, false);
                    const auto r23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dacc[2] + dacc2[2]), __float_as_uint(dacc[3] + dacc2[3]), false, false);
                    const float s01 = __uint_as_float(r01[0]) + __uint_as_float(r01[1]);
                    const float s23 = __uint_as_float(r23[0]) + __uint_as_float(r23[1]);
                    const auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
                    const float outv = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
#else
                    const float *arow = s_wave + (l >> 2) * (kSlot + 8) + kMelOff + slot; // A[row l][k = slot] of K step 0
                    const float4 *bq = (const float4 *)(s_dct + lane * kDctRow);
                    const float4 dq0 = bq[0], dq1 = bq[1], dq2 = bq[2];
                    const float dctb[12] = {dq0.x, dq0.y, dq0.z, dq0.w, dq1.x, dq1.y, dq1.z, dq1.w, dq2.x, dq2.y, dq2.z, dq2.w};
                    // two accumulator chains (even and odd K steps): the 40-cycle dependent latency of the instruction is
                    // covered by the other chain; each chain is an fmaf chain in ascending m, the two are added at the end
                    f32x4 dacc = {0.f, 0.f, 0.f, 0.f}, dacc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < kDctSteps; j += 2) {
                        dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j], dctb[j], dacc, 0, 0, 0);
                        dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + 4], dctb[j + 1], dacc2, 0, 0, 0);
                    }
                    const float outv = dacc[0] + dacc2[0];
#endif
                    // pitch 16 = compact static scratch: write whole 64-byte rows (zeros beyond cols)
                    if (live && (l < cols || p.feat_pitch == 16)) dst[l] = outv;
                } else {
                    float *melbuf = xb + kMelOff;
                    const int2 *mmeta = (const int2 *)s_mmeta + l;
                    for (int r = 0; r < rounds; ++r) {
                        const int L = p.mel_L[r];
                        const int2 mt = *mmeta;
                        mmeta += 16;
                        const float *mg = mg0 + mt.x;
                        const int fid = mt.y;
                        float acc = 0.f;
                        for (int s = 0; s < L; s += 4) {
                            const float4 w = *(const float4 *)(wrow + s);
                            const float2 m0 = *(const float2 *)(mg + s);
                            const float2 m1 = *(const float2 *)(mg + s + 2);
                            acc += w.x * m0.x;
                            acc += w.y * m0.y;
                            acc += w.z * m1.x;
                            acc += w.w * m1.y;
                        }
                        wrow += L;
                        if (fid >= 0) melbuf[fid] = MFX_LOG(fmaxf(acc, 1e-30f));
                    }
                    wave_sync();
                    // ---- DCT-II + lifter: out[c] = sum_m mel[m] * dct[m][c], ascending m (mfcccpu.cpp:222-232)
                    for (int c0 = 0; c0 < cols; c0 += 16) {
                        const int cc = c0 + l;
                        const bool act = cc < cols;
                        float acc;
                        if (p.dct) {
                            const float *dm = s_dct + (act ? cc : 0) * DS;
                            acc = 0.f;
                            for (int m = 0; m < nb_pad; m += 4) {
                                const float4 mv = *(const float4 *)(melbuf + m);
                                const float4 dv = *(const float4 *)(dm + m);
                                acc += mv.x * dv.x;
                                acc += mv.y * dv.y;
                                acc += mv.z * dv.z;
                                acc += mv.w * dv.w;
                            }
                        } else {
                            acc = melbuf[act ? cc : 0];
                        }
                        if (live && act) dst[cc] = acc;
                    }
                }
                wave_sync();
            }
            MFX_STAMP(6);
        }
        if (n_live <= 0) issue(cnxt.rsrc, lane_off(cnxt, slot)); // empty chunk: nothing was prefetched
        if (FUSE) { // the chunk's statics are on their way to memory: publish it to the delta wave
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            const int kk = c_cur - chunk_base;
            if (lane == 0)
                __hip_atomic_fetch_or(&s_done[kk >> 5], 1u << (kk & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // rotate the pipeline: next -> current, the index drawn a chunk ago -> next, draw another
        c_cur = c_nxt;
        ccur = cnxt;
        asm volatile("" : "+v"(v_nn)); // keep the LDS atomic's result in flight until here
        c_nxt = chunk_of(__builtin_amdgcn_readfirstlane(v_nn));
        cnxt = make_ctx(c_nxt);
        v_nn = next_index();
    }
#ifdef MFX_STAMPS
    if (lane == 0 && p.spec) {
        unsigned long long *o = (unsigned long long *)p.spec + (size_t)(blockIdx.x * kWaves + wave) * 8;
        unsigned long long st_rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1)::"memory");
        st_acc[7] = st_rt1 - st_rt0; // 100 MHz ticks over the same span
        for (int i = 0; i < 8; ++i) o[i] = st_acc[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// k_front1024: the 1024-point transform of a window of at most 512 samples (25 ms at 16 kHz zero padded to 1024:
// BASELINE configs[2]) on the k_front512 core -- 4 frames per wave iteration, 16 lanes per frame, ONE LDS transpose
// per 256-point complex transform -- instead of k_front_reg's wave per frame with three LDS passes (whose waves are
// bound by the number of dependent LDS round trips per frame).
//
// With x[n] = 0 for n >= 512 the 1024-point DFT splits by decimation in frequency over the SAME packed samples
// z[m] = x[2m] + i x[2m+1], m < 256:
//   even bins  X[2k]   = the 512-point real DFT of x = FFT256(z) + the twiddled split of k_front512 (phase E);
//   odd bins   X[2k+1] = U0[k] + W_1024^(2k+1) U1[k],  U_s = FFT256(x[2m+s] W_512^m):  V = FFT256(z W_512^m) = U0 + i U1
//              and conj U_s[k] = U_s[255-k], so with S = V[k] + conj V[255-k], D = V[k] - conj V[255-k],
//              T = (-i W_1024^(2k+1)) D:   X[2k+1] = (S + T) / 2,   X[2(255-k)+1] = conj(S - T) / 2          (phase O)
// -- the same split arithmetic with another twiddle and the partner in lane 15 - l, register 15 - p (a plain row mirror,
// no self-paired bins).  Phase O multiplies the samples by the window taps times W_512^m (a 2 x 2 real table per sample
// pair) and runs first; its 16 magnitudes per lane wait in registers while phase E re-converts the same raw words.
// Magnitudes land de-interleaved in the frame's slot (E[i] = bin 2i, O[i] = bin 2i+1); the mel walk reads two bins of
// each per 8-byte read and adds them in ascending bin order (mfcccpu.cpp:192-220); log, DCT on the matrix pipe as in
// k_front512 (20 K steps: at most 80 filters).
// ------------------------------------------------------------------------------------------------
constexpr int kWavesL = 12;       // waves per block = per CU (3 per SIMD: 168 registers)
constexpr int kSlotL = 672;       // dwords per frame slot: [0, 512) transposes, then E | O magnitudes (2 x 264); log energies
                                  // from 528 (672 = 32 mod 64: neighbouring slots sit on complementary bank halves)
constexpr int kOddOffL = 264;     // O magnitudes inside the slot
constexpr int kMelOffL = 528;
constexpr int kTabStrideO = 68;   // dwords per lane row of the phase-O window table (16 x (A, B, C, D) + pad: 17 16-byte words)
constexpr int kDctStepsL = 20;    // num_banks <= 80
constexpr int kDctRowL = 20;      // dwords per lane row of the B operand table (5 16-byte words: odd)

template <bool ALIGNED, int NM>
__global__ void __launch_bounds__(kWavesL * 64, 3) k_front1024(FrontParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int slot = lane >> 4, l = lane & 15;
    const int cols = p.cols, rounds = p.mel_rounds, RS = p.mel_row_stride;

    float *s_win = smem;                               // [16 l][kTabStride]: E window pairs (w[2n], w[2n+1]) * 0.5 / W2
    float *s_winO = s_win + 16 * kTabStride;           // [16 l][kTabStrideO]: (A, B, C, D) of pair n = l + 16 m
    float *s_tw = s_winO + 16 * kTabStrideO;           // [16 l][kTabStride]: W_256^(l k)
    float *s_splitE = s_tw + 16 * kTabStride;          // [16 l][kSplitStride]: -i W_1024^(2 (l + 16 p))
    float *s_splitO = s_splitE + 16 * kSplitStride;    // [16 l][kSplitStride]: -i W_1024^(2 (l + 16 p) + 1)
    float *s_melw = s_splitO + 16 * kSplitStride;      // [16][RS]
    int *s_mmeta = (int *)(s_melw + 16 * RS);          // [rounds][16] (first bin, filter id): one 8-byte read per round
    float *s_dct = (float *)(s_mmeta + 32 * rounds);   // [64][kDctRowL]: matrix-pipe B operands per lane
    float *s_wave = s_dct + 64 * kDctRowL + wave * (4 * kSlotL);
    float *xb = s_wave + slot * kSlotL;
    int *s_ctr = (int *)(s_dct + 64 * kDctRowL + kWavesL * (4 * kSlotL));
    if (tid == 0) *s_ctr = 0;

    for (int i = tid; i < 256; i += kWavesL * 64) { // HBM tables are [lane][m]
        ((float2 *)(s_win + (i >> 4) * kTabStride))[i & 15] = ((const float2 *)p.winpair)[i];
        ((float4 *)(s_winO + (i >> 4) * kTabStrideO))[i & 15] = ((const float4 *)p.win1024o)[i];
        ((float2 *)(s_tw + (i >> 4) * kTabStride))[i & 15] = ((const float2 *)p.twid_pass)[i];
    }
    for (int i = tid; i < 128; i += kWavesL * 64) { // twid_split holds -i W_1024^e, e <= 512, in natural order
        ((float2 *)(s_splitE + (i & 15) * kSplitStride))[i >> 4] = ((const float2 *)p.twid_split)[2 * i];
        ((float2 *)(s_splitO + (i & 15) * kSplitStride))[i >> 4] = ((const float2 *)p.twid_split)[2 * i + 1];
    }
    for (int i = tid; i < 16 * RS; i += kWavesL * 64) s_melw[i] = p.mel_lane_w[i];
    for (int i = tid; i < 16 * rounds; i += kWavesL * 64) {
        s_mmeta[2 * i] = p.mel_lane_start[i] >> 1; // (index into the even / odd magnitude arrays)
        const int fid = p.mel_lane_fid[i];
        s_mmeta[2 * i + 1] = fid < 0 ? 4 * kDctStepsL : fid; // idle lanes park their value in a word nobody reads
    }
    for (int i = tid; i < 64 * kDctRowL; i += kWavesL * 64) {
        // (MFX_DCT_QUARTERS: B operand of band 20 kb + j on lane (kb = lane >> 4, n = lane & 15); see k_front512)
        const int ln = i / kDctRowL, j = i - ln * kDctRowL, n = ln & 15;
        const int m = MFX_DCT_QUARTERS ? kDctStepsL * (ln >> 4) + j : 4 * j + (ln >> 4);
        s_dct[i] = (p.dct && m < p.num_banks && n < p.dct_len) ? p.dct[m * p.dct_len + n] : 0.f;
    }
    for (int i = lane; i < 4 * kSlotL; i += 64) s_wave[i] = 0.f; // words read before they are written meet zero weights: finite
    __syncthreads();

    // ---- chunk walk: as k_front512 (descriptor per chunk, software pipelined, block-local work counter)
    struct ChunkCtx {
        int64_t out_row;
        int n_live, odd0;
        __amdgpu_buffer_rsrc_t rsrc;
    };
    auto make_ctx = [&](int c) -> ChunkCtx {
        ChunkCtx x;
        const bool valid = c < p.n_chunks;
        const Chunk *chp = p.chunks + (valid ? c : 0);
        const int64_t pcm_off = chp->pcm_off;
        x.out_row = chp->out_row;
        const int n_frames = valid ? chp->n_frames : 0;
        const int64_t rows_left = p.row_limit - x.out_row;
        x.n_live = (int)(rows_left < n_frames ? (rows_left < 0 ? 0 : rows_left) : n_frames);
        const int64_t base_s = ALIGNED ? pcm_off : (pcm_off & ~(int64_t)1);
        x.odd0 = ALIGNED ? 0 : (int)(pcm_off & 1);
        int64_t bytes_left = valid ? (((p.pcm_total - base_s) * 2 + 3) & ~(int64_t)3) : 0; // (whole words: see k_front512)
        if (bytes_left > 0xfffffff0ll) bytes_left = 0xfffffff0ll;
        if (bytes_left < 0) bytes_left = 0;
        const uintptr_t bp = (uintptr_t)(p.pcm + base_s);
        const uint32_t bp_lo = __builtin_amdgcn_readfirstlane((uint32_t)bp);
        const uint32_t bp_hi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
        const uint32_t nbytes = __builtin_amdgcn_readfirstlane((uint32_t)bytes_left);
        x.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)bp_hi << 32) | bp_lo), 0, nbytes, 0x00020000);
        return x;
    };
    auto lane_off = [&](const ChunkCtx &x, int f) -> int {
        const int s = x.odd0 + f * p.shift + 2 * l;
        return ALIGNED ? s * 2 : (s & ~1) * 2;
    };
    auto next_index = [&]() -> int {
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return k;
    };
    auto chunk_of = [&](int k) -> int {
        const long long c = (long long)blockIdx.x + (long long)k * gridDim.x;
        return c < p.n_chunks ? (int)c : p.n_chunks;
    };
    int v_a = next_index(), v_b = next_index(), v_nn = next_index();
    int c_cur = chunk_of(__builtin_amdgcn_readfirstlane(v_a));
    int c_nxt = chunk_of(__builtin_amdgcn_readfirstlane(v_b));
    ChunkCtx ccur = make_ctx(c_cur);
    ChunkCtx cnxt = make_ctx(c_nxt);
    PcmRegs<ALIGNED, NM> cur;
    pcm_issue<ALIGNED, NM>(cur, ccur.rsrc, lane_off(ccur, slot));

    while (c_cur < p.n_chunks) {
        const int64_t out_row = ccur.out_row;
        const int n_live = ccur.n_live;
        const int odd0 = ccur.odd0;
        for (int f0 = 0; f0 < n_live; f0 += 4) {
            const int f = f0 + slot;
            const bool live = f < n_live;
            const bool last = f0 + 4 >= n_live;
            const bool odd = !ALIGNED && ((odd0 + f * p.shift) & 1);
            // Unaligned frames: the two raw words per sample pair are merged into one right away (both phases read the
            // pairs; 2 NM raw registers held across a phase spilled 34 of them) and the next frames are requested here.
            uint32_t dd[ALIGNED ? 1 : NM];
            if (!ALIGNED) {
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const uint32_t d0 = cur.d[2 * m], d1 = cur.d[2 * m + 1];
                    dd[m] = odd ? __builtin_amdgcn_alignbit(d1, d0, 16) : d0;
                }
                pcm_issue<ALIGNED, NM>(cur, last ? cnxt.rsrc : ccur.rsrc, last ? lane_off(cnxt, slot) : lane_off(ccur, f + 4));
            }
            // sample pair m of this lane (n = l + 16 m) as two floats
            auto pair_of = [&](int m, float &x0, float &x1) {
                const uint32_t d = ALIGNED ? cur.d[m] : dd[m];
                x0 = (float)(int)(short)(d & 0xffffu);
                x1 = (float)((int)d >> 16);
            };
            // pass A + inter-pass twiddle + 16 x 16 transpose through the slot + pass B: a[pp] = FFT256(a)[l + 16 pp]
            auto fft256 = [&](float2(&a)[16]) {
                fft16(a);
                {
                    float4 tq[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) tq[k] = lds_read_b128((const float4 *)(s_tw + l * kTabStride) + k);
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        const float4 t = tq[k >> 1];
                        if (k > 0) a[k] = cmul(a[k], make_float2(t.x, t.y));
                        a[k + 1] = cmul(a[k + 1], make_float2(t.z, t.w));
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) ((float2 *)(xb + k * 32))[l ^ (k & 14)] = a[k];
                wave_sync();
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float4 v = ((const float4 *)(xb + l * 32))[j ^ (l >> 1)];
                    a[2 * j] = make_float2(v.x, v.y);
                    a[2 * j + 1] = make_float2(v.z, v.w);
                }
                wave_sync();
                fft16(a);
            };

            // NM > 16 (a window longer than 512 samples, aligned frames only): the second half of the frame folds onto the
            // first before the two transforms,  Y0[m] = z[m] + z[m + 256]  (even bins),  Y1[m] = (z[m] - z[m + 256]) W_512^m
            // (odd bins).  The tables change roles: s_winO holds the window taps of all 32 rows of sample pairs, s_win the
            // twiddles W_512^(l + 16 m).
            constexpr bool FULL = NM > 16;
            static_assert(!FULL || ALIGNED, "windows longer than 512 samples: aligned frames only");
            // products (tap x sample) of the frame's two halves for row m, folded: sum (phase E) or difference (phase O).
            // `fresh` re-reads the raw words opaquely, so that phase E converts them again instead of keeping 64 floats alive.
            auto folded = [&](int m, bool want_sum, bool fresh) -> float2 {
                uint32_t dA = cur.d[m], dB = (m + 16 < NM) ? cur.d[(m + 16 < NM) ? m + 16 : 0] : 0u;
                if (fresh) asm volatile("" : "+v"(dA), "+v"(dB));
                const float2 tA = ((const float2 *)(s_winO + l * kTabStrideO))[m];
                float2 r = make_float2(tA.x * (float)(int)(short)(dA & 0xffffu), tA.y * (float)((int)dA >> 16));
                if (m + 16 < NM) {
                    const float2 tB = ((const float2 *)(s_winO + l * kTabStrideO))[m + 16];
                    const float2 pB = make_float2(tB.x * (float)(int)(short)(dB & 0xffffu), tB.y * (float)((int)dB >> 16));
                    r = want_sum ? make_float2(r.x + pB.x, r.y + pB.y) : make_float2(r.x - pB.x, r.y - pB.y);
                }
                return r;
            };

            // ---- phase O: odd bins
            float magO_k[8], magO_p[8];
            {
                float2 a[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if (FULL) {
                        a[m] = cmul(folded(m, false, false), ((const float2 *)(s_win + l * kTabStride))[m]); // x W_512^(l + 16 m)
                    } else if (m < NM) {
                        float x0, x1;
                        pair_of(m, x0, x1);
                        const float4 t = ((const float4 *)(s_winO + l * kTabStrideO))[m];
                        a[m] = make_float2(t.x * x0 + t.y * x1, t.z * x0 + t.w * x1);
                    } else {
                        a[m] = make_float2(0.f, 0.f);
                    }
                }
                fft256(a);
                float4 csq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) csq[j] = ((const float4 *)(s_splitO + l * kSplitStride))[j];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) {
                    // partner V[255 - k]: register 15 - pp of lane 15 - l
                    const float zr = row_mirror(a[15 - pp].x);
                    const float zi = row_mirror(a[15 - pp].y);
                    const float2 cs = (pp & 1) ? make_float2(csq[pp >> 1].z, csq[pp >> 1].w) : make_float2(csq[pp >> 1].x, csq[pp >> 1].y);
                    const float sr = a[pp].x + zr, si = a[pp].y - zi;
                    const float dr = a[pp].x - zr, di = a[pp].y + zi;
                    const float tr = cs.x * dr - cs.y * di;
                    const float ti = cs.x * di + cs.y * dr;
                    const float xr = sr + tr, xi = si + ti;
                    const float yr = sr - tr, yi = si - ti;
                    magO_k[pp] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi); // the window taps carry 0.5 / W2
                    magO_p[pp] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
                }
            }

            // ---- phase E: even bins = k_front512's transform of the same samples
            float magE_k[8], magE_p[8], magE128;
            {
                float2 a[16];
                constexpr int NW = FULL ? 1 : (NM + 1) / 2;
                float4 wq[NW];
                if (!FULL) {
#pragma unroll
                    for (int m = 0; m < NW; ++m) wq[m] = ((const float4 *)(s_win + l * kTabStride))[m];
                }
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if (FULL) {
                        a[m] = folded(m, true, true);
                    } else if (m < NM) {
                        float x0, x1;
                        pair_of(m, x0, x1);
                        const float2 w = (m & 1) ? make_float2(wq[m >> 1].z, wq[m >> 1].w) : make_float2(wq[m >> 1].x, wq[m >> 1].y);
                        a[m] = make_float2(w.x * x0, w.y * x1);
                    } else {
                        a[m] = make_float2(0.f, 0.f);
                    }
                }
                // the raw words are consumed: prefetch the next 4 frames (of this chunk, or the first of the next chunk)
                if (ALIGNED) pcm_issue<ALIGNED, NM>(cur, last ? cnxt.rsrc : ccur.rsrc, last ? lane_off(cnxt, slot) : lane_off(ccur, f + 4));
                fft256(a);
                const float m128r = a[8].x + a[8].x, m128i = a[8].y + a[8].y;
                magE128 = __builtin_amdgcn_sqrtf(m128r * m128r + m128i * m128i);
                float4 csq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) csq[j] = ((const float4 *)(s_splitE + l * kSplitStride))[j];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) {
                    const float zr = row_partner_own0(a[(16 - pp) & 15].x, a[15 - pp].x);
                    const float zi = row_partner_own0(a[(16 - pp) & 15].y, a[15 - pp].y);
                    const float2 cs = (pp & 1) ? make_float2(csq[pp >> 1].z, csq[pp >> 1].w) : make_float2(csq[pp >> 1].x, csq[pp >> 1].y);
                    const float sr = a[pp].x + zr, si = a[pp].y - zi;
                    const float dr = a[pp].x - zr, di = a[pp].y + zi;
                    const float tr = cs.x * dr - cs.y * di;
                    const float ti = cs.x * di + cs.y * dr;
                    const float xr = sr + tr, xi = si + ti;
                    const float yr = sr - tr, yi = si - ti;
                    magE_k[pp] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
                    magE_p[pp] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
                }
            }

            // ---- magnitudes to the slot: E[i] = |X[2 i]|, i <= 256; O[i] = |X[2 i + 1]|, i < 256
            {
                float *elo = xb + l, *ehi = xb + (144 - l);                        // i = l + 16 p and 256 - l - 16 p
                float *olo = xb + kOddOffL + l, *ohi = xb + kOddOffL + (143 - l);   // i = l + 16 p and 255 - l - 16 p
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) elo[16 * pp] = magE_k[pp];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) ehi[16 * pp] = magE_p[7 - pp];
                if (l == 0) xb[128] = magE128;
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) olo[16 * pp] = magO_k[pp];
#pragma unroll
                for (int pp = 0; pp < 8; ++pp) ohi[16 * pp] = magO_p[7 - pp];
            }
            wave_sync();

            // ---- mel filterbank: per round every lane walks one filter's bins in ascending order
            // (mfcccpu.cpp:192-220); starts are multiples of 4 bins: two even and two odd bins per 8-byte read
            const float *wrow = s_melw + l * RS;
            float *dst = p.feat + (out_row + f0) * (int64_t)p.feat_pitch + slot * p.feat_pitch;
            float *lm = xb + kMelOffL + 8 * slot;
            const int2 *mmeta = (const int2 *)s_mmeta + l;
            for (int r = 0; r < rounds; ++r) {
                const int L = p.mel_L[r];
                const int2 mt = *mmeta;
                mmeta += 16;
                const float *me = xb + mt.x, *mo = xb + kOddOffL + mt.x;
                const int fid = mt.y;
                float acc = 0.f;
                for (int s = 0; s < L; s += 8) {
                    float4 w[2];
                    float2 e[2], o[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) w[q] = *(const float4 *)(wrow + s + 4 * q);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        e[q] = lds_read_b64((const float2 *)(me + (s >> 1) + 2 * q));
                        o[q] = lds_read_b64((const float2 *)(mo + (s >> 1) + 2 * q));
                    }
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        acc += w[q].x * e[q].x;
                        acc += w[q].y * o[q].x;
                        acc += w[q].z * e[q].y;
                        acc += w[q].w * o[q].y;
                    }
                }
                wrow += L;
                lm[fid] = MFX_LOG(fmaxf(acc, 1e-30f)); // (idle lanes: fid names a word nobody reads)
            }
            wave_sync();
            if (!p.dct) {
                // no DCT (ceps_len = 0: filterbank features, up to 80 log mel energies per frame): the frame's row as it is
                if (live)
                    for (int cc = l; cc < cols; cc += 16) dst[cc] = lm[cc];
            } else
            // ---- DCT-II + lifter on the matrix pipe (see k_front512): frame `slot` in rows 4 slot .. 4 slot + 3
            {
#if MFX_DCT_QUARTERS
                // 20 v_mfma_f32_4x4x1_16b_f32 over band quarters (20 bands each: the frame's row holds them back to back,
                // 80-byte quarters), then the cross-row butterfly -- see k_front512
                const float4 *aq = (const float4 *)(s_wave + (lane & 3) * (kSlotL + 8) + kMelOffL + kDctStepsL * slot);
                const float4 *bq = (const float4 *)(s_dct + lane * kDctRowL);
                f32x4 dacc = {0.f, 0.f, 0.f, 0.f}, dacc2 = {0.f, 0.f, 0.f, 0.f};
                float4 av[kDctStepsL / 4], bv[kDctStepsL / 4];
#pragma unroll
                for (int j = 0; j < kDctStepsL / 4; ++j) {
                    av[j] = aq[j];
                    bv[j] = bq[j];
                }
#pragma unroll
                for (int j = 0; j < kDctStepsL / 4; ++j) {
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j].x, bv[j].x, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j].y, bv[j].y, dacc2, 0, 0, 0);
                    dacc = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j].z, bv[j].z, dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[j].w, bv[j].w, dacc2, 0, 0, 0);
                }
                const auto r01 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dacc[0] + dacc2[0]), __float_as_uint(dacc[1] + dacc2[1]), false, false);
                const auto r23 = __builtin_amdgcn_permlane16_swap(__float_as_uint(dacc[2] + dacc2[2]), __float_as_uint(dacc[3] + dacc2[3]), false, false);
                const float s01 = __uint_as_float(r01[0]) + __uint_as_float(r01[1]);
                const float s23 = __uint_as_float(r23[0]) + __uint_as_float(r23[1]);
                const auto rr = __builtin_amdgcn_permlane32_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
                const float outv = __uint_as_float(rr[0]) + __uint_as_float(rr[1]);
#else
                const float *arow = s_wave + (l >> 2) * (kSlotL + 8) + kMelOffL + slot;
                const float4 *bq = (const float4 *)(s_dct + lane * kDctRowL);
                float dctb[kDctStepsL];
#pragma unroll
                for (int j = 0; j < kDctStepsL / 4; ++j) {
                    const float4 t = bq[j];
                    dctb[4 * j] = t.x;
                    dctb[4 * j + 1] = t.y;
                    dctb[4 * j + 2] = t.z;
                    dctb[4 * j + 3] = t.w;
                }
                f32x4 dacc = {0.f, 0.f, 0.f, 0.f}, dacc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < kDctStepsL; j += 2) {
                    dacc = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j], dctb[j], dacc, 0, 0, 0);
                    dacc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(arow[4 * j + 4], dctb[j + 1], dacc2, 0, 0, 0);
                }
                const float outv = dacc[0] + dacc2[0];
#endif
                if (live && (l < cols || p.feat_pitch == 16)) dst[l] = outv;
            }
            wave_sync();
        }
        if (n_live <= 0) pcm_issue<ALIGNED, NM>(cur, cnxt.rsrc, lane_off(cnxt, slot)); // empty chunk: nothing was prefetched
        c_cur = c_nxt;
        ccur = cnxt;
        asm volatile("" : "+v"(v_nn));
        c_nxt = chunk_of(__builtin_amdgcn_readfirstlane(v_nn));
        cnxt = make_ctx(c_nxt);
        v_nn = next_index();
    }
}

// ------------------------------------------------------------------------------------------------
// Generic front end for the short transforms (64..512 points; 512 only when the register kernel above
// cannot take the configuration: stereo, more than 16 columns ...), mono or stereo, any alignment.
// Transforms of 1024 points and more run in k_front_reg below.
// One WAVE per frame (4 waves per block, each walking its own chunks): half-size complex Stockham
// FFT in the wave's own LDS buffers -- radix-4 stages, one radix-2 stage when log2 is odd, only
// wave-level synchronisation -- then the real split and the magnitudes.
//   FUSED: mel -> log -> DCT straight from LDS (no spectrum round trip through HBM), statics out;
//   else : magnitudes to the HBM spectrum buffer (streaming set_input).
// ------------------------------------------------------------------------------------------------
// G = threads that share one frame: 64 (a wave; wave-level synchronisation only).
#ifndef MFX_WAVE_MINW
#define MFX_WAVE_MINW 8 // waves per SIMD the register allocation aims at (8: 64 registers + 56 bytes of scratch, still 9 % faster than 5 / 6 resident blocks: profiles/r03/abx_front_wave_occupancy.txt)
#endif
template <bool FUSED, int G>
__global__ void __launch_bounds__(256, MFX_WAVE_MINW) k_front_wave(FrontParams p)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid / G, lane = tid % G; // 'wave' = frame group inside the block
    constexpr int NG = 256 / G;                               // frame groups per block
    auto group_sync = [&]() {
        if (G == 64)
            wave_sync();
        else
            __syncthreads();
    };
    const int W2 = p.fft_size, M = W2 >> 1;
    const int nb = p.num_banks, dl = p.dct_len;
    // shared tables (FUSED only: the 64-lane mel plan, MelWavePlan), then per wave: two complex buffers of M points and the
    // log mel energies of 4 frames waiting for the DCT (lm_fs4)
    const int RS = FUSED ? p.mel64_row_stride : 0, rounds = FUSED ? p.mel64_rounds : 0;
    const int WR = mel64_rows(nb);                            // weight rows in LDS (lanes that carry a filter)
    float *s_mw = smem;                                       // [WR][RS]
    int *s_mst = (int *)(s_mw + WR * RS);                     // [rounds][64]
    int *s_mfid = s_mst + 64 * rounds;                        // [rounds][64]
    const int FS = FUSED ? lm_fs4(nb) : 0;
    float *s_wave = (float *)(s_mfid + 64 * rounds) + wave * (4 * M + 4 * FS);
    float2 *bufA = (float2 *)s_wave;
    float2 *bufB = bufA + M;
    float *lm = s_wave + 4 * M;                               // [4][FS]
    (void)dl;
    if (FUSED) {
        for (int i = tid; i < WR * RS; i += 256) s_mw[i] = p.mel64_w[i];
        for (int i = tid; i < 64 * rounds; i += 256) {
            s_mst[i] = p.mel64_start[i];
            s_mfid[i] = p.mel64_fid[i];
        }
        for (int i = lane; i < 4 * M + 4 * FS; i += G) s_wave[i] = 0.f; // words read before they are written: finite
    }
    __syncthreads();
    const int dct_ks = p.dct_ksteps, dct_tiles64 = (dl + 63) >> 6;
    const int dct_bytes = (FUSED && p.dct_b4) ? dct_tiles64 * dct_ks * 1024 : 0;
    const __amdgpu_buffer_rsrc_t dct_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dct_b4, 0, dct_bytes, 0x00020000);

    const float2 *tw = (const float2 *)p.twid_half;   // W_M^k, k < M
    const float2 *cs = (const float2 *)p.twid_split;  // -i W_{W2}^k, k <= M
    const int ch_n = p.channels;
    const float scale = p.scale; // 0.5 / W2

    for (int c = blockIdx.x * NG + wave; c < p.n_chunks; c += gridDim.x * NG) {
        const Chunk ch = p.chunks[c];
        const int64_t rows_left = p.row_limit - ch.out_row;
        const int nf = (int)(rows_left < ch.n_frames ? (rows_left < 0 ? 0 : rows_left) : ch.n_frames);
        for (int f = 0; f < nf; ++f) {
            const int64_t s0 = ch.pcm_off + (int64_t)f * p.shift;
            // ---- framing + window: z[n] = (w[2n] x[2n], w[2n+1] x[2n+1]), zero beyond the window
            for (int n = lane; n < M; n += G) {
                float v[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * n + e;
                    float x = 0.f;
                    if (j < p.window_size) {
                        const int64_t s = s0 + j;
                        int xi;
                        if (ch_n == 2)
                            xi = ((int)p.pcm[2 * s] + (int)p.pcm[2 * s + 1]) >> 1; // stereo -> mono (L + R) >> 1
                        else
                            xi = (int)p.pcm[s];
                        x = p.window[j] * (float)xi;
                    }
                    v[e] = x;
                }
                bufA[n] = make_float2(v[0], v[1]);
            }
            group_sync();
            // ---- Stockham autosort FFT of M complex points
            float2 *x = bufA, *y = bufB;
            int len = M, st = 1, lg_st = 0; // st = 1 << lg_st (all sizes are powers of two: shifts, no division)
            while (len > 1) {
                if ((len & 3) == 0) {
                    const int n1 = len >> 2, tstep = st; // M / len == st
                    for (int idx = lane; idx < (M >> 2); idx += G) {
                        const int pp = idx >> lg_st, q = idx & (st - 1);
                        const float2 w1 = tw[pp * tstep], w2 = tw[2 * pp * tstep], w3 = tw[3 * pp * tstep];
                        const float2 a = x[q + st * pp], b = x[q + st * (pp + n1)];
                        const float2 cc = x[q + st * (pp + 2 * n1)], d = x[q + st * (pp + 3 * n1)];
                        const float2 apc = make_float2(a.x + cc.x, a.y + cc.y), amc = make_float2(a.x - cc.x, a.y - cc.y);
                        const float2 bpd = make_float2(b.x + d.x, b.y + d.y);
                        const float2 jbmd = make_float2(-(b.y - d.y), b.x - d.x); // i * (b - d)
                        y[q + st * (4 * pp)] = make_float2(apc.x + bpd.x, apc.y + bpd.y);
                        y[q + st * (4 * pp + 1)] = cmul(make_float2(amc.x - jbmd.x, amc.y - jbmd.y), w1);
                        y[q + st * (4 * pp + 2)] = cmul(make_float2(apc.x - bpd.x, apc.y - bpd.y), w2);
                        y[q + st * (4 * pp + 3)] = cmul(make_float2(amc.x + jbmd.x, amc.y + jbmd.y), w3);
                    }
                    len >>= 2;
                    st <<= 2;
                    lg_st += 2;
                } else {
                    const int n1 = len >> 1, tstep = st;
                    for (int idx = lane; idx < (M >> 1); idx += G) {
                        const int pp = idx >> lg_st, q = idx & (st - 1);
                        const float2 w = tw[pp * tstep];
                        const float2 a = x[q + st * pp], b = x[q + st * (pp + n1)];
                        y[q + st * (2 * pp)] = make_float2(a.x + b.x, a.y + b.y);
                        y[q + st * (2 * pp + 1)] = cmul(make_float2(a.x - b.x, a.y - b.y), w);
                    }
                    len >>= 1;
                    st <<= 1;
                    lg_st += 1;
                }
                group_sync();
                float2 *t = x;
                x = y;
                y = t;
            }
            // ---- real split + magnitude into the other buffer (as floats)
            float *mag = (float *)y;
            float *dst_spec = FUSED ? nullptr : p.spec + (ch.out_row + f) * (int64_t)p.spec_pitch;
            for (int k = lane; k <= M; k += G) {
                const float2 zk = x[k & (M - 1)];
                const float2 zm = x[(M - k) & (M - 1)];
                const float sr = zk.x + zm.x, si = zk.y - zm.y;
                const float dr = zk.x - zm.x, di = zk.y + zm.y;
                const float2 w = cs[k];
                const float xr = sr + (w.x * dr - w.y * di);
                const float xi = si + (w.x * di + w.y * dr);
                const float m = __builtin_amdgcn_sqrtf(xr * xr + xi * xi) * scale;
                if (FUSED)
                    mag[k] = m;
                else
                    dst_spec[k] = m;
            }
            group_sync();
            if (FUSED) {
                // mel walk on the wave's 64 lanes + log (the magnitudes sit in the 2 M floats of the other buffer: the plan
                // reads at most up to word W2 - 1, stale but finite beyond bin M); DCT once per 4 frames and at the chunk's end
                mel64_walk_log(mag, lm + (f & 3) * FS, FS - 1, s_mw, s_mst, s_mfid, p.mel64_L, rounds, RS, lane, WR);
                group_sync();
                if ((f & 3) == 3 || f == nf - 1) {
                    const int g0 = f & ~3;
                    dct4_store<3>(lm, FS, dct_rsrc, dct_bytes, dct_ks, dct_tiles64, p.dct_b4 != nullptr, lane, p.cols, p.feat,
                               (int64_t)p.feat_pitch, ch.out_row + g0, f - g0 + 1);
                    group_sync();
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Long transforms (1024 / 2048 / 4096 points): one WAVE per frame, the half-size complex FFT as three
// Stockham passes whose butterflies (radix 16 / 8 / 4) run in registers.  Every pass reads all of its
// inputs into registers before it writes, so the frame needs ONE LDS buffer of M complex points and the
// three passes cost three LDS round trips (the radix-4 loop of k_front_wave: five or six, two buffers).
//   M =  512:  8 x  8 x 8          M = 1024: 16 x 16 x 4          M = 2048: 16 x 16 x 8
// Pass 1 takes its inputs straight from the PCM loads (lane l needs z[l + (M/R) r]: exactly the strided
// samples it loaded); the last pass has no twiddles.  The real split pairs bins k and M - k: one
// partner fetch gives both magnitudes (|S + T| and |S - T|).  Tables (pass twiddles W_M^k, split twiddles,
// window pairs with the output scale folded in) are shared by the block's waves in LDS.
//   FUSED (M = 512): mel -> log -> DCT from the magnitudes in LDS; else magnitudes to the HBM spectrum.
//   PAIR: mono, even shift / offsets / window length -> two samples per 32-bit load.
// ------------------------------------------------------------------------------------------------
#ifndef MFX_REG10_THREADS
#define MFX_REG10_THREADS 768   // most threads per block of the fused 2048-point build (sets its register budget: 168)
#endif

// window pairs k_front_reg keeps in LDS: the 64-pair rows that carry taps (a 25 ms window zero padded to the transform
// leaves most of the M rows empty)
__host__ __device__ inline int reg_window_pairs(int window_size, int M)
{
    const int n = (((window_size + 1) / 2) + 63) & ~63;
    return n < M ? n : M;
}

template <int R>
__device__ __forceinline__ void fft_r(float2 (&v)[R])
{
    if (R == 16) {
        fft16(reinterpret_cast<float2(&)[16]>(v));
    } else if (R == 8) {
        fft8(reinterpret_cast<float2(&)[8]>(v));
    } else {
        float2 o0, o1, o2, o3;
        dft4(v[0], v[1], v[2], v[3], o0, o1, o2, o3);
        v[0] = o0;
        v[1] = o1;
        v[2] = o2;
        v[3] = o3;
    }
}

// One Stockham pass of radix R over the M points in `buf` (in place: all reads, then all writes), sub-transform
// length LEN before the pass, stride ST = M / LEN.  NB = butterflies per lane.  `v` in/out: with FROM_REGS the
// inputs are already in v (pass 1), otherwise they are read from buf.
// Index of point i in the wave's complex LDS buffer (LP = 3: radix-8 passes, LP = 4: radix-16 passes).
template <int LP>
__device__ __forceinline__ int pad_idx(int i)
{
#ifdef MFX_REG_PADDED
    return i + (i >> LP);
#else
    // XOR swizzle of the complex buffer (no padding): the strided writes of the first pass (R consecutive points per
    // lane: without it all 16 lanes of a write group fall on one bank pair) spread over the banks, and every run of 16 /
    // 32 consecutive points -- the later passes' and the real split's accesses -- stays a permutation inside its own
    // 16-point block.  Simulated against the LDS access rules for all passes (8.8.8 / 16.16.4 / 16.16.8): 152 / 304 / 608
    // LDS cycles per transform against 224 / 384 / 768 with the padded layout of round 1 (144 / 288 / 576 conflict free).
    return LP == 3 ? (i ^ ((i >> 4) & 7) ^ (((i >> 6) & 1) << 3)) : (i ^ ((i >> 4) & 15));
#endif
}

// s_tw: this pass's twiddles W_LEN^(pp k) laid out [k - 1][pp], pp < LEN / R (unused by the last pass)
// TWREG: the pass's twiddles are already in registers (twr[k - 1], one butterfly per lane: they depend on the lane only)
template <int M, int R, int LEN, bool FROM_REGS, int LP, bool TWREG = false>
__device__ __forceinline__ void stockham_pass(float2 *buf, const float2 *s_tw, int lane, float2 (&v)[M / 64],
                                              const float2 *twr = nullptr)
{
    constexpr int ST = M / LEN, N1 = LEN / R, NB = M / R / 64;
    static_assert(NB >= 1, "a pass needs at least one butterfly per lane");
    if (!FROM_REGS) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const int idx = lane + 64 * b, pp = idx / ST, q = idx % ST;
#pragma unroll
            for (int r = 0; r < R; ++r) v[b * R + r] = buf[pad_idx<LP>(q + ST * (pp + r * N1))];
        }
        wave_sync();
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int idx = lane + 64 * b, pp = idx / ST, q = idx % ST;
        float2(&w)[R] = reinterpret_cast<float2(&)[R]>(v[b * R]);
        fft_r<R>(w);
        // The last pass leaves Z[lane + 64 j], j = b + NB k, in the lane's registers: the real split takes its own
        // bins (j < NV / 2) from there, only the upper half -- the partners Z[M - k] -- goes through LDS.
        if (LEN != R) buf[pad_idx<LP>(q + ST * (R * pp))] = w[0];
#pragma unroll
        for (int k = 1; k < R; ++k) {
            if (LEN == R) {
                if (k >= R / 2) buf[pad_idx<LP>(q + ST * (R * pp + k))] = w[k];
            } else {
                buf[pad_idx<LP>(q + ST * (R * pp + k))] = cmul(w[k], TWREG ? twr[k - 1] : s_tw[(k - 1) * N1 + pp]); // W_LEN^(pp k)
            }
        }
    }
    wave_sync();
}

// HALF: a short window zero padded to the transform -- at most M samples at 1024 points (BASELINE configs[2]), at most
// 1280 at 2048 points (25 ms at 44.1 kHz, configs[4]: 10 of the 16 rows of sample pairs): the
// upper half of every lane's sample pairs is zero at compile time and pass 1 sheds the arithmetic on it.
template <int LOG2M, bool FUSED, bool PAIR, bool HALF>
__global__ void __launch_bounds__(LOG2M >= 11 ? 256 : (LOG2M == 10 && FUSED) ? MFX_REG10_THREADS : 1024) k_front_reg(FrontParams p)
{
    constexpr int M = 1 << LOG2M, NV = M / 64;
    constexpr int R1 = (LOG2M == 9) ? 8 : 16, R2 = R1, R3 = M / (R1 * R2);
#ifdef MFX_REG_PADDED
    constexpr int LP = (LOG2M == 9) ? 3 : 4, MP = M + (M >> LP); // padded buffer (pad_idx)
#else
    constexpr int LP = (LOG2M == 9) ? 3 : 4, MP = M;             // swizzled buffer (pad_idx)
#endif
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, n_waves = blockDim.x >> 6;
    int lane = tid & 63;
    const int nb = p.num_banks, dl = p.dct_len;
    // shared tables, then one buffer of M complex points (+ mel scratch) per wave
    constexpr int NT1 = (R1 - 1) * (M / R1), NT2 = (R2 - 1) * (M / (R1 * R2));
    float2 *s_tw = (float2 *)smem;                 // pass 1 [R1-1][M/R1], then pass 2 [R2-1][M/(R1 R2)]; M slots reserved
    float2 *s_cs = s_tw + M;                       // [M/2 + 1]  -i W_{2M}^k (one per bin pair), padded to even
    float2 *s_win = s_cs + (M / 2 + 2);            // [nwin] (w[2n], w[2n+1]) * 0.5 / W2: whole 64-pair rows that carry taps
    const int nwin = reg_window_pairs(p.window_size, M);
    // FUSED: the mel walk's per-lane weight rows and plan (MelWavePlan), then per wave the complex buffer and the
    // log mel energies of 4 frames (the DCT runs on the matrix pipe once per 4 frames)
    const int RS = FUSED ? p.mel64_row_stride : 0, rounds = FUSED ? p.mel64_rounds : 0;
    const int WR = mel64_rows(nb);                             // weight rows in LDS (lanes that carry a filter)
    float *s_mw = (float *)(s_win + nwin);                     // [WR][RS]
    int *s_mst = (int *)(s_mw + WR * RS);                      // [rounds][64]
    int *s_mfid = s_mst + 64 * rounds;                         // [rounds][64]
    const int nbp = FUSED ? lm_stride(nb) : 0;
    float *s_wave = (float *)(s_mfid + 64 * rounds) + wave * (2 * MP + 4 * nbp);
    float2 *buf = (float2 *)s_wave;
    float *lm = s_wave + 2 * MP;                               // [4][nbp]
    int *s_ctr = (int *)((float *)(s_mfid + 64 * rounds) + n_waves * (2 * MP + 4 * nbp)); // block-local work counter
    if (tid == 0) *s_ctr = 0;
    (void)dl;

    const float scale = p.scale; // 0.5 / W2, a power of two: folded into the window taps (exact)
    static_assert(NT1 + NT2 <= M, "pass tables fit the reserved slots");
    for (int i = tid; i < M; i += blockDim.x) {
        if (i < NT1 + NT2) s_tw[i] = ((const float2 *)p.twid_reg)[i];
        if (i < nwin) {
            const float2 wv = ((const float2 *)p.window)[i];
            s_win[i] = make_float2(wv.x * scale, wv.y * scale);
        }
    }
    for (int i = tid; i <= M / 2; i += blockDim.x) s_cs[i] = ((const float2 *)p.twid_split)[i];
    if (FUSED) {
        for (int i = tid; i < WR * RS; i += blockDim.x) s_mw[i] = p.mel64_w[i];
        for (int i = tid; i < 64 * rounds; i += blockDim.x) {
            s_mst[i] = p.mel64_start[i];
            s_mfid[i] = p.mel64_fid[i];
        }
        for (int i = lane; i < 4 * nbp; i += 64) lm[i] = 0.f; // words the walk never writes meet zero operands: keep them finite
    }
    __syncthreads();

    // 1024 points, one word per sample pair: 40 registers are free at 16 waves per CU -- the twiddles of passes 1 and 2
    // (7 + 7 complex values that depend on the lane only) live in registers instead of being read from LDS every frame.
    // 2048 points, stereo / odd-offset build: 15 + 15 values, inside the 168-register budget of 12 waves per CU.
#ifdef MFX_NO_TWREG
    constexpr bool TWREG = false, TWREG_2 = false;
#else
#ifndef MFX_TWREG10
#define MFX_TWREG10 2
#endif
    constexpr bool TW10 = LOG2M == 10 && FUSED && !PAIR;
    constexpr bool TWREG = (LOG2M == 9 && PAIR && FUSED && HALF) // (a full-length window needs the registers for its samples)
                           || (TW10 && MFX_TWREG10 >= 1);
    constexpr bool TWREG_2 = TWREG && (!TW10 || MFX_TWREG10 >= 2);
#endif
    float2 tw1[R1 - 1], tw2[R2 - 1];
    if (TWREG) {
#pragma unroll
        for (int k = 1; k < R1; ++k) tw1[k - 1] = s_tw[(k - 1) * (M / R1) + lane];                      // W_M^(lane k)
    }
    if (TWREG_2) {
#pragma unroll
        for (int k = 1; k < R2; ++k) tw2[k - 1] = s_tw[NT1 + (k - 1) * (M / (R1 * R2)) + lane / R1];  // W_(M/R1)^(pp k), pp = lane / R1
    }
    constexpr bool META_REG = FUSED && LOG2M == 9; // (the 2048-point builds have no register to spare)
    int mst0 = 0, mfid0 = -1, mst1 = 0, mfid1 = -1;
    if (META_REG) {
        if (rounds > 0) mst0 = s_mst[lane], mfid0 = s_mfid[lane];
        if (rounds > 1) mst1 = s_mst[64 + lane], mfid1 = s_mfid[64 + lane];
    }
    const int ch_n = p.channels, W = p.window_size;
    // PREFETCH: the raw samples of the NEXT frame are requested while this frame's mel stage runs, so their latency never
    // shows: the builds that load one 32-bit word per sample pair, and the short-window 2048-point stereo / odd-offset
    // build (two words per pair, 10 rows: 20 registers -- with all 16 rows it spilled 71 registers, C5 1.18 ms against 1.02;
    // the 1024-point stereo build has no registers left for it at 16 waves per CU).
    constexpr int NJ = !HALF ? NV : LOG2M == 10 ? 10 : NV / 2;   // rows of 64 sample pairs that can carry window taps
    constexpr bool PREFETCH = (PAIR && (FUSED || LOG2M == 9)) || (!PAIR && LOG2M == 10 && FUSED && HALF);
    uint32_t raw[PAIR ? NJ : 2 * NJ];
    struct __attribute__((aligned(4))) Pair32 {
        uint32_t x, y;
    };
    typedef uint32_t __attribute__((aligned(2))) u32_a2;
    auto issue = [&](int64_t s0) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = lane + 64 * j;
            if (PAIR) {
                raw[j] = 0u;
                if (2 * n < W) raw[j] = ((const uint32_t *)(p.pcm + s0))[n];
            } else {
                raw[2 * j] = raw[2 * j + 1] = 0u;
                if (2 * n < W) {
                    const bool has1 = 2 * n + 1 < W;
                    if (ch_n == 2) { // (see the conversion below for the layouts)
                        const Pair32 dd = *(const Pair32 *)((const uint32_t *)p.pcm + (s0 + 2 * n) - (has1 ? 0 : 1));
                        raw[2 * j] = dd.x;
                        raw[2 * j + 1] = dd.y;
                    } else {
                        raw[2 * j] = *(const u32_a2 *)(p.pcm + s0 + 2 * n - (has1 ? 0 : 1));
                    }
                }
            }
        }
    };
    // Block b owns chunks b, b + B, b + 2 B, ...; its waves draw from that list through a counter in LDS (as in
    // k_front512), one draw ahead: waves that the SIMD arbiter favours take more chunks instead of finishing early, and
    // the last chunks of the grid do not wait for one wave's fixed share (C3 0.313 -> 0.308 ms).  The 2048-point builds
    // keep the fixed round-robin deal: they are at their register budget, and the draw's bookkeeping spilled 9 more
    // registers there (C5 0.723 -> 0.759 ms).
    constexpr bool DRAW = LOG2M != 10;
    int fixed_c = blockIdx.x * n_waves + wave; // (!DRAW: wave w of the grid takes chunks w, w + W, w + 2 W, ...)
    auto draw = [&]() -> int {
        if (!DRAW) {
            const int cc = fixed_c < p.n_chunks ? fixed_c : p.n_chunks;
            if (fixed_c < p.n_chunks) fixed_c += gridDim.x * n_waves;
            return cc;
        }
        int k = 0;
        if ((tid & 63) == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        k = __builtin_amdgcn_readfirstlane(k);
        const long long cc = (long long)blockIdx.x + (long long)k * gridDim.x;
        return cc < p.n_chunks ? (int)cc : p.n_chunks;
    };
    bool fetched = false; // raw holds the frame about to be worked on
    int c = draw(), c_next = draw();
    for (; c < p.n_chunks; c = c_next, c_next = draw()) {
        const Chunk ch = p.chunks[c];
        const int64_t rows_left = p.row_limit - ch.out_row;
        const int nf = (int)(rows_left < ch.n_frames ? (rows_left < 0 ? 0 : rows_left) : ch.n_frames);
        // the chunk after this one (its first frame is requested during this chunk's last)
        const bool more = c_next < p.n_chunks;
        const Chunk chn = p.chunks[more ? c_next : c];
        const int64_t rows_left_n = p.row_limit - chn.out_row;
        const bool next_has = more && chn.n_frames > 0 && rows_left_n > 0;
        for (int f = 0; f < nf; ++f) {
#ifndef MFX_REG_NO_LAUNDER
            // 2048 points: every frame re-derives its LDS addresses from the lane number (one or two vector instructions
            // each) instead of holding a hundred loop-invariant addresses in registers -- 99 registers instead of 215, so
            // 10 - 12 waves fit a CU instead of 8 (C5: 1.02 -> 0.92 ms; at equal occupancy the extra instructions cost
            // 6 %; at 1024 points, where 16 waves fit anyway, they cost 17 % and the build keeps its addresses)
            if (LOG2M == 10 && FUSED) asm volatile("" : "+v"(lane));
#endif
            const int64_t s0 = ch.pcm_off + (int64_t)f * p.shift;
            if (PREFETCH && !fetched) issue(s0);
            // ---- framing + window, straight into the registers of pass 1: z[n], n = lane + 64 j
            float2 v[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int n = lane + 64 * j;
                if (j >= NJ) { // 2 n >= M >= W: no taps here
                    v[j] = make_float2(0.f, 0.f);
                    continue;
                }
                float x0 = 0.f, x1 = 0.f;
                if (PREFETCH && PAIR) {
                    const uint32_t d = raw[j];
                    x0 = (float)(int)(short)(d & 0xffffu);
                    x1 = (float)((int)d >> 16);
                } else if (PREFETCH) { // (words past the window were set to zero)
                    const bool has1 = 2 * n + 1 < W;
                    if (ch_n == 2) {
                        const uint32_t d0 = has1 ? raw[2 * j] : raw[2 * j + 1], d1 = raw[2 * j + 1];
                        x0 = stereo_mean(d0);
                        if (has1) x1 = stereo_mean(d1);
                    } else {
                        const uint32_t d = raw[2 * j];
                        x0 = (float)(int)(short)(has1 ? (d & 0xffffu) : (d >> 16));
                        if (has1) x1 = (float)((int)d >> 16);
                    }
                } else if (2 * n < W) { // loaded where it is consumed
                    if (PAIR) {
                        const uint32_t d = ((const uint32_t *)(p.pcm + s0))[n];
                        x0 = (float)(int)(short)(d & 0xffffu);
                        x1 = (float)((int)d >> 16);
                    } else if (ch_n == 2) {
                        // interleaved stereo: sample s is one aligned 32-bit word (L | R << 16); mono = (L + R) >> 1.
                        // The pair (s, s + 1) comes as ONE 8-byte load at 4-byte alignment (consecutive lanes then
                        // cover 512 contiguous bytes); an odd window's last pair re-reads its own sample instead
                        // of the one past the frame.
                        const bool has1 = 2 * n + 1 < W;
                        const uint32_t *w32 = (const uint32_t *)p.pcm + (s0 + 2 * n) - (has1 ? 0 : 1);
                        const Pair32 dd = *(const Pair32 *)w32;
                        const uint32_t d0 = has1 ? dd.x : dd.y, d1 = dd.y;
                        x0 = stereo_mean(d0);
                        if (has1) x1 = stereo_mean(d1);
                    } else {
                        // mono at an odd sample offset: the pair as ONE 4-byte load at 2-byte alignment
                        const bool has1 = 2 * n + 1 < W;
                        const uint32_t d = *(const u32_a2 *)(p.pcm + s0 + 2 * n - (has1 ? 0 : 1));
                        x0 = (float)(int)(short)(has1 ? (d & 0xffffu) : (d >> 16));
                        if (has1) x1 = (float)((int)d >> 16);
                    }
                }
                const float2 w = 64 * j < nwin ? s_win[n] : make_float2(0.f, 0.f); // (rows past the window: no table)
                v[j] = make_float2(w.x * x0, w.y * x1);
            }
            // Pass 1 wants, per butterfly b, its R1 inputs z[pp + (M/R1) r] contiguous in v: with NB1 = M/R1/64
            // butterflies per lane, z[lane + 64 j] is input r = j / NB1 of butterfly b = j % NB1.
            {
                constexpr int NB1 = M / R1 / 64;
                if (NB1 > 1) {
                    float2 t[NV];
#pragma unroll
                    for (int j = 0; j < NV; ++j) t[(j % NB1) * R1 + j / NB1] = v[j];
#pragma unroll
                    for (int j = 0; j < NV; ++j) v[j] = t[j];
                }
            }
            stockham_pass<M, R1, M, true, LP, TWREG>(buf, s_tw, lane, v, tw1);
            stockham_pass<M, R2, M / R1, false, LP, TWREG_2>(buf, s_tw + NT1, lane, v, tw2);
            stockham_pass<M, R3, R3, false, LP>(buf, s_tw, lane, v);

            // ---- real split over the bin pairs (k, M - k), k = lane + 64 j <= M/2, and the magnitudes
            constexpr int NP = M / 128; // pairs per lane (+ the self-paired k = M/2 on lane 0)
            float mag_lo[NP + 1], mag_hi[NP + 1];
#pragma unroll
            for (int j = 0; j <= NP; ++j) {
                const int k = (j < NP) ? lane + 64 * j : M / 2;
                // own bin from the last pass's registers (v[b R3 + kk] = Z[lane + 64 (b + NB3 kk)]); the partner from LDS.
                // Z[0] pairs with itself (lower half, not in LDS), Z[M/2] too (upper half: read back by lane 0 ... all lanes)
                constexpr int NB3 = M / R3 / 64;
                float2 zm = buf[pad_idx<LP>(j < NP ? (((M - k) & (M - 1)) | (M / 2)) : M / 2)];
                const float2 zk = j < NP ? v[(j % NB3) * R3 + j / NB3] : zm;
                if (j == 0 && lane == 0) zm = zk;
                const float sr = zk.x + zm.x, si = zk.y - zm.y;
                const float dr = zk.x - zm.x, di = zk.y + zm.y;
                const float2 w = s_cs[k];
                const float tr = w.x * dr - w.y * di, ti = w.x * di + w.y * dr;
                const float ar = sr + tr, ai = si + ti, br = sr - tr, bi = si - ti;
                mag_lo[j] = __builtin_amdgcn_sqrtf(ar * ar + ai * ai); // |X[k]| / W2
                mag_hi[j] = __builtin_amdgcn_sqrtf(br * br + bi * bi); // |X[M - k]| / W2
            }
            // Request the next frame of this wave (the chunk's next one, or the first frame of the wave's next chunk) here,
            // where few registers are live: the words arrive under the mel / DCT stage (or the spectrum stores).
            if (PREFETCH) {
                fetched = f + 1 < nf || next_has;
                if (f + 1 < nf)
                    issue(s0 + p.shift);
                else if (next_has)
                    issue(chn.pcm_off);
            }
            wave_sync();
#if defined(MFX_ABLATE_REG) && MFX_ABLATE_REG >= 1
            if (FUSED) { // dev-only: stop after the magnitudes (keeps them live)
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j <= NP; ++j) acc += mag_lo[j] + mag_hi[j];
                if (lane < p.cols) (p.feat + (ch.out_row + f) * (int64_t)p.feat_pitch)[lane] = acc;
                continue;
            }
#endif
            if (FUSED) {
                float *mag = (float *)buf; // in place: every complex point has been read
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int k = lane + 64 * j;
                    mag[k] = mag_lo[j];
                    mag[M - k] = mag_hi[j];
                }
                if (lane == 0) mag[M / 2] = mag_lo[NP];
                wave_sync();
                // ---- mel filterbank: per round every lane walks ONE filter's bins in ascending order, one chain of
                // multiply-adds (mfcccpu.cpp:192-220).  Weights come from the lane's own zero-padded row (16-byte reads,
                // disjoint bank quads), magnitudes as 8-byte reads from even starts the host spread over the banks.
                {
                    float *lmf = lm + (f & 3) * nbp;
                    const float *wrow = s_mw + (lane < WR ? lane : WR - 1) * RS;
                    auto one_round = [&](int r, int st, int fid) {
                        const int L = p.mel64_L[r];
                        const float *mg = mag + st;
                        float acc = 0.f;
                        int s2 = 0;
                        if (LOG2M == 10) // (2048 points: 12 reads in flight per trip, half the dependent round trips: C5 -1 %)
                            for (; s2 + 16 <= L; s2 += 16) {
                                float4 w[4];
                                float2 mm[8];
#pragma unroll
                                for (int q = 0; q < 4; ++q) w[q] = lds_read_b128((const float4 *)(wrow + s2 + 4 * q));
#pragma unroll
                                for (int q = 0; q < 8; ++q) mm[q] = lds_read_b64((const float2 *)(mg + s2 + 2 * q));
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    acc += w[q].x * mm[2 * q].x;
                                    acc += w[q].y * mm[2 * q].y;
                                    acc += w[q].z * mm[2 * q + 1].x;
                                    acc += w[q].w * mm[2 * q + 1].y;
                                }
                            }
                        for (; s2 < L; s2 += 8) {
                            const float4 w0 = lds_read_b128((const float4 *)(wrow + s2));
                            const float4 w1 = lds_read_b128((const float4 *)(wrow + s2 + 4));
                            float2 mm[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) mm[q] = lds_read_b64((const float2 *)(mg + s2 + 2 * q));
                            acc += w0.x * mm[0].x;
                            acc += w0.y * mm[0].y;
                            acc += w0.z * mm[1].x;
                            acc += w0.w * mm[1].y;
                            acc += w1.x * mm[2].x;
                            acc += w1.y * mm[2].y;
                            acc += w1.z * mm[3].x;
                            acc += w1.w * mm[3].y;
                        }
                        wrow += L;
                        lmf[fid >= 0 ? fid : nbp - 1] = MFX_LOG(fmaxf(acc, 1e-30f)); // idle lane: the row's spare word
                    };
                    // (META_REG: the first two rounds' starts / filter ids wait in registers -- one dependent LDS round trip
                    // less per round; the waves are bound by the number of those, not by LDS bytes)
                    int r = 0;
                    if (META_REG) {
                        if (rounds > 0) one_round(0, mst0, mfid0);
                        if (rounds > 1) one_round(1, mst1, mfid1);
                        r = 2;
                    }
                    for (; r < rounds; ++r) one_round(r, s_mst[r * 64 + lane], s_mfid[r * 64 + lane]);
                }
                wave_sync();
                // ---- every 4th frame (and at the chunk's end): DCT-II + lifter of the waiting frames on the matrix pipe,
                // D[row][c] = sum_m A[row][m] B[m][c] with frame g in rows 4g..4g+3 (all lanes read valid energies), so
                // register 0 of the result is out[g][c] on lane (g, c); tiles of 16 output columns, K steps of 4 bands,
                // two accumulator chains.  B operands come from L1 / L2 (p.dct_b), A operands from the rows above.
                if ((f & 3) == 3 || f == nf - 1) {
                    const int g0 = f & ~3, gcount = f - g0 + 1;
                    const int gi = lane >> 4, n = lane & 15;
                    float *orow = p.feat + (ch.out_row + g0 + (gi < gcount ? gi : 0)) * (int64_t)p.feat_pitch;
                    if (p.dct_b) {
                        const float *arow = lm + (n >> 2) * nbp + gi;
                        const int ks = p.dct_ksteps;
                        // up to TG tiles of 16 columns share one walk over the K steps: the A operands (the frames' log
                        // energies) are read once for all of them, and a batch of 8 steps waits once for its operands
                        // (TG = 3 where the register budget allows: the 2048-point builds)
                        constexpr int TG = LOG2M == 10 ? 3 : 1;
                        for (int t0 = 0; t0 < p.dct_tiles; t0 += TG) {
                            const int nt = p.dct_tiles - t0 < TG ? p.dct_tiles - t0 : TG;
                            const float *bp = p.dct_b + (int64_t)t0 * ks * 64 + lane;
                            f32x4 d0[TG], d1[TG];
#pragma unroll
                            for (int tt = 0; tt < TG; ++tt) d0[tt] = d1[tt] = f32x4{0.f, 0.f, 0.f, 0.f};
                            // K steps in batches of 8: the batch's operand loads (8 per tile from L1 / L2, 8 from LDS) are all
                            // in flight before its first matrix instruction; steps past the matrix get zero operands
                            for (int j0 = 0; j0 < ks; j0 += 8) {
                                float bv[TG][8], av[8];
#pragma unroll
                                for (int u = 0; u < 8; ++u) av[u] = arow[j0 + u < ks ? 4 * (j0 + u) : 0];
#pragma unroll
                                for (int tt = 0; tt < TG; ++tt)
#pragma unroll
                                    for (int u = 0; u < 8; ++u)
                                        bv[tt][u] = (tt < nt && j0 + u < ks) ? bp[((int64_t)tt * ks + j0 + u) * 64] : 0.f;
#pragma unroll
                                for (int tt = 0; tt < TG; ++tt) {
                                    if (tt >= nt) break;
#pragma unroll
                                    for (int u = 0; u < 8; u += 2) {
                                        d0[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[tt][u], d0[tt], 0, 0, 0);
                                        d1[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u + 1], bv[tt][u + 1], d1[tt], 0, 0, 0);
                                    }
                                }
                            }
#pragma unroll
                            for (int tt = 0; tt < TG; ++tt) {
                                const int col = 16 * (t0 + tt) + n;
                                if (tt < nt && gi < gcount && col < p.cols) orow[col] = d0[tt][0] + d1[tt][0];
                            }
                        }
                    } else { // no DCT: the log mel energies are the features
                        for (int g = 0; g < gcount; ++g)
                            for (int cc = lane; cc < p.cols; cc += 64)
                                (p.feat + (ch.out_row + g0 + g) * (int64_t)p.feat_pitch)[cc] = lm[g * nbp + cc];
                    }
                    wave_sync();
                }
            } else {
                float *dst = p.spec + (ch.out_row + f) * (int64_t)p.spec_pitch;
#pragma unroll
                for (int j = 0; j < NP; ++j) {
                    const int k = lane + 64 * j;
                    dst[k] = mag_lo[j];
                    dst[M - k] = mag_hi[j];
                }
                if (lane == 0) dst[M / 2] = mag_lo[NP];
            }
        }
    }
}

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

int g_num_cus = 0;
int num_cus()
{
    if (g_num_cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    return g_num_cus;
}

} // namespace

size_t front512_lds_bytes(const FrontParams &p)
{
    size_t f = 2 * 16 * kTabStride + 16 * kSplitStride;  // window pairs, pass twiddles, split twiddles
    f += (size_t)16 * p.mel_row_stride;                  // per-lane mel weights
    f += (size_t)32 * p.mel_rounds;                      // per-lane bin starts + filter ids
    f += !p.dct ? 0 : p.dct_mode == 1 ? (size_t)64 * kDctRow : (size_t)p.cols * p.dct_stride; // DCT table (either form)
    f += kWaves * 4 * kSlot;                             // 4 frame slots per wave
    f += 4;                                              // block-local work counter
    return f * sizeof(float);
}

size_t front512_delta_lds_bytes(const FrontParams &p)
{
    const size_t delta_floats = (size_t)delta_wave_lds_floats(p.dl1, p.dl2);
    size_t f = front512_lds_bytes(p) / sizeof(float);
    if (delta_floats > (size_t)4 * kSlot) f += delta_floats - 4 * kSlot; // the delta wave's region grows past its frame slots
    f += (size_t)p.done_words;
    return f * sizeof(float);
}

namespace {

template <bool A, int NM>
hipError_t launch512_delta(const FrontParams &p, hipStream_t stream)
{
    const size_t lds = front512_delta_lds_bytes(p);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_front512<A, false, NM, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_front512<A, false, NM, true>), dim3(p.n_blocks), dim3(kThreads), lds, stream, p);
    return hipGetLastError();
}

template <bool A, bool S, int NM, bool STUFF = false, bool CH2 = false>
hipError_t launch512(const FrontParams &p_in, hipStream_t stream)
{
    FrontParams p = p_in;
    if (S) { // spectrum only: no mel / DCT tables in LDS
        p.mel_rounds = 0;
        p.mel_row_stride = 0;
        p.dct = nullptr;
        p.dct_mode = 0;
    }
    const size_t lds = front512_lds_bytes(p);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_front512<A, S, NM, false, STUFF, CH2>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    // one block per CU; with fewer work items than that, one item per block (spread over the CUs: a small streaming
    // block is latency, not throughput)
    const int cap = num_cus() * (32 / kWaves) / 2; // 16 waves per CU
    int blocks = p.n_chunks < cap ? p.n_chunks : cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_front512<A, S, NM, false, STUFF, CH2>), dim3(blocks), dim3(kThreads), lds, stream, p);
    return hipGetLastError();
}

} // namespace

size_t front1024_lds_bytes(const FrontParams &p)
{
    size_t f = 2 * 16 * kTabStride + 16 * kTabStrideO + 2 * 16 * kSplitStride; // window pairs (E, O), pass twiddles, split twiddles (E, O)
    f += (size_t)16 * p.mel_row_stride + (size_t)32 * p.mel_rounds;             // per-lane mel weights, bin starts + filter ids
    f += (size_t)64 * kDctRowL;                                                 // matrix-pipe operands of the DCT
    f += (size_t)kWavesL * 4 * kSlotL + 4;                                      // 4 frame slots per wave, work counter
    return f * sizeof(float);
}

// (windows longer than 512 samples run on aligned frames only: launch_front1024 refuses the others)
bool front1024_supported(int fft_size, int window_size, int num_banks, int cols, int channels, int ceps_len)
{
    // (with a DCT at most 16 columns -- the quartered matrix-pipe form; without one the log energies of up to 80 filters)
    return fft_size == 1024 && window_size > 0 && window_size <= 1024 && channels <= 1 && num_banks >= 1 &&
           num_banks <= 4 * kDctStepsL && (ceps_len > 0 ? cols <= 16 : cols == num_banks);
}

namespace {
template <bool A, int NM>
hipError_t launch1024(const FrontParams &p, hipStream_t stream)
{
    const size_t lds = front1024_lds_bytes(p);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_front1024<A, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int blocks = (p.n_chunks + kWavesL - 1) / kWavesL;
    if (blocks > num_cus()) blocks = num_cus(); // one block of 12 waves per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_front1024<A, NM>), dim3(blocks), dim3(kWavesL * 64), lds, stream, p);
    return hipGetLastError();
}
} // namespace

hipError_t launch_front1024(const FrontParams &p, bool aligned, int nm16, hipStream_t stream)
{
    if (p.n_chunks <= 0) return hipSuccess;
    if (nm16 > 16) { // NM = rows of 16 sample pairs that carry window taps: 24 covers W <= 768, 32 the full 1024
        if (!aligned) return hipErrorInvalidValue;
        return nm16 <= 24 ? launch1024<true, 24>(p, stream) : launch1024<true, 32>(p, stream);
    }
    const bool nm13 = nm16 <= 13;
    if (aligned) return nm13 ? launch1024<true, 13>(p, stream) : launch1024<true, 16>(p, stream);
    return nm13 ? launch1024<false, 13>(p, stream) : launch1024<false, 16>(p, stream);
}

bool front512_supported(int fft_size, int window_size, int num_banks, int cols, int channels)
{
    // (256 / 128 / 64 points: the zero-stuffed forms of the same kernel, FrontParams::stuff = 512 / fft_size)
    return (fft_size == 512 || fft_size == 256 || fft_size == 128 || fft_size == 64) && window_size <= fft_size && window_size > 0 && channels <= 2 && num_banks >= 1 &&
           num_banks <= 128 && cols <= 128;
}

const char *front512_kernel_name(bool to_spectrum, bool aligned, int nm16)
{
    (void)to_spectrum;
    (void)aligned;
    (void)nm16;
    return "k_front512";
}

hipError_t launch_front512(const FrontParams &p, bool to_spectrum, bool aligned, int nm16, hipStream_t stream)
{
    if (p.n_chunks <= 0) return hipSuccess;
    // NM = number of 32-sample rows that carry window taps: 13 covers W <= 416 (25 ms at 16 kHz)
    const bool nm13 = nm16 <= 13;
    if (p.channels == 2) { // interleaved stereo (any offsets): nm16 = rows that carry taps, of 16 (stuffed) or 32 samples
        if (p.stuff) {
            if (to_spectrum) return nm13 ? launch512<true, true, 13, true, true>(p, stream) : launch512<true, true, 16, true, true>(p, stream);
            return nm13 ? launch512<true, false, 13, true, true>(p, stream) : launch512<true, false, 16, true, true>(p, stream);
        }
        if (to_spectrum) return nm13 ? launch512<true, true, 13, false, true>(p, stream) : launch512<true, true, 16, false, true>(p, stream);
        return nm13 ? launch512<true, false, 13, false, true>(p, stream) : launch512<true, false, 16, false, true>(p, stream);
    }
    if (p.stuff) { // 256-point transforms, zero-stuffed: nm16 = rows of 16 samples (200 taps: 13)
        if (to_spectrum) return nm13 ? launch512<true, true, 13, true>(p, stream) : launch512<true, true, 16, true>(p, stream);
        return nm13 ? launch512<true, false, 13, true>(p, stream) : launch512<true, false, 16, true>(p, stream);
    }
    if (to_spectrum) {
        if (aligned) return nm13 ? launch512<true, true, 13>(p, stream) : launch512<true, true, 16>(p, stream);
        return nm13 ? launch512<false, true, 13>(p, stream) : launch512<false, true, 16>(p, stream);
    }
    if (aligned) return nm13 ? launch512<true, false, 13>(p, stream) : launch512<true, false, 16>(p, stream);
    return nm13 ? launch512<false, false, 13>(p, stream) : launch512<false, false, 16>(p, stream);
}

hipError_t launch_front512_delta(const FrontParams &p, bool aligned, int nm16, hipStream_t stream)
{
    if (p.n_chunks <= 0 || p.n_blocks <= 0) return hipSuccess;
    const bool nm13 = nm16 <= 13;
    if (aligned) return nm13 ? launch512_delta<true, 13>(p, stream) : launch512_delta<true, 16>(p, stream);
    return nm13 ? launch512_delta<false, 13>(p, stream) : launch512_delta<false, 16>(p, stream);
}

namespace {

bool use_front_reg(const FrontParams &p) { return p.fft_size == 1024 || p.fft_size == 2048 || p.fft_size == 4096; }

// LDS floats of k_front_reg: shared tables + per wave one complex buffer (+ mel scratch when fused)
size_t front_reg_lds_floats(const FrontParams &p, bool fused, int n_waves)
{
    const size_t M = (size_t)p.fft_size >> 1;
    size_t f = 2 * M + 2 * (M / 2 + 2) + 2 * (size_t)reg_window_pairs(p.window_size, (int)M); // pass twiddles, split twiddles, window pairs
    if (fused) f += (size_t)mel64_rows(p.num_banks) * p.mel64_row_stride + (size_t)128 * p.mel64_rounds; // lane weight rows, starts + filter ids
#ifdef MFX_REG_PADDED
    const size_t MP = M + (M >> (p.fft_size == 1024 ? 3 : 4)); // padded buffer (pad_idx)
#else
    const size_t MP = M;
#endif
    f += (size_t)n_waves * (2 * MP + (fused ? 4 * lm_stride(p.num_banks) : 0)) + 4; // (+ the block's work counter)
    return f;
}

// waves per block of k_front_reg: as many of 16 / 8 / 4 as the CU's 160 KB of LDS allows (0: does not fit)
int front_reg_waves(const FrontParams &p, bool fused)
{
    // (4096 points: 4 waves -- the kernel is built for 256 threads there, its 32 points per lane need the registers)
    // (2048 points fused: one block of 10 waves -- tables + 12 buffers are all the LDS holds, and the throughput is flat
    // from 10 waves up: C5 0.880 / 0.887 / 0.891 ms at 10 / 11 / 12 waves, 1.04 at 9, 1.10 at 8)
    const int top = p.fft_size >= 4096 ? 4 : (p.fft_size == 2048 && fused) ? (MFX_REG10_THREADS / 64 < 10 ? MFX_REG10_THREADS / 64 : 10) : 16;
    // The block size that puts most waves on a CU (every block carries its own copy of the tables).  1024 points: at
    // 16 waves per CU two blocks of 8 beat one block of 16 (C3: 0.368 against 0.418 ms; 4 x 4: 0.431, 2 x 9 and 1 x 10
    // do not fit twice and lose) -- blocks that run out of step with each other spread their LDS phases.
    int best = 0, best_total = 0;
    for (int nw = top; nw >= 4; --nw) {
        const size_t lds = front_reg_lds_floats(p, fused, nw) * sizeof(float);
        if (lds > 160 * 1024) continue;
        const int cu_waves = p.fft_size == 2048 ? MFX_REG10_THREADS / 64 : 16; // (2048 points: one block per CU)
        int per_cu = p.fft_size <= 2048 ? (int)((160 * 1024) / lds) : 1;
        if (per_cu * nw > cu_waves) per_cu = cu_waves / nw;
        if (per_cu < 1) continue;
        const int total = per_cu * nw;
        if (total > best_total || (total == best_total && p.fft_size <= 2048 && nw >= cu_waves / 2)) {
            best_total = total;
            best = nw;
        }
    }
#ifdef MFX_REG_NW // dev builds only (tools/build_variant.sh): a fixed wave count for the A/B of block shapes
    if (MFX_REG_NW >= 4 && MFX_REG_NW <= top) best = MFX_REG_NW;
#endif
    return best;
}

template <int LOG2M, bool FUSED, bool PAIR, bool HALF>
hipError_t launch_reg_inst(const FrontParams &p, int nw, size_t lds, int blocks, hipStream_t stream)
{
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_front_reg<LOG2M, FUSED, PAIR, HALF>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((k_front_reg<LOG2M, FUSED, PAIR, HALF>), dim3(blocks), dim3(64 * nw), lds, stream, p);
    return hipGetLastError();
}

template <int LOG2M, bool FUSED>
hipError_t launch_reg(const FrontParams &p, int nw, hipStream_t stream)
{
    const size_t lds = front_reg_lds_floats(p, FUSED, nw) * sizeof(float);
    int blocks = (p.n_chunks + nw - 1) / nw;
    const int per_cu = (int)((160 * 1024) / lds);
    const int cap = num_cus() * (per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu);
    if (blocks > cap) blocks = cap;
    // the short-window builds: 1024 points with at most 512 samples (25 ms at 16 kHz zero padded to 1024: BASELINE
    // configs[2]), fused 2048 points with at most 1280 (25 ms at 44.1 kHz: configs[4])
    constexpr bool H = LOG2M == 9 || (LOG2M == 10 && FUSED);
    if (H && p.window_size <= (LOG2M == 9 ? 512 : 1280))
        return p.pair_ok ? launch_reg_inst<LOG2M, FUSED, true, H>(p, nw, lds, blocks, stream)
                         : launch_reg_inst<LOG2M, FUSED, false, H>(p, nw, lds, blocks, stream);
    return p.pair_ok ? launch_reg_inst<LOG2M, FUSED, true, false>(p, nw, lds, blocks, stream)
                     : launch_reg_inst<LOG2M, FUSED, false, false>(p, nw, lds, blocks, stream);
}

} // namespace

size_t front_wave_lds_bytes(const FrontParams &p, bool fused)
{
    if (use_front_reg(p)) {
        const int nw = front_reg_waves(p, fused);
        return nw ? front_reg_lds_floats(p, fused, nw) * sizeof(float) : (size_t)1 << 30;
    }
    const int M = p.fft_size >> 1;
    size_t f = 0;
    if (fused) f += (size_t)mel64_rows(p.num_banks) * p.mel64_row_stride + (size_t)128 * p.mel64_rounds; // lane weight rows, starts + filter ids
    f += 4 * ((size_t)4 * M + (fused ? 4 * (size_t)lm_fs4(p.num_banks) : 0));
    return f * sizeof(float);
}

hipError_t launch_front_generic(const FrontParams &p, bool fused, hipStream_t stream)
{
    if (p.n_chunks <= 0) return hipSuccess;
    if (use_front_reg(p)) { // long transforms: register-pass kernel
        const int nw = front_reg_waves(p, fused);
        if (nw == 0) return hipErrorInvalidValue;
        if (p.fft_size == 1024) return fused ? launch_reg<9, true>(p, nw, stream) : launch_reg<9, false>(p, nw, stream);
        if (p.fft_size == 2048) return fused ? launch_reg<10, true>(p, nw, stream) : launch_reg<10, false>(p, nw, stream);
        if (fused) return hipErrorInvalidValue; // callers fuse up to 2048 points only
        return launch_reg<11, false>(p, nw, stream);
    }
    const size_t lds = front_wave_lds_bytes(p, fused);
    const void *fn = fused ? (const void *)k_front_wave<true, 64> : (const void *)k_front_wave<false, 64>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int blocks = (p.n_chunks + 3) / 4;
    // persistent blocks, as many as are RESIDENT at once (registers: 6 per CU for the fused build; the LDS may allow fewer): a
    // grid of 8 per CU where 5 fit ran in two rounds -- 8 kHz / 256 points 2.46 ms per 2 M frames against 1.78 with 8 resident
    static thread_local const void *cached_fn = nullptr; // (one query per launch shape, not per launch)
    static thread_local size_t cached_lds = 0;
    static thread_local int cached_per_cu = 0;
    int per_cu = cached_per_cu;
    if (cached_fn != fn || cached_lds != lds) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = 4;
        }
        cached_fn = fn;
        cached_lds = lds;
        cached_per_cu = per_cu;
    }
    const int cap = num_cus() * (per_cu > 8 ? 8 : per_cu);
    if (blocks > cap) blocks = cap;
    if (fused)
        hipLaunchKernelGGL((k_front_wave<true, 64>), dim3(blocks), dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL((k_front_wave<false, 64>), dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError();
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
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)k_melcep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    int64_t blocks = ((p.n_rows + 3) / 4 + nw - 1) / nw;
    // persistent blocks: as many as are resident at once (registers allow 6 blocks of 4 waves; see launch_front_generic)
    static thread_local int cached_nw = -1, cached_per_cu = 0; // (one query per launch shape, not per launch)
    static thread_local size_t cached_lds = 0;
    int per_cu = cached_per_cu;
    if (cached_nw != nw || cached_lds != lds) {
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)k_melcep, 64 * nw, lds) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            per_cu = (int)std::min<size_t>(4, (160 * 1024) / lds);
        }
        cached_nw = nw;
        cached_lds = lds;
        cached_per_cu = per_cu;
    }
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
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fast16 ? (const void *)k_delta<true, kDeltaRows> : (const void *)k_delta<false, RW>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
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
