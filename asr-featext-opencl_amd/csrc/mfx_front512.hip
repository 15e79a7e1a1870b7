// mfx_front512.hip -- the two register kernels on 16 lanes per frame, 4 frames per wave iteration:
//   k_front512   512-point transforms (BASELINE configs[0], [1], [3]) and, zero-stuffed, 256 / 128 / 64 points; stereo builds
//   k_front1024  1024-point transforms of a short window on the same core (BASELINE configs[2])
// and their launchers.  Replaces segmenter.cl kernelSegmentWindow + AppleFFT + mfcc.cl kernelTranspose / kernelFilter + the DCT
// slot (mfccopencl.cpp:315-358).  Numerics follow the reference CPU path (mfcccpu.cpp): frames = window * int16 sample (one
// rounding), unnormalised forward DFT, magnitude / W2, two-row triangular mel table walked in ascending bin order,
// log(max(., 1e-30)), DCT as a k-ordered dot product.  See DESIGN.md section 5.
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
    const int block_id = xcd_block_id(); // (consecutive ids, i.e. consecutive chunks, on one XCD's L2)
    auto next_index = [&]() -> int {
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return k;
    };
    auto chunk_of = [&](int k) -> int {
        if (FUSE) return k < chunk_cnt ? chunk_base + k : p.n_chunks; // the block's own contiguous list, in order
        const long long c = (long long)block_id + (long long)k * gridDim.x;
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
// pair).  NO magnitude waits in registers (round 4: that is what lets the aligned builds run 4 waves per SIMD): phase E runs
// first and stores its magnitudes straight into the slot; phase O re-converts the same raw words, transposes BESIDE the E
// magnitudes one component at a time, and stores its own.  Magnitudes lie de-interleaved in the frame's slot (E[i] = bin 2i,
// O[i] = bin 2i+1); the mel walk reads two bins of each per 8-byte read and adds them in ascending bin order
// (mfcccpu.cpp:192-220), keeps its log energies in registers and lays them over the E magnitudes when every lane has finished
// reading; DCT on the matrix pipe as in k_front512 (20 K steps: at most 80 filters).
// ------------------------------------------------------------------------------------------------
constexpr int kSlotL = 544;       // dwords per frame slot: E magnitudes [0, 264) | O magnitudes [264, 528).  Phase E transposes through
                                  // [0, 512) before any magnitude lands; phase O, with the E magnitudes in place, through
                                  // [272, 528) one component at a time; the log energies overlay the E magnitudes once the mel
                                  // walk is over (544 = 32 mod 64: neighbouring slots sit on complementary bank halves)
constexpr int kOddOffL = 264;     // O magnitudes inside the slot
constexpr int kTrOffL = 272;      // phase O's transposition (256 dwords)
constexpr int kMelOffL = 0;       // log energies (frame `slot` staggered by 8 slot words)
constexpr int kTabStrideO = 68;   // dwords per lane row of the phase-O window table (16 x (A, B, C, D) + pad: 17 16-byte words)
constexpr int kDctStepsL = 20;    // num_banks <= 80
constexpr int kDctRowL = 20;      // dwords per lane row of the B operand table (5 16-byte words: odd)

// WAVES per block = per CU: 16 (4 per SIMD, 128 registers: the aligned builds of windows up to 512 samples, when the tables
// leave room for 16 x 4 slots) or 12 (3 per SIMD, 168 registers: every other build)
template <bool ALIGNED, int NM, int WAVES>
__global__ void __launch_bounds__(WAVES * 64, WAVES / 4) k_front1024(FrontParams p)
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
    int *s_ctr = (int *)(s_dct + 64 * kDctRowL + WAVES * (4 * kSlotL));
    if (tid == 0) *s_ctr = 0;

    for (int i = tid; i < 256; i += WAVES * 64) { // HBM tables are [lane][m]
        ((float2 *)(s_win + (i >> 4) * kTabStride))[i & 15] = ((const float2 *)p.winpair)[i];
        ((float4 *)(s_winO + (i >> 4) * kTabStrideO))[i & 15] = ((const float4 *)p.win1024o)[i];
        ((float2 *)(s_tw + (i >> 4) * kTabStride))[i & 15] = ((const float2 *)p.twid_pass)[i];
    }
    for (int i = tid; i < 128; i += WAVES * 64) { // twid_split holds -i W_1024^e, e <= 512, in natural order
        ((float2 *)(s_splitE + (i & 15) * kSplitStride))[i >> 4] = ((const float2 *)p.twid_split)[2 * i];
        ((float2 *)(s_splitO + (i & 15) * kSplitStride))[i >> 4] = ((const float2 *)p.twid_split)[2 * i + 1];
    }
    for (int i = tid; i < 16 * RS; i += WAVES * 64) s_melw[i] = p.mel_lane_w[i];
    for (int i = tid; i < 16 * rounds; i += WAVES * 64) {
        s_mmeta[2 * i] = p.mel_lane_start[i] >> 1; // (index into the even / odd magnitude arrays)
        const int fid = p.mel_lane_fid[i];
        s_mmeta[2 * i + 1] = fid < 0 ? 4 * kDctStepsL : fid; // idle lanes park their value in a word nobody reads
    }
    for (int i = tid; i < 64 * kDctRowL; i += WAVES * 64) {
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
    const int block_id = xcd_block_id(); // (consecutive ids, i.e. consecutive chunks, on one XCD's L2)
    auto next_index = [&]() -> int {
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return k;
    };
    auto chunk_of = [&](int k) -> int {
        const long long c = (long long)block_id + (long long)k * gridDim.x;
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
            // (`fresh`: the second phase re-reads the raw word opaquely and converts it again -- 26 floats kept alive across a
            // phase cost more registers than the conversions cost issue slots)
            auto pair_of = [&](int m, float &x0, float &x1, bool fresh) {
                uint32_t d = ALIGNED ? cur.d[m] : dd[m];
                if (fresh) asm volatile("" : "+v"(d));
                x0 = (float)(int)(short)(d & 0xffffu);
                x1 = (float)((int)d >> 16);
            };
            // pass A + inter-pass twiddles of the 256-point transform; the 16 x 16 transposition through the slot and pass B
            // follow in each phase (a[pp] = FFT256(a)[l + 16 pp] after them)
            auto fft256_head = [&](float2(&a)[16]) {
                fft16(a);
                float4 tq[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) tq[k] = lds_read_b128((const float4 *)(s_tw + l * kTabStride) + k);
#pragma unroll
                for (int k = 0; k < 16; k += 2) {
                    const float4 t = tq[k >> 1];
                    if (k > 0) a[k] = cmul(a[k], make_float2(t.x, t.y));
                    a[k + 1] = cmul(a[k + 1], make_float2(t.z, t.w));
                }
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

            // ---- phase E: even bins = k_front512's transform of the same samples; the magnitudes go straight to the slot
            // (E[i] = |X[2 i]|, i <= 256: the transposition's words are consumed by then)
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
                        a[m] = folded(m, true, false);
                    } else if (m < NM) {
                        float x0, x1;
                        pair_of(m, x0, x1, false);
                        const float2 w = (m & 1) ? make_float2(wq[m >> 1].z, wq[m >> 1].w) : make_float2(wq[m >> 1].x, wq[m >> 1].y);
                        a[m] = make_float2(w.x * x0, w.y * x1);
                    } else {
                        a[m] = make_float2(0.f, 0.f);
                    }
                }
                fft256_head(a);
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
                const float m128r = a[8].x + a[8].x, m128i = a[8].y + a[8].y;
                if (l == 0) xb[128] = __builtin_amdgcn_sqrtf(m128r * m128r + m128i * m128i);
                float4 csq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) csq[j] = ((const float4 *)(s_splitE + l * kSplitStride))[j];
                float *elo = xb + l, *ehi = xb + (144 - l); // i = l + 16 p and 256 - l - 16 p
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
                    elo[16 * pp] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi); // the window taps carry 0.5 / W2
                    ehi[16 * (7 - pp)] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
                }
            }

            // ---- phase O: odd bins.  The E magnitudes already sit in [0, 264) of the slot, so this transposition goes through
            // [kTrOffL, kTrOffL + 256) one component at a time: word (row k, column c) of a component at 16 k + (c ^ 4 (k >> 2))
            // (4-byte stores of a row land on 16 consecutive banks; the 16-byte reads of rows l fall on distinct bank quads)
            {
                float2 a[16];
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    if (FULL) {
                        a[m] = cmul(folded(m, false, true), ((const float2 *)(s_win + l * kTabStride))[m]); // x W_512^(l + 16 m)
                    } else if (m < NM) {
                        float x0, x1;
                        pair_of(m, x0, x1, true);
                        const float4 t = ((const float4 *)(s_winO + l * kTabStrideO))[m];
                        a[m] = make_float2(t.x * x0 + t.y * x1, t.z * x0 + t.w * x1);
                    } else {
                        a[m] = make_float2(0.f, 0.f);
                    }
                }
                // the raw words are consumed: prefetch the next 4 frames (of this chunk, or the first of the next chunk)
                if (ALIGNED) pcm_issue<ALIGNED, NM>(cur, last ? cnxt.rsrc : ccur.rsrc, last ? lane_off(cnxt, slot) : lane_off(ccur, f + 4));
                fft256_head(a);
                float *xt = xb + kTrOffL;
                const float4 *xrow = (const float4 *)(xt + l * 16);
#pragma unroll
                for (int k = 0; k < 16; ++k) xt[k * 16 + (l ^ (4 * (k >> 2)))] = a[k].x;
                wave_sync();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = xrow[j ^ (l >> 2)];
                    a[4 * j].x = v.x;
                    a[4 * j + 1].x = v.y;
                    a[4 * j + 2].x = v.z;
                    a[4 * j + 3].x = v.w;
                }
                wave_sync();
#pragma unroll
                for (int k = 0; k < 16; ++k) xt[k * 16 + (l ^ (4 * (k >> 2)))] = a[k].y;
                wave_sync();
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 v = xrow[j ^ (l >> 2)];
                    a[4 * j].y = v.x;
                    a[4 * j + 1].y = v.y;
                    a[4 * j + 2].y = v.z;
                    a[4 * j + 3].y = v.w;
                }
                wave_sync();
                fft16(a);
                float4 csq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) csq[j] = ((const float4 *)(s_splitO + l * kSplitStride))[j];
                float *olo = xb + kOddOffL + l, *ohi = xb + kOddOffL + (143 - l); // O[i] = |X[2 i + 1]|: i = l + 16 p and 255 - l - 16 p
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
                    olo[16 * pp] = __builtin_amdgcn_sqrtf(xr * xr + xi * xi);
                    ohi[16 * (7 - pp)] = __builtin_amdgcn_sqrtf(yr * yr + yi * yi);
                }
            }
            wave_sync();

            // ---- mel filterbank: per round every lane walks one filter's bins in ascending order
            // (mfcccpu.cpp:192-220); starts are multiples of 4 bins: two even and two odd bins per 8-byte read.  The log
            // energies wait in registers (at most 5 rounds) and overlay the magnitudes once every lane has finished reading.
            const float *wrow = s_melw + l * RS;
            float *dst = p.feat + (out_row + f0) * (int64_t)p.feat_pitch + slot * p.feat_pitch;
            float *lm = xb + kMelOffL + 8 * slot;
            const int2 *mmeta = (const int2 *)s_mmeta + l;
            float le[kDctStepsL / 4];
#pragma unroll
            for (int r = 0; r < kDctStepsL / 4; ++r) {
                le[r] = 0.f;
                if (r < rounds) {
                    const int L = p.mel_L[r];
                    const int first = mmeta[16 * r].x;
                    const float *me = xb + first, *mo = xb + kOddOffL + first;
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
                    le[r] = MFX_LOG(fmaxf(acc, 1e-30f));
                }
            }
            wave_sync();
#pragma unroll
            for (int r = 0; r < kDctStepsL / 4; ++r)
                if (r < rounds) lm[mmeta[16 * r].y] = le[r]; // (idle lanes: the filter id names a word nobody reads)
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
    if (hipError_t e = allow_dynamic_lds((const void *)k_front512<A, false, NM, true>, lds); e != hipSuccess) return e;
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
    if (hipError_t e = allow_dynamic_lds((const void *)k_front512<A, S, NM, false, STUFF, CH2>, lds); e != hipSuccess) return e;
    // one block per CU; with fewer work items than that, one item per block (spread over the CUs: a small streaming
    // block is latency, not throughput)
    const int cap = num_cus() * (32 / kWaves) / 2; // 16 waves per CU
    int blocks = p.n_chunks < cap ? p.n_chunks : cap;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_front512<A, S, NM, false, STUFF, CH2>), dim3(blocks), dim3(kThreads), lds, stream, p);
    return hipGetLastError();
}

} // namespace

size_t front1024_lds_bytes(const FrontParams &p, int waves)
{
    size_t f = 2 * 16 * kTabStride + 16 * kTabStrideO + 2 * 16 * kSplitStride; // window pairs (E, O), pass twiddles, split twiddles (E, O)
    f += (size_t)16 * p.mel_row_stride + (size_t)32 * p.mel_rounds;             // per-lane mel weights, bin starts + filter ids
    f += (size_t)64 * kDctRowL;                                                 // matrix-pipe operands of the DCT
    f += (size_t)waves * 4 * kSlotL + 4;                                        // 4 frame slots per wave, work counter
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
template <bool A, int NM, int WAVES>
hipError_t launch1024(const FrontParams &p, hipStream_t stream)
{
    const size_t lds = front1024_lds_bytes(p, WAVES);
    if (hipError_t e = allow_dynamic_lds((const void *)k_front1024<A, NM, WAVES>, lds); e != hipSuccess) return e;
    int blocks = (p.n_chunks + WAVES - 1) / WAVES;
    if (blocks > num_cus()) blocks = num_cus(); // one block per CU
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL((k_front1024<A, NM, WAVES>), dim3(blocks), dim3(WAVES * 64), lds, stream, p);
    return hipGetLastError();
}
} // namespace

int front1024_waves(const FrontParams &p, bool aligned, int nm16, int max_waves)
{
    return max_waves >= 16 && aligned && nm16 <= 16 && front1024_lds_bytes(p, 16) <= (size_t)160 * 1024 ? 16 : 12;
}

hipError_t launch_front1024(const FrontParams &p, bool aligned, int nm16, hipStream_t stream, int max_waves)
{
    if (p.n_chunks <= 0) return hipSuccess;
    if (nm16 > 16) { // NM = rows of 16 sample pairs that carry window taps: 24 covers W <= 768, 32 the full 1024
        if (!aligned) return hipErrorInvalidValue;
        return nm16 <= 24 ? launch1024<true, 24, 12>(p, stream) : launch1024<true, 32, 12>(p, stream);
    }
    const bool nm13 = nm16 <= 13;
    if (front1024_waves(p, aligned, nm16, max_waves) == 16) return nm13 ? launch1024<true, 13, 16>(p, stream) : launch1024<true, 16, 16>(p, stream);
    if (aligned) return nm13 ? launch1024<true, 13, 12>(p, stream) : launch1024<true, 16, 12>(p, stream);
    return nm13 ? launch1024<false, 13, 12>(p, stream) : launch1024<false, 16, 12>(p, stream);
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

} // namespace mfx
