// mfx_front2048.hip -- k_front2048: the fused front end for 2048-point transforms of a short window
// (BASELINE configs[4]: 44.1 kHz, 25 ms = 1102 samples zero padded to 2048, stereo or mono, 128 mel, 40 MFCC).
//
// Replaces, for that shape, the reference stages segmenter.cl kernelSegmentWindow + AppleFFT (one LDS kernel up to
// 2048 points, AppleFFT/fft_kernelstring.cpp:165-177) + mfcc.cl kernelTranspose / kernelFilter + the DCT slot
// (mfccopencl.cpp:315-358); numerics follow mfcccpu.cpp:192-232.
//
// Shape of the work (round 3; k_front_reg ran one WAVE per frame with three Stockham passes and ~11 dependent LDS round
// trips per frame -- DESIGN.md section 7 found it bound by exactly that):
//   * TWO frames per wave, 32 lanes per frame.  The real 2048-point DFT of a frame is the complex 1024-point DFT of
//     z[n] = x[2n] + i x[2n+1] plus a real split; 1024 = 32 x 32:
//        lane n1 : Y[n1][k2] = sum_n2 z[n1 + 32 n2] W_32^(n2 k2)    32-point DFT in registers (only n2 < 18 carry taps:
//                                                                    the zero inputs fold away at compile time)
//        twiddle : Y *= W_1024^(n1 k2)                               table in LDS, two values per 16-byte read
//        LDS     : 32 x 32 transposition, real parts then imaginary parts through ONE 4 KB plane per frame
//                  (16-byte writes, 4-byte reads, XOR-swizzled: both conflict free)
//        lane k2 : Z[k2 + 32 k1] = sum_n1 Y[n1][k2] W_32^(n1 k1)    second 32-point DFT in registers
//     i.e. two passes and ONE transposition where the 16.16.4 factorisation had three passes and three.
//   * real split over the bin pairs (k, 1024 - k): lane l holds k = l + 32 k1 (k1 < 16) in its lower registers; the
//     partner Z[1024 - k] is register 31 - k1 of lane 32 - l of the same frame: 32 ds_bpermute (no LDS memory), all in
//     flight at once.  |S + T| and |S - T| are the two magnitudes (as k_front512).
//   * magnitudes back into the frame's plane, mel filters walked on the frame's 32 lanes (rounds of 32 filters, longest
//     first, one ascending chain of multiply-adds per filter = the reference's order, mfcccpu.cpp:206-217), log, and the
//     DCT-II + lifter on the matrix pipe once per 4 frames (v_mfma_f32_16x16x4_f32: an exact k-ordered fmaf chain).
//   * the samples of the NEXT two frames are requested right after the current ones are converted (registers just
//     freed), chunks are drawn from a block-local LDS counter, PCM goes through buffer loads with a per-chunk descriptor
//     (hardware range check).
// A frame costs ~5 dependent LDS round trips per PAIR of frames, and with 12 waves per CU 24 frames are in flight per CU
// (k_front_reg: 10).
#include "mfx_kernels.h"

#include <hip/hip_runtime.h>

#include "mfx_dev.h"
#include "mfx_launch.h"

namespace mfx {

namespace {

// Dev-only in-kernel stamps (-DMFX_STAMPS): per-wave cycle sums per phase, written by lane 0 to p.spec (unused by the
// fused path); tools/stamps2048.py reads them.  Never part of a timed build.
#ifdef MFX_STAMPS
#define MFX_STAMP2(i)                                                                  \
    do {                                                                               \
        unsigned long long t_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                             \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");     \
        __builtin_amdgcn_sched_barrier(0);                                             \
        st_acc[i] += t_ - st_last;                                                     \
        st_last = t_;                                                                  \
    } while (0)
#else
#define MFX_STAMP2(i)
#endif

#ifndef MFX_W2048
#define MFX_W2048 12
#endif
constexpr int kW2048 = MFX_W2048;        // waves per block; one block per CU (LDS: ~33 KB of tables + 10.5 KB per wave)
constexpr int kPlane = 1040;      // floats per frame plane: 1024 transposition words / 1025 magnitudes + finite slack
constexpr int kRows2048 = 18;     // rows of 32 sample pairs that can carry window taps: W <= 1152 (25 ms at 44.1 kHz)
constexpr int kRows2048L = 20;    // the long-window build: W <= 1280 (25 ms at 48 kHz = 1200 taps)
constexpr int kRows2048F = 32;    // the full-window build: W <= 2048 (n_fft = win_length = 2048, the audio-analysis default)
__host__ __device__ inline int rows2048(int window_size)
{
    return window_size <= 64 * kRows2048 ? kRows2048 : window_size <= 64 * kRows2048L ? kRows2048L : kRows2048F;
}

// cos / sin of 2 pi e / 32
__host__ __device__ constexpr float c32(int e)
{
    constexpr float q[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                            0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                            0.19509032201612826785f, 0.0f};
    e &= 31;
    return e <= 8 ? q[e] : e <= 16 ? -q[16 - e] : e <= 24 ? -q[e - 16] : q[32 - e];
}
__host__ __device__ constexpr float s32(int e) { return c32(e - 8); }

// a * W_32^E, constants as literals ("x * K + y * K2": v_mul + v_fmac with literal operands, full issue rate)
template <int E>
__device__ __forceinline__ float2 mul_w32(float2 a)
{
    constexpr int e = E & 31;
    if (e == 0) return a;
    if (e == 8) return make_float2(a.y, -a.x);
    if (e == 16) return make_float2(-a.x, -a.y);
    if (e == 24) return make_float2(-a.y, a.x);
    constexpr float c = c32(e), s = s32(e), ns = -s32(e);
    return make_float2(a.x * c + a.y * s, a.y * c + a.x * ns); // (c - i s)(x + i y)
}

// 32-point forward DFT in registers, natural order in and out: n = 8 a + b, k = c + 4 d,
//   X[c + 4 d] = sum_b W_8^(b d) [ W_32^(b c) sum_a x[8 a + b] W_4^(a c) ].
// Inputs that are compile-time zeros fold away after inlining (pass 1: only x[0..17] carry data).
__device__ __forceinline__ void fft32(float2 (&x)[32])
{
    float2 y[32]; // y[8 c + b]
#pragma unroll
    for (int b = 0; b < 8; ++b) dft4(x[b], x[8 + b], x[16 + b], x[24 + b], y[b], y[8 + b], y[16 + b], y[24 + b]);
#define MFX_TW32(c, b) y[8 * c + b] = mul_w32<c * b>(y[8 * c + b])
    MFX_TW32(1, 1); MFX_TW32(1, 2); MFX_TW32(1, 3); MFX_TW32(1, 4); MFX_TW32(1, 5); MFX_TW32(1, 6); MFX_TW32(1, 7);
    MFX_TW32(2, 1); MFX_TW32(2, 2); MFX_TW32(2, 3); MFX_TW32(2, 4); MFX_TW32(2, 5); MFX_TW32(2, 6); MFX_TW32(2, 7);
    MFX_TW32(3, 1); MFX_TW32(3, 2); MFX_TW32(3, 3); MFX_TW32(3, 4); MFX_TW32(3, 5); MFX_TW32(3, 6); MFX_TW32(3, 7);
#undef MFX_TW32
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float2 t[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) t[b] = y[8 * c + b];
        fft8(t);
#pragma unroll
        for (int d = 0; d < 8; ++d) x[c + 4 * d] = t[d];
    }
}

// LDS floats: shared tables, then per wave two planes and the log mel energies of 4 frames
__host__ __device__ inline size_t front2048_table_floats(int rounds, int row_stride, int rows)
{
    return 2 * (size_t)(32 * rows)           // window pairs
           + 4 * (size_t)(16 * 32)            // pass twiddles, two per 16-byte word
           + 4 * (size_t)(8 * 32) + 4         // split twiddles, two per 16-byte word; + the self-paired bin 512
           + (size_t)32 * row_stride          // mel weight rows
           + (size_t)64 * rounds;             // starts + filter ids
}

// SPLIT: the DCT in its split form (p.dct_split != 0: <= 40 columns, bands a multiple of 32) -- a build of its own, so that
// neither form carries the other's code and scalar registers
// NR: rows of 32 sample pairs that carry window taps (18: W <= 1152; 20: W <= 1280; 32: any window) -- the zero rows fold away at compile time
// CH: 0 = mono, frames on aligned sample pairs (even offsets, even shift: one 32-bit word per pair); 1 = interleaved stereo (one
// word per sample, any alignment); 2 = mono at ANY alignment (odd shifts -- 441 samples = 10 ms at 44.1 kHz -- or odd
// offsets): the two aligned words that cover a pair are loaded and funnel-shifted by the frame's parity
template <int CH, bool SPLIT, int NR>
__global__ void __launch_bounds__(kW2048 * 64, (kW2048 + 3) / 4) k_front2048(FrontParams p)
{
    constexpr int M = 1024;
    constexpr bool STEREO = CH == 1, ANY = CH == 2;
    // depth of the DCT's operand ring: what the build's registers allow without a spill
    constexpr int kRingSplit = ((STEREO || ANY) && NR >= kRows2048F) ? 3 : (ANY || (STEREO && NR > kRows2048)) ? 5 : 7;
    constexpr int kRingTile = ((STEREO || ANY) && NR >= kRows2048F) ? 3 : ANY ? 5 : 7;
    constexpr int NWORD = (STEREO || ANY) ? 2 * NR : NR; // raw 32-bit words per lane and frame
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n_waves = blockDim.x >> 6;
    const int l = lane & 31, half = lane >> 5;
    const int nb = p.num_banks;
    const int RS = p.mel32_row_stride, rounds = p.mel32_rounds;
    const int lmFS = lm_fs4(nb);
    const int lm_wave = 4 * lmFS; // floats of the 4 waiting frames

    float2 *s_win = (float2 *)smem;                                // [18][32] (w[2n], w[2n+1]) * 0.5 / W2
    float4 *s_tw = (float4 *)(s_win + 32 * NR);                    // [16][32] (W_1024^(n1 2j), W_1024^(n1 (2j+1)))
    float4 *s_cs = s_tw + 16 * 32;                                 // [8][32]  (cs[l + 64 j], cs[l + 64 j + 32]), cs[k] = -i W_2048^k
    float2 *s_cs512 = (float2 *)(s_cs + 8 * 32);                   // cs[512] (+ pad)
    float *s_mw = (float *)(s_cs512 + 2);                          // [32][RS]
    int *s_mst = (int *)(s_mw + 32 * RS);                          // [rounds][32]
    int *s_mfid = s_mst + 32 * rounds;                             // [rounds][32]
    float *s_wave = (float *)(s_mfid + 32 * rounds) + wave * (2 * kPlane + lm_wave);
    float *plane = s_wave + half * kPlane;                         // this frame's plane
    float *lm = s_wave + 2 * kPlane;                               // [4][lmFS] (lm_fs4)
    int *s_ctr = (int *)((float *)(s_mfid + 32 * rounds) + n_waves * (2 * kPlane + lm_wave));
    if (tid == 0) *s_ctr = 0;

    const float scale = p.scale; // 0.5 / W2 (a power of two: exact)
    for (int i = tid; i < 32 * NR; i += blockDim.x) {
        const float2 wv = 2 * i < p.fft_size ? ((const float2 *)p.window)[i] : make_float2(0.f, 0.f); // (zero padded to W2)
        s_win[i] = make_float2(wv.x * scale, wv.y * scale);
    }
    for (int i = tid; i < 16 * 32; i += blockDim.x) {
        const int j = i >> 5, n1 = i & 31;
        const float2 a = ((const float2 *)p.twid_half)[(n1 * 2 * j) & (M - 1)];
        const float2 b = ((const float2 *)p.twid_half)[(n1 * (2 * j + 1)) & (M - 1)];
        s_tw[i] = make_float4(a.x, a.y, b.x, b.y);
    }
    for (int i = tid; i < 8 * 32; i += blockDim.x) {
        const int j = i >> 5, ll = i & 31;
        const float2 a = ((const float2 *)p.twid_split)[ll + 64 * j];
        const float2 b = ((const float2 *)p.twid_split)[ll + 64 * j + 32];
        s_cs[i] = make_float4(a.x, a.y, b.x, b.y);
    }
    if (tid < 2) s_cs512[tid] = tid == 0 ? ((const float2 *)p.twid_split)[512] : make_float2(0.f, 0.f);
    for (int i = tid; i < 32 * RS; i += blockDim.x) s_mw[i] = p.mel32_w[i];
    for (int i = tid; i < 32 * rounds; i += blockDim.x) {
        s_mst[i] = p.mel32_start[i];
        s_mfid[i] = p.mel32_fid[i];
    }
    for (int i = lane; i < 2 * kPlane + lm_wave; i += 64) s_wave[i] = 0.f; // (words the walk may read but nothing writes)
    __syncthreads();

    // ---- per-lane constants
    // transposition plane, value (n1, k2) at word n1 * 32 + (((k2 >> 2) ^ n1) & 7) * 4 + (k2 & 3):
    //   lane n1 writes its 8 chunks of 4 consecutive k2 as 16-byte words (8 consecutive lanes: 8 distinct bank quads),
    //   lane k2 reads word (n1, k2) for n1 = 0..31 (32 lanes: a permutation of one 32-word row)
    int wr_off[8], rd_off[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        wr_off[j] = l * 32 + ((j ^ l) & 7) * 4;
        rd_off[j] = j * 32 + ((((l >> 2) ^ j) & 7) * 4) + (l & 3); // row n1 = j + 8 m: + 256 m words
    }
    const int part_addr = (((32 - l) & 31) | (lane & 32)) * 4; // ds_bpermute: partner lane of the real split
    const bool lane0 = l == 0;

    // ---- chunk walk: block b owns chunks b, b + B, ...; its waves draw from that list through a counter in LDS
    struct ChunkCtx {
        int64_t out_row;
        int n_live;
        int odd0; // ANY: the chunk starts on an odd sample (its descriptor on the even one before)
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
        // buffer descriptor over [chunk start, end of PCM): out-of-range lanes read 0 (whole 32-bit words: see k_front512)
        x.odd0 = ANY ? (int)(pcm_off & 1) : 0;
        const int64_t el0 = STEREO ? pcm_off * 2 : ANY ? (pcm_off & ~(int64_t)1) : pcm_off;
        int64_t bytes_left = valid ? (((p.pcm_total - el0) * 2 + 3) & ~(int64_t)3) : 0;
        if (bytes_left > 0xfffffff0ll) bytes_left = 0xfffffff0ll;
        if (bytes_left < 0) bytes_left = 0;
        const uintptr_t bp = (uintptr_t)(p.pcm + el0);
        const uint32_t bp_lo = __builtin_amdgcn_readfirstlane((uint32_t)bp);
        const uint32_t bp_hi = __builtin_amdgcn_readfirstlane((uint32_t)(bp >> 32));
        const uint32_t nbytes = __builtin_amdgcn_readfirstlane((uint32_t)bytes_left);
        x.rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(((uintptr_t)bp_hi << 32) | bp_lo), 0, nbytes, 0x00020000);
        return x;
    };
    const int block_id = xcd_block_id(); // (consecutive ids, i.e. consecutive chunks, on one XCD's L2)
    auto draw = [&]() -> int {
        int k = 0;
        if (lane == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        k = __builtin_amdgcn_readfirstlane(k);
        const long long cc = (long long)block_id + (long long)k * gridDim.x;
        return cc < p.n_chunks ? (int)cc : p.n_chunks;
    };
    // raw words of (frame f, this lane): sample pair n = l + 32 j at byte (f S + 2 n) * (STEREO ? 4 : 2)
    uint32_t raw[NWORD];
    auto issue = [&](const ChunkCtx &x, int f) {
        // (ANY: the frame's first sample s = odd0 + f S rounded down to an even one; the pair's two words from there)
        const int voff = ANY ? (((x.odd0 + f * p.shift) & ~1) + 2 * l) * 2 : (f * p.shift + 2 * l) * (STEREO ? 4 : 2);
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            // (the row's constant goes in as the scalar offset: added to the lane's offset by the address unit, no vector add)
            if (STEREO || ANY) {
                const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(x.rsrc, voff, (STEREO ? 256 : 128) * j, 0);
                raw[2 * j] = d[0];
                raw[2 * j + 1] = d[1];
            } else {
                raw[j] = __builtin_amdgcn_raw_buffer_load_b32(x.rsrc, voff, 128 * j, 0);
            }
        }
    };

#ifdef MFX_STAMPS
    unsigned long long st_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_last, st_rt0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt0)::"memory");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    // Stagger: the waves of a block run the same program and, started together, stay in step -- all of a SIMD's waves in
    // the FFT (vector pipe), then all in the mel walk (LDS), then all in the DCT (matrix pipe), each phase taking three
    // times what one wave needs (in-kernel stamps: 6.5 cycles per vector instruction in pass 1, 108 per matrix
    // instruction).  Waves w, w + 4, w + 8 share a SIMD: the second and third group start a third / two thirds of an
    // iteration late, so that one SIMD's waves are in different phases.
#ifndef MFX_STAGGER2048
#define MFX_STAGGER2048 0
#endif
    if (MFX_STAGGER2048 > 0) {
        const int naps = (wave >> 2) * (MFX_STAGGER2048 / 64 / 100);
        for (int i = 0; i < naps; ++i) __builtin_amdgcn_s_sleep(100); // 100 x 64 cycles
    }
    // the DCT's B operands [tiles][ksteps][64] through a buffer descriptor (offsets past the table return 0)
    const int dct_ks = p.dct_ksteps, dct_tiles64 = (p.dct_len + 63) >> 6; // bands / 4; groups of 64 output columns
    // (the split form: nb / 8 K-groups of pass A, then nb / 32 of pass B, 1 KB each; ONE descriptor for whichever table this
    // launch uses -- a second one costs scalar registers the kernel does not have)
    const int dct_bytes = SPLIT      ? ((p.num_banks >> 3) + (p.dct_split > 1 ? (p.num_banks >> 5) : 0)) * 1024
                          : p.dct_b4 ? dct_tiles64 * dct_ks * 1024
                                     : 0;
    const __amdgpu_buffer_rsrc_t dct_rsrc =
        __builtin_amdgcn_make_buffer_rsrc((void *)(SPLIT ? p.dct_b4s : p.dct_b4), 0, dct_bytes, 0x00020000);
    int c_cur = draw(), c_nxt = draw();
    ChunkCtx ccur = make_ctx(c_cur), cnxt = make_ctx(c_nxt);
    issue(ccur, half);
    while (c_cur < p.n_chunks) {
        const int64_t out_row = ccur.out_row;
        const int n_live = ccur.n_live;
        for (int f0 = 0; f0 < n_live; f0 += 2) {
            const int f = f0 + half;
            const bool last = f0 + 2 >= n_live;
            MFX_STAMP2(0);
#ifdef MFX_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (stamp builds only: the wait for the prefetched samples on its own)
            MFX_STAMP2(12);
#endif
            // ---- framing + window: z[n2] = (w[2n] x[2n], w[2n+1] x[2n+1]), n = l + 32 n2
            float2 z[32];
            // (ANY: the frame starts on an odd sample: its pair is the upper half of the first word and the lower half of the second)
            const uint32_t par_shift = ANY ? (uint32_t)(((ccur.odd0 + f * p.shift) & 1) << 4) : 0u;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                if (j >= NR) {
                    z[j] = make_float2(0.f, 0.f);
                    continue;
                }
                float x0, x1;
                if (ANY) {
                    const uint32_t d = __builtin_amdgcn_alignbit(raw[2 * j + 1], raw[2 * j], par_shift);
                    x0 = (float)(int)(short)(d & 0xffffu);
                    x1 = (float)((int)d >> 16);
                } else if (STEREO) { // one 32-bit word per sample (L | R << 16); mono = (L + R) >> 1 as the reference driver's caller
                    const uint32_t d0 = raw[2 * j], d1 = raw[2 * j + 1];
                    x0 = stereo_mean(d0);
                    x1 = stereo_mean(d1);
                } else {
                    const uint32_t d = raw[j];
                    x0 = (float)(int)(short)(d & 0xffffu);
                    x1 = (float)((int)d >> 16);
                }
                const float2 w = s_win[l + 32 * j];
                z[j] = make_float2(w.x * x0, w.y * x1);
            }
            MFX_STAMP2(1);
            // ---- pass 1 + inter-pass twiddle
            fft32(z);
#pragma unroll
            for (int j0 = 0; j0 < 16; j0 += 4) { // (4 table words in flight at a time: 16 registers, not 64)
                float4 tq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) tq[j] = lds_read_b128(s_tw + (j0 + j) * 32 + l);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j0 + j > 0) z[2 * (j0 + j)] = cmul(z[2 * (j0 + j)], make_float2(tq[j].x, tq[j].y));
                    z[2 * (j0 + j) + 1] = cmul(z[2 * (j0 + j) + 1], make_float2(tq[j].z, tq[j].w));
                }
            }
            // ---- 32 x 32 transposition through the frame's plane: real parts, then imaginary parts.  The LDS executes a
            // wave's instructions in order, so the imaginary parts' writes need not wait for the real parts' reads.
            MFX_STAMP2(2);
            float zr[32], zi[32];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *(float4 *)(plane + wr_off[j]) = make_float4(z[4 * j].x, z[4 * j + 1].x, z[4 * j + 2].x, z[4 * j + 3].x);
            wave_sync();
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) zr[n1] = plane[rd_off[n1 & 7] + 256 * (n1 >> 3)];
            wave_sync();
#pragma unroll
            for (int j = 0; j < 8; ++j)
                *(float4 *)(plane + wr_off[j]) = make_float4(z[4 * j].y, z[4 * j + 1].y, z[4 * j + 2].y, z[4 * j + 3].y);
            wave_sync();
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) zi[n1] = plane[rd_off[n1 & 7] + 256 * (n1 >> 3)];
            wave_sync();
#pragma unroll
            for (int n1 = 0; n1 < 32; ++n1) z[n1] = make_float2(zr[n1], zi[n1]);

            MFX_STAMP2(3);
            // ---- pass 2: lane k2 = l now holds Z[l + 32 k1] in z[k1]
            fft32(z);
            MFX_STAMP2(4);

            // ---- real split over the pairs (k, 1024 - k), k = l + 32 k1, k1 < 16.  Z[1024 - k] is register 31 - k1 of
            // lane 32 - l; lane 0 pairs with itself: register 32 - k1 (and Z[0] with itself), so it sends its registers
            // shifted by one.
            // The magnitudes go straight into the frame's plane (every transposition word has been read), 8 pairs at a time.
#pragma unroll
            for (int h8 = 0; h8 < 2; ++h8) {
                float2 pz[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int j = 31 - (8 * h8 + i);
                    const float sx = lane0 ? z[(j + 1) & 31].x : z[j].x;
                    const float sy = lane0 ? z[(j + 1) & 31].y : z[j].y;
                    pz[i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(part_addr, __float_as_int(sx)));
                    pz[i].y = __int_as_float(__builtin_amdgcn_ds_bpermute(part_addr, __float_as_int(sy)));
                }
                float4 cq[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) cq[j] = lds_read_b128(s_cs + (4 * h8 + j) * 32 + l);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k1 = 8 * h8 + i;
                    const float2 zk = z[k1], zm = pz[i];
                    const float2 w = (i & 1) ? make_float2(cq[i >> 1].z, cq[i >> 1].w) : make_float2(cq[i >> 1].x, cq[i >> 1].y);
                    const float sr = zk.x + zm.x, si = zk.y - zm.y;
                    const float dr = zk.x - zm.x, di = zk.y + zm.y;
                    const float tr = w.x * dr - w.y * di, ti = w.x * di + w.y * dr;
                    const float ar = sr + tr, ai = si + ti, br = sr - tr, bi = si - ti;
                    plane[l + 32 * k1] = __builtin_amdgcn_sqrtf(ar * ar + ai * ai);     // |X[k]| / W2
                    plane[M - l - 32 * k1] = __builtin_amdgcn_sqrtf(br * br + bi * bi); // |X[1024 - k]| / W2
                }
            }
            MFX_STAMP2(10);
            { // bin 512 pairs with itself: Z[512] is register 16 of lane 0 (S = 2 Re, T = cs[512] * 2i Im)
                const float2 zk = z[16], w = *s_cs512;
                const float sr = zk.x + zk.x, di = zk.y + zk.y;
                const float tr = -w.y * di, ti = w.x * di;
                const float ar = sr + tr;
                if (lane0) plane[M / 2] = __builtin_amdgcn_sqrtf(ar * ar + ti * ti);
            }
            // the next two frames of this chunk or, from its last iteration, the first two of the next chunk: requested here,
            // unconditionally, where few registers are live -- the words arrive under the mel walk and the DCT
            if (last)
                issue(cnxt, half);
            else
                issue(ccur, f + 2);
            MFX_STAMP2(11);
            const int st_first = s_mst[l]; // first bin of this lane's filter in round 0 of the mel walk (requested before the sync)
            wave_sync();

            MFX_STAMP2(5);
            // ---- mel filterbank on the frame's 32 lanes: per round every lane walks ONE filter's bins in ascending order,
            // one chain of multiply-adds (mfcccpu.cpp:206-217); weights from the lane's own zero-padded row (16-byte
            // reads), magnitudes as 8-byte reads from even starts spread over the banks by the host
            {
                float *lmf = lm + (f & 3) * lmFS;
                const float *wrow = s_mw + l * RS;
#if defined(MFX_ABLATE2048) && (MFX_ABLATE2048 == 1 || MFX_ABLATE2048 == 3)
                for (int r = 0; r < 0; ++r) { // dev-only ablation: no mel walk
#else
                int st = st_first;
                for (int r = 0; r < rounds; ++r) {
#endif
                    // (the next round's first bin is in flight during this round's trips: a round starts with ONE dependent LDS
                    // round trip -- its first magnitudes -- instead of two; the filter id is only needed at the round's end)
                    const int st_next = s_mst[(r + 1 < rounds ? r + 1 : r) * 32 + l], fid = s_mfid[r * 32 + l];
                    const int L = p.mel32_L[r];
                    const float *mg = plane + st;
                    float acc = 0.f;
                    int s2 = 0;
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
                    lmf[fid >= 0 ? fid : lmFS - 1] = MFX_LOG(fmaxf(acc, 1e-30f)); // idle lane: the spare word
                    st = st_next;
                }
            }
            wave_sync();

            MFX_STAMP2(6);
            // ---- every 4th frame (and at the chunk's end): DCT-II + lifter of the waiting frames on the matrix pipe,
            // D[row][c] = sum_m A[row][m] B[m][c] with frame g in rows 4g..4g+3, so register 0 of the result is out[g][c]
            // on lane (g, c); tiles of 16 columns, K steps of 4 bands, two accumulator chains (as k_front_reg)
#if defined(MFX_ABLATE2048) && (MFX_ABLATE2048 == 2 || MFX_ABLATE2048 == 3)
            if (false) { // dev-only ablation: no DCT, no stores
#else
            if ((f0 & 2) || last) {
#endif
                const int g0 = f0 & ~3, gcount = (n_live - g0) < 4 ? (n_live - g0) : 4;
                if (SPLIT) {
                    // <= 40 columns, bands a multiple of 32 (build_dct_mfma_operands4_split): the 16 blocks of an instruction
                    // are (column group, band part) pairs instead of 16 column groups of which 6 - 8 would be empty.
                    // Pass A: lane = (band half kb = lane >> 5, column lane & 31); one instruction adds one band of EACH half.
                    const int nbk = p.num_banks;
                    float ra[4];
                    dct_mfma4s<kRingSplit>(lm + (lane & 3) * lmFS + (lane >> 5) * (nbk >> 1), dct_rsrc, dct_bytes, lane, 0, nbk >> 3, ra);
                    // the halves meet: after the swap lanes 0..31 hold (frame 0 | frame 2), lanes 32..63 (frame 1 | frame 3)
                    const auto a01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ra[0]), __float_as_uint(ra[1]), false, false);
                    const auto a23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(ra[2]), __float_as_uint(ra[3]), false, false);
                    const float oA0 = __uint_as_float(a01[0]) + __uint_as_float(a01[1]);
                    const float oA1 = __uint_as_float(a23[0]) + __uint_as_float(a23[1]);
                    {
                        const int col = lane & 31, fa = lane >> 5;
                        if (col < p.cols) {
                            if (fa < gcount) (p.feat + (out_row + g0 + fa) * (int64_t)p.feat_pitch)[col] = oA0;
                            if (fa + 2 < gcount) (p.feat + (out_row + g0 + fa + 2) * (int64_t)p.feat_pitch)[col] = oA1;
                        }
                    }
                    if (p.dct_split > 1) {
                        // Pass B: lane = (band eighth kb = lane >> 3, column 32 + (lane & 7))
                        float rb[4];
                        dct_mfma4s<4>(lm + (lane & 3) * lmFS + (lane >> 3) * (nbk >> 3), dct_rsrc, dct_bytes, lane, nbk >> 3, nbk >> 5, rb);
                        const auto b01 = __builtin_amdgcn_permlane32_swap(__float_as_uint(rb[0]), __float_as_uint(rb[1]), false, false);
                        const auto b23 = __builtin_amdgcn_permlane32_swap(__float_as_uint(rb[2]), __float_as_uint(rb[3]), false, false);
                        const float s01 = __uint_as_float(b01[0]) + __uint_as_float(b01[1]); // rows 0,1: frame 0; rows 2,3: frame 1
                        const float s23 = __uint_as_float(b23[0]) + __uint_as_float(b23[1]); // rows 0,1: frame 2; rows 2,3: frame 3
                        const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(s01), __float_as_uint(s23), false, false);
                        // 16-lane rows now hold frames 0, 2, 1, 3; lanes l and l + 8 of a row the last two band parts
                        const float t = __uint_as_float(q[0]) + __uint_as_float(q[1]);
                        const float oB = t + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(t), 0x128, 0xf, 0xf, true)); // row_ror:8
                        const int col = 32 + (lane & 7), row = lane >> 4, fb = ((row & 1) << 1) | (row >> 1);
                        if (!(lane & 8) && col < p.cols && fb < gcount) (p.feat + (out_row + g0 + fb) * (int64_t)p.feat_pitch)[col] = oB;
                    }
                } else if (p.dct_b4) {
                    const float *arow = lm + (lane & 3) * lmFS;
                    for (int tile = 0; tile < dct_tiles64; ++tile) {
                        float res[4];
                        dct_mfma4<kRingTile>(arow, dct_rsrc, dct_bytes, lane, tile, dct_ks, res);
                        const int col = 64 * tile + lane;
                        if (col < p.cols) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < gcount) (p.feat + (out_row + g0 + i) * (int64_t)p.feat_pitch)[col] = res[i];
                        }
                    }
                } else { // no DCT: the log mel energies are the features
                    for (int g = 0; g < gcount; ++g)
                        for (int cc = lane; cc < p.cols; cc += 64)
                            (p.feat + (out_row + g0 + g) * (int64_t)p.feat_pitch)[cc] = lm[g * lmFS + cc];
                }
                wave_sync();
            }
            MFX_STAMP2(7);
        }
        if (n_live <= 0) issue(cnxt, half); // empty chunk: its last iteration never ran, nothing was requested
        c_cur = c_nxt;
        ccur = cnxt;
        c_nxt = draw();
        cnxt = make_ctx(c_nxt);
        MFX_STAMP2(8);
    }
#ifdef MFX_STAMPS
    if (lane == 0 && p.spec) {
        unsigned long long *o = (unsigned long long *)p.spec + (size_t)(blockIdx.x * n_waves + wave) * 14;
        unsigned long long st_rt1;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_rt1)::"memory");
        st_acc[9] = st_rt1 - st_rt0; // 100 MHz ticks over the same span
        for (int i = 0; i < 14; ++i) o[i] = st_acc[i];
    }
#endif
}

} // namespace

bool front2048_supported(int fft_size, int window_size, int num_banks, int cols, int channels)
{
    return fft_size == 2048 && window_size > 0 && window_size <= 64 * kRows2048F && (channels == 1 || channels == 2) &&
           num_banks >= 1 && num_banks <= 256 && cols >= 1;
}

namespace {
int lm_fs(const FrontParams &p) { return lm_fs4(p.num_banks); }
size_t lds_bytes_2048(const FrontParams &p, int n_waves)
{
    const size_t f = front2048_table_floats(p.mel32_rounds, p.mel32_row_stride, rows2048(p.window_size)) +
                     (size_t)n_waves * (2 * kPlane + 4 * (size_t)lm_fs(p)) + 4;
    return f * sizeof(float);
}
// waves per block (one block per CU): as many of 12 as the CU's 160 KB of LDS hold, at least 6 (0: does not fit)
int waves_2048(const FrontParams &p)
{
    for (int nw = kW2048; nw >= 6; --nw)
        if (lds_bytes_2048(p, nw) <= 160 * 1024) return nw;
    return 0;
}
} // namespace

size_t front2048_lds_bytes(const FrontParams &p)
{
    const int nw = waves_2048(p);
    return nw ? lds_bytes_2048(p, nw) : (size_t)1 << 30;
}

hipError_t launch_front2048(const FrontParams &p, int num_cus, hipStream_t stream)
{
    if (p.n_chunks <= 0) return hipSuccess;
    const int ch = p.channels == 2 ? 1 : p.pair_ok ? 0 : 2; // (mono off the aligned pairs: the any-alignment build)
    const int nw = waves_2048(p);
    if (nw == 0) return hipErrorInvalidValue;
    const size_t lds = lds_bytes_2048(p, nw);
    const bool split = p.dct_split != 0 && p.dct_b4s != nullptr;
    const int rows = rows2048(p.window_size);
    const bool wide = rows == kRows2048L;
    int blocks = (p.n_chunks + nw - 1) / nw;
    if (blocks > num_cus) blocks = num_cus; // one block of up to 12 waves per CU
    if (blocks < 1) blocks = 1;
    hipError_t err = hipSuccess;
    auto go = [&](auto kern) {
        err = allow_dynamic_lds((const void *)kern, lds);
        if (err == hipSuccess) hipLaunchKernelGGL(kern, dim3(blocks), dim3(nw * 64), lds, stream, p);
    };
    if (rows == kRows2048F) {
        if (ch == 1) { if (split) go(k_front2048<1, true, kRows2048F>); else go(k_front2048<1, false, kRows2048F>); }
        else if (ch == 2) { if (split) go(k_front2048<2, true, kRows2048F>); else go(k_front2048<2, false, kRows2048F>); }
        else { if (split) go(k_front2048<0, true, kRows2048F>); else go(k_front2048<0, false, kRows2048F>); }
    } else if (wide) {
        if (ch == 1) { if (split) go(k_front2048<1, true, kRows2048L>); else go(k_front2048<1, false, kRows2048L>); }
        else if (ch == 2) { if (split) go(k_front2048<2, true, kRows2048L>); else go(k_front2048<2, false, kRows2048L>); }
        else { if (split) go(k_front2048<0, true, kRows2048L>); else go(k_front2048<0, false, kRows2048L>); }
    } else {
        if (ch == 1) { if (split) go(k_front2048<1, true, kRows2048>); else go(k_front2048<1, false, kRows2048>); }
        else if (ch == 2) { if (split) go(k_front2048<2, true, kRows2048>); else go(k_front2048<2, false, kRows2048>); }
        else { if (split) go(k_front2048<0, true, kRows2048>); else go(k_front2048<0, false, kRows2048>); }
    }
    if (err != hipSuccess) return err;
    return hipGetLastError();
}

} // namespace mfx
