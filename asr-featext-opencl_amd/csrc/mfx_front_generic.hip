// mfx_front_generic.hip -- the one-WAVE-per-frame front ends and their launcher:
//   k_front_wave  64 .. 512 points, what k_front512 refuses (more than 128 filters or columns, MFX_ENGINE_NO_STUFF256)
//   k_front_reg   1024 / 2048 / 4096 points beyond k_front1024's / k_front2048's limits (stereo at 1024, more than 80 filters,
//                 long windows on unaligned frames, 4096 points, MFX_ENGINE_NO_FRONT*)
// Same reference stages and numerics as mfx_front512.hip (mfcccpu.cpp:187-232).  See DESIGN.md section 5.
#include "mfx_kernels.h"

#include <hip/hip_runtime.h>

#include "mfx_dev.h"
#include "mfx_launch.h"

#include <algorithm>
#include <cstdlib>

namespace mfx {

namespace {

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
    const int nbp = FUSED ? lm_fs4(nb) : 0; // row pitch of the 4 waiting frames' log energies (dct_mfma4's operand layout)
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
    // the DCT's B operands [tiles of 64 columns][bands / 4][lane][4] through a buffer descriptor (offsets past the table return 0)
    const int dct_tiles64 = (p.dct_len + 63) >> 6;
    const int dct_bytes = (FUSED && p.dct_b4) ? dct_tiles64 * p.dct_ksteps * 1024 : 0;
    const __amdgpu_buffer_rsrc_t dct_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dct_b4, 0, dct_bytes, 0x00020000);

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
    const int block_id = xcd_block_id(); // (consecutive ids, i.e. consecutive chunks, on one XCD's L2)
    int fixed_c = block_id * n_waves + wave; // (!DRAW: wave w of the grid takes chunks w, w + W, w + 2 W, ...)
    auto draw = [&]() -> int {
        if (!DRAW) {
            const int cc = fixed_c < p.n_chunks ? fixed_c : p.n_chunks;
            if (fixed_c < p.n_chunks) fixed_c += gridDim.x * n_waves;
            return cc;
        }
        int k = 0;
        if ((tid & 63) == 0) k = __hip_atomic_fetch_add(s_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        k = __builtin_amdgcn_readfirstlane(k);
        const long long cc = (long long)block_id + (long long)k * gridDim.x;
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
                // ---- every 4th frame (and at the chunk's end): DCT-II + lifter of the waiting frames on the matrix pipe, 64
                // columns and 4 frames per v_mfma_f32_4x4x1 (dct_mfma4, mfx_dev.h: one band per instruction, an fmaf chain in
                // ascending m) -- round 4: the 16x16x4 form this kernel kept until then used a quarter of its rows
                if ((f & 3) == 3 || f == nf - 1) {
                    const int g0 = f & ~3, gcount = f - g0 + 1;
                    dct4_store<3>(lm, nbp, dct_rsrc, dct_bytes, p.dct_ksteps, dct_tiles64, p.dct_b4 != nullptr, lane, p.cols, p.feat,
                                  (int64_t)p.feat_pitch, ch.out_row + g0, gcount);
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

} // namespace

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
    f += (size_t)n_waves * (2 * MP + (fused ? 4 * lm_fs4(p.num_banks) : 0)) + 4; // (+ the block's work counter)
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
    if (hipError_t e = allow_dynamic_lds((const void *)k_front_reg<LOG2M, FUSED, PAIR, HALF>, lds); e != hipSuccess) return e;
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
    if (hipError_t e = allow_dynamic_lds(fn, lds); e != hipSuccess) return e;
    int blocks = (p.n_chunks + 3) / 4;
    // persistent blocks, as many as are RESIDENT at once (registers: 6 per CU for the fused build; the LDS may allow fewer): a
    // grid of 8 per CU where 5 fit ran in two rounds -- 8 kHz / 256 points 2.46 ms per 2 M frames against 1.78 with 8 resident
    const int per_cu = blocks_per_cu(fn, 256, lds, 4); // (one query per device and launch shape, not per launch)
    const int cap = num_cus() * (per_cu > 8 ? 8 : per_cu);
    if (blocks > cap) blocks = cap;
    if (fused)
        hipLaunchKernelGGL((k_front_wave<true, 64>), dim3(blocks), dim3(256), lds, stream, p);
    else
        hipLaunchKernelGGL((k_front_wave<false, 64>), dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError();
}

} // namespace mfx
