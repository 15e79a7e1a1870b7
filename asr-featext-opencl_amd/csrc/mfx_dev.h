// mfx_dev.h -- device helpers shared by the gfx950 kernel translation units (mfx_front512.hip, mfx_front_generic.hip, mfx_front2048.hip, mfx_tail.hip):
// wave-level LDS ordering, LDS reads the optimiser must keep whole, the register FFT butterflies, the fast log.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mfx {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------

// All cross-lane traffic inside a wave goes through LDS instructions of that same wave, which the
// LDS executes in issue order; only the compiler has to be kept from reordering around it.
// (L + R) >> 1 of one stereo sample word (L | R << 16) as a float: the two sign-extended halves in ONE sub-word add
// (left to itself the compiler builds the overflow-free average from xor / and / 16-bit shift / 16-bit add: five
// operations per sample, each at the slow issue rate -- k_front2048 0.346 -> 0.332 ms on C5, profiles/r03/abx_c5_cvt_sdwa_add.txt)
__device__ __forceinline__ float stereo_mean(uint32_t d)
{
    int s;
    asm("v_add_u32_sdwa %0, sext(%1), sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1"
        : "=v"(s)
        : "v"(d));
    return (float)(s >> 1);
}

// Blocks are dealt to the 8 XCDs round robin (blocks b and b + 8 share one, whichever it is: MI355X_MICROARCH.md, Workgroup
// dispatch), each XCD with its own L2.  The logical id of this block, chosen so that CONSECUTIVE ids share an XCD (bijective for
// any grid size): consecutive chunks -- neighbouring 16-frame pieces of one utterance, W - S samples in common -- are then
// fetched through one L2 instead of two.  A speed choice only: nothing depends on where a block really runs.
__device__ __forceinline__ int xcd_block_id()
{
#ifdef MFX_NO_XCD_REMAP
    return blockIdx.x;
#else
    const int n = gridDim.x, b = blockIdx.x, x = b & 7, q = n >> 3, r = n & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (b >> 3);
#endif
}

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// 8-byte LDS read that the load/store optimiser must not fuse with a neighbour: ds_read2_b64 costs
// 8 LDS cycles for 16 bytes per lane where two ds_read_b64 cost 2 + 2 (MI355X_MICROARCH.md, LDS table).
__device__ __forceinline__ float2 lds_read_b64(const float2 *p)
{
    const float2 v = *p;
    asm volatile("" ::: "memory"); // a compiler-level fence between neighbouring reads keeps them apart
    return v;
}

// 16-byte LDS read kept whole (the optimiser otherwise splits a read whose halves are used apart and re-pairs the
// pieces as ds_read2_b64, which costs twice the LDS cycles of ds_read_b128)
__device__ __forceinline__ float4 lds_read_b128(const float4 *p)
{
    typedef float v4f __attribute__((ext_vector_type(4)));
    v4f v = *(const v4f *)p;
    asm("" : "+v"(v)); // (not volatile: free to move)
    return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 w)
{
    return make_float2(a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x);
}

// 4-point forward DFT (W4 = -i)
__device__ __forceinline__ void dft4(float2 a0, float2 a1, float2 a2, float2 a3, float2 &o0, float2 &o1,
                                     float2 &o2, float2 &o3)
{
    float2 b0 = make_float2(a0.x + a2.x, a0.y + a2.y);
    float2 b1 = make_float2(a0.x - a2.x, a0.y - a2.y);
    float2 b2 = make_float2(a1.x + a3.x, a1.y + a3.y);
    float2 b3 = make_float2(a1.x - a3.x, a1.y - a3.y);
    o0 = make_float2(b0.x + b2.x, b0.y + b2.y);
    o2 = make_float2(b0.x - b2.x, b0.y - b2.y);
    o1 = make_float2(b1.x + b3.y, b1.y - b3.x); // b1 - i*b3
    o3 = make_float2(b1.x - b3.y, b1.y + b3.x); // b1 + i*b3
}

// 16-point forward DFT held entirely in registers, natural order in and out (4 x 4 Cooley-Tukey).
// Inputs that are compile-time zeros are folded away by the compiler after inlining.
__device__ __forceinline__ void fft16(float2 (&x)[16])
{
    constexpr float C1 = 0.92387953251128673848f; // cos(pi/8)
    constexpr float S1 = 0.38268343236508978178f; // sin(pi/8)
    constexpr float R = 0.70710678118654752440f;  // sqrt(1/2)
    float2 y[16];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2)
        dft4(x[n2], x[4 + n2], x[8 + n2], x[12 + n2], y[0 + n2], y[4 + n2], y[8 + n2], y[12 + n2]);
    // y[4*k1 + n2] *= W16^(n2*k1)
    // (written with negative constants instead of subtracted products: "a * K + b * (-K2)" compiles to v_mul + v_fmamk with
    // literal constants, which issue at full rate; "a * K - b * K2" takes a VOP3 fma with a negated operand and the
    // constant in a scalar register, which issues at half rate -- tools/ubench/valu_forms.hip)
    constexpr float nC1 = -C1, nS1 = -S1, nR = -R;
    float2 t;
    t = y[4 + 1];  y[4 + 1]  = make_float2(t.x * C1 + t.y * S1, t.y * C1 + t.x * nS1);    // W^1
    t = y[4 + 2];  y[4 + 2]  = make_float2(R * (t.x + t.y), R * (t.y - t.x));             // W^2
    t = y[4 + 3];  y[4 + 3]  = make_float2(t.x * S1 + t.y * C1, t.y * S1 + t.x * nC1);    // W^3
    t = y[8 + 1];  y[8 + 1]  = make_float2(R * (t.x + t.y), R * (t.y - t.x));             // W^2
    t = y[8 + 2];  y[8 + 2]  = make_float2(t.y, -t.x);                                    // W^4
    t = y[8 + 3];  y[8 + 3]  = make_float2(R * (t.y - t.x), nR * (t.x + t.y));            // W^6
    t = y[12 + 1]; y[12 + 1] = make_float2(t.x * S1 + t.y * C1, t.y * S1 + t.x * nC1);    // W^3
    t = y[12 + 2]; y[12 + 2] = make_float2(R * (t.y - t.x), nR * (t.x + t.y));            // W^6
    t = y[12 + 3]; y[12 + 3] = make_float2(t.x * nC1 + t.y * nS1, t.x * S1 + t.y * nC1);  // W^9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
        dft4(y[4 * k1], y[4 * k1 + 1], y[4 * k1 + 2], y[4 * k1 + 3], x[k1], x[k1 + 4], x[k1 + 8], x[k1 + 12]);
}

// 8-point forward DFT in registers, natural order in and out
__device__ __forceinline__ void fft8(float2 (&x)[8])
{
    constexpr float R = 0.70710678118654752440f;
    float2 e[4], o[4];
    dft4(x[0], x[2], x[4], x[6], e[0], e[1], e[2], e[3]);
    dft4(x[1], x[3], x[5], x[7], o[0], o[1], o[2], o[3]);
    float2 t;
    t = o[1]; o[1] = make_float2(R * (t.x + t.y), R * (t.y - t.x));   // W8^1
    t = o[2]; o[2] = make_float2(t.y, -t.x);                          // W8^2 = -i
    t = o[3]; o[3] = make_float2(R * (t.y - t.x), -R * (t.x + t.y));  // W8^3
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = make_float2(e[k].x + o[k].x, e[k].y + o[k].y);
        x[k + 4] = make_float2(e[k].x - o[k].x, e[k].y - o[k].y);
    }
}

// Natural log of a positive normal float.  v_log_f32 is accurate to 1 ulp of log2(x); the product
// with ln 2 keeps the absolute error near 1e-7 * |log x|, far inside the parity tolerance (the
// reference uses libm logf, mfcccpu.cpp:212).  -DMFX_EXACT_LOG selects the library logf instead.
#ifdef MFX_EXACT_LOG
#define MFX_LOG(x) logf(x)
#else
#define MFX_LOG(x) (__builtin_amdgcn_logf(x) * 0.69314718055994530942f)
#endif

// Log mel energies of 4 consecutive frames wait in LDS for the DCT on the matrix pipe: rows of lm_stride(nb) floats,
// stride = 8 mod 32 so that the 4 frames' operand reads fall on distinct banks, at least one spare word per row (idle
// lanes of the mel walk park their value there).
__host__ __device__ inline int lm_stride(int nb)
{
    int x = ((nb + 3) & ~3) + 1;
    while ((x & 31) != 8) ++x;
    return x;
}


// Log mel energies of 4 frames waiting for the DCT: lm[g * FS + m]; FS = 4 q with q odd and FS > nb (the last word parks
// idle lanes): the four frames' 16-byte operand reads then fall on distinct bank quads.
__host__ __device__ inline int lm_fs4(int nb)
{
    int q = (nb + 4) / 4 + ((nb + 4) % 4 ? 1 : 0);
    if ((q & 1) == 0) ++q;
    return 4 * q;
}

// DCT-II + lifter of 4 frames on the matrix pipe, 64 output columns at a time: v_mfma_f32_4x4x1_16b_f32 computes 16
// independent 4 x 4 outer products, D_b[i][j] += A_b[i] * B_b[j]; with i = frame, b = lane / 4, j = lane % 4 one
// instruction adds ONE band m to out[frame i][column 4 b + j] for all 64 columns and all 4 frames:
//     A operand of lane l: lm[frame l & 3][m]          B operand of lane l: dct[m][64 tile + l]
//     result register i of lane l: out[frame i][64 tile + l]
// Every product of the instruction is used when the row has 64 columns (40 here: 62 %), where the 16 x 16 x 4 form with
// 4 frames used a quarter of its rows -- and on gfx950 f32 matrix instructions do not overlap the vector pipe
// (SQ_VALU_MFMA_COEXEC_CYCLES = 0: they run on the same FP32 units), so their cycles are the SIMD's cycles: 8 per band
// for four frames here against 32 per 4 bands and 16 columns before (C5: 3072 -> 1024 SIMD cycles per four frames).
// Bands are accumulated in ascending order in two chains (even / odd m), as mfcccpu.cpp:222-232 sums them up to that
// association.  Operands come four bands at a time (one 16-byte LDS read, one 16-byte buffer load; B laid out
// [tile][band / 4][lane][4], bands past the table return 0), software pipelined without branches: the next four bands'
// operands are requested before the current four instructions issue.
// KS > 0: the number of four-band batches is a compile-time constant and the whole ring is unrolled -- every wait is then
// counted exactly (around a loop's back edge the compiler waits for ALL outstanding loads: the ring would be drained once
// per turn); KS = 0: a loop over runtime `ks` for the other filterbank sizes.
template <int KS, int DEPTH>
__device__ __forceinline__ void dct_mfma4_impl(const float *arow, __amdgpu_buffer_rsrc_t rsrc, int table_bytes, int lane, int tile,
                                               int ks, float (&res)[4])
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef float v4f __attribute__((ext_vector_type(4)));
    f32x4 d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
    // A ring of DEPTH operand pairs in flight: four instructions are 32 SIMD cycles, an operand load from L1 / L2 takes
    // 300 - 500 -- with one batch ahead the wave waited out every load (stamps: 84 cycles per instruction).
    // (DEPTH: 6 in k_front2048 -- 8 spills 37 registers in its stereo build; 3 in k_melcep / k_front_wave, whose
    // occupancy is set by registers, not LDS)
    v4f av[DEPTH];
    u32x4 bv[DEPTH];
    // (plain 16-byte LDS read: the lds_read_b128 wrapper's register constraint would wait for the data where it is
    // requested, not where it is used)
    auto fetch = [&](int slot, int j4) {
        av[slot] = *(const v4f *)(arow + 4 * (j4 < ks ? j4 : 0));
        bv[slot] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, lane * 16, j4 < ks ? ((tile * ks + j4) * 1024) : table_bytes, 0);
    };
    auto mac = [&](int slot) {
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[slot][0], __uint_as_float(bv[slot][0]), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[slot][1], __uint_as_float(bv[slot][1]), d1, 0, 0, 0);
        d0 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[slot][2], __uint_as_float(bv[slot][2]), d0, 0, 0, 0);
        d1 = __builtin_amdgcn_mfma_f32_4x4x1f32(av[slot][3], __uint_as_float(bv[slot][3]), d1, 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) fetch(i, i);
    if (KS > 0) {
#pragma unroll
        for (int j4 = 0; j4 < KS; ++j4) {
            __builtin_amdgcn_sched_barrier(0);
            mac(j4 % DEPTH);
            if (j4 + DEPTH < KS) fetch(j4 % DEPTH, j4 + DEPTH);
        }
    } else {
        for (int j4 = 0; j4 < ks; j4 += DEPTH) {
#pragma unroll
            for (int i = 0; i < DEPTH; ++i) {
                __builtin_amdgcn_sched_barrier(0);
                mac(i);
                fetch(i, j4 + DEPTH + i);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) res[i] = d0[i] + d1[i];
}

template <int DEPTH>
__device__ __forceinline__ void dct_mfma4(const float *arow, __amdgpu_buffer_rsrc_t rsrc, int table_bytes, int lane, int tile,
                                          int ks, float (&res)[4])
{
    // the filterbank sizes of the BASELINE configurations (128 / 80 / 40 bands) get the unrolled form
    if (ks == 32)
        dct_mfma4_impl<32, DEPTH>(arow, rsrc, table_bytes, lane, tile, ks, res);
    else if (ks == 20)
        dct_mfma4_impl<20, DEPTH>(arow, rsrc, table_bytes, lane, tile, ks, res);
    else if (ks == 10)
        dct_mfma4_impl<10, DEPTH>(arow, rsrc, table_bytes, lane, tile, ks, res);
    else
        dct_mfma4_impl<0, DEPTH>(arow, rsrc, table_bytes, lane, tile, ks, res);
}

// The split form of k_front2048 (build_dct_mfma_operands4_split): `group0` = first 1 KB K-group of the pass in the table,
// `ks` K-groups of 4 bands; C5's 128 bands (16 / 4 groups) get the unrolled form.
template <int DEPTH>
__device__ __forceinline__ void dct_mfma4s(const float *arow, __amdgpu_buffer_rsrc_t rsrc, int table_bytes, int lane, int group0,
                                           int ks, float (&res)[4])
{
    // (dct_mfma4_impl addresses K-group tile * ks + j4: with ks a divisor of group0 the pass starts at tile = group0 / ks)
    if (ks == 16 && group0 == 0)
        dct_mfma4_impl<16, DEPTH>(arow, rsrc, table_bytes, lane, 0, 16, res);
    else if (ks == 4 && group0 == 16)
        dct_mfma4_impl<4, DEPTH>(arow, rsrc, table_bytes, lane, 4, 4, res);
    else
        dct_mfma4_impl<0, DEPTH>(arow, rsrc, table_bytes, lane, group0 / ks, ks, res);
}

// One frame's mel filterbank on the 64 lanes of a wave (MelWavePlan, lanes = 64) + log: per round every lane walks ONE
// filter's bins in ascending order, one chain of multiply-adds (mfcccpu.cpp:206-217); weights from the lane's own
// zero-padded row (16-byte reads, disjoint bank quads), magnitudes as 8-byte reads from even starts the host spread over
// the banks.  mag must hold finite values up to the plan's last read.  The log energy of filter fid goes to lmf[fid]; idle
// lanes park theirs in lmf[park].  s_mw holds mel64_rows(nb) weight rows: filters are dealt to lanes 0, 1, ... of each
// round, so lanes >= nb never carry one -- they walk the last staged row (same addresses: a broadcast) and park a finite value.
__host__ __device__ __forceinline__ int mel64_rows(int num_banks) { return num_banks < 64 ? (num_banks > 0 ? num_banks : 1) : 64; }

__device__ __forceinline__ void mel64_walk_log(const float *mag, float *lmf, int park, const float *s_mw, const int *s_mst,
                                               const int *s_mfid, const int *Lr, int rounds, int RS, int lane, int w_rows)
{
    const float *wrow = s_mw + (lane < w_rows ? lane : w_rows - 1) * RS;
    for (int r = 0; r < rounds; ++r) {
        const int st = s_mst[r * 64 + lane], fid = s_mfid[r * 64 + lane];
        const int L = Lr[r];
        const float *mg = mag + st;
        float acc = 0.f;
        for (int s2 = 0; s2 < L; s2 += 8) {
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
        lmf[fid >= 0 ? fid : park] = MFX_LOG(fmaxf(acc, 1e-30f));
    }
}

// DCT of the (up to) 4 frames whose log energies wait in lm[4][FS] (lm_fs4), rows out_row0 .. out_row0 + count - 1 of
// feat: 64 columns per pass on the matrix pipe (dct_mfma4), or -- without a DCT -- the log energies themselves.
template <int DEPTH>
__device__ __forceinline__ void dct4_store(const float *lm, int FS, __amdgpu_buffer_rsrc_t rsrc, int table_bytes, int ks,
                                           int tiles64, bool has_dct, int lane, int cols, float *feat, int64_t feat_pitch,
                                           int64_t out_row0, int count)
{
    if (has_dct) {
        const float *arow = lm + (lane & 3) * FS;
        for (int tile = 0; tile < tiles64; ++tile) {
            float res[4];
            dct_mfma4<DEPTH>(arow, rsrc, table_bytes, lane, tile, ks, res);
            const int col = 64 * tile + lane;
            if (col < cols) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < count) (feat + (out_row0 + i) * feat_pitch)[col] = res[i];
            }
        }
    } else {
        for (int g = 0; g < count; ++g)
            for (int cc = lane; cc < cols; cc += 64) (feat + (out_row0 + g) * feat_pitch)[cc] = lm[g * FS + cc];
    }
}

} // namespace
} // namespace mfx
