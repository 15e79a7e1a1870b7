// mfx_kernels.h -- launch interface of the gfx950 kernels (implemented in mfx_front512.hip, mfx_front_generic.hip, mfx_front2048.hip, mfx_tail.hip: one translation unit per kernel family).
//
// Kernel inventory and the reference stage each one replaces:
//   spectrum512 / fused512   segmenter.cl kernelSegmentWindow + AppleFFT fft0 + mfcc.cl kernelTranspose
//                            (+ mfcc.cl kernelFilter + the DCT slot when fused)
//   spectrum_generic         same three stages for any power-of-two FFT length
//   melcep                   mfcc.cl kernelFilter + DCT slot (mfccopencl.cpp:315-358) from a stored spectrum
//   delta                    delta.cl kernelDelta x2 + the staging copies of mfccopencl.cpp:360-387
//   norm_stats / norm_apply  norm.cl kernelSum + kernelFinalizeSum / kernelNormalize
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace mfx {

// A run of consecutive frames of one utterance (or of the streaming carry buffer).
struct Chunk {
    int64_t pcm_off;  // sample index (per channel) of the first sample of the chunk's first frame
    int64_t out_row;  // destination row of the chunk's first frame
    int32_t n_frames;
    int32_t pad;
};

// One independent series of feature rows (an utterance, or the current streaming block) for the
// delta and normalisation kernels.
struct Segment {
    int64_t src_row0;   // first row of the series inside the static-feature buffer
    int64_t out_row0;   // first output row
    int32_t n_out;      // rows to produce
    int32_t shift;      // padded[i] = src[clamp(i + shift, lo, hi)] (rows relative to src_row0)
    int32_t lo, hi;
    int32_t static_off; // static part of output row r is src row r + static_off
    int32_t pad;        // normalisation: rows the statistics cover (0 = all n_out rows)
};

// One tile of the delta stage fused into the 512-point kernel: <= 64 consecutive output rows of one
// utterance, all of them inside the owning block's row range.
struct DeltaTile {
    int64_t out_row0;   // first output row (absolute) = first row of the statics scratch the tile owns
    int64_t seg_row0;   // first row of the tile's utterance (statics scratch and output use the same rows)
    int32_t n_rows;
    int32_t r0;         // out_row0 - seg_row0
    int32_t shift, lo, hi, static_off; // the utterance's Segment fields (padded[i] = src[clamp(i + shift, lo, hi)])
    int32_t dep_lo, dep_hi; // block-local chunk indices whose statics the tile reads (inclusive)
};

struct FrontParams {
    const int16_t *pcm;
    int64_t pcm_total;        // int16 elements readable behind `pcm` (all channels)
    const Chunk *chunks;
    int32_t n_chunks;
    int32_t channels;         // 1 or 2 (generic kernels only)
    int32_t pair_ok;          // mono, even shift, even window length, even chunk offsets: 2 samples per 32-bit load
    int64_t row_limit;        // frames whose destination row is >= row_limit are skipped
    int32_t window_size;      // W
    int32_t shift;            // S
    int32_t fft_size;         // W2
    // outputs: exactly one of {spec, feat} is used by a launch
    float *spec;              // [rows][spec_pitch] magnitudes |X[k]|/W2, k = 0..W2/2
    int32_t spec_pitch;
    float *feat;              // [rows][feat_pitch], static features written at columns [0, cols)
    int32_t feat_pitch;
    // tables (device pointers)
    const float *window;      // [W2] zero padded                      (generic)
    const float *winpair;     // [16][16][2] window laid out per lane  (512 fast path, k_front1024 phase E)
    const float *win1024o;    // [16][16][4] window x W_512^m as (A, B, C, D) per sample pair (k_front1024 phase O)
    const float *twid_pass;   // [16][16][2] W_256^(l*k)               (512 fast path)
    const float *twid_half;   // [W2/2][2]   W_{W2/2}^k, k < W2/2       (generic Stockham, radix 4 needs 3k)
    const float *twid_reg;    // k_front_reg: pass tables [R1-1][M/R1][2] then [R1-1][M/R1^2][2] (W2 >= 1024)
    const float *twid_split;  // [W2/2+1][2] -i * W_{W2}^k             (real split)
    const float *mel_w;       // [2][W2]
    const int32_t *mel_beg;   // [nb+2]
    const float *dct;         // [nb][dct_len] or nullptr when ceps_len == 0
    // 512 fast path: mel filters dealt to the 16 lanes of a frame in `mel_rounds` rounds, round r
    // padded to mel_L[r] bins (multiple of 4); lane j's weights for all rounds are one row of mel_lane_w
    const float *mel_lane_w;      // [16][mel_row_stride]
    const int32_t *mel_lane_start;// [mel_rounds][16] first bin of the lane's filter in that round
    const int32_t *mel_lane_fid;  // [mel_rounds][16] filter index or -1
    const float *dct_t;           // [cols][dct_stride] transposed DCT matrix, rows zero padded to nb_pad
    int32_t dct_mode;             // 0: DCT from the LDS mel scratch; 1: on the matrix pipe, v_mfma_f32_16x16x4_f32 per
                                  //    4 frames (cols <= 16, num_banks <= 40; matrix from `dct`, held in registers)
    int32_t mel_rounds, mel_row_stride, dct_stride, nb_pad;
    int32_t mel_L[8];
    // wave-per-frame lane plan (k_front_reg / k_front_wave fused): filters dealt to the 64 lanes of the frame's wave in rounds
    const float *mel64_w;         // [64][mel64_row_stride]
    const int32_t *mel64_start;   // [mel64_rounds][64]
    const int32_t *mel64_fid;     // [mel64_rounds][64]
    int32_t mel64_rounds, mel64_row_stride;
    int32_t mel64_L[8];
    // two-frames-per-wave lane plan (k_front2048): filters dealt to the 32 lanes of a frame in rounds of 32
    const float *mel32_w;         // [32][mel32_row_stride]
    const int32_t *mel32_start;   // [mel32_rounds][32]
    const int32_t *mel32_fid;     // [mel32_rounds][32]
    int32_t mel32_rounds, mel32_row_stride;
    int32_t mel32_L[8];
    // DCT on the matrix pipe: B operands [dct_tiles][dct_ksteps][64] (build_dct_mfma_operands), read from L1 / L2
    const float *dct_b;
    const float *dct_b4;          // 4x4x1 form: [ceil(dct_len / 64)][dct_ksteps][64][4] (k_front2048, build_dct_mfma_operands4)
    int32_t stuff;                // k_front512: 0, or 512 / fft_size = 2, 4, 8: 256 / 128 / 64-point transforms in the zero-stuffed form
    const float *dct_b4s;         // k_front2048: the split form (build_dct_mfma_operands4_split) or nullptr
    int32_t dct_split;            // dct_split_mode(): 0 none, 1 pass A (<= 32 columns), 2 passes A + B (<= 40 columns)
    int32_t dct_tiles, dct_ksteps;
    int32_t num_banks;
    int32_t dct_len;
    int32_t cols;             // dct_len, or num_banks when ceps_len == 0
    float scale;              // 1/W2 (0.5/W2 where the split's 1/2 is folded in)
    // fused delta stage (512 fast path, launch_front512_delta): block b walks the chunks
    // [blk_chunk_off[b], blk_chunk_off[b+1]) of `chunks` in order (its own rows plus <= D halo rows either
    // side, statics to the compact scratch `feat`, pitch 16) while its last wave turns finished statics
    // into whole [static | d | dd] rows of `out` for the tiles [blk_tile_off[b], blk_tile_off[b+1]).
    const int32_t *blk_chunk_off;
    const int32_t *blk_tile_off;
    const DeltaTile *tiles;
    float *out;
    int32_t out_pitch;
    int32_t dl1, dl2;
    int32_t n_blocks;
    int32_t done_words;       // LDS words of the per-block "chunk finished" bitmap
    int32_t *err_flag;        // set nonzero if the delta wave gave up waiting (never expected)
};

struct MelcepParams {
    const float *spec;
    int32_t spec_pitch;
    int64_t n_rows;
    float *feat;
    int32_t feat_pitch;
    int32_t fft_size;
    const float *mel_w;
    const int32_t *mel_beg;
    const float *dct;
    int32_t num_banks, dct_len, cols;
    // 64-lane mel plan (MelWavePlan), one per table of a sweep, all padded to the same row stride
    const float *mel64_w;         // [n_tables][64][mel64_row_stride]
    const int32_t *mel64_start;   // [n_tables][mel64_rounds][64]
    const int32_t *mel64_fid;     // [n_tables][mel64_rounds][64]
    const int32_t *mel64_L;       // [n_tables][8] bins per lane and round (device memory)
    int32_t mel64_rounds, mel64_row_stride;
    int32_t mag_floats;           // LDS floats of a wave's magnitude buffer: >= spec_pitch and > the plans' last read, x4
    const float *dct_b4;          // DCT operands of the 4x4x1 form (build_dct_mfma_operands4) or nullptr
    int32_t dct_ksteps;
    // VTLN sweep: n_tables (>= 1) warped filterbanks over the same spectrum in one launch; table a is
    // mel_w + a * mel_w_stride / mel_beg + a * mel_beg_stride and writes feat + a * feat_table_stride
    int32_t n_tables;
    int32_t mel_beg_stride;
    int64_t mel_w_stride;
    int64_t feat_table_stride;
};

struct DeltaParams {
    const float *src;      // static features, [rows][src_pitch]
    int32_t src_pitch;
    float *out;            // [rows][out_pitch]: [static | delta | acc]
    int32_t out_pitch;
    const Segment *segs;
    int32_t n_segs;
    int32_t cols;
    int32_t l1, l2;        // l2 == 0: first order only; l1 == 0: copy statics only
    int32_t tiles_per_seg_max;
    int32_t inline_seg;    // nonzero: ignore segs and use seg0 (single streaming block)
    Segment seg0;
};

struct NormParams {
    float *data;           // [rows][pitch], normalised in place at column offset col0
    int32_t pitch;
    int32_t col0;
    int32_t cols;
    const Segment *segs;   // uses out_row0 / n_out (rows to normalise) only
    int32_t n_segs;
    int32_t row_off;       // extra row offset added to out_row0
    int32_t norm_type;     // MFX_NORM_*
    float *stats;          // [n_segs][2][cols]: mean, scale (persist across calls for use_last_stats)
    int32_t inline_seg;    // nonzero: ignore segs and use seg0
    Segment seg0;
    int32_t max_rows;      // largest row count of any segment (sizes the grids)
    int32_t chunks;        // set by the launcher: row chunks per segment
    double *partial;       // [n_segs][chunks][4][cols] scratch, needed when max_rows > 4096 (norm_partial_doubles)
    int32_t groups;        // launch_norm_fused only: column groups col0 + g * cols normalised in ONE launch (0 / 1: one)
    int64_t group_stats_stride; // floats between the statistics of consecutive groups
};

// All launchers are asynchronous on `stream` and return the launch status.
hipError_t launch_front512(const FrontParams &p, bool to_spectrum, bool aligned, int nm16, hipStream_t stream);
// fused front end + delta stage (p.blk_chunk_off etc. filled in); statics only pass through p.feat
hipError_t launch_front512_delta(const FrontParams &p, bool aligned, int nm16, hipStream_t stream);
size_t front512_delta_lds_bytes(const FrontParams &p);
// fused = mel/log/DCT in the same kernel (statics to p.feat); else magnitudes to p.spec
hipError_t launch_front_generic(const FrontParams &p, bool fused, hipStream_t stream);
size_t front_wave_lds_bytes(const FrontParams &p, bool fused);
hipError_t launch_melcep(const MelcepParams &p, hipStream_t stream);
hipError_t launch_delta(const DeltaParams &p, hipStream_t stream);
// LDS of k_melcep with n_waves waves per block (the launcher takes as many of 4 as fit)
size_t melcep_lds_bytes(const MelcepParams &p, int n_waves);
hipError_t launch_norm_stats(const NormParams &p, hipStream_t stream);
// doubles of NormParams::partial for n_segs segments of at most max_rows rows (0: none needed)
size_t norm_partial_doubles(int n_segs, int max_rows, int cols);
hipError_t launch_norm_apply(const NormParams &p, hipStream_t stream);
// copy of a small block (even byte count) by a kernel; either side may be page-locked host memory
hipError_t launch_copy_small(void *dst, const void *src, size_t bytes, hipStream_t stream);
// statistics + apply in one launch, for segments of at most 48 KB of rows (norm_fused_fits); same bits as the pair above
bool norm_fused_fits(int max_rows, int cols);
hipError_t launch_norm_fused(const NormParams &p, hipStream_t stream);

// dynamic LDS bytes one block of the 512-point kernel needs for these parameters
size_t front512_lds_bytes(const FrontParams &p);

// true when the 512-point fast path can take this configuration
bool front512_supported(int fft_size, int window_size, int num_banks, int cols, int channels);
// k_front1024: 1024-point transform of a window of at most 512 samples on the k_front512 core (two 256-point complex
// transforms per frame: even and odd bins), mel -> log -> DCT fused, statics out
bool front1024_supported(int fft_size, int window_size, int num_banks, int cols, int channels, int ceps_len);
size_t front1024_lds_bytes(const FrontParams &p, int waves = 12);
int front1024_waves(const FrontParams &p, bool aligned, int nm16, int max_waves = 16); // 16 waves per CU where the build and the LDS allow, else 12
hipError_t launch_front1024(const FrontParams &p, bool aligned, int nm16, hipStream_t stream, int max_waves = 16);

// k_front2048 (mfx_front2048.hip): 2048-point transform of a window of at most 1152 samples, two frames per wave (32 lanes
// each, 32 x 32 two-pass FFT), mono (aligned sample pairs) or interleaved stereo, mel -> log -> DCT fused, statics out
bool front2048_supported(int fft_size, int window_size, int num_banks, int cols, int channels);
size_t front2048_lds_bytes(const FrontParams &p);
hipError_t launch_front2048(const FrontParams &p, int num_cus, hipStream_t stream);

// symbol name of the dominant kernel for rocprofv3 (depends on the instantiation chosen)
const char *front512_kernel_name(bool to_spectrum, bool aligned, int nm16);

} // namespace mfx
