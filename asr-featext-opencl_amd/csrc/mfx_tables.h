// mfx_tables.h -- host-side construction of the constant tables the HIP kernels consume.
//
// The tables are *data* of the feature extractor, defined by the reference's CPU back end; they
// are built once per handle (the mel table again on every set_alpha change) in the reference's
// float32 expression order and uploaded to HBM.
#pragma once
#include <cstdint>
#include <vector>

namespace mfx {

// Smallest power of two >= v (reference: ceil2, mfcccpu.cpp:10-20).
uint32_t ceil_pow2(uint32_t v);

// Frames that fit in `samples` samples: floor((samples - (W - S)) / S), integer arithmetic.
// Equals ParamBase::estimated_window_count (parambase.cpp:16-19) wherever that float32
// expression is exact (samples < 2^24).
int64_t frame_count(int64_t samples, int window_size, int shift);

// The reference's float32 version, bit-for-bit (used by the streaming state machine).
int estimated_window_count_f32(int samples, int window_size, int shift);

struct MelTable {
    std::vector<float> weights;   // [2][fft_size]: row 0 even-numbered filters, row 1 odd-numbered
    std::vector<int32_t> beg;     // [num_banks + 2] first bin of filter i (= rounded centre i)
};

// Triangular mel filterbank with optional VTLN bilinear warp (reference: MfccCpu::refresh_filters,
// mfcccpu.cpp:24-60).  Filter m covers bins [beg[m], beg[m+2]) of weights[m % 2].
void build_mel_table(int num_banks, int fft_size, float sample_rate, float low_freq, float high_freq,
                     float alpha, MelTable &out);

// DCT-II with sinusoidal lifter, c0 (if wanted) as the LAST column
// (reference: mfcccpu.cpp:118-136).  Row-major [num_banks][dct_len].
void build_dct_matrix(int num_banks, int ceps_len, bool want_c0, float lift_coef, std::vector<float> &out);

// exp(-2*pi*i*k/n) for k in [0, count), evaluated in double and rounded once to float.
void build_twiddles(int n, int count, std::vector<float> &re_im_interleaved);

} // namespace mfx

namespace mfx {

// Work plan of the mel stage of the 512-point kernel: the filters are dealt to the 16 lanes that
// share a frame, `rounds` at a time (longest filters first so that a round's padding is small).
// Lane j's weights for round r start at w[j * row_stride + sum(L[0..r))], cover bins
// [start[r][j], start[r][j] + L[r]) and are zero outside the filter's own span.
struct MelLanePlan {
    int rounds = 0;
    int row_stride = 0;            // floats; a multiple of 4 with row_stride / 4 odd (LDS banks)
    int L[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<float> w;          // [16][row_stride]
    std::vector<int32_t> start;    // [rounds][16]
    std::vector<int32_t> fid;      // [rounds][16], -1 = idle lane
};

// false when the plan does not fit the kernel's limits (more than 8 rounds, or a padded span that
// would read past `max_read_bin`)
// align: starts are multiples of it (2: two bins per 8-byte read of one magnitude array; 4: the de-interleaved even / odd
// arrays of k_front1024, two bins of each per 8-byte read)
bool build_mel_lane_plan(const MelTable &t, int num_banks, int fft_size, int max_read_bin, MelLanePlan &out, int align = 2);

// The same plan for the wave-per-frame kernels (64 lanes share a frame; k_front_reg): rounds of 64 filters, longest
// first, every lane walks its filter's bins in ascending order (the reference's summation order, mfcccpu.cpp:192-220).
//   w     : [64][row_stride], row_stride / 4 odd: the 16 lanes of a 16-byte access group read disjoint bank quads
//   start : [rounds][64] first bin read (even), fid : [rounds][64] filter or -1, L[r] bins per lane in round r (x8)
struct MelWavePlan {
    int rounds = 0;
    int row_stride = 0;
    int L[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<float> w;
    std::vector<int32_t> start;
    std::vector<int32_t> fid;
};
// false when it does not fit (more than 8 rounds, or a read past max_read_bin)
// lanes: 64 (one frame per wave, k_front_reg) or 32 (two frames per wave, k_front2048: the two halves of the wave walk
// the same filters of their own frames); the arrays are [.][lanes]
bool build_mel_wave_plan(const MelTable &t, int num_banks, int fft_size, int max_read_bin, MelWavePlan &out, int lanes = 64);

// B operands of the DCT on the matrix pipe (v_mfma_f32_16x16x4_f32), for every tile of 16 output columns and every
// K step of 4 mel bands: out[(tile * ksteps + j) * 64 + lane] = dct[4 j + (lane >> 4)][16 tile + (lane & 15)],
// zero beyond the matrix.  ksteps = ceil(num_banks / 4), tiles = ceil(dct_len / 16).
void build_dct_mfma_operands(const std::vector<float> &dct, int num_banks, int dct_len, int &tiles, int &ksteps,
                             std::vector<float> &out);

// B operands of the DCT on v_mfma_f32_4x4x1_16b_f32 (k_front2048): 64 output columns per tile, four bands per 16-byte load,
// out[((tile * ks + j4) * 64 + lane) * 4 + u] = dct[4 j4 + u][64 tile + lane], ks = ceil(num_banks / 4), zero beyond the matrix.
void build_dct_mfma_operands4(const std::vector<float> &dct, int num_banks, int dct_len, std::vector<float> &out);

// k_front2048 with at most 40 output columns and a multiple of 32 bands: the 64-column tile of the 4x4x1 form would be
// 37 - 50 % empty.  The 16 blocks of an instruction are dealt to (column group, band part) instead: pass A = 8 groups of 4
// columns x 2 band halves (columns 0..31), pass B = 2 groups x 8 band eighths (columns 32..39); the parts are summed across
// the wave afterwards.  mode 0: not applicable, 1: pass A alone (dct_len <= 32), 2: A + B.
//   out = [nb / 8 groups of pass A][64][4] then [nb / 32 groups of pass B][64][4]:
//   A: out[(g * 64 + lane) * 4 + u] = dct[(lane >> 5) * nb / 2 + 4 g + u][lane & 31]
//   B: out[((nb / 8 + g) * 64 + lane) * 4 + u] = dct[(lane >> 3) * nb / 8 + 4 g + u][32 + (lane & 7)]
int dct_split_mode(int num_banks, int dct_len);
void build_dct_mfma_operands4_split(const std::vector<float> &dct, int num_banks, int dct_len, std::vector<float> &out);

// Transposed, padded DCT matrix for the 512-point kernel: [cols][stride], stride / 4 odd,
// row c = column c of the [num_banks][dct_len] matrix followed by zeros.
void build_dct_transposed(const std::vector<float> &dct, int num_banks, int dct_len, int &stride, int &nb_pad,
                          std::vector<float> &out);

} // namespace mfx
