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
