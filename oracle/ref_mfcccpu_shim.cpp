// ref_mfcccpu_shim.cpp -- extern "C" door onto the REAL MfccCpu member functions (TEST INFRASTRUCTURE ONLY).
//
// oracle/Makefile compiles /root/reference/mfcccpu.cpp IN PLACE (never copied) with
//     -ffunction-sections -fdata-sections -fvisibility=hidden
// (the vendored include/fftw3.h is enough to compile it) and links it, this file and the five
// reference translation units that need no FFTW into oracle/_ref/libref_mfcccpu.so with
// -Wl,--gc-sections.  The only code in mfcccpu.cpp that references libfftw3f is
//     MfccCpu::MfccCpu / ~MfccCpu   (fftwf_alloc_*, fftwf_plan_many_dft_r2c, fftwf_free, ...; mfcccpu.cpp:109-116,170-173)
//     MfccCpu::fft                  (fftwf_execute, mfcccpu.cpp:187-190) and its two callers set_input / flush
// Nothing here references them (nor the vtable, whose only user is the constructor), so the
// linker drops those sections and NO fftwf_* symbol remains undefined in the library: no FFTW,
// and no stand-in for it, is involved.  `make ref` checks that with nm.
//
// What runs from the reference, unmodified, through QUALIFIED (non-virtual) calls:
//     MfccCpu::refresh_filters  mfcccpu.cpp:24-60     MfccCpu::normalize        :265-282
//     MfccCpu::filter           :192-220              MfccCpu::apply            :371-425
//     MfccCpu::dct              :222-232              MfccCpu::get_output_data  :427-444 (+ get_output :62-71)
//     MfccCpu::do_delta         :234-263              MfccCpu::set_window       :182-185
//   + the real ParamBase / MfccBase constructors (parambase.cpp:4-14, mfccbase.cpp:3-31) and the real
//     SegmenterCPU / DeltaCPU / NormalizerCPU members.
// What does NOT run and is stated here instead (each line cites what it follows):
//     * the constructor body mfcccpu.cpp:94-157: capacity arithmetic, buffer allocation, the
//       DCT-II + lifter matrix (:118-136).  The object is built on raw zeroed storage.
//     * set_input / flush (:338-345, :350-368): four lines of bookkeeping around the real
//       SegmenterCPU calls; the FFT result is WRITTEN INTO m_fft BY THE CALLER (tests compute the
//       DFT in double precision and round to float: the transform FFTW is defined to compute).
// This file contains no signal arithmetic of its own apart from the DCT matrix block.
//
// NB (toolchain): g++ binds the reference's unqualified log/exp/atan/sin/cos/sqrt on floats
// (mfcccpu.cpp:21-22,37,203,212) to the C double functions, the reference's own toolchain (MSVC)
// to the float overloads.  oracle/mfcc_oracle.c has a switch for either binding; tests compare
// bit for bit under the g++ binding and report what the MSVC binding moves.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <stdexcept>

// The members the functions read are private/protected; the translation unit that DEFINES the
// functions is compiled untouched, only this door looks inside.  Access keywords do not change
// the object layout under the Itanium ABI (members are laid out in declaration order).
#define private public
#define protected public
#include "mfcccpu.h"
#undef private
#undef protected

namespace {

struct Probe : public MfccBase {
    using MfccBase::MfccBase;
    void set_window(const float *) override {}
    int set_input(const short *, int) override { return 0; }
    int flush() override { return 0; }
    void apply() override {}
    void get_output_data(float *, int) override {}
};
static_assert(sizeof(Probe) <= sizeof(MfccCpu), "MfccBase subobject must fit the MfccCpu storage");

struct Handle {
    void *storage;
    MfccCpu *m;
    int cap_rows; // allocation rows (reference: m_window_limit; here + slack, SURVEY B8 / DESIGN B8)
};

int error_code(const std::exception &e)
{
    const char *w = e.what();
    if (std::strstr(w, "buffer is too small")) return -1;
    if (std::strstr(w, "window count is too small")) return -2;
    if (std::strstr(w, "Processed samples")) return -3;
    if (std::strstr(w, "Window count too high")) return -4;
    return -9;
}

} // namespace

extern "C" {

void *refm_new(int input_buffer_size, int window_size, int shift, int num_banks, float sample_rate, float low_freq,
               float high_freq, int ceps_len, int want_c0, float lift_coef, int norm, int dyn, int delta_l1,
               int delta_l2, int norm_after_dyn)
{
    Handle *h = new Handle();
    h->storage = std::calloc(1, sizeof(MfccCpu));
    // real ParamBase + MfccBase constructors on the front of the storage (single inheritance: the
    // MfccBase subobject of an MfccCpu starts at offset 0)
    new (h->storage) Probe(input_buffer_size, window_size, shift, num_banks, sample_rate, low_freq, high_freq,
                           ceps_len, want_c0 != 0, lift_coef, (Normalizer::norm_t)norm, (ParamBase::dyn_t)dyn,
                           delta_l1, delta_l2, norm_after_dyn != 0);
    MfccCpu *m = reinterpret_cast<MfccCpu *>(h->storage);
    h->m = m;
    new (&m->segmenter) SegmenterCPU();
    new (&m->normalizer) NormalizerCPU();
    new (&m->normalizer_delta) NormalizerCPU();
    new (&m->normalizer_acc) NormalizerCPU();
    new (&m->delta) DeltaCPU();
    new (&m->delta_acc) DeltaCPU();

    // ---- constructor body, mfcccpu.cpp:94-157, stated (the real one needs libfftw3f) ----
    if (m->m_dyn != ParamBase::DYN_NONE) // :95-98
        m->m_window_limit = m->m_input_window_limit + 2 + 3 * (m->m_delta_l1 + m->m_delta_l2);
    else { // :99-103
        m->m_delta_l1 = m->m_delta_l2 = 0;
        m->m_window_limit = m->m_input_window_limit + 2;
    }
    // The reference sizes every buffer from m_window_limit and a steady-state block can overrun
    // that; the allocations below carry slack rows, m_window_limit keeps its value.
    const int cap = m->m_window_limit + m->m_window_size / m->m_shift + 4;
    h->cap_rows = cap;
    m->m_buffer_size = m->m_window_limit * m->m_shift + m->m_window_size - m->m_shift; // :104
    const int D = m->m_delta_l1 + m->m_delta_l2;
    m->segmenter.init(m->m_window_size, m->m_shift, cap + 2 * (m->m_window_size / m->m_shift + 1), D); // :107 (+ slack)
    m->m_window_size2 = m->segmenter.m_window_size2; // :94 -- the segmenter's own ceil2 of the same value (segmentercpu.cpp:39)
    m->m_data_length = m->m_window_limit * m->m_window_size2;                               // :105
    m->m_data = (float *)std::calloc((size_t)cap * m->m_window_size2, sizeof(float));       // :109-110 (zeroed once)
    m->m_fft = (fftwf_complex *)std::calloc((size_t)cap * m->m_window_size2, sizeof(fftwf_complex)); // :111
    m->m_mel_energies = new float[(size_t)m->m_num_banks * cap]();                          // :112
    m->m_fft_plan = nullptr;                                                                 // :114 never created
    if (m->m_ceps_len > 0) { // :118-136
        m->m_mfcc = new float[(size_t)m->m_dct_len * cap]();
        m->m_dct_matrix = new float[m->m_num_banks * m->m_dct_len];
        std::memset(m->m_dct_matrix, 0, m->m_num_banks * m->m_dct_len * sizeof(float));
        float normfact = sqrt(2.0 / m->m_num_banks);
        for (int iy = 0; iy < m->m_num_banks; iy++)
            for (int ix = 1; ix <= m->m_ceps_len; ix++) {
                float lifter = (1 + lift_coef / 2 * sinf((float)M_PI * (float)ix / lift_coef));
                m->m_dct_matrix[m->m_dct_len * iy + ix - 1] =
                    lifter * normfact * cosf((float)M_PI * ix * (iy + 0.5f) / m->m_num_banks);
            }
        if (m->m_want_c0)
            for (int iy = 0; iy < m->m_num_banks; iy++) m->m_dct_matrix[m->m_dct_len * iy + m->m_ceps_len] = normfact;
    }
    const int cols = m->m_ceps_len > 0 ? m->m_dct_len : m->m_num_banks;
    if (m->m_norm != Normalizer::NORM_NONE) { // :137-145
        m->normalizer.init(m->m_norm, cols);
        if (m->m_dyn == ParamBase::DYN_DELTA || m->m_dyn == ParamBase::DYN_ACC) m->normalizer_delta.init(m->m_norm, cols);
        if (m->m_dyn == ParamBase::DYN_ACC) m->normalizer_acc.init(m->m_norm, cols);
    }
    if (m->m_dyn != ParamBase::DYN_NONE) { // :146-157
        int rows = cap + 2 * D;
        m->delta.init(cols, cap + 2 * m->m_delta_l2, m->m_delta_l1);
        if (m->m_dyn == ParamBase::DYN_ACC) m->delta_acc.init(cols, cap, m->m_delta_l2);
        m->m_delta_in = new float[(size_t)cols * rows]();
    }
    m->MfccCpu::refresh_filters(); // :159 (real)
    return h;
}

void refm_free(void *p)
{
    Handle *h = static_cast<Handle *>(p);
    MfccCpu *m = h->m;
    // as ~MfccCpu (:164-179) minus the fftwf_* calls
    m->segmenter.cleanup();
    m->normalizer.cleanup();
    m->normalizer_delta.cleanup();
    m->normalizer_acc.cleanup();
    m->delta.cleanup();
    m->delta_acc.cleanup();
    std::free(m->m_data);
    std::free(m->m_fft);
    delete[] m->m_filters; // refresh_filters leaks every earlier pair (:26-28); only the last is freed here too
    delete[] m->m_filter_beg;
    delete[] m->m_mel_energies;
    delete[] m->m_dct_matrix;
    delete[] m->m_mfcc;
    delete[] m->m_delta_in;
    std::free(h->storage);
    delete h;
}

// ---- geometry the real constructors / members computed ----
int refm_input_buffer_size(void *p) { return static_cast<Handle *>(p)->m->get_input_buffer_size(); }
int refm_window_limit(void *p) { return static_cast<Handle *>(p)->m->m_window_limit; }
int refm_cap_rows(void *p) { return static_cast<Handle *>(p)->cap_rows; }
int refm_fft_size(void *p) { return static_cast<Handle *>(p)->m->m_window_size2; }
int refm_ewc(void *p, int samples) { return static_cast<Handle *>(p)->m->estimated_window_count(samples); }
int refm_output_width(void *p) { return static_cast<Handle *>(p)->m->MfccBase::get_output_data_width(); }
int refm_was_flushed(void *p) { return static_cast<Handle *>(p)->m->segmenter.was_flushed(); }
int refm_last_block(void *p) { return static_cast<Handle *>(p)->m->m_last_block; }

void refm_set_window(void *p, const float *w) { static_cast<Handle *>(p)->m->MfccCpu::set_window(w); }
void refm_set_alpha(void *p, float a) { static_cast<Handle *>(p)->m->set_alpha(a); }

// MfccCpu::set_input, mfcccpu.cpp:338-345, WITHOUT its fft() call: on return > 0 the caller must put the
// transform of frame rows [0, *wcnd) into refm_fft().
int refm_set_input_nofft(void *p, const short *data, int samples, int *wcnd)
{
    MfccCpu *m = static_cast<Handle *>(p)->m;
    *wcnd = 0;
    if (samples > m->m_input_buffer_size) return -1; // :338-339
    int window_count = 0, window_count_no_delta = 0;
    try {
        m->segmenter.set_input(data, m->m_data, samples, window_count, window_count_no_delta); // :341 (real)
    } catch (const std::exception &e) {
        return error_code(e);
    }
    if (window_count <= 0) return 0; // :342-343
    *wcnd = window_count_no_delta;   // :344 fft(window_count_no_delta)
    return window_count;
}

// MfccCpu::flush, mfcccpu.cpp:348-369, WITHOUT its fft() call.
int refm_flush_nofft(void *p, int *wcnd)
{
    MfccCpu *m = static_cast<Handle *>(p)->m;
    *wcnd = 0;
    if (m->m_last_block) return 0; // :350-351
    m->m_last_block = true;        // :352
    int window_count = 0, window_count_no_delta = 0;
    m->segmenter.flush(m->m_data, window_count, window_count_no_delta); // :364 (real)
    if (window_count <= 0) return 0;
    *wcnd = window_count_no_delta;
    return window_count;
}

float *refm_data(void *p) { return static_cast<Handle *>(p)->m->m_data; }               // [cap][W2] frames
float *refm_fft(void *p) { return (float *)static_cast<Handle *>(p)->m->m_fft; }        // [cap][W2] complex
float *refm_mel(void *p) { return static_cast<Handle *>(p)->m->m_mel_energies; }        // [cap][nb]
float *refm_mfcc(void *p) { return static_cast<Handle *>(p)->m->m_mfcc; }               // [cap][dct_len] or NULL
float *refm_dct_matrix(void *p) { return static_cast<Handle *>(p)->m->m_dct_matrix; }   // [nb][dct_len] or NULL
float *refm_filters(void *p) { return static_cast<Handle *>(p)->m->m_filters; }         // [2][W2]
int *refm_filter_beg(void *p) { return static_cast<Handle *>(p)->m->m_filter_beg; }     // [nb+2]
float *refm_delta_in(void *p) { return static_cast<Handle *>(p)->m->m_delta_in; }
float *refm_delta_out(void *p) { return static_cast<Handle *>(p)->m->delta.get_output_buffer(); }
float *refm_acc_out(void *p) { return static_cast<Handle *>(p)->m->delta_acc.get_output_buffer(); }
// normaliser statistics: group 0/1/2 = static / delta / acc instance; which 0 = mean, 1 = var (CVN), 2 = minmax
float *refm_norm_stats(void *p, int group, int which)
{
    MfccCpu *m = static_cast<Handle *>(p)->m;
    NormalizerCPU *n = group == 0 ? &m->normalizer : group == 1 ? &m->normalizer_delta : &m->normalizer_acc;
    return which == 0 ? n->m_mean : which == 1 ? n->m_var : n->m_minmax;
}

// ---- the real member functions ----
void refm_refresh_filters(void *p)
{
    MfccCpu *m = static_cast<Handle *>(p)->m;
    delete[] m->m_filters; // the reference leaks these on every call (:26-28); free them here, results unchanged
    delete[] m->m_filter_beg;
    m->MfccCpu::refresh_filters();
}
static void drop_tables(MfccCpu *m)
{
    // filter() re-allocates both tables through refresh_filters (:194) without freeing
    delete[] m->m_filters;
    delete[] m->m_filter_beg;
    m->m_filters = nullptr;
    m->m_filter_beg = nullptr;
}
void refm_filter(void *p, int window_count)
{
    MfccCpu *m = static_cast<Handle *>(p)->m;
    drop_tables(m);
    m->MfccCpu::filter(window_count);
}
void refm_dct(void *p, int window_count) { static_cast<Handle *>(p)->m->MfccCpu::dct(window_count); }
void refm_do_delta(void *p, int window_count, int first_call, int last_call)
{
    static_cast<Handle *>(p)->m->MfccCpu::do_delta(window_count, first_call != 0, last_call != 0);
}
void refm_normalize(void *p, int window_count, int use_last_stats)
{
    static_cast<Handle *>(p)->m->MfccCpu::normalize(window_count, use_last_stats != 0);
}
int refm_apply(void *p)
{
    MfccCpu *m = static_cast<Handle *>(p)->m;
    drop_tables(m);
    try {
        m->MfccCpu::apply();
    } catch (const std::exception &e) {
        return error_code(e);
    }
    return 0;
}
int refm_get_output_data(void *p, float *out, int window_count)
{
    try {
        static_cast<Handle *>(p)->m->MfccCpu::get_output_data(out, window_count);
    } catch (const std::exception &e) {
        return error_code(e);
    }
    return 0;
}

} // extern "C"
