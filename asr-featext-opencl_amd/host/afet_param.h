// afet_param.h -- source-compatible mirror of the reference's parameterizer interface, for C++
// callers that want to swap `new MfccOpenCL(...)` for `new MfccHip(...)` and change nothing else.
//
// Mirrors (interface only -- names, argument order and meaning, defaults, error behaviour):
//     Normalizer::norm_t          normalizer.h:5
//     class ParamBase             parambase.h:6-33   (+ parambase.cpp:4-19)
//     class MfccBase              mfccbase.h:6-52    (+ mfccbase.cpp:3-43)
//     class MfccOpenCL            mfccopencl.h:12-74 -> class MfccHip here
// All compute happens behind the C ABI of include/mfx.h (libmfcchip.so); nothing in this header
// or in mfcchip.cpp does arithmetic on samples or features.
#ifndef AFET_PARAM_H
#define AFET_PARAM_H

#include <stdexcept>

// -DAFET_USE_REFERENCE_HEADERS: take ParamBase / MfccBase / Normalizer from the reference's OWN headers
// (parambase.h:6-33, mfccbase.h:6-52, normalizer.h:5; add -I<reference>) and link the reference's parambase.cpp /
// mfccbase.cpp -- this is how MfccHip drops into the reference tree itself (INTEGRATION.md section 1;
// tests/test_host.py::test_mfcchip_builds_against_reference_headers).  Without it the mirror below stands in.
#ifdef AFET_USE_REFERENCE_HEADERS
#include "mfccbase.h"
struct mfx_handle;
#else

struct mfx_handle;

namespace Normalizer {
enum norm_t { NORM_NONE, NORM_CMN, NORM_CVN, NORM_MINMAX };
}

// Abstract streaming feature extractor: set_window once, then per file
//   while (samples) { n = set_input(pcm, cnt); set_alpha(a); apply(); get_output_data(out, n); }
//   n = flush(); if (n > 0) { set_alpha(a); apply(); get_output_data(out, n); }
// exactly as the reference driver does (ASR_OCL.cpp:152-301).
class ParamBase {
public:
    enum dyn_t { DYN_NONE, DYN_DELTA, DYN_ACC };

    ParamBase(int input_buffer_size, int window_size, int shift, Normalizer::norm_t norm, dyn_t dyn);
    virtual ~ParamBase() {}

    // largest block set_input accepts: whole frames that fit in the requested size (parambase.cpp:12-13)
    int get_input_buffer_size() const { return m_input_buffer_size; }
    // floor(float(samples - (W - S)) / S), float32 arithmetic as the reference (parambase.cpp:16-19)
    int estimated_window_count(int samples) const;
    void set_alpha(float alpha) { m_alpha = alpha; }   // not virtual (parambase.h:25): apply() forwards m_alpha

    virtual void set_window(const float *window) = 0;
    virtual int set_input(const short *data, int samples) = 0;
    virtual int flush() = 0;
    virtual void apply() = 0;
    virtual int get_output_data_width() const = 0;
    virtual void get_output_data(float *data_out, int window_count) = 0;

protected:
    int m_input_buffer_size, m_input_window_limit, m_window_size, m_shift;
    float m_alpha;
    Normalizer::norm_t m_norm;
    dyn_t m_dyn;
    bool m_last_block;
};

class MfccBase : public ParamBase {
public:
    MfccBase(int input_buffer_size, int window_size, int shift, int num_banks, float sample_rate, float low_freq,
             float high_freq, int ceps_len, bool want_c0, float lift_coef,
             Normalizer::norm_t norm = Normalizer::NORM_NONE, dyn_t dyn = DYN_NONE, int delta_l1 = 1,
             int delta_l2 = 1, bool norm_after_dyn = true);
    virtual ~MfccBase() {}

    int get_output_data_width() const override;

protected:
    int m_num_banks, m_ceps_len, m_dct_len, m_delta_l1, m_delta_l2;
    float m_sample_rate, m_low_freq, m_high_freq, m_lift_coef;
    bool m_want_c0, m_norm_after_dyn;
};

#endif // AFET_USE_REFERENCE_HEADERS

// The MI355X back end.  `hip_device` takes the place of MfccOpenCL's trailing cl_device_id
// (mfccopencl.h:60).  Errors surface as std::runtime_error with the reference's messages.
class MfccHip : public MfccBase {
public:
    MfccHip(int input_buffer_size, int window_size, int shift, int num_banks, float sample_rate, float low_freq,
            float high_freq, int ceps_len, bool want_c0, float lift_coef,
            Normalizer::norm_t norm = Normalizer::NORM_NONE, dyn_t dyn = DYN_NONE, int delta_l1 = 1,
            int delta_l2 = 1, bool norm_after_dyn = true, int hip_device = 0, bool bug_compat = true, int engine = 0);
    ~MfccHip() override;
    MfccHip(const MfccHip &) = delete;
    MfccHip &operator=(const MfccHip &) = delete;

    // set_alpha is ParamBase's own (non-virtual, stores m_alpha): apply() hands m_alpha to the library
    void set_window(const float *window) override;
    int set_input(const short *data, int samples) override;
    int flush() override;
    void apply() override;
    void get_output_data(float *data_out, int window_count) override;

    // VTLN sweep over the current block's spectrum (the caller's alpha loop, ASR_OCL.cpp:236-243, in
    // one call); block i of the result is read with get_output_data_alpha(i, ...)
    void apply_alphas(const float *alphas, int n_alpha);
    void get_output_data_alpha(int alpha_index, float *data_out, int window_count);

    // Many files in one launch sequence (mfx_batch_plan / mfx_batch_run_host): utterance u = samples [offsets[u],
    // offsets[u] + lengths[u]) of one PCM array, its rows start at out_rows[u]; returns the total number of rows
    long long batch_plan(int n_utt, const long long *offsets, const long long *lengths, long long *out_rows);
    void batch_run_host(const short *pcm, long long samples_total, float *out);
    long long batch_frames(long long samples) const;

    // upper bound on the rows one set_input()/flush() can deliver (the reference's own bound,
    // estimated_window_count(get_input_buffer_size()), can be exceeded: SURVEY B6)
    int max_frames_out() const;
    mfx_handle *handle() const { return m_handle; }

private:
    void check(int status) const;
    mfx_handle *m_handle;
};

#endif // AFET_PARAM_H
