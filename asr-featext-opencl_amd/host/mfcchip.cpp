// mfcchip.cpp -- MfccHip: the reference's parameterizer interface forwarded to the C ABI.
#include <cmath>
#include <string>

#include "../../include/mfx.h"
#include "afet_param.h"

// ---- ParamBase / MfccBase bookkeeping (parambase.cpp:4-19, mfccbase.cpp:3-43); with the reference's own headers
// the reference's own parambase.cpp / mfccbase.cpp are linked instead ----
#ifndef AFET_USE_REFERENCE_HEADERS

ParamBase::ParamBase(int input_buffer_size, int window_size, int shift, Normalizer::norm_t norm, dyn_t dyn)
    : m_window_size(window_size), m_shift(shift), m_alpha(1), m_norm(norm), m_dyn(dyn), m_last_block(false)
{
    m_input_window_limit = estimated_window_count(input_buffer_size);
    m_input_buffer_size = m_input_window_limit * m_shift + m_window_size - m_shift;
}

int ParamBase::estimated_window_count(int samples) const
{
    return (int)std::floor(float(samples - (m_window_size - m_shift)) / m_shift);
}

MfccBase::MfccBase(int input_buffer_size, int window_size, int shift, int num_banks, float sample_rate,
                   float low_freq, float high_freq, int ceps_len, bool want_c0, float lift_coef,
                   Normalizer::norm_t norm, dyn_t dyn, int delta_l1, int delta_l2, bool norm_after_dyn)
    : ParamBase(input_buffer_size, window_size, shift, norm, dyn),
      m_num_banks(num_banks),
      m_ceps_len(ceps_len),
      m_dct_len(want_c0 ? ceps_len + 1 : ceps_len),
      m_delta_l1(dyn != DYN_NONE ? delta_l1 : 0),
      m_delta_l2(dyn == DYN_ACC ? delta_l2 : 0),
      m_sample_rate(sample_rate),
      m_low_freq(low_freq),
      m_high_freq(high_freq),
      m_lift_coef(lift_coef),
      m_want_c0(want_c0),
      m_norm_after_dyn(norm_after_dyn)
{
}

int MfccBase::get_output_data_width() const
{
    const int cols = m_ceps_len > 0 ? m_dct_len : m_num_banks;
    return m_dyn == DYN_ACC ? 3 * cols : m_dyn == DYN_DELTA ? 2 * cols : cols;
}

#endif // AFET_USE_REFERENCE_HEADERS

// ---- MfccHip ----

MfccHip::MfccHip(int input_buffer_size, int window_size, int shift, int num_banks, float sample_rate,
                 float low_freq, float high_freq, int ceps_len, bool want_c0, float lift_coef,
                 Normalizer::norm_t norm, dyn_t dyn, int delta_l1, int delta_l2, bool norm_after_dyn, int hip_device,
                 bool bug_compat, int engine)
    : MfccBase(input_buffer_size, window_size, shift, num_banks, sample_rate, low_freq, high_freq, ceps_len, want_c0,
               lift_coef, norm, dyn, delta_l1, delta_l2, norm_after_dyn),
      m_handle(nullptr)
{
    mfx_config cfg = {};
    cfg.input_buffer_size = input_buffer_size;
    cfg.window_size = window_size;
    cfg.shift = shift;
    cfg.num_banks = num_banks;
    cfg.sample_rate = sample_rate;
    cfg.low_freq = low_freq;
    cfg.high_freq = high_freq;
    cfg.ceps_len = ceps_len;
    cfg.want_c0 = want_c0 ? 1 : 0;
    cfg.lift_coef = lift_coef;
    cfg.norm = (int)norm;
    cfg.dyn = (int)dyn;
    cfg.delta_l1 = delta_l1;
    cfg.delta_l2 = delta_l2;
    cfg.norm_after_dyn = norm_after_dyn ? 1 : 0;
    cfg.bug_compat = bug_compat ? 1 : 0;
    cfg.engine = engine;
    const int rc = mfx_create(&cfg, hip_device, &m_handle);
    if (rc != MFX_OK) throw std::runtime_error(std::string("MfccHip: ") + mfx_status_string(rc));
}

MfccHip::~MfccHip() { mfx_destroy(m_handle); }

void MfccHip::check(int status) const
{
    if (status == MFX_OK) return;
    // the library's message for this call when it left one (the reference's fixed strings among them), else the
    // generic text of the status code
    const char *m = mfx_last_error(m_handle);
    throw std::runtime_error((m && *m) ? m : mfx_status_string(status));
}

void MfccHip::set_window(const float *window) { check(mfx_set_window(m_handle, window)); }

int MfccHip::set_input(const short *data, int samples)
{
    int32_t frames = 0;
    check(mfx_set_input(m_handle, data, samples, &frames));
    m_last_block = false;
    return frames;
}

int MfccHip::flush()
{
    int32_t frames = 0;
    check(mfx_flush(m_handle, &frames));
    m_last_block = true;
    return frames;
}

void MfccHip::apply()
{
    // the caller's set_alpha (ParamBase's, not virtual) only stored m_alpha: it takes effect here, as in the
    // reference, whose apply() rebuilds the filterbank from m_alpha every time (mfcccpu.cpp:194)
    check(mfx_set_alpha(m_handle, m_alpha));
    check(mfx_apply(m_handle));
}

void MfccHip::get_output_data(float *data_out, int window_count)
{
    check(mfx_get_output_data(m_handle, data_out, window_count));
}

void MfccHip::apply_alphas(const float *alphas, int n_alpha) { check(mfx_apply_alphas(m_handle, alphas, n_alpha)); }

void MfccHip::get_output_data_alpha(int alpha_index, float *data_out, int window_count)
{
    check(mfx_get_output_data_alpha(m_handle, alpha_index, data_out, window_count));
}

int MfccHip::max_frames_out() const { return mfx_max_frames_out(m_handle); }

long long MfccHip::batch_plan(int n_utt, const long long *offsets, const long long *lengths, long long *out_rows)
{
    static_assert(sizeof(long long) == sizeof(int64_t), "64-bit long long");
    int64_t total = 0;
    check(mfx_batch_plan(m_handle, n_utt, (const int64_t *)offsets, (const int64_t *)lengths, (int64_t *)out_rows, &total));
    return (long long)total;
}

void MfccHip::batch_run_host(const short *pcm, long long samples_total, float *out)
{
    check(mfx_batch_run_host(m_handle, pcm, (int64_t)samples_total, out));
}

long long MfccHip::batch_frames(long long samples) const { return (long long)mfx_batch_frames(m_handle, (int64_t)samples); }
