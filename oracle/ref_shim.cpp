// ref_shim.cpp -- extern "C" door onto the REAL reference objects (TEST INFRASTRUCTURE ONLY).
//
// Compiled by oracle/Makefile together with the reference's own translation units, which are
// read in place from /root/reference (never copied):
//     parambase.cpp mfccbase.cpp segmentercpu.cpp deltacpu.cpp normalizercpu.cpp
// into oracle/_ref/libref_stages.so.  (mfcccpu.cpp has its own door, ref_mfcccpu_shim.cpp ->
// oracle/_ref/libref_mfcccpu.so, which links it without its FFTW-calling constructor.)  The shim adds
// no arithmetic of its own; it only forwards calls so that ctypes can reach the C++ classes.
#include <stdexcept>

#include "deltacpu.h"
#include "mfccbase.h"
#include "normalizercpu.h"
#include "segmentercpu.h"

namespace {
// ParamBase is abstract (parambase.h:27-32); the probe supplies empty bodies so that the
// reference's own ParamBase/MfccBase constructors and accessors can be exercised.
struct Probe : public MfccBase {
    using MfccBase::MfccBase;
    void set_window(const float *) override {}
    int set_input(const short *, int) override { return 0; }
    int flush() override { return 0; }
    void apply() override {}
    void get_output_data(float *, int) override {}
};
} // namespace

extern "C" {

// ---- ParamBase / MfccBase (parambase.cpp, mfccbase.cpp) ----
void *ref_base_new(int input_buffer_size, int window_size, int shift, int num_banks, float sample_rate,
                   float low_freq, float high_freq, int ceps_len, int want_c0, float lift_coef, int norm,
                   int dyn, int delta_l1, int delta_l2, int norm_after_dyn)
{
    return new Probe(input_buffer_size, window_size, shift, num_banks, sample_rate, low_freq, high_freq,
                     ceps_len, want_c0 != 0, lift_coef, (Normalizer::norm_t)norm, (ParamBase::dyn_t)dyn,
                     delta_l1, delta_l2, norm_after_dyn != 0);
}
void ref_base_free(void *p) { delete static_cast<Probe *>(p); }
int ref_base_input_buffer_size(void *p) { return static_cast<Probe *>(p)->get_input_buffer_size(); }
int ref_base_ewc(void *p, int samples) { return static_cast<Probe *>(p)->estimated_window_count(samples); }
int ref_base_output_width(void *p) { return static_cast<Probe *>(p)->get_output_data_width(); }

// ---- SegmenterCPU (segmentercpu.cpp) ----
void *ref_seg_new(int window_size, int shift, int window_limit, int deltasize)
{
    SegmenterCPU *s = new SegmenterCPU();
    s->init(window_size, shift, window_limit, deltasize);
    return s;
}
void ref_seg_free(void *p)
{
    SegmenterCPU *s = static_cast<SegmenterCPU *>(p);
    s->cleanup();
    delete s;
}
void ref_seg_set_window(void *p, const float *w) { static_cast<SegmenterCPU *>(p)->set_window(w); }
// returns 0, or -2 / -3 for the two std::runtime_error sites (segmentercpu.cpp:65,71)
int ref_seg_set_input(void *p, const short *in, float *out, int samples, int *wc, int *wcnd)
{
    try {
        static_cast<SegmenterCPU *>(p)->set_input(in, out, samples, *wc, *wcnd);
    } catch (const std::runtime_error &e) {
        return e.what()[0] == 'C' ? -2 : -3;
    }
    return 0;
}
void ref_seg_flush(void *p, float *out, int *wc, int *wcnd) { static_cast<SegmenterCPU *>(p)->flush(out, *wc, *wcnd); }
int ref_seg_remaining(void *p) { return static_cast<SegmenterCPU *>(p)->get_remaining_samples(); }
int ref_seg_samples(void *p) { return static_cast<SegmenterCPU *>(p)->get_samples(); }
int ref_seg_is_flushed(void *p) { return static_cast<SegmenterCPU *>(p)->is_flushed(); }
int ref_seg_was_flushed(void *p) { return static_cast<SegmenterCPU *>(p)->was_flushed(); }
int ref_seg_ewc(void *p, int samples) { return static_cast<SegmenterCPU *>(p)->estimated_window_count(samples); }

// ---- DeltaCPU (deltacpu.cpp) ----
void *ref_delta_new(int dim, int window_limit, int delta_size)
{
    DeltaCPU *d = new DeltaCPU();
    d->init(dim, window_limit, delta_size);
    return d;
}
void ref_delta_free(void *p)
{
    DeltaCPU *d = static_cast<DeltaCPU *>(p);
    d->cleanup();
    delete d;
}
void ref_delta_apply(void *p, const float *data, int window_count) { static_cast<DeltaCPU *>(p)->apply(data, window_count); }
float *ref_delta_output(void *p) { return static_cast<DeltaCPU *>(p)->get_output_buffer(); }

// ---- NormalizerCPU (normalizercpu.cpp) ----
void *ref_norm_new(int norm_type, int dim)
{
    NormalizerCPU *n = new NormalizerCPU();
    n->init((Normalizer::norm_t)norm_type, dim);
    return n;
}
void ref_norm_free(void *p)
{
    NormalizerCPU *n = static_cast<NormalizerCPU *>(p);
    n->cleanup();
    delete n;
}
void ref_norm_normalize(void *p, float *data, int window_count, int use_last_stats)
{
    static_cast<NormalizerCPU *>(p)->normalize(data, window_count, use_last_stats != 0);
}

} // extern "C"
