/*
 * mfcc_oracle.h -- CPU restatement of the reference MFCC front end (TEST INFRASTRUCTURE).
 *
 * This is the parity checker for the HIP path, not a product code path.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * It restates, in plain C99, the reference's CPU back end:
 *   mfcccpu.cpp (orchestration, mel table, filter, DCT, delta staging, output interleave),
 *   segmentercpu.cpp, deltacpu.cpp, normalizercpu.cpp, parambase.cpp, mfccbase.cpp.
 * Each function below cites the reference file:line it follows.
 *
 * PINNING STATUS: pinned to the reference's own compiled code, function by function.
 *   - oracle/_ref/libref_stages.so: the reference's parambase / mfccbase / segmentercpu / deltacpu / normalizercpu
 *     translation units, compiled in place from /root/reference;
 *   - oracle/_ref/libref_mfcccpu.so (round 3): the reference's mfcccpu.cpp as well -- MfccCpu::refresh_filters, filter, dct,
 *     do_delta, normalize, apply, get_output_data run as compiled; only its constructor / destructor / fft(), the sole
 *     users of libfftw3f (absent from this image), are never referenced and are dropped by the linker (--gc-sections; no
 *     fftwf_* symbol remains, no stand-in for FFTW is written).  See oracle/ref_mfcccpu_shim.cpp.
 *   Both libraries exist in TWO builds, one per binding of the reference's UNQUALIFIED libm calls on floats
 *   (mfcccpu.cpp:21-22,37,203,212; abs in normalizercpu.cpp:66):
 *     - _ref/libref_mfcccpu_f32.so, libref_stages_f32.so (round 4): compiled with `-include math.h -include stdlib.h`, which
 *       makes those calls select the FLOAT overloads -- the selection the reference's own toolchain (MSVC) makes.  This
 *       restatement's DEFAULT arithmetic is BIT-IDENTICAL to that build on every case of tests/refcases.py (MINMAX
 *       included), on filter() + dct() over synthetic spectra and on a randomised sweep (tests/test_ref_mfcccpu.py);
 *       committed vectors tests/golden/ref_mfcccpu_vectors_f32.npz.  Same overload selection as MSVC, not the same C
 *       runtime: logf / expf / atanf / sinf / cosf / sqrtf are glibc's.
 *     - _ref/libref_mfcccpu.so, libref_stages.so: plain g++ (the C double functions, int abs(int)); this restatement under
 *       orc_set_libm_binding(o, 1) is BIT-IDENTICAL to it (tests/golden/ref_mfcccpu_vectors.npz).
 *   Not pinned by execution, "by definition" instead: the transform at the FFTW call site (mfcccpu.cpp:114,187-190 -- the
 *   DFT, checked against numpy float64) and the DCT + lifter matrix block of the constructor (mfcccpu.cpp:118-136, explicit
 *   sinf / cosf: no toolchain question).
 */
#ifndef MFCC_ORACLE_H
#define MFCC_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_NORM_NONE = 0, ORC_NORM_CMN = 1, ORC_NORM_CVN = 2, ORC_NORM_MINMAX = 3 }; /* normalizer.h:5 */
enum { ORC_DYN_NONE = 0, ORC_DYN_DELTA = 1, ORC_DYN_ACC = 2 };                         /* parambase.h:9 */

enum {
    ORC_OK = 0,
    ORC_ERR_BUFFER_TOO_SMALL = -1, /* "Can't process data, buffer is too small"        mfcccpu.cpp:339 */
    ORC_ERR_WINDOW_COUNT = -2,     /* "Can't process data, window count is too small"  segmentercpu.cpp:65, mfcccpu.cpp:396 */
    ORC_ERR_PROCESSED = -3,        /* "Processed samples <= 0, this should never happen" segmentercpu.cpp:71 */
    ORC_ERR_WINDOW_HIGH = -4,      /* "Window count too high"                          mfcccpu.cpp:430 */
    ORC_ERR_CONFIG = -5
};

typedef struct {
    /* MfccBase ctor arguments, mfccbase.h:21-35 */
    int input_buffer_size;
    int window_size;
    int shift;
    int num_banks;
    float sample_rate;
    float low_freq;
    float high_freq;
    int ceps_len;
    int want_c0;
    float lift_coef;
    int norm;
    int dyn;
    int delta_l1;
    int delta_l2;
    int norm_after_dyn;
    /* oracle-only knob: 0 = double-precision FFT rounded to float (checker default),
     * 1 = float32 half-size real FFT (used for the timed CPU baseline). */
    int fft_mode;
} orc_config;

typedef struct orc_mfcc orc_mfcc;

orc_mfcc *orc_create(const orc_config *cfg);
void orc_destroy(orc_mfcc *o);

void orc_set_window(orc_mfcc *o, const float *window);
int orc_set_input(orc_mfcc *o, const short *data, int samples); /* >=0 frames, <0 error */
int orc_flush(orc_mfcc *o);
void orc_set_alpha(orc_mfcc *o, float alpha);
int orc_apply(orc_mfcc *o);
int orc_get_output_data_width(const orc_mfcc *o);
int orc_get_output_data(orc_mfcc *o, float *data_out, int window_count);
int orc_get_input_buffer_size(const orc_mfcc *o);
int orc_estimated_window_count(const orc_mfcc *o, int samples);
int orc_window_limit(const orc_mfcc *o);
int orc_fft_size(const orc_mfcc *o);
/* When nonzero (default), get_output_data reads the static block at the offset the reference
 * uses (was_flushed() ? 0 : D), which reproduces reference behaviour B1 for files
 * consumed by exactly one set_input.  When zero the flush block reads the correct rows. */
void orc_set_bug_compat(orc_mfcc *o, int on);
/* How the reference's unqualified log/exp/atan/sin/cos/sqrt on floats (mfcccpu.cpp:21-22,37,203,212) bind:
 * 0 (default) = float overloads, as under the reference's own toolchain (MSVC); 1 = the C double functions, as
 * under g++ -- the binding of oracle/_ref/libref_mfcccpu.so, against which this mode is compared bit for bit. */
void orc_set_libm_binding(orc_mfcc *o, int use_double);

/* stage taps (valid after set_input/flush resp. apply), for stage-by-stage parity tests */
const float *orc_tap_frames(const orc_mfcc *o);   /* [window_limit][W2]            */
const float *orc_tap_fft(const orc_mfcc *o);      /* [window_limit][W2] complex as float pairs */
const float *orc_tap_mel(const orc_mfcc *o);      /* [window_limit][num_banks]     */
const float *orc_tap_mfcc(const orc_mfcc *o);     /* [window_limit][dct_len] or NULL */
const float *orc_tap_filters(const orc_mfcc *o);  /* [2][W2]                       */
const int *orc_tap_filter_beg(const orc_mfcc *o); /* [num_banks+2]                 */
const float *orc_tap_dct_matrix(const orc_mfcc *o); /* [num_banks][dct_len] or NULL */
/* normaliser statistics after apply(): group 0/1/2 = static/delta/delta-delta instance, which 0 = mean, 1 = multiplier;
 * cols floats each (normalizercpu.cpp:22-67) */
const float *orc_tap_norm_stats(const orc_mfcc *o, int group, int which);

/* MfccCpu::filter (mfcccpu.cpp:192-220) / MfccCpu::dct (:222-232) alone, on the first `rows` rows of the spectrum
 * buffer orc_tap_fft points at (tests write caller-made spectra there) */
void orc_stage_filter(orc_mfcc *o, int rows);
void orc_stage_dct(orc_mfcc *o, int rows);

/* ---- standalone stage functions (compared 1:1 with the real reference objects in oracle/_ref) ---- */

/* parambase.cpp:16-19 */
int orc_ewc(int samples, int window_size, int shift);
/* mfccbase.cpp:33-43 */
int orc_output_width(int num_banks, int ceps_len, int want_c0, int dyn);
/* deltacpu.cpp:16-29 : data has window_count + 2*delta_size rows of dim floats */
void orc_delta_apply(const float *data, int dim, int window_count, int delta_size, float *out);
/* normalizercpu.cpp:22-89 : in place; mean/var/minmax are the persistent stats (dim floats each) */
void orc_normalize(int norm_type, float *data, int dim, int window_count, int use_last_stats,
                   float *mean, float *var, float *minmax);
/* segmentercpu.cpp:17-28 */
void orc_segment(const short *pcm, const float *window, int window_size, int window_size2,
                 int shift, int window_count, float *data_out);
/* forward real DFT of `howmany` rows of length n (power of two); out[b*n + k] complex for
 * k = 0..n/2 (rest of the row untouched) -- semantics of the fftwf_plan_many_dft_r2c call at
 * mfcccpu.cpp:114. mode as orc_config.fft_mode. */
void orc_rfft_rows(const float *in, float *out_complex, int n, int howmany, int mode);

/* Whole-utterance convenience used by tests/bench: runs the reference call sequence
 * (ASR_OCL.cpp:149-301: set_window; loop{set_input; apply; get_output_data}; flush; apply;
 * get_output_data) with blocks of at most `block_samples` samples.  `out` must hold
 * orc_ewc(samples)*width floats.  Returns total frames or <0. */
int orc_run_utterance(const orc_config *cfg, const float *window, float alpha, int bug_compat,
                      const short *pcm, int samples, int block_samples, float *out);

/* Timed CPU-baseline helper: n_utt utterances of utt_samples samples each, contiguous in pcm,
 * processed with n_threads OpenMP threads (one extractor instance per thread, utterances
 * sharded round-robin).  Returns total frames written to out ([n_utt*frames_per_utt][width]). */
long long orc_run_batch(const orc_config *cfg, const float *window, const short *pcm, int n_utt,
                        int utt_samples, float *out, int n_threads);

/* Timed variant (see mfcc_oracle.c): setup excluded, batch repeated `reps` times. */
long long orc_bench_batch(const orc_config *cfg, const float *window, const short *pcm, int n_utt,
                          int utt_samples, int n_threads, int reps, double *seconds);

#ifdef __cplusplus
}
#endif
#endif
