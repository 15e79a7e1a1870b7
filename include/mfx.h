/*
 * mfx.h -- C ABI of libmfcchip.so, the MI355X (gfx950) MFCC front end.
 *
 * This is the drop-in boundary for the reference's OpenCL back end (MfccOpenCL + SegmenterOpenCL +
 * clFFT/AppleFFT + NormalizerOpenCL + DeltaOpenCL).  Every entry point replaces one method of the
 * reference's parameterizer interface; the reference file:line it stands for is cited next to it
 * (paths relative to the reference repository).  Plain C types only: pointers, sizes, POD structs.
 *
 * Conventions
 *   - Every function returns MFX_OK (0) or a negative mfx_status; nothing throws.
 *   - mfx_last_error(h) returns the message for the last failure on that handle; the strings are
 *     the ones the reference throws as std::runtime_error, so a C++ wrapper can re-throw them.
 *   - A handle is bound to one HIP device and one stream and must be used from one thread at a
 *     time (the reference has no locking either: ASR_OCL.cpp:52,365-366).  Handles are independent.
 *   - Host pointers passed to the streaming calls are consumed before the call returns, so the
 *     caller may reuse one buffer for PCM-in and features-out as the reference driver does
 *     (ASR_OCL.cpp:160-161,231,243).
 *   - There is no CPU fallback: if the HIP runtime or a gfx950 device is missing, mfx_create fails.
 */
#ifndef MFX_H
#define MFX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: mfx_config.batch_norm_stats (default 0 = the reference's block statistics; version 1 libraries took them over all T
 *    rows) and mfx_config.engine / tail_split, all carved out of reserved[] -- a zeroed reserved[] is still valid. */
#define MFX_ABI_VERSION 2

typedef enum {
    MFX_OK = 0,
    MFX_ERR_BUFFER_TOO_SMALL = -1, /* "Can't process data, buffer is too small"           mfcccpu.cpp:339, mfccopencl.cpp:473 */
    MFX_ERR_WINDOW_COUNT = -2,     /* "Can't process data, window count is too small"     segmentercpu.cpp:65, mfcccpu.cpp:396 */
    MFX_ERR_PROCESSED = -3,        /* "Processed samples <= 0, this should never happen"  segmentercpu.cpp:71 */
    MFX_ERR_WINDOW_HIGH = -4,      /* "Window count too high"                             mfcccpu.cpp:430 */
    MFX_ERR_CONFIG = -5,           /* invalid mfx_config                                                     */
    MFX_ERR_DEVICE = -6,           /* HIP runtime / device error (message carries hipGetErrorString)         */
    MFX_ERR_ARG = -7,              /* NULL or out-of-range argument                                          */
    MFX_ERR_STATE = -8             /* call out of sequence (e.g. get_output_data before apply)               */
} mfx_status;

/* Normalizer::norm_t, normalizer.h:5 */
enum { MFX_NORM_NONE = 0, MFX_NORM_CMN = 1, MFX_NORM_CVN = 2, MFX_NORM_MINMAX = 3 };
/* ParamBase::dyn_t, parambase.h:9 */
enum { MFX_DYN_NONE = 0, MFX_DYN_DELTA = 1, MFX_DYN_ACC = 2 };

/* Constructor arguments of MfccBase (mfccbase.h:21-35) / MfccOpenCL (mfccopencl.h:45-60), in the
 * same order, followed by extensions (all zero = reference behaviour). */
typedef struct mfx_config {
    int32_t input_buffer_size; /* "sample_limit": max samples per set_input (ASR_OCL.cpp:132,563)    */
    int32_t window_size;       /* W, samples                                                         */
    int32_t shift;             /* S, samples                                                         */
    int32_t num_banks;         /* mel filters                                                        */
    float sample_rate;
    float low_freq;
    float high_freq;
    int32_t ceps_len;          /* 0 = output log mel energies                                        */
    int32_t want_c0;           /* c0 appended as LAST column (mfcccpu.cpp:133-135)                   */
    float lift_coef;
    int32_t norm;              /* MFX_NORM_*                                                         */
    int32_t dyn;               /* MFX_DYN_*                                                          */
    int32_t delta_l1;
    int32_t delta_l2;
    int32_t norm_after_dyn;
    /* ---- extensions ---- */
    int32_t fft_size;          /* 0 = ceil2(window_size) as the reference (mfcccpu.cpp:94); else a
                                  power of two >= window_size (zero padded)                          */
    int32_t channels;          /* 0/1 = mono; 2 = interleaved stereo, downmixed (L+R)>>1 (batch API) */
    int32_t bug_compat;        /* 1 = reproduce reference behaviour B1 (static rows of a flush that
                                  follows exactly one set_input are read D rows early,
                                  mfcccpu.cpp:439 + segmentercpu.cpp:97-106); 0 = correct rows       */
    int32_t batch_norm_stats;  /* batch entries, norm after dyn: 0 = statistics as the reference computes them for
                                  an utterance it consumes as one block -- over the T - D rows that block delivers,
                                  re-used for the D flush rows (mfcccpu.cpp:377-388,395-407, normalizercpu.cpp:22-27);
                                  1 = over all T rows of the utterance.  Reference parity holds for utterances of at
                                  most input_buffer_size samples (longer files are several blocks in the reference,
                                  each with its own statistics: use the streaming entries for those)                  */
    int32_t engine;            /* MFX_ENGINE_* bits: which of two equivalent kernels serves a configuration (0 = the
                                  library's choice).  For cross-checks between kernels and A/B measurements; results
                                  agree to float32 rounding either way                                               */
    int32_t tail_split;        /* fused batch front ends: the last `tail_split` chunks of every wave of the grid are cut
                                  into 4-frame pieces so that the launch ends evenly; 0 = default (2), -1 = off        */
    int32_t reserved[2];
} mfx_config;

/* mfx_config.engine bits */
#define MFX_ENGINE_NO_FRONT1024 1 /* 1024-point short-window configurations stay on the generic long-transform kernel  */
#define MFX_ENGINE_NO_FRONT2048 4 /* 2048-point short-window configurations stay on the generic long-transform kernel  */
#define MFX_ENGINE_STREAM_KERNELS 8 /* batch entries run the streaming interface's kernels (spectrum through HBM, then the
                                      mel / DCT kernel) instead of the fused front ends: the rows are then the SAME BITS
                                      that set_input / apply / get_output_data deliver for a file consumed as one block
                                      (the fused kernels agree with them to float32 rounding, ~1e-6 of scale)           */
#define MFX_ENGINE_FUSE_DELTA 2   /* 512-point batch path: delta / delta-delta computed by a wave of the front-end kernel
                                     instead of the separate delta kernel (slower on MI355X)        */

#define MFX_ENGINE_NORM_TWO_KERNELS 16 /* normaliser: statistics and apply as two launches also where one block's LDS holds a
                                          segment's rows (the one-launch form computes the same bits)                    */
#define MFX_ENGINE_DMA_SMALL_BLOCKS 32  /* streaming interface: blocks under 1 MB are copied by DMA commands (as larger ones are)
                                           instead of by a copy kernel through the pinned staging buffers                 */
#define MFX_ENGINE_NO_DCT_SPLIT 64      /* 2048-point fused kernel: the DCT as one 64-column tile also where at most 40 columns
                                           are wanted (else: column groups x band parts, summed across the wave)          */
#define MFX_ENGINE_NO_STUFF256 128      /* 256-point transforms stay on the one-wave-per-frame kernel instead of the zero-stuffed
                                           form of the 512-point kernel                                                   */

#define MFX_ENGINE_FRONT1024_12_WAVES 256 /* 1024-point fused kernel: the 12-waves-per-CU build also where the 16-wave build fits
                                            (aligned frames, window <= 512 samples, tables small enough): the same bits      */

typedef struct mfx_handle mfx_handle;

/* ---- lifetime: replaces `new MfccOpenCL(..., cl_device_id)` (ASR_OCL.cpp:140-143,
 *      mfccopencl.cpp:98-233) and `delete param` (ASR_OCL.cpp:323) ---- */
int mfx_create(const mfx_config *cfg, int hip_device, mfx_handle **out);
void mfx_destroy(mfx_handle *h);
const char *mfx_last_error(const mfx_handle *h);
/* message for a status code when no handle exists (mfx_create failure) */
const char *mfx_status_string(int status);
int mfx_abi_version(void);

/* ---- streaming parameterizer interface, one function per ParamBase method (parambase.h:23-32) ---- */

/* ParamBase::set_window (parambase.h:27; SegmenterOpenCL::set_window segmenteropencl.cpp:110-118):
 * window_size floats, copied. */
int mfx_set_window(mfx_handle *h, const float *window);
/* ParamBase::set_input (parambase.h:28; MfccOpenCL::set_input mfccopencl.cpp:472-480): uploads
 * `samples` int16 samples, frames + windows them and runs the FFT.  *frames_out = frames that the
 * next apply()/get_output_data() will deliver (0 when the block is still too short). */
int mfx_set_input(mfx_handle *h, const int16_t *pcm, int32_t samples, int32_t *frames_out);
/* ParamBase::flush (parambase.h:29; mfccopencl.cpp:482-493) */
int mfx_flush(mfx_handle *h, int32_t *frames_out);
/* ParamBase::set_alpha (parambase.h:25): VTLN warp factor used by the next apply() */
int mfx_set_alpha(mfx_handle *h, float alpha);
/* ParamBase::apply (parambase.h:30; mfccopencl.cpp:495-549): filterbank -> log -> DCT -> delta ->
 * normalisation for the current block; may be repeated with different alpha on one FFT result
 * (ASR_OCL.cpp:236-243). */
int mfx_apply(mfx_handle *h);
/* ParamBase::get_output_data_width (parambase.h:31; mfccbase.cpp:33-43) */
int mfx_get_output_data_width(const mfx_handle *h);
/* ParamBase::get_output_data (parambase.h:32; mfccopencl.cpp:551-569): `frames` rows of
 * get_output_data_width() floats, row-major [static | delta | delta-delta]. */
int mfx_get_output_data(mfx_handle *h, float *data_out, int32_t frames);
/* VTLN sweep: the reference's alpha loop (ASR_OCL.cpp:236-243: one set_input, then set_alpha + apply +
 * get_output_data per warp factor; warp mfcccpu.cpp:36-38) as ONE call.  All n_alpha warped filterbanks
 * are applied to the stored spectrum of the current block in one launch per stage; results are the
 * same as n_alpha rounds of mfx_set_alpha + mfx_apply.  Read block a with mfx_get_output_data_alpha.
 * The handle's own alpha (mfx_set_alpha) is not changed.  Normalisation statistics are kept per
 * alpha, so a flush block re-uses the statistics of the same alpha (normalizercpu.cpp use_last_stats). */
int mfx_apply_alphas(mfx_handle *h, const float *alphas, int32_t n_alpha);
int mfx_get_output_data_alpha(mfx_handle *h, int32_t alpha_index, float *data_out, int32_t frames);
/* ParamBase::get_input_buffer_size / estimated_window_count (parambase.h:23-24, parambase.cpp:12-19) */
int mfx_get_input_buffer_size(const mfx_handle *h);
int mfx_estimated_window_count(const mfx_handle *h, int32_t samples);
/* True upper bound on the frames one set_input()/flush() can return.  The reference sizes its
 * output buffer from estimated_window_count(get_input_buffer_size()) (ASR_OCL.cpp:157-161), which
 * a steady-state block can exceed; size output buffers with this instead. */
int mfx_max_frames_out(const mfx_handle *h);
/* FFT length in use (mfcccpu.cpp:94) */
int mfx_fft_size(const mfx_handle *h);

/* ---- batch interface: many independent utterances per call (the reference processes its file
 *      list one utterance at a time through the loop at ASR_OCL.cpp:163-321; this runs the same
 *      per-utterance computation -- whole-utterance semantics, i.e. what a multi-block streaming
 *      run delivers -- for all of them in one launch sequence) ---- */

/* Frames of one utterance of `samples` samples per channel (= estimated_window_count, but with
 * integer arithmetic so it stays exact above 2^24 samples; parambase.cpp:16-19). */
int64_t mfx_batch_frames(const mfx_handle *h, int64_t samples);

/* Describe a batch: utterance u occupies samples [offsets[u], offsets[u]+lengths[u]) of the PCM
 * array (per channel; for stereo the array holds 2*that many interleaved int16).  Feature rows of
 * utterance u start at row out_rows[u] of the output (computed here: prefix sum of frame counts).
 * Arrays are host pointers, copied.  Returns total rows in *total_rows. */
int mfx_batch_plan(mfx_handle *h, int32_t n_utt, const int64_t *offsets, const int64_t *lengths,
                   int64_t *out_rows, int64_t *total_rows);

/* Run the planned batch on DEVICE pointers: d_pcm (int16, HBM) -> d_out (float [total_rows][width],
 * HBM).  Asynchronous on the handle's stream; nothing is copied to or from the host.  This is the
 * entry the roofline numbers are measured on.  d_pcm must be 4-byte aligned; when the number of int16
 * elements is odd the kernels read the 32-bit word that holds the last sample whole (2 bytes past the
 * last element, inside any device allocation; that half-word only meets a zero window tap).
 * With normalisation on, an utterance's statistics are the reference's for a file consumed as one
 * block (mfx_config.batch_norm_stats). */
int mfx_batch_run_device(mfx_handle *h, const int16_t *d_pcm, int64_t pcm_samples_total, float *d_out);

/* Opt-in pipelining of consecutive batches: with enable=1 the delta / normalisation tail of a batch runs
 * on a second internal stream, so it overlaps the front end of the NEXT mfx_batch_run_device call.
 * Results of a batch are then complete only after mfx_synchronize() (or a device-wide synchronise),
 * not in order on the handle's stream.  Off by default (strict stream order). */
int mfx_batch_overlap(mfx_handle *h, int enable);

/* Convenience: same, from/to HOST buffers (pinned staging + H2D, run, D2H, synchronises). */
int mfx_batch_run_host(mfx_handle *h, const int16_t *pcm, int64_t pcm_samples_total, float *out);

/* Page-locked host memory for the caller's PCM / feature buffers (mfx_batch_run_host and the streaming entries DMA
 * straight from / to such buffers; pageable ones go through the handle's staging).  NULL on failure. */
void *mfx_alloc_pinned(size_t bytes);
void mfx_free_pinned(void *p);

/* ---- stream / timing plumbing ---- */
/* Use an existing hipStream_t (e.g. torch's current stream) instead of the handle's own. */
int mfx_set_stream(mfx_handle *h, void *hip_stream);
int mfx_synchronize(mfx_handle *h);
/* HIP-event timing of the dominant kernel over the launches since the last reset: number of
 * launches and their summed device time in milliseconds (events recorded on the handle's stream
 * around that kernel only).  enable=1 turns recording on; it is off by default. */
int mfx_profile_enable(mfx_handle *h, int enable);
int mfx_profile_read(mfx_handle *h, int32_t *launches, double *kernel_ms, int reset);
/* name of the dominant (front-end) kernel of the batch entries as it appears in rocprofv3's kernel trace: what the ONE
 * dispatch rule of the library (choose_front, mfx_api.cpp) picks for this handle */
const char *mfx_dominant_kernel_name(const mfx_handle *h);
/* A PLANNING handle: mfx_create's own configuration checks, host-built tables and LDS sums with every device call left
 * out -- it answers mfx_dominant_kernel_name and the geometry accessors (mfx_get_output_data_width,
 * mfx_get_input_buffer_size, mfx_estimated_window_count, mfx_max_frames_out, mfx_fft_size) without a GPU and computes
 * nothing; every other entry fails on it with MFX_ERR_DEVICE.  The shape -> kernel table of DESIGN.md is pinned through it
 * (tests/test_host.py).  The reference has no analogue: it chooses its back end by a CLI switch (ASR_OCL.cpp:132-146).
 * mfx_plan_set_aligned: whether the batch's frames lie on aligned sample pairs (mfx_batch_plan derives that from the
 * caller's offsets on a real handle; default 1). */
int mfx_plan_create(const mfx_config *cfg, mfx_handle **out);
int mfx_plan_set_aligned(mfx_handle *h, int aligned);

/* ---- host-side table builders (no device needed; the same code fills the tables the kernels
 *      read, exposed so that CPU-only tests can compare them with the oracle bit for bit) ---- */
/* mel table of MfccCpu::refresh_filters (mfcccpu.cpp:24-60): weights [2][fft_size], beg [num_banks+2] */
int mfx_host_mel_table(int32_t num_banks, int32_t fft_size, float sample_rate, float low_freq, float high_freq,
                       float alpha, float *weights, int32_t *beg);
/* DCT-II + lifter matrix (mfcccpu.cpp:118-136): [num_banks][ceps_len + (want_c0 ? 1 : 0)] */
int mfx_host_dct_matrix(int32_t num_banks, int32_t ceps_len, int32_t want_c0, float lift_coef, float *matrix);
/* Lane plan of the mel walk of the fused kernels (lanes = 16: 512-point kernel, and with fft_size = 1024 the short-window
 * 1024-point kernel, whose starts are multiples of 4 bins; lanes = 64: long-transform kernel): filters dealt to the lanes
 * in rounds, longest first; returns the number of rounds.  Test / inspection aid. */
int mfx_host_mel_lane_plan(int32_t lanes, int32_t num_banks, int32_t fft_size, const float *weights, const int32_t *beg,
                           int32_t max_read_bin, int32_t *L, int32_t *row_stride, int32_t *start, int32_t *fid, float *w,
                           int64_t w_cap);
/* Operands of the DCT on the matrix pipe, [tile][K step][lane]; returns their count.  Test / inspection aid. */
int64_t mfx_host_dct_mfma_operands(int32_t num_banks, int32_t dct_len, const float *matrix, float *out, int64_t out_cap,
                                   int32_t *tiles, int32_t *ksteps);
/* frame count, integer arithmetic (parambase.cpp:16-19 without the float32 division) */
int64_t mfx_host_frame_count(int64_t samples, int32_t window_size, int32_t shift);

/* ---- test taps (device -> host copies of intermediate tables; used by the parity tests) ---- */
/* kind: 0 = mel table [2][fft_size] floats, 1 = filter_beg [num_banks+2] int32,
 *       2 = DCT matrix [num_banks][dct_len] floats, 3 = magnitude spectrum of the current block
 *       [frames_with_context][fft_size/2+1] floats.  Returns element count or <0. */
int64_t mfx_debug_read(mfx_handle *h, int kind, void *dst, int64_t dst_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MFX_H */
