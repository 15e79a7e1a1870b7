/*
 * mfcc_oracle.c -- CPU restatement of the reference MFCC front end (TEST INFRASTRUCTURE ONLY).
 * See mfcc_oracle.h for scope and pinning status.  Citations are file:line in /root/reference.
 *
 * Arithmetic follows the reference's float32 expression order.  Where the reference calls an
 * unqualified libm name on a float (log/exp/sin/cos/atan/sqrt in mfcccpu.cpp:21-22,36,203,212; abs in
 * normalizercpu.cpp:66) the float overload is used by default, which is what the reference's own toolchain (MSVC,
 * global <cmath> overloads) resolves to -- and under which this file is bit-identical to the reference built here with
 * that overload selection (oracle/_ref/libref_mfcccpu_f32.so: oracle/Makefile ref_f32).  orc_set_libm_binding(o, 1)
 * selects the C double functions and int abs instead, which is how plain g++ compiles those lines -- and under which
 * this file is bit-identical to oracle/_ref/libref_mfcccpu.so (tests/test_ref_mfcccpu.py, both).
 */
#include "mfcc_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846264338
#endif

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* mfcccpu.cpp:10-20 (ceil2) */
static unsigned int orc_ceil2(unsigned int v)
{
    v--;
    v |= v >> 1;
    v |= v >> 2;
    v |= v >> 4;
    v |= v >> 8;
    v |= v >> 16;
    v++;
    return v;
}

/* parambase.cpp:16-19, segmentercpu.h:33-36: float32 division, then floor */
int orc_ewc(int samples, int window_size, int shift)
{
    return (int)floorf((float)(samples - (window_size - shift)) / (float)shift);
}

/* mfccbase.cpp:28,33-43 */
int orc_output_width(int num_banks, int ceps_len, int want_c0, int dyn)
{
    int dct_len = want_c0 ? ceps_len + 1 : ceps_len;
    int cols = ceps_len > 0 ? dct_len : num_banks;
    if (dyn == ORC_DYN_DELTA) return cols * 2;
    if (dyn == ORC_DYN_ACC) return cols * 3;
    return cols;
}

/* mfcccpu.cpp:21-22.  `dbl` selects how the reference's UNQUALIFIED libm names bind when their argument is a
 * float: 0 = the float overloads (MSVC, the reference's own toolchain -- the checker's default), 1 = the C double
 * functions (g++ with <cmath> only, i.e. how oracle/_ref/libref_mfcccpu.so computes). */
static float hz2mel(float f, int dbl) { return dbl ? (float)(1127 * log((double)(f / 700 + 1))) : 1127 * logf(f / 700 + 1); }
static float mel2hz(float f, int dbl) { return dbl ? (float)(700 * (exp((double)(f / 1127)) - 1)) : 700 * (expf(f / 1127) - 1); }

/* ------------------------------------------------------------------------------------------ */
/* stage functions                                                                            */
/* ------------------------------------------------------------------------------------------ */

/* segmentercpu.cpp:17-28.  Columns j >= window_size of each row are left untouched (the
 * reference zeroes the frame buffer once, mfcccpu.cpp:110). */
void orc_segment(const short *pcm, const float *window, int window_size, int window_size2,
                 int shift, int window_count, float *data_out)
{
    for (int i = 0; i < window_count; i++)
        for (int j = 0; j < window_size; j++)
            data_out[(size_t)window_size2 * i + j] = window[j] * pcm[(size_t)i * shift + j];
}

/* deltacpu.cpp:16-29 */
void orc_delta_apply(const float *data, int dim, int window_count, int delta_size, float *out)
{
    for (int i = 0; i < window_count; i++)
        for (int j = 0; j < dim; j++) {
            float num = 0, den = 0;
            for (int l = 1; l <= delta_size; l++) {
                num += l * (data[(size_t)dim * (i + delta_size + l) + j] -
                            data[(size_t)dim * (i + delta_size - l) + j]);
                den += l * l;
            }
            out[(size_t)dim * i + j] = num / (2 * den);
        }
}

/* normalizercpu.cpp:22-89.  Statistics in double, applied in float.  abs_int: the unqualified abs() of
 * normalizercpu.cpp:66 bound to int abs(int) (g++) instead of the float overload (MSVC). */
static void normalize_impl(int norm_type, float *data, int dim, int window_count, int use_last_stats,
                           float *mean, float *var, float *minmax, int abs_int)
{
    if (!use_last_stats) {
        switch (norm_type) {
        case ORC_NORM_CMN: /* normalizercpu.cpp:28-36 */
            for (int i = 0; i < dim; i++) {
                double sum = 0;
                for (int j = 0; j < window_count; j++) sum += data[(size_t)dim * j + i];
                mean[i] = (float)(sum / window_count);
            }
            break;
        case ORC_NORM_CVN: /* normalizercpu.cpp:37-51 */
            for (int i = 0; i < dim; i++) {
                double sum = 0, sum2 = 0;
                for (int j = 0; j < window_count; j++) {
                    float v = data[(size_t)dim * j + i];
                    sum += v;
                    sum2 += v * v; /* float product, double accumulate */
                }
                mean[i] = (float)(sum / window_count);
                var[i] = (float)sqrt((window_count - 1) / (sum2 - sum * (sum / window_count)));
            }
            break;
        case ORC_NORM_MINMAX: /* normalizercpu.cpp:52-68 */
            for (int i = 0; i < dim; i++) {
                double sum = 0;
                float minv = 3.402823466e+38f, maxv = -3.402823466e+38f;
                for (int j = 0; j < window_count; j++) {
                    float v = data[(size_t)dim * j + i];
                    sum += v;
                    minv = v < minv ? v : minv;
                    maxv = v > maxv ? v : maxv;
                }
                mean[i] = (float)(sum / window_count);
                float a = fabsf(minv - mean[i]), b = fabsf(maxv - mean[i]);
                if (abs_int) {
                    a = (float)abs((int)(minv - mean[i]));
                    b = (float)abs((int)(maxv - mean[i]));
                }
                minmax[i] = 1.f / (a > b ? a : b);
            }
            break;
        default:
            break;
        }
    }
    switch (norm_type) { /* normalizercpu.cpp:71-88 */
    case ORC_NORM_CMN:
        for (int i = 0; i < window_count; i++)
            for (int j = 0; j < dim; j++) data[(size_t)dim * i + j] -= mean[j];
        break;
    case ORC_NORM_CVN:
        for (int i = 0; i < window_count; i++)
            for (int j = 0; j < dim; j++)
                data[(size_t)dim * i + j] = (data[(size_t)dim * i + j] - mean[j]) * var[j];
        break;
    case ORC_NORM_MINMAX:
        for (int i = 0; i < window_count; i++)
            for (int j = 0; j < dim; j++)
                data[(size_t)dim * i + j] = (data[(size_t)dim * i + j] - mean[j]) * minmax[j];
        break;
    default:
        break;
    }
}

void orc_normalize(int norm_type, float *data, int dim, int window_count, int use_last_stats,
                   float *mean, float *var, float *minmax)
{
    normalize_impl(norm_type, data, dim, window_count, use_last_stats, mean, var, minmax, 0);
}

/* ------------------------------------------------------------------------------------------ */
/* FFT: stands where the reference calls FFTW (mfcccpu.cpp:114,187-190).  FFTW is a           */
/* third-party dependency that is absent here (only include/fftw3.h is vendored, no version   */
/* pin); what it computes at that call site is the unnormalised forward real DFT              */
/*      out[b][k] = sum_j in[b][j] * exp(-2*pi*i*j*k/n),  k = 0..n/2.                         */
/* ------------------------------------------------------------------------------------------ */

/* mode 0: complex radix-2 in double, result rounded to float */
static void rfft_f64(const float *in, float *out, int n, int howmany)
{
    int logn = 0;
    while ((1 << logn) < n) logn++;
    double *re = (double *)malloc(sizeof(double) * n);
    double *im = (double *)malloc(sizeof(double) * n);
    double *wr = (double *)malloc(sizeof(double) * (n / 2 + 1));
    double *wi = (double *)malloc(sizeof(double) * (n / 2 + 1));
    for (int k = 0; k < n / 2; k++) {
        wr[k] = cos(2.0 * M_PI * k / n);
        wi[k] = -sin(2.0 * M_PI * k / n);
    }
    for (int b = 0; b < howmany; b++) {
        const float *x = in + (size_t)b * n;
        for (int i = 0; i < n; i++) { /* bit reversal */
            int r = 0;
            for (int t = 0; t < logn; t++) r |= ((i >> t) & 1) << (logn - 1 - t);
            re[r] = x[i];
            im[r] = 0.0;
        }
        for (int len = 2; len <= n; len <<= 1) {
            int half = len >> 1, step = n / len;
            for (int s = 0; s < n; s += len)
                for (int k = 0; k < half; k++) {
                    double cr = wr[k * step], ci = wi[k * step];
                    double ar = re[s + k], ai = im[s + k];
                    double br = re[s + k + half], bi = im[s + k + half];
                    double tr = br * cr - bi * ci, ti = br * ci + bi * cr;
                    re[s + k] = ar + tr;
                    im[s + k] = ai + ti;
                    re[s + k + half] = ar - tr;
                    im[s + k + half] = ai - ti;
                }
        }
        float *o = out + (size_t)b * n * 2;
        for (int k = 0; k <= n / 2; k++) {
            o[2 * k] = (float)re[k];
            o[2 * k + 1] = (float)im[k];
        }
    }
    free(re);
    free(im);
    free(wr);
    free(wi);
}

/* mode 1: float32, half-size complex Stockham (radix 4, one radix-2 step when log2 is odd) +
 * real split.  Used for the timed CPU baseline. */
typedef struct {
    int n;      /* real length */
    float *tw;  /* W_m^k, m = n/2, k < m, interleaved */
    float *tws; /* W_n^k, k <= n/4 ... split twiddles, k < m/2+1 */
    float *buf; /* 2 * m complex scratch */
} rfft32_plan;

static rfft32_plan *rfft32_create(int n)
{
    rfft32_plan *p = (rfft32_plan *)malloc(sizeof(*p));
    int m = n / 2;
    p->n = n;
    p->tw = (float *)malloc(sizeof(float) * 2 * (m > 0 ? m : 1));
    p->tws = (float *)malloc(sizeof(float) * 2 * (m + 1));
    p->buf = (float *)malloc(sizeof(float) * 4 * (m > 0 ? m : 1));
    for (int k = 0; k < m; k++) {
        p->tw[2 * k] = (float)cos(2.0 * M_PI * k / m);
        p->tw[2 * k + 1] = (float)-sin(2.0 * M_PI * k / m);
    }
    for (int k = 0; k <= m; k++) {
        p->tws[2 * k] = (float)cos(2.0 * M_PI * k / n);
        p->tws[2 * k + 1] = (float)-sin(2.0 * M_PI * k / n);
    }
    return p;
}

static void rfft32_destroy(rfft32_plan *p)
{
    if (!p) return;
    free(p->tw);
    free(p->tws);
    free(p->buf);
    free(p);
}

/* one frame: x real[n] -> o complex[0..n/2] */
static void rfft32_exec(rfft32_plan *p, const float *xin, float *o)
{
    const int n = p->n, m = n / 2;
    float *x = p->buf, *y = p->buf + 2 * m;
    memcpy(x, xin, sizeof(float) * n); /* z[j] = x[2j] + i x[2j+1] */
    int len = m, s = 1;
    const float *tw = p->tw;
    while (len > 1) {
        if ((len & 3) == 0 && len >= 4) {
            int n1 = len / 4, tstep = m / len;
            for (int pp = 0; pp < n1; pp++) {
                float w1r = tw[2 * pp * tstep], w1i = tw[2 * pp * tstep + 1];
                float w2r = tw[4 * pp * tstep], w2i = tw[4 * pp * tstep + 1];
                float w3r = tw[6 * pp * tstep], w3i = tw[6 * pp * tstep + 1];
                const float *xa = x + 2 * s * pp, *xb = x + 2 * s * (pp + n1);
                const float *xc = x + 2 * s * (pp + 2 * n1), *xd = x + 2 * s * (pp + 3 * n1);
                float *y0 = y + 2 * s * (4 * pp), *y1 = y0 + 2 * s, *y2 = y1 + 2 * s, *y3 = y2 + 2 * s;
                for (int q = 0; q < s; q++) {
                    float ar = xa[2 * q], ai = xa[2 * q + 1], br = xb[2 * q], bi = xb[2 * q + 1];
                    float cr = xc[2 * q], ci = xc[2 * q + 1], dr = xd[2 * q], di = xd[2 * q + 1];
                    float apcr = ar + cr, apci = ai + ci, amcr = ar - cr, amci = ai - ci;
                    float bpdr = br + dr, bpdi = bi + di;
                    float jr = -(bi - di), ji = br - dr; /* j*(b-d) */
                    y0[2 * q] = apcr + bpdr;
                    y0[2 * q + 1] = apci + bpdi;
                    float t1r = amcr - jr, t1i = amci - ji;
                    y1[2 * q] = t1r * w1r - t1i * w1i;
                    y1[2 * q + 1] = t1r * w1i + t1i * w1r;
                    float t2r = apcr - bpdr, t2i = apci - bpdi;
                    y2[2 * q] = t2r * w2r - t2i * w2i;
                    y2[2 * q + 1] = t2r * w2i + t2i * w2r;
                    float t3r = amcr + jr, t3i = amci + ji;
                    y3[2 * q] = t3r * w3r - t3i * w3i;
                    y3[2 * q + 1] = t3r * w3i + t3i * w3r;
                }
            }
            len /= 4;
            s *= 4;
        } else {
            int n1 = len / 2, tstep = m / len;
            for (int pp = 0; pp < n1; pp++) {
                float wr = tw[2 * pp * tstep], wi = tw[2 * pp * tstep + 1];
                const float *xa = x + 2 * s * pp, *xb = x + 2 * s * (pp + n1);
                float *y0 = y + 2 * s * (2 * pp), *y1 = y0 + 2 * s;
                for (int q = 0; q < s; q++) {
                    float ar = xa[2 * q], ai = xa[2 * q + 1], br = xb[2 * q], bi = xb[2 * q + 1];
                    y0[2 * q] = ar + br;
                    y0[2 * q + 1] = ai + bi;
                    float tr = ar - br, ti = ai - bi;
                    y1[2 * q] = tr * wr - ti * wi;
                    y1[2 * q + 1] = tr * wi + ti * wr;
                }
            }
            len /= 2;
            s *= 2;
        }
        float *t = x;
        x = y;
        y = t;
    }
    /* real split: X[k] = 0.5*((Z[k]+conj Z[m-k]) - i W_n^k (Z[k]-conj Z[m-k])) */
    const float *ts = p->tws;
    o[0] = x[0] + x[1];
    o[1] = 0.f;
    o[2 * m] = x[0] - x[1];
    o[2 * m + 1] = 0.f;
    for (int k = 1; k < m; k++) {
        float ar = x[2 * k], ai = x[2 * k + 1];
        float br = x[2 * (m - k)], bi = -x[2 * (m - k) + 1];
        float sr = ar + br, si = ai + bi, dr = ar - br, di = ai - bi;
        float wr = ts[2 * k], wi = ts[2 * k + 1];
        /* -i * W * d = (wi*dr + wr*di) + i*(wi*di - wr*dr) ... expand (-i)(wr+ i wi)(dr + i di) */
        float pr = wr * dr - wi * di, pi = wr * di + wi * dr; /* W*d */
        o[2 * k] = 0.5f * (sr + pi);
        o[2 * k + 1] = 0.5f * (si - pr);
    }
}

void orc_rfft_rows(const float *in, float *out_complex, int n, int howmany, int mode)
{
    if (mode == 0) {
        rfft_f64(in, out_complex, n, howmany);
    } else {
        rfft32_plan *p = rfft32_create(n);
        for (int b = 0; b < howmany; b++)
            rfft32_exec(p, in + (size_t)b * n, out_complex + (size_t)b * n * 2);
        rfft32_destroy(p);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* the extractor object: MfccCpu + SegmenterCPU + DeltaCPU x2 + NormalizerCPU x3              */
/* ------------------------------------------------------------------------------------------ */

struct orc_mfcc {
    orc_config cfg;
    /* ParamBase / MfccBase members (parambase.h:11-18, mfccbase.h:9-19) */
    int input_buffer_size, input_window_limit, window_size, shift;
    float alpha;
    int last_block;
    int num_banks, ceps_len, dct_len, delta_l1, delta_l2;
    /* MfccCpu members (mfcccpu.h:17-34) */
    int buffer_size, window_limit, cap_rows, data_length, window_size2;
    float *data, *fft, *mel, *mfcc, *dct_matrix, *filters, *delta_in;
    int *filter_beg;
    /* SegmenterCPU members (segmentercpu.h:7-17) */
    short *tmpbuffer;
    size_t tmp_capacity;
    float *window;
    int deltasize, remaining_samples, samples, flushed, last_calc_flushed;
    /* DeltaCPU outputs (deltacpu.h:10) */
    float *delta_out, *acc_out;
    /* NormalizerCPU stats x3 (normalizercpu.h:7-9): [static, delta, acc] */
    float *n_mean[3], *n_var[3], *n_minmax[3];
    int bug_compat;
    int libm_double; /* see hz2mel */
    rfft32_plan *plan32;
};

static int cols_of(const orc_mfcc *o) { return o->ceps_len > 0 ? o->dct_len : o->num_banks; }

/* mfcccpu.cpp:24-60 */
static void refresh_filters(orc_mfcc *o)
{
    const int nb = o->num_banks, W2 = o->window_size2;
    const float sr = o->cfg.sample_rate;
    float *centers = (float *)malloc(sizeof(float) * (nb + 2));
    memset(o->filters, 0, sizeof(float) * 2 * W2);

    const int dbl = o->libm_double;
    float minmel = hz2mel(o->cfg.low_freq, dbl), maxmel = hz2mel(o->cfg.high_freq, dbl);
    for (int i = 0; i < nb + 2; i++) {
        float f = mel2hz(i / (float)(nb + 1) * (maxmel - minmel) + minmel, dbl);
        float w = 2 * (float)M_PI * f / sr;
        if (dbl) /* float * double -> double all the way to the assignment */
            w = (float)(w + 2 * atan(((1 - o->alpha) * sin((double)w)) / (1 - (1 - o->alpha) * cos((double)w))));
        else
            w = w + 2 * atanf(((1 - o->alpha) * sinf(w)) / (1 - (1 - o->alpha) * cosf(w)));
        centers[i] = sr * w / (2 * (float)M_PI);
        o->filter_beg[i] = (int)floor((double)(centers[i] * W2 / sr) + 0.5);
    }
    for (int i = 0; i < nb; i++) {
        float cl = centers[i], cc = centers[i + 1], cr = centers[i + 2];
        int il = (int)floor((double)(W2 * cl / sr) + 0.5);
        int ir = (int)floor((double)(W2 * cr / sr) + 0.5);
        for (int j = il; j < ir; j++) {
            float lowslope = (j * sr / (W2)-cl) / (cc - cl);
            float highslope = (j * sr / (W2)-cr) / (cc - cr);
            float m = lowslope < highslope ? lowslope : highslope;
            if (j >= 0 && j < W2) o->filters[(i % 2) * W2 + j] = m > 0.0f ? m : 0.0f;
        }
    }
    free(centers);
}

orc_mfcc *orc_create(const orc_config *cfg)
{
    if (!cfg || cfg->window_size <= 0 || cfg->shift <= 0 || cfg->num_banks <= 0) return NULL;
    orc_mfcc *o = (orc_mfcc *)calloc(1, sizeof(*o));
    o->cfg = *cfg;
    o->bug_compat = 1;
    /* ParamBase ctor, parambase.cpp:4-14 */
    o->window_size = cfg->window_size;
    o->shift = cfg->shift;
    o->alpha = 1;
    o->last_block = 0;
    o->input_window_limit = orc_ewc(cfg->input_buffer_size, cfg->window_size, cfg->shift);
    o->input_buffer_size = o->input_window_limit * o->shift + o->window_size - o->shift;
    /* MfccBase ctor, mfccbase.cpp:18-30 */
    o->num_banks = cfg->num_banks;
    o->ceps_len = cfg->ceps_len;
    o->delta_l1 = cfg->dyn != ORC_DYN_NONE ? cfg->delta_l1 : 0;
    o->delta_l2 = cfg->dyn == ORC_DYN_ACC ? cfg->delta_l2 : 0;
    o->dct_len = cfg->want_c0 ? cfg->ceps_len + 1 : cfg->ceps_len;
    /* MfccCpu ctor, mfcccpu.cpp:94-105 */
    o->window_size2 = (int)orc_ceil2((unsigned int)(float)o->window_size);
    const int D = o->delta_l1 + o->delta_l2;
    if (cfg->dyn != ORC_DYN_NONE)
        o->window_limit = o->input_window_limit + 2 + 3 * D;
    else
        o->window_limit = o->input_window_limit + 2;
    if (o->window_limit <= 0) {
        free(o);
        return NULL;
    }
    /* Capacity: the reference sizes every buffer from window_limit (mfcccpu.cpp:104-112,
     * segmentercpu.cpp:40-41).  A steady-state block can need up to ~W/S more frames and W more
     * samples than that (the reference then writes past its buffers: W - S > 2S, or dyn off), so the
     * allocations here carry that slack; window_limit itself keeps the reference value. */
    o->cap_rows = o->window_limit + o->window_size / o->shift + 4;
    o->buffer_size = o->window_limit * o->shift + o->window_size - o->shift;
    o->data_length = o->cap_rows * o->window_size2;
    /* SegmenterCPU::init, segmentercpu.cpp:30-44 */
    o->deltasize = D;
    o->remaining_samples = 0;
    o->samples = 0;
    o->flushed = 1;
    o->last_calc_flushed = 0;
    o->tmp_capacity = (size_t)o->cap_rows * o->shift + 2 * (size_t)o->window_size;
    o->tmpbuffer = (short *)calloc(o->tmp_capacity, sizeof(short));
    o->window = (float *)calloc(o->window_size, sizeof(float));
    /* mfcccpu.cpp:109-112 */
    o->data = (float *)calloc((size_t)o->data_length, sizeof(float));
    o->fft = (float *)calloc((size_t)o->data_length * 2, sizeof(float));
    o->mel = (float *)calloc((size_t)o->num_banks * o->cap_rows, sizeof(float));
    /* DCT-II + lifter matrix, mfcccpu.cpp:118-136 */
    if (o->ceps_len > 0) {
        const int nb = o->num_banks, dl = o->dct_len;
        const float lift_coef = cfg->lift_coef;
        o->mfcc = (float *)calloc((size_t)dl * o->cap_rows, sizeof(float));
        o->dct_matrix = (float *)calloc((size_t)nb * dl, sizeof(float));
        float normfact = (float)sqrt(2.0 / nb);
        for (int iy = 0; iy < nb; iy++)
            for (int ix = 1; ix <= o->ceps_len; ix++) {
                float lifter = (1 + lift_coef / 2 * sinf((float)M_PI * (float)ix / lift_coef));
                o->dct_matrix[dl * iy + ix - 1] =
                    lifter * normfact * cosf((float)M_PI * ix * (iy + 0.5f) / nb);
            }
        if (cfg->want_c0)
            for (int iy = 0; iy < nb; iy++) o->dct_matrix[dl * iy + o->ceps_len] = normfact;
    }
    const int cols = cols_of(o);
    /* normalizers, mfcccpu.cpp:137-145 + normalizercpu.cpp:6-13 */
    for (int k = 0; k < 3; k++) {
        o->n_mean[k] = (float *)calloc(cols, sizeof(float));
        o->n_var[k] = (float *)calloc(cols, sizeof(float));
        o->n_minmax[k] = (float *)calloc(cols, sizeof(float));
    }
    /* deltas, mfcccpu.cpp:146-157 */
    if (cfg->dyn != ORC_DYN_NONE) {
        int rows = o->cap_rows + 2 * D;
        o->delta_out = (float *)calloc((size_t)cols * (o->cap_rows + 2 * o->delta_l2), sizeof(float));
        if (cfg->dyn == ORC_DYN_ACC)
            o->acc_out = (float *)calloc((size_t)cols * o->cap_rows, sizeof(float));
        o->delta_in = (float *)calloc((size_t)cols * rows, sizeof(float));
    }
    o->filters = (float *)calloc((size_t)2 * o->window_size2, sizeof(float));
    o->filter_beg = (int *)calloc(o->num_banks + 2, sizeof(int));
    refresh_filters(o); /* mfcccpu.cpp:159 */
    if (cfg->fft_mode == 1) o->plan32 = rfft32_create(o->window_size2);
    return o;
}

void orc_destroy(orc_mfcc *o)
{
    if (!o) return;
    free(o->tmpbuffer);
    free(o->window);
    free(o->data);
    free(o->fft);
    free(o->mel);
    free(o->mfcc);
    free(o->dct_matrix);
    free(o->filters);
    free(o->filter_beg);
    free(o->delta_in);
    free(o->delta_out);
    free(o->acc_out);
    for (int k = 0; k < 3; k++) {
        free(o->n_mean[k]);
        free(o->n_var[k]);
        free(o->n_minmax[k]);
    }
    rfft32_destroy(o->plan32);
    free(o);
}

void orc_set_bug_compat(orc_mfcc *o, int on) { o->bug_compat = on; }
void orc_set_libm_binding(orc_mfcc *o, int use_double)
{
    o->libm_double = use_double != 0;
    refresh_filters(o);
}
void orc_set_alpha(orc_mfcc *o, float alpha) { o->alpha = alpha; } /* parambase.h:25 */
int orc_get_input_buffer_size(const orc_mfcc *o) { return o->input_buffer_size; }
int orc_estimated_window_count(const orc_mfcc *o, int samples) { return orc_ewc(samples, o->window_size, o->shift); }
int orc_window_limit(const orc_mfcc *o) { return o->window_limit; }
int orc_fft_size(const orc_mfcc *o) { return o->window_size2; }
int orc_get_output_data_width(const orc_mfcc *o)
{
    return orc_output_width(o->num_banks, o->ceps_len, o->cfg.want_c0, o->cfg.dyn);
}

/* segmentercpu.cpp:51-54 */
void orc_set_window(orc_mfcc *o, const float *window) { memcpy(o->window, window, sizeof(float) * o->window_size); }

/* mfcccpu.cpp:187-190.  The reference executes its FFTW plan over the whole capacity; only the
 * first window_count rows are ever read afterwards, so only those are transformed here. */
static void do_fft(orc_mfcc *o, int window_count)
{
    const int W2 = o->window_size2;
    if (o->plan32) {
        for (int b = 0; b < window_count; b++)
            rfft32_exec(o->plan32, o->data + (size_t)b * W2, o->fft + (size_t)b * W2 * 2);
    } else {
        rfft_f64(o->data, o->fft, W2, window_count);
    }
}

/* SegmenterCPU::set_input, segmentercpu.cpp:56-95 (memmove where the reference memcpy's an
 * overlapping range). Returns 0 or an error; counts through the out parameters. */
static int seg_set_input(orc_mfcc *o, const short *data_in, int samples, int *window_count, int *wcnd)
{
    const int D = o->deltasize, W = o->window_size, S = o->shift;
    o->last_calc_flushed = o->flushed;
    if (o->last_calc_flushed) {
        if ((size_t)samples > o->tmp_capacity) return ORC_ERR_BUFFER_TOO_SMALL;
        memcpy(o->tmpbuffer, data_in, sizeof(short) * samples);
        *wcnd = orc_ewc(samples, W, S);
        *window_count = *wcnd - D;
        if (*window_count <= 0) return ORC_ERR_WINDOW_COUNT;
        orc_segment(o->tmpbuffer, o->window, W, o->window_size2, S, *wcnd, o->data);
        int processed = (*window_count - D) * S + W - S;
        /* with fewer than 2 D frames in a first block the reference's carry-over starts BEFORE its buffer
         * (segmentercpu.cpp:72-73: m_tmpbuffer + samples - m_remaining_samples < m_tmpbuffer) unless its `processed <= 0`
         * guard happens to fire: undefined behaviour there; refused here (and by the product) with that guard's error. */
        if (processed <= 0 || *window_count < D) return ORC_ERR_PROCESSED;
        o->remaining_samples = samples - processed + W - S;
        memmove(o->tmpbuffer, o->tmpbuffer + samples - o->remaining_samples, sizeof(short) * o->remaining_samples);
        o->flushed = 0;
    } else {
        if ((size_t)samples + o->remaining_samples > o->tmp_capacity) return ORC_ERR_BUFFER_TOO_SMALL;
        memcpy(o->tmpbuffer + o->remaining_samples, data_in, sizeof(short) * samples);
        samples += o->remaining_samples;
        *wcnd = orc_ewc(samples, W, S);
        *window_count = *wcnd - 2 * D;
        if (*window_count > 0)
            orc_segment(o->tmpbuffer, o->window, W, o->window_size2, S, *wcnd, o->data);
        else
            *window_count = 0;
        int processed = *window_count * S + W - S;
        o->remaining_samples = samples - processed + W - S;
        memmove(o->tmpbuffer, o->tmpbuffer + samples - o->remaining_samples, sizeof(short) * o->remaining_samples);
    }
    o->samples = samples;
    return ORC_OK;
}

/* MfccCpu::set_input, mfcccpu.cpp:338-346.  Deviation (documented in DESIGN.md, "B7"): the
 * reference never clears m_last_block after flush() (parambase.h:18, mfcccpu.cpp:350-352), so a
 * second file on the same object runs apply()'s flush branch forever; here a new set_input
 * starts a new stream. */
int orc_set_input(orc_mfcc *o, const short *data, int samples)
{
    if (samples > o->input_buffer_size) return ORC_ERR_BUFFER_TOO_SMALL;
    o->last_block = 0;
    int window_count = 0, wcnd = 0;
    int rc = seg_set_input(o, data, samples, &window_count, &wcnd);
    if (rc != ORC_OK) return rc;
    if (window_count <= 0) return 0;
    do_fft(o, wcnd);
    return window_count;
}

/* MfccCpu::flush mfcccpu.cpp:348-369 + SegmenterCPU::flush segmentercpu.cpp:97-106 */
int orc_flush(orc_mfcc *o)
{
    if (o->last_block) return 0;
    o->last_block = 1;
    o->flushed = 1;
    int wcnd = orc_ewc(o->remaining_samples, o->window_size, o->shift);
    int window_count = wcnd - o->deltasize;
    if (window_count <= 0) return 0;
    orc_segment(o->tmpbuffer, o->window, o->window_size, o->window_size2, o->shift, wcnd, o->data);
    do_fft(o, wcnd);
    return window_count;
}

/* mfcccpu.cpp:192-220.  B2 (out-of-bounds read of filter_beg[nb+2]) is not reproduced: the
 * boundary test stops at the last table entry. */
static void do_filter(orc_mfcc *o, int window_count)
{
    refresh_filters(o);
    const int nb = o->num_banks, W2 = o->window_size2;
    for (int i = 0; i < window_count; i++) {
        float sum[2] = {0, 0};
        int curf = 0;
        int lastf = o->filter_beg[nb + 1];
        for (int j = o->filter_beg[0]; j <= lastf; j++) {
            const float *c = o->fft + 2 * ((size_t)W2 * i + j);
            float v = o->libm_double ? (float)(sqrt((double)(c[0] * c[0] + c[1] * c[1])) / W2)
                                     : sqrtf(c[0] * c[0] + c[1] * c[1]) / W2;
            while (curf + 1 <= nb + 1 && j == o->filter_beg[curf + 1]) {
                curf++;
                if (curf >= 2) {
                    int sumidx = curf % 2;
                    float s = sum[sumidx];
                    s = s > 1e-30f ? s : 1e-30f;
                    o->mel[(size_t)nb * i + curf - 2] = o->libm_double ? (float)log((double)s) : logf(s);
                    sum[sumidx] = 0;
                }
            }
            sum[0] += o->filters[j] * v;
            sum[1] += o->filters[W2 + j] * v;
        }
    }
}

/* mfcccpu.cpp:222-232 */
static void do_dct(orc_mfcc *o, int window_count)
{
    const int nb = o->num_banks, dl = o->dct_len;
    for (int i = 0; i < window_count; i++)
        for (int j = 0; j < dl; j++) {
            float sum = 0;
            for (int k = 0; k < nb; k++) sum += o->mel[(size_t)nb * i + k] * o->dct_matrix[dl * k + j];
            o->mfcc[(size_t)dl * i + j] = sum;
        }
}

/* test hooks: MfccCpu::filter / MfccCpu::dct on the first `rows` rows of the spectrum buffer (which the caller may have
 * filled through orc_tap_fft) -- the counterpart of oracle/ref_mfcccpu_shim.cpp's refm_filter / refm_dct */
void orc_stage_filter(orc_mfcc *o, int rows) { do_filter(o, rows); }
void orc_stage_dct(orc_mfcc *o, int rows) { if (o->ceps_len > 0) do_dct(o, rows); }

/* mfcccpu.cpp:234-263 */
static void do_delta(orc_mfcc *o, int window_count, int first_call, int last_call)
{
    if (o->cfg.dyn == ORC_DYN_NONE) return;
    if (first_call && last_call) return;
    const int cols = cols_of(o), D = o->delta_l1 + o->delta_l2;
    const float *src = o->ceps_len > 0 ? o->mfcc : o->mel;
    if (first_call) {
        memcpy(o->delta_in + (size_t)cols * D, src, sizeof(float) * cols * (window_count + D));
        for (int i = 0; i < D; i++) memcpy(o->delta_in + (size_t)cols * i, src, sizeof(float) * cols);
    } else if (last_call) {
        memcpy(o->delta_in, src, sizeof(float) * cols * (window_count + D));
        for (int i = 0; i < D; i++)
            memcpy(o->delta_in + (size_t)cols * (i + window_count + D),
                   src + (size_t)cols * (window_count + D - 1), sizeof(float) * cols);
    } else
        memcpy(o->delta_in, src, sizeof(float) * cols * (window_count + 2 * D));

    orc_delta_apply(o->delta_in, cols, window_count + 2 * o->delta_l2, o->delta_l1, o->delta_out);
    if (o->cfg.dyn == ORC_DYN_ACC)
        orc_delta_apply(o->delta_out, cols, window_count, o->delta_l2, o->acc_out);
}

/* row offset of the first static output row inside mfcc/mel: mfcccpu.cpp:274,439 use
 * was_flushed() ? 0 : D.  With bug_compat off, a flush block always reads at D (fixes B1). */
static int static_offset_rows(const orc_mfcc *o)
{
    const int D = o->delta_l1 + o->delta_l2;
    int at_zero = o->last_calc_flushed;
    if (!o->bug_compat && o->last_block) at_zero = 0;
    return at_zero ? 0 : D;
}

/* mfcccpu.cpp:265-282 */
static void do_normalize(orc_mfcc *o, int window_count, int use_last_stats)
{
    const int norm = o->cfg.norm;
    if (norm == ORC_NORM_NONE) return;
    const int cols = cols_of(o);
    float *src = o->ceps_len > 0 ? o->mfcc : o->mel;
    if (o->cfg.norm_after_dyn) {
        normalize_impl(norm, src + (size_t)static_offset_rows(o) * cols, cols, window_count, use_last_stats,
                      o->n_mean[0], o->n_var[0], o->n_minmax[0], o->libm_double);
        if (o->cfg.dyn == ORC_DYN_DELTA || o->cfg.dyn == ORC_DYN_ACC)
            normalize_impl(norm, o->delta_out + (size_t)o->delta_l2 * cols, cols, window_count, use_last_stats,
                          o->n_mean[1], o->n_var[1], o->n_minmax[1], o->libm_double);
        if (o->cfg.dyn == ORC_DYN_ACC)
            normalize_impl(norm, o->acc_out, cols, window_count, use_last_stats, o->n_mean[2], o->n_var[2],
                          o->n_minmax[2], o->libm_double);
    } else
        normalize_impl(norm, src, cols, window_count, use_last_stats, o->n_mean[0], o->n_var[0], o->n_minmax[0], o->libm_double);
}

/* mfcccpu.cpp:371-425 */
int orc_apply(orc_mfcc *o)
{
    const int D = o->delta_l1 + o->delta_l2;
    const int norm = o->cfg.norm, nad = o->cfg.norm_after_dyn, dyn = o->cfg.dyn;
    if (o->last_block) {
        int wcnd = orc_ewc(o->remaining_samples, o->window_size, o->shift);
        int wc = wcnd - D;
        if (wc > 0) {
            do_filter(o, wcnd);
            if (o->ceps_len > 0) do_dct(o, wcnd);
            if (!nad && norm != ORC_NORM_NONE) do_normalize(o, wcnd, 1);
            if (dyn != ORC_DYN_NONE) do_delta(o, wc, 0, 1);
            if (nad && norm != ORC_NORM_NONE) do_normalize(o, wc, 1);
        }
    } else if (o->last_calc_flushed) {
        int wcnd = orc_ewc(o->samples, o->window_size, o->shift);
        int wc = wcnd - D;
        if (wc <= 0) return ORC_ERR_WINDOW_COUNT;
        do_filter(o, wcnd);
        if (o->ceps_len > 0) do_dct(o, wcnd);
        if (!nad && norm != ORC_NORM_NONE) do_normalize(o, wcnd, 0);
        if (dyn != ORC_DYN_NONE) do_delta(o, wc, 1, 0);
        if (nad && norm != ORC_NORM_NONE) do_normalize(o, wc, 0);
    } else {
        int wcnd = orc_ewc(o->samples, o->window_size, o->shift);
        int wc = wcnd - 2 * D;
        if (wc > 0) {
            do_filter(o, wcnd);
            if (o->ceps_len > 0) do_dct(o, wcnd);
            if (!nad && norm != ORC_NORM_NONE) do_normalize(o, wcnd, 0);
            if (dyn != ORC_DYN_NONE) do_delta(o, wc, 0, 0);
            if (nad && norm != ORC_NORM_NONE) do_normalize(o, wc, 0);
        }
    }
    return ORC_OK;
}

/* mfcccpu.cpp:62-71 (B3's chained comparison only matters for pitch 1; row copy is equivalent) */
static void get_output(float *data_out, const float *buff, int width, int height, int spitch, int dpitch)
{
    for (int i = 0; i < height; i++)
        memcpy(data_out + (size_t)i * dpitch, buff + (size_t)i * spitch, sizeof(float) * width);
}

/* mfcccpu.cpp:427-444 */
int orc_get_output_data(orc_mfcc *o, float *data_out, int window_count)
{
    if (window_count > o->cap_rows) return ORC_ERR_WINDOW_HIGH;
    const int cols = cols_of(o), pitch = orc_get_output_data_width(o);
    const float *src = o->ceps_len > 0 ? o->mfcc : o->mel;
    get_output(data_out, src + (size_t)static_offset_rows(o) * cols, cols, window_count, cols, pitch);
    if (o->cfg.dyn == ORC_DYN_DELTA || o->cfg.dyn == ORC_DYN_ACC)
        get_output(data_out + cols, o->delta_out + (size_t)o->delta_l2 * cols, cols, window_count, cols, pitch);
    if (o->cfg.dyn == ORC_DYN_ACC) get_output(data_out + 2 * cols, o->acc_out, cols, window_count, cols, pitch);
    return ORC_OK;
}

/* persistent statistics of the three normaliser instances (mfcccpu.h:31-33: static, delta, delta-delta), as the last
 * apply() left them: which = 0 the means, 1 the multipliers (normalizercpu.cpp m_var for CVN, m_minmax for MINMAX) */
const float *orc_tap_norm_stats(const orc_mfcc *o, int group, int which)
{
    if (group < 0 || group > 2) return 0;
    if (which == 0) return o->n_mean[group];
    return o->cfg.norm == ORC_NORM_MINMAX ? o->n_minmax[group] : o->n_var[group];
}
const float *orc_tap_frames(const orc_mfcc *o) { return o->data; }
const float *orc_tap_fft(const orc_mfcc *o) { return o->fft; }
const float *orc_tap_mel(const orc_mfcc *o) { return o->mel; }
const float *orc_tap_mfcc(const orc_mfcc *o) { return o->mfcc; }
const float *orc_tap_filters(const orc_mfcc *o) { return o->filters; }
const int *orc_tap_filter_beg(const orc_mfcc *o) { return o->filter_beg; }
const float *orc_tap_dct_matrix(const orc_mfcc *o) { return o->dct_matrix; }

/* ------------------------------------------------------------------------------------------ */
/* drivers                                                                                    */
/* ------------------------------------------------------------------------------------------ */

/* per-file loop of ASR_OCL.cpp:227-301 on an existing object */
static int run_stream(orc_mfcc *o, float alpha, const short *pcm, int samples, int block_samples, float *out)
{
    const int width = orc_get_output_data_width(o);
    int limit = o->input_buffer_size;
    if (block_samples > 0 && block_samples < limit) limit = block_samples;
    int total = 0;
    while (samples > 0) {
        int n_in = samples < limit ? samples : limit;
        int n = orc_set_input(o, pcm, n_in);
        if (n < 0) return n;
        orc_set_alpha(o, alpha);
        int rc = orc_apply(o);
        if (rc < 0) return rc;
        rc = orc_get_output_data(o, out + (size_t)total * width, n);
        if (rc < 0) return rc;
        total += n;
        pcm += n_in;
        samples -= n_in;
    }
    int n = orc_flush(o);
    if (n > 0) {
        orc_set_alpha(o, alpha);
        int rc = orc_apply(o);
        if (rc < 0) return rc;
        rc = orc_get_output_data(o, out + (size_t)total * width, n);
        if (rc < 0) return rc;
        total += n;
    }
    return total;
}

int orc_run_utterance(const orc_config *cfg, const float *window, float alpha, int bug_compat,
                      const short *pcm, int samples, int block_samples, float *out)
{
    orc_mfcc *o = orc_create(cfg);
    if (!o) return ORC_ERR_CONFIG;
    orc_set_bug_compat(o, bug_compat);
    orc_set_window(o, window);
    int total = run_stream(o, alpha, pcm, samples, block_samples, out);
    orc_destroy(o);
    return total;
}

long long orc_run_batch(const orc_config *cfg, const float *window, const short *pcm, int n_utt,
                        int utt_samples, float *out, int n_threads)
{
    const int width = orc_output_width(cfg->num_banks, cfg->ceps_len, cfg->want_c0, cfg->dyn);
    const int fpu = orc_ewc(utt_samples, cfg->window_size, cfg->shift);
    long long total = 0;
    int failed = 0;
    if (n_threads < 1) n_threads = 1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads) reduction(+ : total)
#endif
    {
        orc_mfcc *o = orc_create(cfg);
        if (o) {
            orc_set_bug_compat(o, 0);
            orc_set_window(o, window);
#ifdef _OPENMP
            int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#else
            int tid = 0, nt = 1;
#endif
            for (int u = tid; u < n_utt; u += nt) { /* round-robin shard, ASR_OCL.cpp:340-368 file queue */
                int n = run_stream(o, 1.0f, pcm + (size_t)u * utt_samples, utt_samples, 0,
                                   out + (size_t)u * fpu * width);
                if (n < 0) {
                    failed = 1;
                    break;
                }
                total += n;
            }
            orc_destroy(o);
        } else
            failed = 1;
    }
    return failed ? -1 : total;
}

/* Timed variant for bench.py's cpu_baseline leg: each thread builds its extractor once, then the
 * batch is processed `reps` times; only the processing is timed (max over threads).  Features go to
 * a per-thread scratch row block (the point is the arithmetic, not keeping 256 copies of the
 * output).  Returns frames processed in total, seconds through *seconds. */
long long orc_bench_batch(const orc_config *cfg, const float *window, const short *pcm, int n_utt,
                          int utt_samples, int n_threads, int reps, double *seconds)
{
    const int width = orc_output_width(cfg->num_banks, cfg->ceps_len, cfg->want_c0, cfg->dyn);
    const int fpu = orc_ewc(utt_samples, cfg->window_size, cfg->shift);
    long long total = 0;
    int failed = 0;
    double tmax = 0.0;
    if (n_threads < 1) n_threads = 1;
    if (fpu <= 0) return -1;
#ifdef _OPENMP
#pragma omp parallel num_threads(n_threads) reduction(+ : total) reduction(max : tmax)
#endif
    {
        orc_mfcc *o = orc_create(cfg);
        float *scratch = (float *)malloc(sizeof(float) * (size_t)(fpu + 8) * width);
        if (o && scratch) {
            orc_set_bug_compat(o, 0);
            orc_set_window(o, window);
#ifdef _OPENMP
            int tid = omp_get_thread_num(), nt = omp_get_num_threads();
#pragma omp barrier
            double t0 = omp_get_wtime();
#else
            int tid = 0, nt = 1;
            double t0 = 0;
#endif
            for (int r = 0; r < reps; r++)
                for (int u = tid; u < n_utt; u += nt) {
                    int n = run_stream(o, 1.0f, pcm + (size_t)u * utt_samples, utt_samples, 0, scratch);
                    if (n < 0) {
                        failed = 1;
                        break;
                    }
                    total += n;
                }
#ifdef _OPENMP
            tmax = omp_get_wtime() - t0;
#endif
        } else
            failed = 1;
        free(scratch);
        orc_destroy(o);
    }
    if (seconds) *seconds = tmax;
    return failed ? -1 : total;
}
