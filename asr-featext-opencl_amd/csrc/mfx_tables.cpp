// mfx_tables.cpp -- see mfx_tables.h.  Compiled with -ffp-contract=off so that the float32
// expression order below is what actually executes.
#include "mfx_tables.h"

#include <algorithm>
#include <cmath>
#include <functional>
#include <vector>

namespace mfx {

static const float kPiF = (float)3.14159265358979323846264338;

uint32_t ceil_pow2(uint32_t v)
{
    uint32_t p = 1;
    while (p < v) p <<= 1;
    return p;
}

int64_t frame_count(int64_t samples, int window_size, int shift)
{
    int64_t num = samples - (int64_t)(window_size - shift);
    // floor division (num may be negative for very short inputs)
    int64_t q = num / shift;
    if ((num % shift != 0) && ((num < 0) != (shift < 0))) --q;
    return q;
}

int estimated_window_count_f32(int samples, int window_size, int shift)
{
    return (int)std::floor((float)(samples - (window_size - shift)) / (float)shift);
}

namespace {
inline float mel_of_hz(float f) { return 1127 * std::log(f / 700 + 1); }  // float overloads
inline float hz_of_mel(float m) { return 700 * (std::exp(m / 1127) - 1); }
inline int round_bin(float centre_hz, int fft_size, float sample_rate)
{
    // reference: floor(c * W2 / sr + 0.5) with the sum taken in double (mfcccpu.cpp:40,47-49)
    return (int)std::floor((double)(centre_hz * fft_size / sample_rate) + 0.5);
}
} // namespace

void build_mel_table(int num_banks, int fft_size, float sample_rate, float low_freq, float high_freq,
                     float alpha, MelTable &out)
{
    const int npts = num_banks + 2;
    std::vector<float> centre(npts);
    out.weights.assign((size_t)2 * fft_size, 0.0f);
    out.beg.assign(npts, 0);

    const float mel_lo = mel_of_hz(low_freq), mel_hi = mel_of_hz(high_freq);
    const float one_minus_alpha = 1 - alpha;
    for (int i = 0; i < npts; ++i) {
        float hz = hz_of_mel(i / float(num_banks + 1) * (mel_hi - mel_lo) + mel_lo);
        float omega = 2 * kPiF * hz / sample_rate;
        // VTLN bilinear warp; identity for alpha == 1 (mfcccpu.cpp:36-38)
        omega = omega + 2 * std::atan((one_minus_alpha * std::sin(omega)) / (1 - one_minus_alpha * std::cos(omega)));
        centre[i] = sample_rate * omega / (2 * kPiF);
        out.beg[i] = round_bin(centre[i], fft_size, sample_rate);
    }
    for (int m = 0; m < num_banks; ++m) {
        const float left = centre[m], mid = centre[m + 1], right = centre[m + 2];
        const int first = round_bin(left, fft_size, sample_rate);
        const int last = round_bin(right, fft_size, sample_rate);
        float *row = out.weights.data() + (size_t)(m & 1) * fft_size;
        for (int bin = first; bin < last; ++bin) {
            if (bin < 0 || bin >= fft_size) continue;
            float hz = bin * sample_rate / (fft_size);
            float rising = (hz - left) / (mid - left);
            float falling = (hz - right) / (mid - right);
            row[bin] = std::max(0.0f, std::min(rising, falling));
        }
    }
}

void build_dct_matrix(int num_banks, int ceps_len, bool want_c0, float lift_coef, std::vector<float> &out)
{
    const int dct_len = ceps_len + (want_c0 ? 1 : 0);
    out.assign((size_t)num_banks * dct_len, 0.0f);
    const float norm = (float)std::sqrt(2.0 / num_banks);
    for (int bank = 0; bank < num_banks; ++bank) {
        float *row = out.data() + (size_t)bank * dct_len;
        for (int c = 1; c <= ceps_len; ++c) {
            float lifter = (1 + lift_coef / 2 * sinf(kPiF * (float)c / lift_coef));
            row[c - 1] = lifter * norm * cosf(kPiF * c * (bank + 0.5f) / num_banks);
        }
        if (want_c0) row[ceps_len] = norm;
    }
}

void build_twiddles(int n, int count, std::vector<float> &t)
{
    t.resize((size_t)2 * count);
    for (int k = 0; k < count; ++k) {
        double ang = -2.0 * 3.14159265358979323846264338 * (double)k / (double)n;
        t[2 * k] = (float)std::cos(ang);
        t[2 * k + 1] = (float)std::sin(ang);
    }
}

} // namespace mfx

namespace mfx {

// Start bins for the filters of one round: cands[j] lists lane j's admissible starts (its own first, then earlier ones);
// a start occupies residue (start / align) mod residues.  Returns one start per lane such that as many lanes as possible
// hold a residue of their own (maximum bipartite matching, augmenting paths; lanes left over keep their own start).
static std::vector<int> match_starts(const std::vector<std::vector<int>> &cands, int align, int residues)
{
    const int n = (int)cands.size();
    std::vector<int> owner(residues, -1), choice(n, -1);
    std::vector<char> seen;
    auto res = [&](int start) { return (start / align) % residues; };
    std::function<bool(int)> augment = [&](int j) {
        for (size_t c = 0; c < cands[j].size(); ++c) {
            const int q = res(cands[j][c]);
            if (seen[q]) continue;
            seen[q] = 1;
            if (owner[q] < 0 || augment(owner[q])) {
                owner[q] = j;
                choice[j] = (int)c;
                return true;
            }
        }
        return false;
    };
    for (int j = 0; j < n; ++j) {
        if (cands[j].empty()) continue;
        seen.assign(residues, 0);
        augment(j);
    }
    std::vector<int> pick(n, 0);
    for (int j = 0; j < n; ++j)
        if (!cands[j].empty()) pick[j] = cands[j][choice[j] < 0 ? 0 : choice[j]];
    return pick;
}

static int stride_4odd(int n)
{
    int q = (n + 3) / 4;
    if ((q & 1) == 0) ++q;
    return 4 * q;
}

bool build_mel_lane_plan(const MelTable &t, int num_banks, int fft_size, int max_read_bin, MelLanePlan &out, int align)
{
    std::vector<int> order(num_banks);
    for (int m = 0; m < num_banks; ++m) order[m] = m;
    auto span = [&](int m) { return t.beg[m + 2] - t.beg[m]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return span(a) > span(b); });
    out.rounds = (num_banks + 15) / 16;
    if (out.rounds > 8) return false;
    out.start.assign((size_t)16 * out.rounds, 0);
    out.fid.assign((size_t)16 * out.rounds, -1);
    int total = 0;
    for (int r = 0; r < out.rounds; ++r) {
        // Starts are multiples of `align` bins (an 8- or 16-byte LDS read).  The 16 lanes of a frame read at start + s
        // simultaneously; they fall on distinct bank groups when (start / align) mod 16 differs from lane to lane, so a
        // clashing filter may begin a few reads early (zero weights: the sum keeps its bits) -- as long as the round does
        // not get longer for it (a two-way bank conflict costs one LDS cycle per read, a longer round a whole 8-bin trip
        // on every lane).  Which filter moves where is a maximum bipartite matching of the round's filters to residues.
        int natural = 8;
        for (int j = 0; j < 16; ++j) {
            const int idx = r * 16 + j;
            if (idx >= num_banks) continue;
            const int m = order[idx];
            natural = std::max(natural, (t.beg[m + 2] - (t.beg[m] & ~(align - 1)) + 7) & ~7);
        }
        std::vector<std::vector<int>> cands(16);
        for (int j = 0; j < 16; ++j) {
            const int idx = r * 16 + j;
            if (idx >= num_banks) continue;
            const int m = order[idx];
            for (int d = 0; d < 16; ++d) {
                const int cand = (t.beg[m] & ~(align - 1)) - align * d;
                if (cand < 0 || t.beg[m + 2] - cand > natural) break;
                cands[j].push_back(cand);
            }
        }
        const std::vector<int> pick = match_starts(cands, align, 16);
        int longest = 0;
        for (int j = 0; j < 16; ++j) {
            const int idx = r * 16 + j;
            if (idx >= num_banks) continue;
            const int m = order[idx];
            out.start[r * 16 + j] = pick[j];
            out.fid[r * 16 + j] = m;
            longest = std::max(longest, t.beg[m + 2] - pick[j]);
        }
        out.L[r] = std::max(8, (longest + 7) & ~7);
        total += out.L[r];
    }
    out.row_stride = stride_4odd(total);
    out.w.assign((size_t)16 * out.row_stride, 0.0f);
    int base = 0;
    for (int r = 0; r < out.rounds; ++r) {
        for (int j = 0; j < 16; ++j) {
            const int m = out.fid[r * 16 + j];
            if (m < 0) continue;
            const int start = out.start[r * 16 + j];
            if (start + out.L[r] - 1 > max_read_bin) return false;
            const float *row = t.weights.data() + (size_t)(m & 1) * fft_size;
            float *dst = out.w.data() + (size_t)j * out.row_stride + base;
            for (int k = t.beg[m]; k < t.beg[m + 2]; ++k) dst[k - start] = row[k];
        }
        base += out.L[r];
    }
    return true;
}

bool build_mel_wave_plan(const MelTable &t, int num_banks, int fft_size, int max_read_bin, MelWavePlan &out, int lanes)
{
    if (lanes != 64 && lanes != 32) return false;
    std::vector<int> order(num_banks);
    for (int m = 0; m < num_banks; ++m) order[m] = m;
    auto span = [&](int m) { return t.beg[m + 2] - t.beg[m]; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return span(a) > span(b); });
    out.rounds = (num_banks + lanes - 1) / lanes;
    if (out.rounds > 8) return false;
    out.start.assign((size_t)lanes * out.rounds, 0);
    out.fid.assign((size_t)lanes * out.rounds, -1);
    int total = 0;
    for (int r = 0; r < out.rounds; ++r) {
        // An 8-byte read is served in groups of 32 lanes over 32 bank pairs: a group is conflict free when
        // (start / 2) mod 32 differs from lane to lane, so a clashing filter begins a few pairs early (zero weights) where
        // the round does not get longer for it; the assignment is a maximum matching per group of 32 lanes.
        int natural = 8;
        for (int j = 0; j < lanes; ++j) {
            const int idx = r * lanes + j;
            if (idx >= num_banks) continue;
            const int m = order[idx];
            natural = std::max(natural, (t.beg[m + 2] - (t.beg[m] & ~1) + 7) & ~7);
        }
        int longest = 0;
        for (int g0 = 0; g0 < lanes; g0 += 32) {
            std::vector<std::vector<int>> cands(32);
            for (int j = g0; j < g0 + 32; ++j) {
                const int idx = r * lanes + j;
                if (idx >= num_banks) continue;
                const int m = order[idx];
                for (int d = 0; d < 32; ++d) {
                    const int cand = (t.beg[m] & ~1) - 2 * d;
                    if (cand < 0 || t.beg[m + 2] - cand > natural) break;
                    cands[j - g0].push_back(cand);
                }
            }
            const std::vector<int> pick = match_starts(cands, 2, 32);
            for (int j = g0; j < g0 + 32; ++j) {
                const int idx = r * lanes + j;
                if (idx >= num_banks) continue;
                const int m = order[idx];
                out.start[r * lanes + j] = pick[j - g0];
                out.fid[r * lanes + j] = m;
                longest = std::max(longest, t.beg[m + 2] - pick[j - g0]);
            }
        }
        out.L[r] = std::max(8, (longest + 7) & ~7);
        total += out.L[r];
    }
    out.row_stride = stride_4odd(total);
    out.w.assign((size_t)lanes * out.row_stride, 0.0f);
    int base = 0;
    for (int r = 0; r < out.rounds; ++r) {
        for (int j = 0; j < lanes; ++j) {
            const int m = out.fid[r * lanes + j];
            if (m < 0) continue;
            const int start = out.start[r * lanes + j];
            if (start + out.L[r] - 1 > max_read_bin) return false;
            const float *row = t.weights.data() + (size_t)(m & 1) * fft_size;
            float *dst = out.w.data() + (size_t)j * out.row_stride + base;
            for (int k = t.beg[m]; k < t.beg[m + 2]; ++k) dst[k - start] = row[k];
        }
        base += out.L[r];
    }
    return true;
}

void build_dct_mfma_operands(const std::vector<float> &dct, int num_banks, int dct_len, int &tiles, int &ksteps,
                             std::vector<float> &out)
{
    tiles = (dct_len + 15) / 16;
    ksteps = (num_banks + 3) / 4;
    out.assign((size_t)tiles * ksteps * 64, 0.0f);
    for (int tl = 0; tl < tiles; ++tl)
        for (int j = 0; j < ksteps; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = 4 * j + (lane >> 4), c = 16 * tl + (lane & 15);
                if (m < num_banks && c < dct_len) out[((size_t)tl * ksteps + j) * 64 + lane] = dct[(size_t)m * dct_len + c];
            }
}

void build_dct_mfma_operands4(const std::vector<float> &dct, int num_banks, int dct_len, std::vector<float> &out)
{
    const int tiles = (dct_len + 63) / 64, ks = (num_banks + 3) / 4;
    out.assign((size_t)tiles * ks * 64 * 4, 0.0f);
    for (int tl = 0; tl < tiles; ++tl)
        for (int j4 = 0; j4 < ks; ++j4)
            for (int lane = 0; lane < 64; ++lane)
                for (int u = 0; u < 4; ++u) {
                    const int m = 4 * j4 + u, c = 64 * tl + lane;
                    if (m < num_banks && c < dct_len) out[(((size_t)tl * ks + j4) * 64 + lane) * 4 + u] = dct[(size_t)m * dct_len + c];
                }
}

int dct_split_mode(int num_banks, int dct_len)
{
    if (num_banks % 32 != 0 || num_banks > 256 || dct_len < 1) return 0;
    return dct_len <= 32 ? 1 : dct_len <= 40 ? 2 : 0;
}

void build_dct_mfma_operands4_split(const std::vector<float> &dct, int num_banks, int dct_len, std::vector<float> &out)
{
    const int mode = dct_split_mode(num_banks, dct_len);
    out.clear();
    if (mode == 0) return;
    const int H = num_banks / 2, E = num_banks / 8, ga = H / 4, gb = mode == 2 ? E / 4 : 0;
    out.assign((size_t)(ga + gb) * 64 * 4, 0.0f);
    for (int g = 0; g < ga; ++g)
        for (int lane = 0; lane < 64; ++lane)
            for (int u = 0; u < 4; ++u) {
                const int m = (lane >> 5) * H + 4 * g + u, c = lane & 31;
                if (c < dct_len) out[((size_t)g * 64 + lane) * 4 + u] = dct[(size_t)m * dct_len + c];
            }
    for (int g = 0; g < gb; ++g)
        for (int lane = 0; lane < 64; ++lane)
            for (int u = 0; u < 4; ++u) {
                const int m = (lane >> 3) * E + 4 * g + u, c = 32 + (lane & 7);
                if (c < dct_len) out[((size_t)(ga + g) * 64 + lane) * 4 + u] = dct[(size_t)m * dct_len + c];
            }
}

void build_dct_transposed(const std::vector<float> &dct, int num_banks, int dct_len, int &stride, int &nb_pad,
                          std::vector<float> &out)
{
    nb_pad = (num_banks + 3) & ~3;
    stride = stride_4odd(nb_pad);
    out.assign((size_t)dct_len * stride, 0.0f);
    for (int m = 0; m < num_banks; ++m)
        for (int c = 0; c < dct_len; ++c) out[(size_t)c * stride + m] = dct[(size_t)m * dct_len + c];
}

} // namespace mfx
