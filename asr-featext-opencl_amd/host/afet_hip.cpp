// afet_hip.cpp -- command-line driver over MfccHip: the reference's per-file loop
// (process_files_worker, ASR_OCL.cpp:109-338) with its own minimal RIFF/PCM16 reader in place of
// libsndfile and the option names of the reference's (commented-out) option table
// (ASR_OCL.cpp:569-669).  Output is the reference's text format (ASR_OCL.cpp:252-260):
//     | <frame time> | v0 | v1 | ... |
//
//   afet_hip [options] in1.wav out1.txt [in2.wav out2.txt ...]
// Inputs: RIFF/WAVE or NIST SPHERE (the reference's sample1.wav), 16-bit PCM.  --htk writes HTK
// parameter files instead of text (the reference's binary branch is a stub).
//
// Files that the reference's loop would consume as ONE block (no longer than --sample-limit; one alpha) are not pushed
// through set_input / apply / get_output_data one by one: a worker drains the queue into batches of up to --batch-mb of
// PCM, packs them into one pinned buffer and runs ONE mfx_batch_plan + mfx_batch_run_host per batch (a 7 s file is ~1 us
// of kernel work against ~130 us of per-call overhead), while helper threads read the next batch's files and format /
// write the previous batch's rows.  The extractor is created with MFX_ENGINE_STREAM_KERNELS, so the rows are the same
// bits the per-file loop delivers, and the reference's single-block flush behaviour (B1, --bug-compat) is applied to them:
// the outputs are byte-identical to the per-file loop's (--batch-mb 0 selects that loop; tests compare the two).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <atomic>
#include <charconv>
#include <future>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "afet_param.h"
#include "../../include/mfx.h"

namespace {

struct Options {
    float window_ms = 25.f, shift_ms = 10.f;
    int banks = 15, ceps = 12, norm = 2, dyn = 0, l1 = 3, l2 = 3;   // reference main() defaults (ASR_OCL.cpp:560)
    float low = 64.f, high = 0.f, lift = 22.f;
    bool c0 = true, norm_after_dyn = true;
    float alpha_min = 1.f, alpha_max = 1.f, alpha_step = 1.f;
    int sample_limit = 10000000, device = 0;
    bool bug_compat = true;
    bool htk = false; // binary output in HTK parameter-file format instead of the reference's text rows
    int format_threads = 4; // threads that format the text rows of a block (per worker)
    int batch_mb = 16;      // PCM per batch of files (0: the per-file loop only); small enough that a few thousand files pipeline
    int io_threads = 12;    // helper threads of a worker that read files and format / write rows of a batch
    std::vector<int> devices; // --devs a,b,...: one worker (own MfccHip, own thread) per entry
};

struct Wav {
    int sample_rate = 0, channels = 0;
    std::vector<int16_t> pcm; // interleaved
};

uint32_t rd32(const unsigned char *p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// NIST SPHERE (the reference ships one: sample1.wav): "NIST_1A\n   <header bytes>\n" followed by
// "key -type value" lines up to "end_head"; 16-bit PCM, sample_byte_format 01 = little endian, 10 = big.
Wav read_sphere(const std::vector<unsigned char> &b, const std::string &path)
{
    // "NIST_1A\n" then the header length as text on a line of its own: parse a bounded, terminated copy
    const std::string len_field(reinterpret_cast<const char *>(b.data()) + 8, std::min<size_t>(b.size() - 8, 8));
    const size_t hdr = (size_t)std::max(0L, std::atol(len_field.c_str()));
    if (hdr < 16 || hdr > b.size()) throw std::runtime_error("bad SPHERE header in \"" + path + "\"");
    const std::string text(reinterpret_cast<const char *>(b.data()), hdr);
    auto field = [&](const char *key, const std::string &dflt) -> std::string {
        size_t p = text.find(std::string("\n") + key + " ");
        if (p == std::string::npos) return dflt;
        p = text.find(' ', p + 1 + std::strlen(key) + 1); // skip "-i" / "-sN"
        if (p == std::string::npos) return dflt;
        const size_t e = text.find('\n', p);
        return text.substr(p + 1, e - p - 1);
    };
    Wav w;
    w.channels = std::atoi(field("channel_count", "1").c_str());
    w.sample_rate = std::atoi(field("sample_rate", "0").c_str());
    const long count = std::atol(field("sample_count", "0").c_str());
    if (std::atoi(field("sample_n_bytes", "2").c_str()) != 2 || field("sample_coding", "pcm").find("pcm") != 0)
        throw std::runtime_error("only 16-bit PCM SPHERE files are supported");
    const bool big = field("sample_byte_format", "01") == "10";
    size_t n = std::min<size_t>((b.size() - hdr) / 2, (size_t)std::max<long>(count, 0) * std::max(w.channels, 1));
    w.pcm.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const unsigned char *q = &b[hdr + 2 * i];
        w.pcm[i] = (int16_t)(big ? ((q[0] << 8) | q[1]) : (q[0] | (q[1] << 8)));
    }
    if (w.channels < 1 || w.sample_rate <= 0 || w.pcm.empty()) throw std::runtime_error("Error while loading \"" + path + "\"");
    return w;
}

Wav read_wav(const std::string &path)
{
    FILE *f = std::fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("Can't open \"" + path + "\"");
    std::vector<unsigned char> b;
    std::fseek(f, 0, SEEK_END);
    const long fsz = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    if (fsz > 0) { // one read of the whole file
        b.resize((size_t)fsz);
        b.resize(std::fread(b.data(), 1, (size_t)fsz, f));
    } else { // (not seekable)
        unsigned char buf[65536];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + n);
    }
    std::fclose(f);
    if (b.size() >= 16 && std::memcmp(b.data(), "NIST_1A", 7) == 0) return read_sphere(b, path);
    if (b.size() < 12 || std::memcmp(b.data(), "RIFF", 4) != 0 || std::memcmp(b.data() + 8, "WAVE", 4) != 0)
        throw std::runtime_error("\"" + path + "\" is not a RIFF/WAVE or NIST SPHERE file");
    Wav w;
    size_t pos = 12;
    bool have_fmt = false;
    while (pos + 8 <= b.size()) {
        const uint32_t sz = rd32(&b[pos + 4]);
        const unsigned char *body = b.data() + pos + 8;
        if (std::memcmp(&b[pos], "fmt ", 4) == 0 && sz >= 16) {
            if (pos + 8 + 16 > b.size()) throw std::runtime_error("truncated fmt chunk in \"" + path + "\"");
            if (rd16(body) != 1 || rd16(body + 14) != 16) throw std::runtime_error("only 16-bit PCM is supported");
            w.channels = rd16(body + 2);
            w.sample_rate = (int)rd32(body + 4);
            have_fmt = true;
        } else if (std::memcmp(&b[pos], "data", 4) == 0) {
            const size_t avail = std::min<size_t>(sz, b.size() - pos - 8);
            w.pcm.resize(avail / 2);
            if (!w.pcm.empty()) std::memcpy(w.pcm.data(), body, w.pcm.size() * 2);
        }
        pos += (size_t)8 + sz + (sz & 1);
    }
    // (a fmt chunk that announces 0 channels or a rate of 0 is refused like an unreadable file: the downmix divides by the
    // channel count -- found reading the parser beside tools/fuzz_all.py's finds, round 4)
    if (!have_fmt || w.channels < 1 || w.sample_rate <= 0 || w.pcm.empty()) throw std::runtime_error("Error while loading \"" + path + "\"");
    return w;
}

// HTK parameter file: big-endian header {int32 nSamples, int32 sampPeriod [100 ns], int16 sampSize, int16 parmKind}
// then nSamples vectors of big-endian float32 (the reference's own binary branch is an empty stub,
// ASR_OCL.cpp:212,315-319).
void put_be32(FILE *f, uint32_t v)
{
    unsigned char b[4] = {(unsigned char)(v >> 24), (unsigned char)(v >> 16), (unsigned char)(v >> 8), (unsigned char)v};
    std::fwrite(b, 1, 4, f);
}
void put_be16(FILE *f, uint16_t v)
{
    unsigned char b[2] = {(unsigned char)(v >> 8), (unsigned char)v};
    std::fwrite(b, 1, 2, f);
}
void write_htk_header(FILE *f, uint32_t n_frames, const Options &o, int width)
{
    uint16_t kind = o.ceps > 0 ? 6 /* MFCC */ : 7 /* FBANK */;
    if (o.ceps > 0 && o.c0) kind |= 0x2000;  // _0
    if (o.dyn >= 1) kind |= 0x0100;          // _D
    if (o.dyn >= 2) kind |= 0x0200;          // _A
    if (o.norm == 1) kind |= 0x0800;         // _Z (zero mean)
    put_be32(f, n_frames);
    put_be32(f, (uint32_t)std::llround(o.shift_ms * 1e4));
    put_be16(f, (uint16_t)(4 * width));
    put_be16(f, kind);
}
void write_rows_htk(FILE *out, const float *rows, int n, int width, std::vector<char> &buf)
{
    const size_t count = (size_t)n * width;
    buf.resize(count * 4);
    for (size_t i = 0; i < count; ++i) { // big endian, one write for the block
        uint32_t u;
        std::memcpy(&u, &rows[i], 4);
        buf[4 * i] = (char)(u >> 24);
        buf[4 * i + 1] = (char)(u >> 16);
        buf[4 * i + 2] = (char)(u >> 8);
        buf[4 * i + 3] = (char)u;
    }
    std::fwrite(buf.data(), 1, buf.size(), out);
}

// "%f" of a double into p (std::to_chars with fixed notation and precision 6 is specified to produce what printf("%f")
// prints in the C locale, i.e. the reference's text, ASR_OCL.cpp:254-257 -- only without a format parse per value)
inline char *put_f(char *p, char *end, double v)
{
    if (!(v == v) || v - v != 0.0) return p + std::snprintf(p, (size_t)(end - p), "%f", v); // nan / inf as printf spells them
    return std::to_chars(p, end, v, std::chars_format::fixed, 6).ptr;
}

// "%f" of a FLOAT, exactly as printf prints it, without going through a general double formatter: the value is
// m * 2^e with a 24-bit m, so round(value * 10^6) is one 64-bit product, one shift and a round-half-to-even on the exact
// remainder (ties do occur: 1/128 = 0.0078125).  Values of 2^40 and more, NaN and infinities take the general path.
inline char *put_f32(char *p, char *end, float vf)
{
    uint32_t u;
    std::memcpy(&u, &vf, 4);
    const uint32_t ex = (u >> 23) & 0xffu;
    if (ex >= 127 + 40) return put_f(p, end, (double)vf);
    uint64_t m = u & 0x7fffffu;
    int e;
    if (ex == 0) {
        e = -149; // denormal (or zero)
    } else {
        m |= 0x800000u;
        e = (int)ex - 150;
    }
    uint64_t q;
    const uint64_t P = m * 1000000ull; // < 2^44
    if (e >= 0) {
        q = P << e; // e <= 16 here: < 2^60
    } else {
        const int sh = -e;
        if (sh >= 64) {
            q = 0;
        } else {
            q = P >> sh;
            const uint64_t rem = P & ((1ull << sh) - 1), half = 1ull << (sh - 1);
            if (rem > half || (rem == half && (q & 1))) ++q;
        }
    }
    if (u >> 31) *p++ = '-';
    const uint64_t ip = q / 1000000ull;
    uint32_t fp = (uint32_t)(q % 1000000ull);
    p = std::to_chars(p, end, ip).ptr;
    *p++ = '.';
    for (int i = 5; i >= 0; --i) {
        p[i] = (char)('0' + fp % 10);
        fp /= 10;
    }
    return p + 6;
}

// rows [r0, r1) formatted into p; returns the end
char *format_rows(char *p, char *end, const float *rows, int r0, int r1, int width, int first_frame, long double t0, long double dt)
{
    for (int f = r0; f < r1; ++f) {
        *p++ = '|';
        *p++ = ' ';
        p = put_f(p, end, (double)(t0 + (first_frame + f) * dt));
        *p++ = ' ';
        *p++ = '|';
        for (int i = 0; i < width; ++i) {
            *p++ = ' ';
            p = put_f32(p, end, rows[(size_t)width * f + i]);
            *p++ = ' ';
            *p++ = '|';
        }
        *p++ = '\n';
    }
    return p;
}

void write_rows(FILE *out, const float *rows, int n, int width, int first_frame, long double t0, long double dt,
                std::vector<char> &buf, int threads)
{
    // a value needs at most 1 + 39 + 1 + 6 characters plus " |"; the rows are formatted into one buffer (large blocks: by a
    // few threads, each into its own share of the buffer) and written in order
    const size_t per_row = ((size_t)width + 1) * 52 + 4;
    buf.resize((size_t)n * per_row + 16);
    const int T = (n >= 256 && threads > 1) ? std::min(threads, 4) : 1;
    if (T == 1) {
        char *p = format_rows(buf.data(), buf.data() + buf.size(), rows, 0, n, width, first_frame, t0, dt);
        std::fwrite(buf.data(), 1, (size_t)(p - buf.data()), out);
        return;
    }
    std::vector<std::thread> th;
    std::vector<size_t> used((size_t)T, 0);
    for (int t = 0; t < T; ++t) {
        const int r0 = (int)((int64_t)n * t / T), r1 = (int)((int64_t)n * (t + 1) / T);
        char *base = buf.data() + (size_t)r0 * per_row;
        th.emplace_back([=, &used] { used[t] = (size_t)(format_rows(base, base + (size_t)(r1 - r0) * per_row, rows, r0, r1, width, first_frame, t0, dt) - base); });
    }
    for (auto &x : th) x.join();
    for (int t = 0; t < T; ++t) std::fwrite(buf.data() + (size_t)((int64_t)n * t / T) * per_row, 1, used[t], out);
}

// --timing: wall time per phase, summed over the files of all workers (dev aid; printed at exit)
struct Timing {
    std::atomic<long long> wait_read{0}, prep{0}, device{0}, write{0}, files{0};
    std::atomic<long long> prep_wall{0}, write_wall{0}, wait_write{0}, batches{0}; // batched mode: wall time of the stages
    std::atomic<long long> t_main{0}, t_created{0}, t_loop_end{0}, t_destroyed{0};  // process milestones (last worker to pass)
    bool on = false;
} g_time;
inline long long now_ns()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (long long)ts.tv_sec * 1000000000ll + ts.tv_nsec;
}

// buffers a worker keeps from file to file (the row buffer alone is ~10 MB at the reference's default sample limit)
struct Scratch {
    std::vector<float> rows;
    std::vector<int16_t> mono;
    std::vector<char> text;
};

void process_file(MfccHip &param, const Options &o, const Wav &w, const std::string &in, const std::string &out_name,
                  float sample_rate, Scratch &sc)
{
    if ((float)w.sample_rate != sample_rate)
        throw std::runtime_error("File \"" + in + "\" has incorrect sample rate");
    // Multi-channel input: ONE policy across the product -- the downmix of the batch kernels (channels = 2 in the C
    // ABI: mono = (L + R) >> 1 in integer arithmetic), applied here on the host because the streaming
    // interface is mono like the reference's (which has no downmix at all: it reads `frames` shorts into a mono-sized
    // buffer, ASR_OCL.cpp:229-231).  More than two channels: the first two.
    const long long t_begin = g_time.on ? now_ns() : 0;
    long long t_dev = 0, t_wr = 0;
    std::vector<int16_t> &mono = sc.mono;
    mono.resize(w.pcm.size() / w.channels);
    for (size_t i = 0; i < mono.size(); ++i)
        mono[i] = w.channels >= 2 ? (int16_t)(((int)w.pcm[i * w.channels] + (int)w.pcm[i * w.channels + 1]) >> 1)
                                  : w.pcm[i * w.channels];

    const int limit = param.get_input_buffer_size();
    const int width = param.get_output_data_width();
    const int rows_cap = std::max(param.estimated_window_count(limit), 0) + 64;
    std::vector<float> &rows = sc.rows;
    if (rows.size() < (size_t)rows_cap * width) rows.resize((size_t)rows_cap * width);
    // Frame time column.  The reference divides its window / shift IN MILLISECONDS by the sample rate
    // (ASR_OCL.cpp:225-226: cfg.shift / cfg.sample_rate, 0.5f * cfg.window_size / cfg.sample_rate, float arithmetic):
    // 0.000625 s per frame at 10 ms / 16 kHz instead of 0.01 s.  --bug-compat 1 (default) prints
    // what the reference prints, --bug-compat 0 the times in seconds.
    const long double dt = o.bug_compat ? (long double)(o.shift_ms / sample_rate) : o.shift_ms / 1000.0L;
    const long double t0 = o.bug_compat ? (long double)(0.5f * o.window_ms / sample_rate) : 0.5L * o.window_ms / 1000.0L;

    std::vector<std::pair<float, FILE *>> outs;
    int idx = 0;
    for (float a = o.alpha_min; a <= o.alpha_max; a = o.alpha_min + (++idx) * o.alpha_step) {
        std::string name = out_name;
        if (o.alpha_max - o.alpha_min >= o.alpha_step) name += "." + std::to_string(a);
        FILE *fo = std::fopen(name.c_str(), o.htk ? "wb" : "w");
        if (!fo) throw std::runtime_error("Can't create output file: " + name);
        if (o.htk) write_htk_header(fo, 0, o, width); // frame count patched at the end
        outs.emplace_back(a, fo);
    }
    size_t pos = 0;
    int total = 0;
    std::vector<float> alphas;
    for (auto &oa : outs) alphas.push_back(oa.first);
    // One alpha: the reference's set_alpha + apply + get_output_data (ASR_OCL.cpp:236-243).  Several:
    // the whole sweep over the block's stored spectrum in one call (mfx_apply_alphas).
    auto emit = [&](int n) {
        if (n <= 0) return;
        long long t0e = g_time.on ? now_ns() : 0;
        if (alphas.size() > 1) param.apply_alphas(alphas.data(), (int)alphas.size());
        for (size_t i = 0; i < outs.size(); ++i) {
            if (alphas.size() > 1) {
                param.get_output_data_alpha((int)i, rows.data(), n);
            } else {
                param.set_alpha(alphas[i]);
                param.apply();
                param.get_output_data(rows.data(), n);
            }
            const long long t1e = g_time.on ? now_ns() : 0;
            t_dev += t1e - t0e;
            if (o.htk)
                write_rows_htk(outs[i].second, rows.data(), n, width, sc.text);
            else
                write_rows(outs[i].second, rows.data(), n, width, total, t0, dt, sc.text, o.format_threads);
            t0e = g_time.on ? now_ns() : 0;
            t_wr += t0e - t1e;
        }
    };
    while (pos < mono.size()) {
        const int n_in = (int)std::min<size_t>(mono.size() - pos, (size_t)limit);
        const long long ts0 = g_time.on ? now_ns() : 0;
        const int n = param.set_input(mono.data() + pos, n_in);
        if (g_time.on) t_dev += now_ns() - ts0;
        emit(n);
        total += n;
        pos += n_in;
    }
    const long long tf0 = g_time.on ? now_ns() : 0;
    const int n = param.flush();
    if (g_time.on) t_dev += now_ns() - tf0;
    if (n > 0) emit(n);
    total += n;
    for (auto &oa : outs) {
        if (o.htk) {
            std::fseek(oa.second, 0, SEEK_SET);
            write_htk_header(oa.second, (uint32_t)total, o, width);
        }
        std::fclose(oa.second);
    }
    if (g_time.on) {
        const long long t_end = now_ns();
        g_time.device += t_dev;
        g_time.write += t_wr;
        g_time.prep += (t_end - t_begin) - t_dev - t_wr; // downmix, opening and closing the outputs
        ++g_time.files;
    }
    static std::mutex print_lock;
    std::lock_guard<std::mutex> g(print_lock);
    std::printf("%s: %d frames x %d\n", in.c_str(), total, width);
}

// ---- batches of whole files -----------------------------------------------------------------------------------------
struct Pinned { // page-locked buffer (DMA straight from / to it), grown on demand
    void *p = nullptr;
    size_t bytes = 0;
    void *get(size_t need)
    {
        if (need > bytes) {
            mfx_free_pinned(p);
            bytes = need + need / 4;
            p = mfx_alloc_pinned(bytes);
            if (!p) throw std::runtime_error("can't allocate pinned host memory");
        }
        return p;
    }
    ~Pinned() { mfx_free_pinned(p); }
};

template <class F>
void parallel_for(size_t n, int threads, F fn)
{
    std::atomic<size_t> next{0};
    auto body = [&](int t) { // t = helper index: callers keep per-helper scratch
        for (size_t i; (i = next.fetch_add(1)) < n;) fn(i, t);
    };
    const int T = (int)std::min<size_t>((size_t)std::max(threads, 1), n);
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(body, t);
    body(0);
    for (auto &x : th) x.join();
}

struct BatchItem {
    size_t file = 0;            // index of the (input, output) pair
    std::vector<int16_t> mono;  // downmixed samples
    std::string error;          // reading / checking failed
    bool stream = false;        // not batchable: goes through the per-file loop
    Wav wav;                    // kept for the per-file loop
    long long off = 0, len = 0, row0 = 0, frames = 0;
};

struct Batch {
    std::vector<BatchItem> items;
    Pinned pcm, rows;
    long long samples = 0, total_rows = 0;
    int width = 0;
};

// Claims files from the shared queue until the batch holds `cap_samples`, reads and downmixes them (io threads) and
// packs the batchable ones back to back (even offsets: aligned 32-bit loads) into the pinned PCM buffer.
bool prepare_batch(Batch &b, const Options &o, const std::vector<std::string> &files, std::atomic<size_t> &next, float sr,
                   long long cap_samples, int limit, int W, int S, int D)
{
    struct WallTimer {
        long long t0 = g_time.on ? now_ns() : 0;
        ~WallTimer()
        {
            if (g_time.on) g_time.prep_wall += now_ns() - t0;
        }
    } wall_timer;
    b.items.clear();
    b.samples = 0;
    // file sizes bound the sample counts before anything is read: claim while the batch has room
    long long claimed = 0;
    while (claimed < cap_samples && b.items.size() < 4096) {
        const size_t i = next.fetch_add(1);
        if (2 * i + 1 >= files.size()) break;
        BatchItem it;
        it.file = i;
        b.items.push_back(std::move(it));
        FILE *f = std::fopen(files[2 * i].c_str(), "rb");
        long sz = 0;
        if (f) {
            std::fseek(f, 0, SEEK_END);
            sz = std::ftell(f);
            std::fclose(f);
        }
        claimed += std::max(sz, 0L) / 2;
    }
    if (b.items.empty()) return false;
    const bool sweep = o.alpha_max - o.alpha_min >= o.alpha_step;
    parallel_for(b.items.size(), o.io_threads, [&](size_t k, int) {
        BatchItem &it = b.items[k];
        try {
            it.wav = read_wav(files[2 * it.file]);
            if ((float)it.wav.sample_rate != sr) throw std::runtime_error("File \"" + files[2 * it.file] + "\" has incorrect sample rate");
            const size_t n = it.wav.pcm.size() / it.wav.channels;
            const long long T = n >= (size_t)W ? (long long)((n - (size_t)(W - S)) / (size_t)S) : 0;
            // the per-file loop keeps: files of more than one block, files too short for the deltas' context (the
            // reference refuses or mangles them: same messages from the same code), alpha sweeps
            it.stream = sweep || n > (size_t)limit || T < 2 * D + 1;
            if (!it.stream) {
                it.mono.resize(n);
                const int ch = it.wav.channels;
                for (size_t s2 = 0; s2 < n; ++s2)
                    it.mono[s2] = ch >= 2 ? (int16_t)(((int)it.wav.pcm[s2 * ch] + (int)it.wav.pcm[s2 * ch + 1]) >> 1) : it.wav.pcm[s2 * ch];
                it.wav.pcm.clear();
                it.wav.pcm.shrink_to_fit();
                it.len = (long long)n;
                it.frames = T;
            }
        } catch (const std::exception &e) {
            it.error = e.what();
        }
    });
    long long pos = 0;
    for (BatchItem &it : b.items)
        if (it.error.empty() && !it.stream) {
            it.off = pos;
            pos += it.len + (it.len & 1);
        }
    b.samples = pos;
    if (pos > 0) {
        int16_t *dst = (int16_t *)b.pcm.get((size_t)(pos + 8) * sizeof(int16_t));
        parallel_for(b.items.size(), o.io_threads, [&](size_t k, int) {
            BatchItem &it = b.items[k];
            if (!it.error.empty() || it.stream) return;
            std::memcpy(dst + it.off, it.mono.data(), (size_t)it.len * sizeof(int16_t));
            if (it.len & 1) dst[it.off + it.len] = 0;
            std::vector<int16_t>().swap(it.mono);
        });
    }
    return true;
}

// One worker of the reference's file queue (process_files_worker over the shared std::list,
// ASR_OCL.cpp:109-338,340-368): its own extractor on its own device, files drawn from a shared index
// until the queue is empty.  The reference runs its workers one after another (:365-366); these run
// concurrently, one per GPU (utterances are independent: SURVEY 8e, no exchange between devices).
void worker(const Options &o, int device, float sr, const std::vector<std::string> &files,
            std::atomic<size_t> &next, std::atomic<int> &failures)
{
    try {
        const long W = (long)(sr * o.window_ms * 1e-3), S = (long)(sr * o.shift_ms * 1e-3);
        // (a header that announces a sample rate of a few Hz -- found by mutating the reference's own sample1.wav, round 4 --
        // makes the window or the shift shorter than one sample: refused here, before anything divides by the shift)
        if (W < 1 || S < 1) throw std::runtime_error("window or shift shorter than one sample at this sample rate");
        // The extractor (HIP start-up, code object load, tables: 0.1-0.3 s of a fresh process) is created on a helper
        // thread while this one claims and reads the first batch of files: reading needs nothing from the device.
        auto make_param = [&] {
            std::unique_ptr<MfccHip> p(new MfccHip(o.sample_limit, (int)W, (int)S, o.banks, sr, o.low, o.high, o.ceps, o.c0, o.lift,
                                                  (Normalizer::norm_t)o.norm, (ParamBase::dyn_t)o.dyn, o.l1, o.l2,
                                                  o.norm_after_dyn, device, o.bug_compat,
                                                  o.batch_mb > 0 ? MFX_ENGINE_STREAM_KERNELS : 0));
            std::vector<float> window((size_t)W);
            for (long i = 0; i < W; ++i) // ASR_OCL.cpp:149-151
                window[i] = (float)(0.56f - 0.46f * std::cos((2.0f * M_PI * i) / W)) / 32768.f;
            p->set_window(window.data());
            if (g_time.on) g_time.t_created = now_ns();
            return p;
        };
        Batch slots[2]; // (declared before the extractor: their pinned buffers outlive nothing of it, and the first fills early)
        std::future<bool> prep;
        const int D = o.dyn == 0 ? 0 : o.dyn == 1 ? o.l1 : o.l1 + o.l2;
        const long long cap = (long long)o.batch_mb * 1024 * 1024 / 2;
        std::unique_ptr<MfccHip> param_owner;
        if (o.batch_mb > 0) {
            // largest block one set_input takes (parambase.cpp:12-13,16-19; the extractor reports the same number)
            const int limit0 = (int)std::floor(float(o.sample_limit - (W - S)) / S) * (int)S + (int)(W - S);
            std::future<std::unique_ptr<MfccHip>> created = std::async(std::launch::async, make_param);
            prep = std::async(std::launch::async, [&, limit0] {
                return prepare_batch(slots[0], o, files, next, sr, cap, limit0, (int)W, (int)S, D);
            });
            try {
                param_owner = created.get();
            } catch (...) {
                prep.wait();
                throw;
            }
            if (param_owner->get_input_buffer_size() != limit0) {
                prep.wait();
                throw std::runtime_error("input block size mismatch");
            }
        } else {
            param_owner = make_param();
        }
        MfccHip &param = *param_owner;
        struct LoopEnd { // (runs before the extractor is destroyed: declared after it)
            ~LoopEnd()
            {
                if (g_time.on) g_time.t_loop_end = now_ns();
            }
        } loop_end;
        Scratch sc;
        if (o.batch_mb > 0) {
            // ---- batches of whole files: read batch k + 1 and write batch k - 1 (helper threads) beside the device calls
            // of batch k (this thread)
            const int limit = param.get_input_buffer_size(), width = param.get_output_data_width();
            const int cols = width / (1 + o.dyn);
            const long double dt = o.bug_compat ? (long double)(o.shift_ms / sr) : o.shift_ms / 1000.0L;
            const long double t0 = o.bug_compat ? (long double)(0.5f * o.window_ms / sr) : 0.5L * o.window_ms / 1000.0L;
            std::vector<std::vector<char>> text_bufs((size_t)std::max(o.io_threads, 1));
            auto write_batch = [&](Batch *b) {
                const long long tww = g_time.on ? now_ns() : 0;
                const float *rows = (const float *)b->rows.p;
                parallel_for(b->items.size(), o.io_threads, [&](size_t k, int t) {
                    const BatchItem &it = b->items[k];
                    if (!it.error.empty() || it.stream) return;
                    try {
                        const long long tw0 = g_time.on ? now_ns() : 0;
                        const std::string &name = files[2 * it.file + 1];
                        FILE *fo = std::fopen(name.c_str(), o.htk ? "wb" : "w");
                        if (!fo) throw std::runtime_error("Can't create output file: " + name);
                        std::vector<char> &text = text_bufs[(size_t)t]; // (grown once per helper, not per file)
                        const float *r = rows + (size_t)it.row0 * b->width;
                        if (o.htk) {
                            write_htk_header(fo, (uint32_t)it.frames, o, b->width);
                            write_rows_htk(fo, r, (int)it.frames, b->width, text);
                        } else {
                            write_rows(fo, r, (int)it.frames, b->width, 0, t0, dt, text, 1);
                        }
                        std::fclose(fo);
                        if (g_time.on) {
                            g_time.write += now_ns() - tw0;
                            ++g_time.files;
                        }
                    } catch (const std::exception &e) {
                        std::fprintf(stderr, "Exception caught %s\n", e.what());
                        ++failures;
                    }
                });
                std::string log;
                for (const BatchItem &it : b->items)
                    if (it.error.empty() && !it.stream)
                        log += files[2 * it.file] + ": " + std::to_string(it.frames) + " frames x " + std::to_string(b->width) + "\n";
                {
                    static std::mutex print_lock;
                    std::lock_guard<std::mutex> g(print_lock);
                    std::fputs(log.c_str(), stdout);
                }
                if (g_time.on) g_time.write_wall += now_ns() - tww;
            };
            int cur = 0;
            std::future<void> writer;
            for (;;) {
                const long long tw0 = g_time.on ? now_ns() : 0;
                const bool have = prep.get();
                if (g_time.on) g_time.wait_read += now_ns() - tw0;
                if (!have) break;
                Batch &b = slots[cur];
                const int nxt = cur ^ 1;
                {
                    const long long t_ww = g_time.on ? now_ns() : 0;
                    if (writer.valid()) writer.get(); // the other slot's rows are on disk: it may be refilled
                    if (g_time.on) g_time.wait_write += now_ns() - t_ww, ++g_time.batches;
                }
                prep = std::async(std::launch::async, [&, nxt] {
                    return prepare_batch(slots[nxt], o, files, next, sr, cap, limit, (int)W, (int)S, D);
                });
                for (const BatchItem &it : b.items)
                    if (!it.error.empty()) { // a bad file does not stop the queue
                        std::fprintf(stderr, "Exception caught %s\n", it.error.c_str());
                        ++failures;
                    }
                // ---- device: one plan + one run for all batchable files
                std::vector<long long> off, len, row0;
                std::vector<BatchItem *> in_batch;
                for (BatchItem &it : b.items)
                    if (it.error.empty() && !it.stream) {
                        off.push_back(it.off);
                        len.push_back(it.len);
                        in_batch.push_back(&it);
                    }
                b.width = width;
                b.total_rows = 0;
                if (!in_batch.empty()) {
                    const long long td0 = g_time.on ? now_ns() : 0;
                    row0.resize(in_batch.size());
                    b.total_rows = param.batch_plan((int)in_batch.size(), off.data(), len.data(), row0.data());
                    float *rows = (float *)b.rows.get((size_t)std::max<long long>(b.total_rows, 1) * width * sizeof(float));
                    param.batch_run_host((const short *)b.pcm.p, b.samples, rows);
                    for (size_t k = 0; k < in_batch.size(); ++k) {
                        BatchItem &it = *in_batch[k];
                        it.row0 = row0[k];
                        if (it.frames != param.batch_frames(it.len)) throw std::runtime_error("frame count mismatch");
                        // The reference's flush after exactly one set_input reads its D static rows D rows early (B1,
                        // mfcccpu.cpp:439 + segmentercpu.cpp:97-106): rows T - D .. T - 1 repeat the statics of rows
                        // T - 2 D .. T - D - 1 (deltas unaffected).  The batch entries deliver the correct rows; the
                        // per-file loop this replaces reproduces the reference when --bug-compat is on, so do we.
                        if (o.bug_compat && D > 0)
                            for (int i = 0; i < D; ++i)
                                std::memcpy(rows + (size_t)(it.row0 + it.frames - D + i) * width,
                                            rows + (size_t)(it.row0 + it.frames - 2 * D + i) * width, (size_t)cols * sizeof(float));
                    }
                    if (g_time.on) g_time.device += now_ns() - td0;
                }
                // ---- files that need the per-file loop (longer than one block, too short, alpha sweeps)
                for (BatchItem &it : b.items)
                    if (it.error.empty() && it.stream) {
                        try {
                            process_file(param, o, it.wav, files[2 * it.file], files[2 * it.file + 1], sr, sc);
                        } catch (const std::exception &e) {
                            std::fprintf(stderr, "Exception caught %s\n", e.what());
                            ++failures;
                        }
                    }
                writer = std::async(std::launch::async, write_batch, &b);
                cur = nxt;
            }
            if (writer.valid()) writer.get();
            return;
        }
        // The worker's next file is claimed and read (its own thread) while the current one is on the GPU.
        // The worker's next file is claimed and read (its own thread) while the current one is on the GPU.
        struct Pending {
            size_t i;
            std::future<Wav> wav;
        };
        auto claim = [&](Pending &p) -> bool {
            p.i = next.fetch_add(1);
            if (2 * p.i + 1 >= files.size()) return false;
            p.wav = std::async(std::launch::async, read_wav, files[2 * p.i]);
            return true;
        };
        Pending cur, nxt;
        bool have = claim(cur);
        while (have) {
            const bool have_next = claim(nxt);
            try {
                const long long tw0 = g_time.on ? now_ns() : 0;
                const Wav w = cur.wav.get();
                if (g_time.on) g_time.wait_read += now_ns() - tw0;
                process_file(param, o, w, files[2 * cur.i], files[2 * cur.i + 1], sr, sc);
            } catch (const std::exception &e) { // a bad file does not stop the queue
                std::fprintf(stderr, "Exception caught %s\n", e.what());
                ++failures;
            }
            cur = std::move(nxt);
            have = have_next;
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Exception caught %s\n", e.what());
        ++failures;
    }
}

} // namespace

int main(int argc, char **argv)
{
    g_time.t_main = now_ns();
    Options o;
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "missing value for %s\n", a.c_str());
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "--window-size") o.window_ms = (float)std::atof(val());
        else if (a == "--shift") o.shift_ms = (float)std::atof(val());
        else if (a == "--banks") o.banks = std::atoi(val());
        else if (a == "--ceps") o.ceps = std::atoi(val());
        else if (a == "--c0") o.c0 = std::atoi(val()) != 0;
        else if (a == "--norm") o.norm = std::atoi(val());
        else if (a == "--dyn") o.dyn = std::atoi(val());
        else if (a == "--l1") o.l1 = std::atoi(val());
        else if (a == "--l2") o.l2 = std::atoi(val());
        else if (a == "--low-freq") o.low = (float)std::atof(val());
        else if (a == "--high-freq") o.high = (float)std::atof(val());
        else if (a == "--lift-coef") o.lift = (float)std::atof(val());
        else if (a == "--norm-after-dyn") o.norm_after_dyn = std::atoi(val()) != 0;
        else if (a == "--alpha") o.alpha_min = o.alpha_max = (float)std::atof(val());
        else if (a == "--alpha-min") o.alpha_min = (float)std::atof(val());
        else if (a == "--alpha-max") o.alpha_max = (float)std::atof(val());
        else if (a == "--alpha-step") o.alpha_step = (float)std::atof(val());
        else if (a == "--sample-limit") o.sample_limit = std::atoi(val());
        else if (a == "--dev") o.device = std::atoi(val());
        else if (a == "--devs") { // comma separated device list; an id may repeat (two workers on one GPU)
            for (const char *q = val(); *q;) {
                o.devices.push_back((int)std::strtol(q, const_cast<char **>(&q), 10));
                if (*q == ',') ++q;
                else if (*q) { std::fprintf(stderr, "bad --devs list\n"); return 2; }
            }
        }
        else if (a == "--bug-compat") o.bug_compat = std::atoi(val()) != 0;
        else if (a == "--htk") o.htk = true;
        else if (a == "--timing") g_time.on = true;
        else if (a == "--format-threads") o.format_threads = std::max(1, std::atoi(val()));
        else if (a == "--batch-mb") o.batch_mb = std::max(0, std::atoi(val()));
        else if (a == "--io-threads") o.io_threads = std::max(1, std::atoi(val()));
        else if (a == "--selftest-format") { // put_f32 against printf("%f") on n random bit patterns + the known hard cases
            const long n = std::atol(val());
            uint64_t st = 0x9E3779B97F4A7C15ull;
            long bad = 0;
            auto check = [&](float v) {
                char a1[128], a2[128];
                char *e1 = put_f32(a1, a1 + sizeof(a1), v);
                *e1 = 0;
                std::snprintf(a2, sizeof(a2), "%f", (double)v);
                if (std::strcmp(a1, a2) != 0 && ++bad < 10) std::fprintf(stderr, "put_f32 mismatch: %s vs %s\n", a1, a2);
            };
            const float hard[] = {0.f, -0.f, 1.f / 128, -1.f / 128, 0.5e-6f, 1.5e-6f, 2.5e-6f, 1e-10f, -1e-10f, 1e-45f, 123456.789f,
                                  -69.07755f, 0.9999995f, 0.99999994f, 9.9999995f, 16777216.f, 1e12f, 1.0995116e12f, 3.4e38f,
                                  -3.4e38f, 0.0078125f, 0.0234375f, 1.0000005f, 4194304.5f, 8388607.5f};
            for (float v : hard) check(v);
            check(std::nanf(""));
            check(INFINITY);
            check(-INFINITY);
            for (long i = 0; i < n; ++i) {
                st ^= st << 13, st ^= st >> 7, st ^= st << 17; // xorshift64
                uint32_t b = (uint32_t)(st >> 16);
                float v;
                std::memcpy(&v, &b, 4);
                check(v);
                // and values in the range features actually take
                check((float)((double)(int32_t)(st >> 33) * 1e-7));
            }
            std::printf("put_f32 selftest: %ld mismatches\n", bad);
            return bad ? 1 : 0;
        }
        else if (a == "--help") {
            std::printf("afet_hip [--window-size ms] [--shift ms] [--banks n] [--ceps n] [--c0 0|1] [--norm 0..3]\n"
                        "         [--dyn 0..2] [--l1 n] [--l2 n] [--low-freq hz] [--high-freq hz] [--lift-coef x]\n"
                        "         [--norm-after-dyn 0|1] [--alpha a | --alpha-min a --alpha-max b --alpha-step s]\n"
                        "         [--sample-limit n] [--dev n | --devs a,b,...] [--bug-compat 0|1] [--htk]  in.wav out.txt [...]\n"
                        "         [--batch-mb n (PCM per batch of whole files; 0 = per-file loop)] [--io-threads n]\n"
                        "  --devs: one worker per listed GPU, files dealt from a shared queue\n"
                        "  inputs: RIFF/WAVE or NIST SPHERE, 16-bit PCM; output: the reference's text rows, or HTK binary\n");
            return 0;
        } else files.push_back(a);
    }
    if (files.empty() || files.size() % 2) {
        std::fprintf(stderr, "usage: afet_hip [options] in.wav out.txt [in2.wav out2.txt ...]  (--help)\n");
        return 2;
    }
    try {
        const Wav first = read_wav(files[0]); // sample rate from the first file (ASR_OCL.cpp:342-358)
        const float sr = (float)first.sample_rate;
        if (o.high <= 0) o.high = sr / 2;
        if (o.devices.empty()) o.devices.push_back(o.device);
        std::atomic<size_t> next{0};
        std::atomic<int> failures{0};
        if (o.devices.size() == 1) {
            worker(o, o.devices[0], sr, files, next, failures);
        } else {
            std::vector<std::thread> pool;
            for (int d : o.devices) pool.emplace_back(worker, std::cref(o), d, sr, std::cref(files), std::ref(next), std::ref(failures));
            for (auto &t : pool) t.join();
        }
        if (g_time.on) g_time.t_destroyed = now_ns();
        if (g_time.on && g_time.files > 0) {
            const double n = (double)g_time.files.load();
            std::fprintf(stderr, "process milestones (ms): start -> extractor created %.1f, file loop %.1f (%.0f files/s inside the loop), "
                                 "extractor destroyed %.1f\n", (g_time.t_created - g_time.t_main) / 1e6,
                         (g_time.t_loop_end - g_time.t_created) / 1e6, n / ((g_time.t_loop_end - g_time.t_created) / 1e9),
                         (g_time.t_destroyed - g_time.t_loop_end) / 1e6);
            std::fprintf(stderr, "timing per file (us): waiting for the reader %.0f, host preparation %.0f, device calls %.0f, "
                                 "formatting + writing %.0f", g_time.wait_read / n / 1e3, g_time.prep / n / 1e3,
                         g_time.device / n / 1e3, g_time.write / n / 1e3);
            if (g_time.batches > 0)
                std::fprintf(stderr, "; %lld batches, stage wall time per file: read + pack %.0f, write %.0f, waiting for the writer %.0f",
                             g_time.batches.load(), g_time.prep_wall / n / 1e3, g_time.write_wall / n / 1e3, g_time.wait_write / n / 1e3);
            std::fprintf(stderr, "\n");
        }
        return failures.load() ? 1 : 0;
    } catch (const std::exception &e) {
        std::fprintf(stderr, "Exception caught %s\n", e.what());
        return 1;
    }
}
