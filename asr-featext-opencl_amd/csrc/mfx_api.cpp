// mfx_api.cpp -- the C ABI of include/mfx.h: handle, streaming state machine, batch planner.
//
// The streaming bookkeeping restates the reference's segmenter and apply() state machines
// (segmentercpu.cpp:56-106 / segmenteropencl.cpp:120-175, mfcccpu.cpp:371-425 /
// mfccopencl.cpp:495-549) on top of device buffers; all arithmetic on samples and features happens
// in the HIP kernels of mfx_front*.hip / mfx_tail.hip.  There is deliberately no CPU compute path in this file.
#include "../../include/mfx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mfx_kernels.h"
#include "mfx_tables.h"

using namespace mfx;

namespace {

const char *kMsgBuffer = "Can't process data, buffer is too small";
const char *kMsgWindow = "Can't process data, window count is too small";
const char *kMsgProcessed = "Processed samples <= 0, this should never happen";
const char *kMsgHigh = "Window count too high";
const char *kMsgPlanning = "planning handle (mfx_plan_create): no device behind it";

constexpr int kChunkFrames = 16; // frames per work item of the front-end kernels

// A PLANNING handle (mfx_plan_create) runs mfx_create's own code -- the predicates, the host-built tables, the LDS sums that
// decide which kernels a shape lands on -- with every device call left out: it can answer mfx_dominant_kernel_name and the
// geometry accessors, and nothing else (no buffer exists; every other entry fails with MFX_ERR_DEVICE).  It is how the
// shape -> kernel table of DESIGN.md section 5 is pinned by a test that needs no GPU.  It computes nothing.
thread_local bool t_planning = false;

template <class T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (count == 0 || t_planning) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

} // namespace

constexpr size_t kSmallBlock = (size_t)1 << 20; // below this a copy kernel replaces the DMA command (streaming interface)

struct mfx_handle {
    mfx_config cfg{};
    int device = 0;
    bool planning = false; // mfx_plan_create: no device behind this handle (see t_planning)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;

    // derived (mfccbase.cpp:18-30, mfcccpu.cpp:94-105)
    int W = 0, S = 0, W2 = 0, nb = 0, ceps = 0, dl = 0, cols = 0, width = 0;
    int l1 = 0, l2 = 0, D = 0;
    int input_window_limit = 0, input_buffer_size = 0, window_limit = 0, cap_rows = 0;
    int spec_pitch = 0;
    int channels = 1;
    bool fast512 = false;
    bool stuff256 = false; // fast512 serving 256-point transforms in the zero-stuffed form
    bool fast1024 = false; // 1024 points, window <= 512 samples: k_front1024 (two 256-point transforms per frame)
    int nm16 = 16;
    float alpha = 1.f, table_alpha = -1.f;
    bool have_window = false;

    // tables in HBM
    DevBuf<float> d_win1024o;
    DevBuf<float> d_window, d_winpair, d_twid_pass, d_twid_half, d_twid_split, d_twid_reg, d_mel_w, d_dct;
    DevBuf<int32_t> d_mel_beg;
    DevBuf<int32_t> d_mel64_L;                       // [8] of the handle's own plan (k_melcep reads L from memory)
    // VTLN sweep: one 64-lane plan per alpha, padded to a common row stride
    DevBuf<float> d_sweep64_w;
    DevBuf<int32_t> d_sweep64_start, d_sweep64_fid, d_sweep64_L;
    int sweep64_rs = 0;
    // 512-point kernel: per-lane mel plan + transposed DCT matrix
    DevBuf<float> d_mel_lane_w, d_dct_t;
    DevBuf<int32_t> d_mel_lane_start, d_mel_lane_fid;
    MelLanePlan plan;
    // wave-per-frame kernels (k_front_reg, fused): 64-lane mel plan + DCT operands for the matrix pipe
    MelWavePlan wplan;
    bool wplan_ok = false;
    DevBuf<float> d_mel64_w, d_dct_b;
    DevBuf<int32_t> d_mel64_start, d_mel64_fid;
    DevBuf<float> d_dct_b4;                          // k_front2048: DCT operands as 16-byte words
    DevBuf<float> d_dct_b4s;                         // k_front2048: the split form for <= 40 columns (or empty)
    int dct_split = 0;
    DevBuf<float> d_mel32_w;                         // k_front2048: the 32-lane plan
    DevBuf<int32_t> d_mel32_start, d_mel32_fid;
    MelWavePlan wplan32;
    bool fast2048 = false, wplan32_ok = false;
    int dct_tiles = 0, dct_ksteps = 0;
    int dct_stride = 0, nb_pad = 0;
    bool fused_ok = false;
    std::vector<float> h_dct;

    // streaming state (segmentercpu.h:7-17, parambase.h:18)
    DevBuf<int16_t> d_carry[2];
    int cur = 0;
    size_t carry_capacity = 0;
    int remaining = 0, samples = 0;
    bool flushed = true, last_calc_flushed = false, last_block = false;
    int block_wcnd = 0;      // frames (with context) the last FFT covered
    int block_frames = 0;    // frames apply() delivers
    bool block_applied = false;
    DevBuf<float> d_spec, d_src, d_blk, d_stats_stream;
    DevBuf<Chunk> d_chunks_stream;
    int stream_chunk_frames = 16;
    int n_chunks_stream_max = 0;
    int16_t *h_stage = nullptr; // pinned
    size_t h_stage_n = 0;
    // small-block handles (every block under 1 MB): the carried tail stays on the HOST, inside the pinned staging buffer, and
    // goes up again in front of the next block -- one copy kernel per set_input instead of copy + device-to-device tail copy
    bool host_tail = false;
    size_t stage_tail_off = 0;  // samples: where the pending tail (h->remaining samples) starts in h_stage
    float *h_out_stage = nullptr;         // pinned staging of get_output_data (allocated on first use)
    size_t h_out_stage_n = 0;
    bool rows_in_stage = false;           // the current block's rows were written straight into h_out_stage by the delta kernel
    hipEvent_t ev_copy[16] = {};          // chunk events of the pipelined device-to-host copy
    // VTLN sweep (mfx_apply_alphas): one filterbank, one static and one output block per alpha
    std::vector<float> sweep_alphas;      // alphas of the tables currently in d_sweep_w
    int sweep_cap = 0;                    // alphas the sweep buffers hold
    int sweep_n = 0;                      // alphas of the last sweep (0: last apply was a plain one)
    DevBuf<float> d_sweep_w, d_sweep_src, d_sweep_blk, d_sweep_stats;
    DevBuf<int32_t> d_sweep_beg;
    DevBuf<Segment> d_sweep_segs;         // [2][sweep_cap]: rows with context, rows delivered

    // batch plan
    int32_t n_utt = 0;
    int64_t total_rows = 0;
    std::vector<int64_t> utt_off, utt_len, utt_row;
    std::vector<Chunk> h_chunks;
    std::vector<int32_t> chunk_utt;      // utterance of every entry of h_chunks
    std::vector<int32_t> utt_chunk0;     // [n_utt + 1] first chunk of every utterance (chunks are in utterance order)
    hipStream_t stream_up = nullptr, stream_dn = nullptr; // sliced mfx_batch_run_host: upload / download beside the kernels
    hipEvent_t ev_up[16] = {}, ev_run[16] = {};
    DevBuf<Chunk> d_chunks;
    DevBuf<Segment> d_segs;
    DevBuf<float> d_stats_batch, d_spec_slab, d_host_out;
    DevBuf<double> d_norm_partial;       // chunk results of the normaliser's statistics (segments longer than 4096 rows)
    DevBuf<int16_t> d_host_pcm;          // mfx_batch_run_host: device copies of the caller's host buffers
    DevBuf<float> d_static16[2]; // compact [rows][16] statics between front end and delta (double buffered for overlap)
    // optional overlap of the delta/normalisation tail of batch i with the front end of batch i+1
    bool overlap = false;
    hipStream_t stream2 = nullptr;
    hipEvent_t ev_front[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
    bool tail_pending[2] = {false, false};
    unsigned batch_seq = 0;
    int tiles_max = 0;
    bool batch_aligned = true;
    // fused delta stage of the 512-point kernel: per-block chunk lists (own rows + halo) and delta tiles
    int num_cus = 256;
    bool fuse_delta_enabled = false; // mfx_config.engine & MFX_ENGINE_FUSE_DELTA opts in to the fused delta stage (measured 1-2 % slower
                                     // than front end + k_delta on C2; kept tested, off by default)
    bool fuse_plan = false;
    int f_blocks = 0, f_done_words = 0;
    int32_t f_nchunks = 0;
    DevBuf<Chunk> d_fchunks;
    DevBuf<int32_t> d_blk_chunk_off, d_blk_tile_off, d_err;
    DevBuf<DeltaTile> d_tiles;

    // profiling of the dominant kernel
    bool prof_on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_events;
    size_t prof_used = 0;
    int prof_launches = 0;
    double prof_ms = 0;
};

namespace {

int fail(mfx_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg;
    return code;
}

int fail_hip(mfx_handle *h, hipError_t e, const char *what)
{
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail(h, MFX_ERR_DEVICE, m);
}

#define HIP_TRY(h, expr)                                         \
    do {                                                         \
        hipError_t _e = (expr);                                  \
        if (_e != hipSuccess) return fail_hip((h), _e, #expr);   \
    } while (0)

template <class T>
hipError_t upload(DevBuf<T> &b, const std::vector<T> &v)
{
    hipError_t e = b.alloc(v.size());
    if (e != hipSuccess || v.empty() || t_planning) return e;
    return hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

void fill_front(const mfx_handle *h, FrontParams &p);

// rebuild the alpha-dependent mel table if needed (the reference re-derives it on every apply(),
// mfcccpu.cpp:194; here only when alpha actually changed)
int refresh_mel(mfx_handle *h)
{
    if (h->table_alpha == h->alpha && h->d_mel_w.p) return MFX_OK;
    MelTable t;
    build_mel_table(h->nb, h->W2, h->cfg.sample_rate, h->cfg.low_freq, h->cfg.high_freq, h->alpha, t);
    // every filter edge must address a computed bin
    for (int v : t.beg)
        if (v < 0 || v > h->W2 / 2) return fail(h, MFX_ERR_CONFIG, "mel filter edge outside [0, fft_size/2]");
    if (!t_planning) HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, upload(h->d_mel_w, t.weights));
    HIP_TRY(h, upload(h->d_mel_beg, t.beg));
    h->fused_ok = false;
    if (h->fast512 && build_mel_lane_plan(t, h->nb, h->W2, /*max_read_bin=*/511 - 32, h->plan)) {
        HIP_TRY(h, upload(h->d_mel_lane_w, h->plan.w));
        HIP_TRY(h, upload(h->d_mel_lane_start, h->plan.start));
        HIP_TRY(h, upload(h->d_mel_lane_fid, h->plan.fid));
        FrontParams probe;
        fill_front(h, probe);
        h->fused_ok = front512_lds_bytes(probe) <= 160 * 1024;
    }
    if (h->fast1024) {
        // bins up to 527 may be read (times zero weights): the slot keeps 264 words for each of the even / odd arrays
        h->fused_ok = false;
        if (build_mel_lane_plan(t, h->nb, h->W2, /*max_read_bin=*/527, h->plan, /*align=*/4)) {
            HIP_TRY(h, upload(h->d_mel_lane_w, h->plan.w));
            HIP_TRY(h, upload(h->d_mel_lane_start, h->plan.start));
            HIP_TRY(h, upload(h->d_mel_lane_fid, h->plan.fid));
            FrontParams probe;
            fill_front(h, probe);
            h->fused_ok = front1024_lds_bytes(probe) <= 160 * 1024;
        }
    }
    // The 64-lane plan (whole filters walked on a wave's lanes): k_front_reg / k_front_wave fused, and k_melcep -- the
    // streaming apply(), sweeps and the spectrum path of the batch entry, i.e. EVERY configuration needs it.  The kernels'
    // magnitude buffers hold W2 floats (bins 0 .. W2/2, finite words beyond).
    h->wplan_ok = false;
    if (!build_mel_wave_plan(t, h->nb, h->W2, /*max_read_bin=*/h->W2 - 1, h->wplan))
        return fail(h, MFX_ERR_CONFIG, "mel filterbank does not fit the kernels' lane plan (more than 512 filters?)");
    HIP_TRY(h, upload(h->d_mel64_w, h->wplan.w));
    HIP_TRY(h, upload(h->d_mel64_start, h->wplan.start));
    HIP_TRY(h, upload(h->d_mel64_fid, h->wplan.fid));
    {
        std::vector<int32_t> L(h->wplan.L, h->wplan.L + 8);
        HIP_TRY(h, upload(h->d_mel64_L, L));
    }
    {   // apply() runs k_melcep for every configuration: its tables (one weight row per filter-carrying lane, mel64_rows)
        // + one wave's buffers must fit the CU's LDS.  rows x row stride is at most ~4 x W2 floats for the reference's
        // triangular banks, so every transform up to 4096 points fits whatever the filter count (4096 points, 48 kHz,
        // 20 filters: 20 rows of 600 floats = 48 KB; refused until round 4, when all 64 lanes' rows were staged); 8192
        // points and more with few, wide filters do not: refuse here, not at the first apply()
        MelcepParams probe;
        std::memset(&probe, 0, sizeof(probe));
        probe.num_banks = h->nb;
        probe.mel64_rounds = h->wplan.rounds;
        probe.mel64_row_stride = h->wplan.row_stride;
        probe.mag_floats = std::max(h->W2, (h->spec_pitch + 3) & ~3);
        if (melcep_lds_bytes(probe, 1) > 160 * 1024)
            return fail(h, MFX_ERR_CONFIG, "mel filterbank does not fit the kernels' LDS (very wide filters on a long transform)");
    }
    h->wplan_ok = true;
    h->wplan32_ok = false;
    if (h->fast2048) { // k_front2048 walks the filters on the 32 lanes of each of a wave's two frames
        if (build_mel_wave_plan(t, h->nb, h->W2, /*max_read_bin=*/1039, h->wplan32, /*lanes=*/32)) {
            HIP_TRY(h, upload(h->d_mel32_w, h->wplan32.w));
            HIP_TRY(h, upload(h->d_mel32_start, h->wplan32.start));
            HIP_TRY(h, upload(h->d_mel32_fid, h->wplan32.fid));
            FrontParams probe;
            std::memset(&probe, 0, sizeof(probe));
            probe.num_banks = h->nb;
            probe.window_size = h->W; // (18 or 20 rows of window taps in LDS)
            probe.dct_ksteps = h->ceps > 0 ? (h->nb + 3) / 4 : 0;
            probe.mel32_rounds = h->wplan32.rounds;
            probe.mel32_row_stride = h->wplan32.row_stride;
            h->wplan32_ok = front2048_lds_bytes(probe) <= 160 * 1024;
        }
    }
    h->table_alpha = h->alpha;
    return MFX_OK;
}

void fill_front(const mfx_handle *h, FrontParams &p)
{
    std::memset(&p, 0, sizeof(p));
    p.channels = h->channels;
    p.pair_ok = (h->channels == 1 && (h->S % 2) == 0 && (h->W % 2) == 0 && h->batch_aligned) ? 1 : 0;
    p.window_size = h->W;
    p.shift = h->S;
    p.fft_size = h->W2;
    p.window = h->d_window.p;
    p.winpair = h->d_winpair.p;
    p.win1024o = h->d_win1024o.p;
    p.twid_pass = h->d_twid_pass.p;
    p.twid_half = h->d_twid_half.p;
    p.twid_reg = h->d_twid_reg.p;
    p.twid_split = h->d_twid_split.p;
    p.mel_w = h->d_mel_w.p;
    p.mel_beg = h->d_mel_beg.p;
    p.dct = h->ceps > 0 ? h->d_dct.p : nullptr;
    p.num_banks = h->nb;
    p.dct_len = h->dl;
    p.cols = h->cols;
    p.scale = 0.5f / (float)h->W2;
    p.mel_lane_w = h->d_mel_lane_w.p;
    p.mel_lane_start = h->d_mel_lane_start.p;
    p.mel_lane_fid = h->d_mel_lane_fid.p;
    p.dct_t = h->d_dct_t.p;
    p.mel_rounds = h->plan.rounds;
    p.mel_row_stride = h->plan.row_stride;
    for (int i = 0; i < 8; ++i) p.mel_L[i] = h->plan.L[i];
    p.dct_mode = (h->ceps > 0 && h->cols <= 16 && h->nb <= 40) ? 1 : 0; // DCT on the matrix pipe
    p.mel64_w = h->d_mel64_w.p;
    p.mel64_start = h->d_mel64_start.p;
    p.mel64_fid = h->d_mel64_fid.p;
    p.mel64_rounds = h->wplan_ok ? h->wplan.rounds : 0;
    p.mel64_row_stride = h->wplan_ok ? h->wplan.row_stride : 0;
    for (int i = 0; i < 8; ++i) p.mel64_L[i] = h->wplan.L[i];
    p.mel32_w = h->d_mel32_w.p;
    p.mel32_start = h->d_mel32_start.p;
    p.mel32_fid = h->d_mel32_fid.p;
    p.mel32_rounds = h->wplan32_ok ? h->wplan32.rounds : 0;
    p.mel32_row_stride = h->wplan32_ok ? h->wplan32.row_stride : 0;
    for (int i = 0; i < 8; ++i) p.mel32_L[i] = h->wplan32.L[i];
    p.dct_b = h->ceps > 0 ? h->d_dct_b.p : nullptr;
    p.stuff = h->stuff256 ? 512 / h->W2 : 0;
    p.dct_b4 = h->ceps > 0 ? h->d_dct_b4.p : nullptr;
    p.dct_b4s = (h->ceps > 0 && h->dct_split) ? h->d_dct_b4s.p : nullptr;
    p.dct_split = h->ceps > 0 ? h->dct_split : 0;
    p.dct_tiles = h->dct_tiles;
    p.dct_ksteps = h->dct_ksteps;
    p.dct_stride = h->dct_stride;
    p.nb_pad = h->nb_pad > 0 ? h->nb_pad : ((h->nb + 3) & ~3);
}

struct ProfScope {
    mfx_handle *h;
    hipEvent_t a = nullptr, b = nullptr;
    explicit ProfScope(mfx_handle *hh) : h(hh)
    {
        if (!h->prof_on) return;
        if (h->prof_used == h->prof_events.size()) {
            hipEvent_t x, y;
            if (hipEventCreate(&x) != hipSuccess || hipEventCreate(&y) != hipSuccess) return;
            h->prof_events.emplace_back(x, y);
        }
        a = h->prof_events[h->prof_used].first;
        b = h->prof_events[h->prof_used].second;
        ++h->prof_used;
        (void)hipEventRecord(a, h->stream);
    }
    ~ProfScope()
    {
        if (b) (void)hipEventRecord(b, h->stream);
    }
};

int prof_collect(mfx_handle *h)
{
    if (h->prof_used == 0) return MFX_OK;
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    for (size_t i = 0; i < h->prof_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, h->prof_events[i].first, h->prof_events[i].second) == hipSuccess) {
            h->prof_ms += ms;
            ++h->prof_launches;
        }
    }
    h->prof_used = 0;
    return MFX_OK;
}

// ---- normalisation helper: stats (unless reused) + apply over one column group
// groups > 1: the column groups col0 + g * cols (g < groups) each with statistics at stats + g * group_stats_stride
int run_norm(mfx_handle *h, float *data, int pitch, int col0, const Segment *segs, int n_segs, const Segment *seg0,
             int row_off, float *stats, bool use_last, int max_rows, int groups = 1, size_t group_stats_stride = 0)
{
    NormParams np;
    std::memset(&np, 0, sizeof(np));
    np.max_rows = max_rows;
    const size_t need = norm_partial_doubles(seg0 ? 1 : n_segs, max_rows, h->cols);
    if (need > h->d_norm_partial.n) { // (sized at create / plan time for the usual shapes: not reached in a timed loop)
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, h->d_norm_partial.alloc(need));
    }
    np.partial = h->d_norm_partial.p;
    np.data = data;
    np.pitch = pitch;
    np.col0 = col0;
    np.cols = h->cols;
    np.segs = segs;
    np.n_segs = n_segs;
    np.row_off = row_off;
    np.norm_type = h->cfg.norm;
    np.stats = stats;
    if (seg0) {
        np.inline_seg = 1;
        np.seg0 = *seg0;
        np.n_segs = 1;
    }
    if (!use_last && !(h->cfg.engine & MFX_ENGINE_NORM_TWO_KERNELS) && norm_fused_fits(max_rows, h->cols)) {
        np.groups = groups;
        np.group_stats_stride = (int64_t)group_stats_stride;
        HIP_TRY(h, launch_norm_fused(np, h->stream)); // short segments: one read of the rows, ONE launch for all groups
        return MFX_OK;
    }
    for (int g = 0; g < groups; ++g) {
        np.col0 = col0 + g * h->cols;
        np.stats = stats + (size_t)g * group_stats_stride;
        if (!use_last) HIP_TRY(h, launch_norm_stats(np, h->stream));
        HIP_TRY(h, launch_norm_apply(np, h->stream));
    }
    return MFX_OK;
}

} // namespace

// ------------------------------------------------------------------------------------------------
// lifetime
// ------------------------------------------------------------------------------------------------

extern "C" int mfx_abi_version(void) { return MFX_ABI_VERSION; }

extern "C" const char *mfx_status_string(int status)
{
    switch (status) {
    case MFX_OK: return "ok";
    case MFX_ERR_BUFFER_TOO_SMALL: return kMsgBuffer;
    case MFX_ERR_WINDOW_COUNT: return kMsgWindow;
    case MFX_ERR_PROCESSED: return kMsgProcessed;
    case MFX_ERR_WINDOW_HIGH: return kMsgHigh;
    case MFX_ERR_CONFIG: return "invalid configuration";
    case MFX_ERR_DEVICE: return "HIP device error";
    case MFX_ERR_ARG: return "invalid argument";
    case MFX_ERR_STATE: return "call out of sequence";
    default: return "unknown status";
    }
}

extern "C" const char *mfx_last_error(const mfx_handle *h) { return h ? h->err.c_str() : "null handle"; }

extern "C" void mfx_destroy(mfx_handle *h)
{
    if (!h) return;
    if (h->planning) { // nothing was allocated
        delete h;
        return;
    }
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto &ev : h->prof_events) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    h->d_window.release();
    h->d_winpair.release();
    h->d_win1024o.release();
    h->d_twid_pass.release();
    h->d_twid_half.release();
    h->d_twid_reg.release();
    h->d_twid_split.release();
    h->d_mel_w.release();
    h->d_dct.release();
    h->d_mel_beg.release();
    h->d_mel64_L.release();
    h->d_sweep64_w.release();
    h->d_sweep64_start.release();
    h->d_sweep64_fid.release();
    h->d_sweep64_L.release();
    h->d_mel_lane_w.release();
    h->d_mel64_w.release();
    h->d_mel64_start.release();
    h->d_mel64_fid.release();
    h->d_dct_b.release();
    h->d_mel32_w.release();
    h->d_dct_b4.release();
    h->d_dct_b4s.release();
    h->d_mel32_start.release();
    h->d_mel32_fid.release();
    h->d_dct_t.release();
    h->d_mel_lane_start.release();
    h->d_mel_lane_fid.release();
    h->d_carry[0].release();
    h->d_carry[1].release();
    h->d_spec.release();
    h->d_src.release();
    h->d_blk.release();
    h->d_stats_stream.release();
    h->d_norm_partial.release();
    h->d_host_pcm.release();
    h->d_host_out.release();
    h->d_fchunks.release();
    h->d_blk_chunk_off.release();
    h->d_blk_tile_off.release();
    h->d_tiles.release();
    h->d_err.release();
    h->d_sweep_w.release();
    h->d_sweep_beg.release();
    h->d_sweep_src.release();
    h->d_sweep_blk.release();
    h->d_sweep_stats.release();
    h->d_sweep_segs.release();
    h->d_chunks_stream.release();
    h->d_chunks.release();
    h->d_segs.release();
    h->d_stats_batch.release();
    h->d_spec_slab.release();
    h->d_static16[0].release();
    h->d_static16[1].release();
    if (h->stream2) {
        (void)hipStreamSynchronize(h->stream2);
        (void)hipStreamDestroy(h->stream2);
    }
    for (int i = 0; i < 2; ++i) {
        if (h->ev_front[i]) (void)hipEventDestroy(h->ev_front[i]);
        if (h->ev_tail[i]) (void)hipEventDestroy(h->ev_tail[i]);
    }
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->h_out_stage) (void)hipHostFree(h->h_out_stage);
    if (h->stream_up) (void)hipStreamDestroy(h->stream_up);
    if (h->stream_dn) (void)hipStreamDestroy(h->stream_dn);
    for (int i = 0; i < 16; ++i) {
        if (h->ev_up[i]) (void)hipEventDestroy(h->ev_up[i]);
        if (h->ev_run[i]) (void)hipEventDestroy(h->ev_run[i]);
    }
    for (auto &e : h->ev_copy)
        if (e) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

namespace {
int create_impl(const mfx_config *cfg, int hip_device, mfx_handle **out);
}

extern "C" int mfx_create(const mfx_config *cfg, int hip_device, mfx_handle **out)
{
    t_planning = false;
    return create_impl(cfg, hip_device, out);
}

/* A planning handle: see t_planning.  No device is touched; only mfx_dominant_kernel_name, the geometry accessors
 * (mfx_get_output_data_width, mfx_get_input_buffer_size, mfx_estimated_window_count, mfx_max_frames_out, mfx_fft_size),
 * mfx_last_error and mfx_destroy are meaningful on it. */
extern "C" int mfx_plan_create(const mfx_config *cfg, mfx_handle **out)
{
    t_planning = true;
    const int rc = create_impl(cfg, -1, out);
    t_planning = false;
    return rc;
}

namespace {
int create_impl(const mfx_config *cfg, int hip_device, mfx_handle **out)
{
    if (!cfg || !out) return MFX_ERR_ARG;
    *out = nullptr;
    if (cfg->window_size <= 0 || cfg->shift <= 0 || cfg->num_banks <= 0 || cfg->ceps_len < 0 ||
        cfg->sample_rate <= 0 || cfg->norm < 0 || cfg->norm > 3 || cfg->dyn < 0 || cfg->dyn > 2 ||
        cfg->channels < 0 || cfg->channels > 2)
        return MFX_ERR_CONFIG;
    if (cfg->ceps_len > 0 && cfg->lift_coef == 0.f) return MFX_ERR_CONFIG; // reference divides by lift_coef
    if (cfg->dyn != MFX_DYN_NONE && cfg->delta_l1 <= 0) return MFX_ERR_CONFIG;
    if (cfg->dyn == MFX_DYN_ACC && cfg->delta_l2 <= 0) return MFX_ERR_CONFIG;

    if (!t_planning) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || hip_device < 0 || hip_device >= ndev)
            return MFX_ERR_DEVICE; // no CPU fallback by design
        if (hipSetDevice(hip_device) != hipSuccess) return MFX_ERR_DEVICE;
    }

    mfx_handle *h = new mfx_handle();
    h->cfg = *cfg;
    h->device = hip_device;
    h->planning = t_planning;
    h->W = cfg->window_size;
    h->S = cfg->shift;
    h->nb = cfg->num_banks;
    h->ceps = cfg->ceps_len;
    h->l1 = cfg->dyn != MFX_DYN_NONE ? cfg->delta_l1 : 0;
    h->l2 = cfg->dyn == MFX_DYN_ACC ? cfg->delta_l2 : 0;
    h->D = h->l1 + h->l2;
    h->dl = cfg->want_c0 ? cfg->ceps_len + 1 : cfg->ceps_len;
    h->cols = h->ceps > 0 ? h->dl : h->nb;
    h->width = h->cols * (cfg->dyn == MFX_DYN_ACC ? 3 : cfg->dyn == MFX_DYN_DELTA ? 2 : 1);
    h->channels = cfg->channels == 2 ? 2 : 1;
    h->W2 = (int)ceil_pow2((uint32_t)h->W);
    if (cfg->fft_size != 0) {
        if (cfg->fft_size < h->W || (cfg->fft_size & (cfg->fft_size - 1)) != 0) {
            delete h;
            return MFX_ERR_CONFIG;
        }
        h->W2 = cfg->fft_size;
    }
    if (h->W2 < 64 || h->W2 > 4096) {
        delete h;
        return MFX_ERR_CONFIG;
    }
    // ParamBase ctor (parambase.cpp:4-14)
    h->input_window_limit = estimated_window_count_f32(cfg->input_buffer_size, h->W, h->S);
    h->input_buffer_size = h->input_window_limit * h->S + h->W - h->S;
    // MfccCpu ctor (mfcccpu.cpp:95-103)
    h->window_limit = h->input_window_limit + 2 + (cfg->dyn != MFX_DYN_NONE ? 3 * h->D : 0);
    if (h->input_window_limit <= 0 || h->window_limit <= 0) {
        delete h;
        return MFX_ERR_CONFIG;
    }
    h->spec_pitch = ((h->W2 / 2 + 1) + 3) & ~3;
    h->fast512 = front512_supported(h->W2, h->W, h->nb, h->cols, h->channels) && !(h->W2 < 512 && (h->cfg.engine & MFX_ENGINE_NO_STUFF256));
    h->stuff256 = h->fast512 && h->W2 < 512; // (256, 128 or 64 points: stuff factor 512 / W2)
    h->fast2048 = !(h->cfg.engine & MFX_ENGINE_NO_FRONT2048) && front2048_supported(h->W2, h->W, h->nb, h->cols, h->channels);
    h->fast1024 = !(h->cfg.engine & MFX_ENGINE_NO_FRONT1024) &&
                  front1024_supported(h->W2, h->W, h->nb, h->cols, h->channels, h->ceps);
    {
        hipDeviceProp_t prop;
        if (!t_planning && hipGetDeviceProperties(&prop, hip_device) == hipSuccess && prop.multiProcessorCount > 0)
            h->num_cus = prop.multiProcessorCount;
        h->fuse_delta_enabled = (h->cfg.engine & MFX_ENGINE_FUSE_DELTA) != 0;
    }
    // rows of the frame that carry window taps: 32 samples per row, or 16 / 8 / 4 in the zero-stuffed forms
    h->nm16 = h->stuff256 ? (h->W + h->W2 / 16 - 1) / (h->W2 / 16) : (h->W + 31) / 32;

    int rc = MFX_OK;
    auto bail = [&](int code) {
        std::string msg = h->err;
        mfx_destroy(h);
        (void)msg;
        return code;
    };
    if (!t_planning) {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) return bail(MFX_ERR_DEVICE);
        h->own_stream = true;
    }

    // ---- constant tables
    {
        std::vector<float> tw;
        build_twiddles(h->W2 / 2, h->W2 / 2, tw); // W_M^k, k < M (radix-4 stages use k, 2k, 3k)
        if (upload(h->d_twid_half, tw) != hipSuccess) return bail(MFX_ERR_DEVICE);
        // Split twiddles -i W_N^k, k <= N / 2.  The zero-stuffed forms (256 / 128 / 64 points on k_front512) run the 512-POINT
        // transform and its split: the kernel stages W_512^k for k < 128 whatever W2 is.  (Until round 4 this table had
        // W2 / 2 + 1 entries: at 128 and 64 points the kernel read 63 / 95 entries past its end.  The split's difference
        // term is rounding noise there, so fresh -- zero -- memory hid it; stale memory with large values did not:
        // found by tools/fuzz_all.py, seed 3 case 17.)
        const int WS = h->stuff256 ? 512 : h->W2;
        std::vector<float> ws;
        build_twiddles(WS, WS / 2 + 1, ws); // W_{WS}^k
        std::vector<float> split(ws.size());
        for (int k = 0; k <= WS / 2; ++k) { // -i * W = (wi, -wr)
            split[2 * k] = ws[2 * k + 1];
            split[2 * k + 1] = -ws[2 * k];
        }
        if (upload(h->d_twid_split, split) != hipSuccess) return bail(MFX_ERR_DEVICE);
        if (h->W2 >= 1024) {
            // k_front_reg: per-pass twiddle tables laid out [k][butterfly] so that the lanes of one LDS read
            // touch consecutive words.  Pass 1 (radix R1 over M points): W_M^(pp k), pp < M/R1; pass 2 (radix R1
            // over M/R1 points): W_M^(pp k R1), pp < M/R1^2.  Same values as the W_M^e table above.
            const int M = h->W2 / 2, R1 = h->W2 == 1024 ? 8 : 16, n1 = M / R1, n2 = M / (R1 * R1);
            std::vector<float> reg((size_t)2 * (R1 - 1) * (n1 + n2));
            size_t o = 0;
            for (int k = 1; k < R1; ++k)
                for (int pp = 0; pp < n1; ++pp, ++o) {
                    const int e = (pp * k) & (M - 1);
                    reg[2 * o] = tw[2 * e];
                    reg[2 * o + 1] = tw[2 * e + 1];
                }
            for (int k = 1; k < R1; ++k)
                for (int pp = 0; pp < n2; ++pp, ++o) {
                    const int e = (pp * k * R1) & (M - 1);
                    reg[2 * o] = tw[2 * e];
                    reg[2 * o + 1] = tw[2 * e + 1];
                }
            if (upload(h->d_twid_reg, reg) != hipSuccess) return bail(MFX_ERR_DEVICE);
        }
        if (h->fast512 || h->fast1024) {
            std::vector<float> full;
            build_twiddles(256, 256, full); // W_256^e
            std::vector<float> pass(16 * 16 * 2);
            for (int l = 0; l < 16; ++l)
                for (int k = 0; k < 16; ++k) {
                    int e = (l * k) & 255;
                    pass[2 * (l * 16 + k)] = full[2 * e];
                    pass[2 * (l * 16 + k) + 1] = full[2 * e + 1];
                }
            if (upload(h->d_twid_pass, pass) != hipSuccess) return bail(MFX_ERR_DEVICE);
        }
        if (h->ceps > 0) {
            std::vector<float> m;
            build_dct_matrix(h->nb, h->ceps, cfg->want_c0 != 0, cfg->lift_coef, m);
            if (upload(h->d_dct, m) != hipSuccess) return bail(MFX_ERR_DEVICE);
            h->h_dct = m;
            {
                std::vector<float> ob;
                build_dct_mfma_operands(m, h->nb, h->dl, h->dct_tiles, h->dct_ksteps, ob);
                if (upload(h->d_dct_b, ob) != hipSuccess) return bail(MFX_ERR_DEVICE);
                build_dct_mfma_operands4(m, h->nb, h->dl, ob); // the 4x4x1 form: k_front2048, k_front_wave, k_melcep
                if (upload(h->d_dct_b4, ob) != hipSuccess) return bail(MFX_ERR_DEVICE);
                h->dct_split = (h->cfg.engine & MFX_ENGINE_NO_DCT_SPLIT) ? 0 : dct_split_mode(h->nb, h->dl);
                if (h->dct_split) {
                    build_dct_mfma_operands4_split(m, h->nb, h->dl, ob);
                    if (upload(h->d_dct_b4s, ob) != hipSuccess) return bail(MFX_ERR_DEVICE);
                }
            }
            if (h->fast512) {
                std::vector<float> mt;
                build_dct_transposed(m, h->nb, h->dl, h->dct_stride, h->nb_pad, mt);
                if (upload(h->d_dct_t, mt) != hipSuccess) return bail(MFX_ERR_DEVICE);
            }
        }
    }
    rc = refresh_mel(h);
    if (rc != MFX_OK) return bail(rc);

    // ---- streaming buffers (capacity as the reference: segmentercpu.cpp:40-41, mfcccpu.cpp:104-112)
    // The reference sizes these from window_limit alone (segmentercpu.cpp:40-41, mfcccpu.cpp:104-112); a
    // steady-state block can need up to ~W/S more frames and W more samples than that (it writes past
    // its buffers when W - S > 2S or dyn is off), so capacity here carries that slack.
    h->cap_rows = h->window_limit + h->W / h->S + 4;
    h->carry_capacity = (size_t)h->cap_rows * h->S + 2 * (size_t)h->W;
    const size_t carry_alloc = (h->carry_capacity + h->W2 + 8) & ~(size_t)1;
    for (int i = 0; i < 2; ++i) {
        if (h->d_carry[i].alloc(carry_alloc) != hipSuccess) return bail(MFX_ERR_DEVICE);
        if (!t_planning && hipMemset(h->d_carry[i].p, 0, carry_alloc * sizeof(int16_t)) != hipSuccess) return bail(MFX_ERR_DEVICE);
    }
    if (h->d_spec.alloc((size_t)h->cap_rows * h->spec_pitch) != hipSuccess) return bail(MFX_ERR_DEVICE);
    if (h->d_src.alloc((size_t)h->cap_rows * h->cols) != hipSuccess) return bail(MFX_ERR_DEVICE);
    if (h->d_blk.alloc((size_t)h->cap_rows * h->width) != hipSuccess) return bail(MFX_ERR_DEVICE);
    if (h->d_stats_stream.alloc((size_t)3 * 2 * h->cols) != hipSuccess) return bail(MFX_ERR_DEVICE);
    if (cfg->norm != MFX_NORM_NONE) { // chunk results of the statistics over a long streaming block
        const size_t need = norm_partial_doubles(1, h->cap_rows, h->cols);
        if (need > 0 && h->d_norm_partial.alloc(need) != hipSuccess) return bail(MFX_ERR_DEVICE);
    }
    if (!t_planning && hipMemset(h->d_stats_stream.p, 0, (size_t)3 * 2 * h->cols * sizeof(float)) != hipSuccess)
        return bail(MFX_ERR_DEVICE);
    {
        // work items of a streaming block: 16 frames, or 4 where a whole block is only a few thousand frames (one 10-s
        // utterance = 62 items of 16 frames would occupy 4 of 256 CUs, every wave running 4 iterations back to back)
        h->stream_chunk_frames = h->cap_rows <= 16384 ? 4 : kChunkFrames;
        const int cf = h->stream_chunk_frames;
        h->n_chunks_stream_max = (h->cap_rows + cf - 1) / cf;
        std::vector<Chunk> ch(h->n_chunks_stream_max);
        for (int i = 0; i < h->n_chunks_stream_max; ++i) {
            ch[i].pcm_off = (int64_t)i * cf * h->S;
            ch[i].out_row = (int64_t)i * cf;
            ch[i].n_frames = cf;
            ch[i].pad = 0;
        }
        if (upload(h->d_chunks_stream, ch) != hipSuccess) return bail(MFX_ERR_DEVICE);
    }
    h->host_tail = !(h->cfg.engine & MFX_ENGINE_DMA_SMALL_BLOCKS) && (h->carry_capacity + 8) * sizeof(int16_t) < kSmallBlock;
    // (+ 16 bytes: small blocks are staged at the destination's alignment; host_tail: tail + block, up to the carry capacity)
    h->h_stage_n = (h->host_tail ? h->carry_capacity : (size_t)h->input_buffer_size) + 8;
    if (!t_planning && hipHostMalloc((void **)&h->h_stage, h->h_stage_n * sizeof(int16_t), hipHostMallocDefault) != hipSuccess)
        return bail(MFX_ERR_DEVICE);

    *out = h;
    return MFX_OK;
}
} // namespace

// ------------------------------------------------------------------------------------------------
// simple accessors
// ------------------------------------------------------------------------------------------------

extern "C" int mfx_get_output_data_width(const mfx_handle *h) { return h ? h->width : MFX_ERR_ARG; }
extern "C" int mfx_get_input_buffer_size(const mfx_handle *h) { return h ? h->input_buffer_size : MFX_ERR_ARG; }
extern "C" int mfx_estimated_window_count(const mfx_handle *h, int32_t samples)
{
    return h ? estimated_window_count_f32(samples, h->W, h->S) : MFX_ERR_ARG;
}
extern "C" int mfx_max_frames_out(const mfx_handle *h) { return h ? h->input_window_limit + h->W / h->S + 3 : MFX_ERR_ARG; }
extern "C" int mfx_fft_size(const mfx_handle *h) { return h ? h->W2 : MFX_ERR_ARG; }

extern "C" int mfx_set_alpha(mfx_handle *h, float alpha)
{
    if (!h) return MFX_ERR_ARG;
    h->alpha = alpha;
    return MFX_OK;
}

extern "C" int mfx_set_stream(mfx_handle *h, void *hip_stream)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->stream) HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return MFX_OK;
}

extern "C" int mfx_synchronize(mfx_handle *h)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->stream2) HIP_TRY(h, hipStreamSynchronize(h->stream2));
    h->tail_pending[0] = h->tail_pending[1] = false;
    if (h->d_err.p) { // the fused delta stage reports a wait that ran out (never expected) instead of hanging
        int32_t flag = 0;
        HIP_TRY(h, hipMemcpy(&flag, h->d_err.p, sizeof(flag), hipMemcpyDeviceToHost));
        if (flag != 0) {
            (void)hipMemset(h->d_err.p, 0, sizeof(flag));
            return fail(h, MFX_ERR_DEVICE, "fused delta stage gave up waiting for its statics");
        }
    }
    return MFX_OK;
}

extern "C" int mfx_profile_enable(mfx_handle *h, int enable)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    int rc = prof_collect(h);
    h->prof_on = enable != 0;
    return rc;
}

extern "C" int mfx_profile_read(mfx_handle *h, int32_t *launches, double *kernel_ms, int reset)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    int rc = prof_collect(h);
    if (rc != MFX_OK) return rc;
    if (launches) *launches = h->prof_launches;
    if (kernel_ms) *kernel_ms = h->prof_ms;
    if (reset) {
        h->prof_launches = 0;
        h->prof_ms = 0;
    }
    return MFX_OK;
}

namespace {
// Which front-end kernel the BATCH entries run for this handle -- the ONE place that decides it (run_batch_range launches what
// this returns; mfx_dominant_kernel_name prints it; DESIGN.md section 5 tabulates it; tests/test_host.py pins the table
// through planning handles).  Order of preference: the three register kernels (4 frames per wave at 512 points and the
// short-window 1024-point case, 2 frames per wave at 2048 points), then the fused one-wave-per-frame kernels while their LDS
// fits (<= 2048 points), then spectrum through an HBM slab + k_melcep.
enum FrontKind { kFront512, kFront1024, kFront2048, kFrontGenFused, kSpec512, kSpecGen };
FrontKind choose_front(const mfx_handle *h)
{
    const bool allow_fused = !(h->cfg.engine & MFX_ENGINE_STREAM_KERNELS); // (else: the streaming interface's kernels)
    if (allow_fused && h->fast512 && h->fused_ok) return kFront512;
    // (k_front1024, windows longer than 512 samples: aligned frames only)
    if (allow_fused && h->fast1024 && h->fused_ok && (h->W <= 512 || h->batch_aligned)) return kFront1024;
    // (2048 points, any window: stereo, mono on aligned sample pairs, mono at any alignment -- three builds)
    if (allow_fused && h->fast2048 && h->wplan32_ok) return kFront2048;
    // (up to 2048 points the fused form saves the spectrum's round trip through HBM -- 8 KB per frame at 2048 points; at 4096
    // points the tables + per-wave buffers no longer leave enough waves per CU)
    if (allow_fused && h->W2 <= 2048 && h->wplan_ok) {
        FrontParams probe;
        fill_front(h, probe);
        if (front_wave_lds_bytes(probe, true) <= 160 * 1024) return kFrontGenFused;
    }
    return h->fast512 ? kSpec512 : kSpecGen;
}
} // namespace

extern "C" const char *mfx_dominant_kernel_name(const mfx_handle *h)
{
    if (!h) return "";
    switch (choose_front(h)) { // names as rocprofv3 prints them
    case kFront512:
    case kSpec512: return "k_front512";
    case kFront1024: return "k_front1024";
    case kFront2048: return "k_front2048";
    default: return h->W2 >= 1024 ? "k_front_reg" : "k_front_wave";
    }
}

/* planning handles only: frames of the batch on aligned sample pairs (even offsets and shift) or not -- what mfx_batch_plan
 * derives from the caller's offsets on a real handle */
extern "C" int mfx_plan_set_aligned(mfx_handle *h, int aligned)
{
    if (!h || !h->planning) return MFX_ERR_ARG;
    h->batch_aligned = aligned != 0;
    return MFX_OK;
}

// ------------------------------------------------------------------------------------------------
// streaming interface
// ------------------------------------------------------------------------------------------------

extern "C" int mfx_set_window(mfx_handle *h, const float *window)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (!window) return fail(h, MFX_ERR_ARG, "invalid argument");
    HIP_TRY(h, hipSetDevice(h->device));
    std::vector<float> padded((size_t)h->W2, 0.f);
    std::memcpy(padded.data(), window, sizeof(float) * h->W);
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, upload(h->d_window, padded));
    if (h->fast1024 && h->W > 512) {
        // k_front1024, window longer than 512 samples: the taps of all 32 rows of sample pairs (with the output scale
        // folded in) and the twiddles W_512^n of the first 16 rows; the kernel folds the frame's halves itself
        const float fold = 0.5f / (float)h->W2;
        std::vector<float> tw512;
        build_twiddles(512, 256, tw512);
        std::vector<float> taps(16 * 32 * 2, 0.f), tw(16 * 16 * 2, 0.f);
        for (int l = 0; l < 16; ++l) {
            for (int m = 0; m < 32; ++m) {
                const int n = l + 16 * m;
                taps[2 * (l * 32 + m)] = padded[2 * n] * fold;
                taps[2 * (l * 32 + m) + 1] = padded[2 * n + 1] * fold;
            }
            for (int m = 0; m < 16; ++m) {
                const int n = l + 16 * m;
                tw[2 * (l * 16 + m)] = tw512[2 * n];
                tw[2 * (l * 16 + m) + 1] = tw512[2 * n + 1];
            }
        }
        HIP_TRY(h, upload(h->d_win1024o, taps));
        HIP_TRY(h, upload(h->d_winpair, tw));
    } else if (h->fast1024) {
        // phase O of k_front1024: (taps of pair n) x W_512^n as the real 2 x 2 form
        //   re = A x0 + B x1,  im = C x0 + D x1,   (A, B, C, D) = (t0 c, -t1 s, t0 s, t1 c),  W_512^n = c + i s
        // with the same output scale folded in (exact: a power of two)
        const float fold = 0.5f / (float)h->W2;
        std::vector<float> tw512;
        build_twiddles(512, 256, tw512);
        std::vector<float> wo(16 * 16 * 4, 0.f);
        for (int l = 0; l < 16; ++l)
            for (int m = 0; m < 16; ++m) {
                const int n = l + 16 * m;
                const float t0 = padded[2 * n] * fold, t1 = padded[2 * n + 1] * fold;
                const float c = tw512[2 * n], sn = tw512[2 * n + 1];
                float *q = &wo[4 * (l * 16 + m)];
                q[0] = t0 * c;
                q[1] = -(t1 * sn);
                q[2] = t0 * sn;
                q[3] = t1 * c;
            }
        HIP_TRY(h, upload(h->d_win1024o, wo));
    }
    if (h->fast512 || (h->fast1024 && h->W <= 512)) {
        // The 512-point kernel's copy carries the output scale 0.5 / W2 (1/2 of the real split, 1/W2 of
        // mfcccpu.cpp:203).  It is a power of two, so scaling the taps instead of the magnitudes changes no
        // bit of the result (every intermediate is the same value times 2^-10) and saves a multiply per bin.
        const float fold = 0.5f / (float)h->W2;
        std::vector<float> wp(16 * 16 * 2, 0.f);
        for (int l = 0; l < 16; ++l)
            for (int m = 0; m < 16; ++m) {
                int n = l + 16 * m;
                if (h->stuff256) { // zero-stuffed forms: packed sample n = (x[n / step], 0) where step divides n, else (0, 0)
                    const int step = 256 / h->W2;
                    if (n % step == 0 && n / step < h->W2) wp[2 * (l * 16 + m)] = padded[n / step] * fold;
                    continue;
                }
                wp[2 * (l * 16 + m)] = padded[2 * n] * fold;
                wp[2 * (l * 16 + m) + 1] = padded[2 * n + 1] * fold;
            }
        HIP_TRY(h, upload(h->d_winpair, wp));
    }
    h->have_window = true;
    return MFX_OK;
}

namespace {

// frame + window + FFT + magnitude over the first `wcnd` frames of the carry buffer
// ---- host side of the streaming copies --------------------------------------------------------------------------
// The drop-in interface hands over pageable host memory that the caller may overwrite on return (ASR_OCL.cpp:160-161,
// 231,243), so a block goes through pinned staging.  The staging copy is split over a few threads when it is large
// (one core moves ~10 GB/s, the link 50) and pipelined with the DMA in chunks; a caller buffer that is itself pinned
// (hipHostMalloc / hipHostRegister, a pinned torch tensor) is used by the DMA directly.
void host_copy(void *dst, const void *src, size_t bytes)
{
    const size_t kMin = (size_t)1 << 20;
    const unsigned nt = (unsigned)std::min<size_t>(4, bytes / kMin);
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    const size_t piece = ((bytes / nt) + 63) & ~(size_t)63;
    std::thread th[3];
    for (unsigned t = 1; t < nt; ++t) {
        const size_t off = piece * t, len = t + 1 == nt ? bytes - off : piece;
        th[t - 1] = std::thread([=] { std::memcpy((char *)dst + off, (const char *)src + off, len); });
    }
    std::memcpy(dst, src, piece);
    for (unsigned t = 1; t < nt; ++t) th[t - 1].join();
}

// page-locked host memory?  dev_ptr (optional): the address a kernel uses for it (the same address for hipHostMalloc memory;
// registered memory reports its own)
bool is_pinned_host(const void *p, void **dev_ptr = nullptr)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError(); // plain pageable memory: not an error
        return false;
    }
    if (a.type != hipMemoryTypeHost) return false;
    if (dev_ptr) *dev_ptr = a.devicePointer ? a.devicePointer : const_cast<void *>(p);
    return true;
}

constexpr size_t kCopyChunk = (size_t)4 << 20;

bool small_block(const mfx_handle *h, size_t bytes)
{
    return bytes > 0 && bytes < kSmallBlock && !(h->cfg.engine & MFX_ENGINE_DMA_SMALL_BLOCKS);
}

// host block -> device, asynchronous on the stream; `src` is free for the caller when this returns
int upload_block(mfx_handle *h, int16_t *d_dst, const int16_t *src, size_t samples, bool *direct)
{
    const size_t bytes = samples * sizeof(int16_t);
    *direct = bytes >= kCopyChunk && is_pinned_host(src);
    if (*direct) { // DMA straight from the caller's pinned buffer; the caller waits for it (wait_upload) before returning
        HIP_TRY(h, hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, h->stream));
        if (!h->ev_copy[0]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copy[0], hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->ev_copy[0], h->stream));
        return MFX_OK;
    }
    if (small_block(h, bytes)) {
        // a small block: into the pinned staging buffer at the destination's alignment, then a copy KERNEL reads it over the
        // link (one launch; a DMA command of this size costs more in latency than in transfer)
        char *stage = (char *)h->h_stage + ((uintptr_t)d_dst & 15);
        std::memcpy(stage, src, bytes);
        HIP_TRY(h, launch_copy_small(d_dst, stage, bytes, h->stream));
        return MFX_OK;
    }
    for (size_t off = 0; off < bytes; off += kCopyChunk) { // staging copy of chunk c+1 runs under the DMA of chunk c
        const size_t len = std::min(kCopyChunk, bytes - off);
        host_copy((char *)h->h_stage + off, (const char *)src + off, len);
        HIP_TRY(h, hipMemcpyAsync((char *)d_dst + off, (char *)h->h_stage + off, len, hipMemcpyHostToDevice, h->stream));
    }
    return MFX_OK;
}

// device rows -> host, returns when `dst` holds them
int download_rows(mfx_handle *h, float *dst, const float *d_src, size_t count)
{
    const size_t bytes = count * sizeof(float);
    if (small_block(h, bytes)) {
        // small: a copy kernel writes the rows into page-locked memory (the caller's buffer if it is pinned, else the
        // staging buffer at the source's alignment), one stream wait, one memcpy
        void *dst_dev = nullptr;
        if (is_pinned_host(dst, &dst_dev)) {
            HIP_TRY(h, launch_copy_small(dst_dev, d_src, bytes, h->stream));
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            return MFX_OK;
        }
        if (h->h_out_stage_n < count + 4) {
            if (h->h_out_stage) (void)hipHostFree(h->h_out_stage);
            h->h_out_stage = nullptr;
            h->h_out_stage_n = 0;
            const size_t want = std::max(count, (size_t)h->cap_rows * h->width) + 4;
            HIP_TRY(h, hipHostMalloc((void **)&h->h_out_stage, want * sizeof(float), hipHostMallocDefault));
            h->h_out_stage_n = want;
        }
        char *stage = (char *)h->h_out_stage + ((uintptr_t)d_src & 15);
        HIP_TRY(h, launch_copy_small(stage, d_src, bytes, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(dst, stage, bytes);
        return MFX_OK;
    }
    if (bytes < kCopyChunk || is_pinned_host(dst)) {
        HIP_TRY(h, hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        return MFX_OK;
    }
    if (h->h_out_stage_n < count) {
        if (h->h_out_stage) (void)hipHostFree(h->h_out_stage);
        h->h_out_stage = nullptr;
        h->h_out_stage_n = 0;
        const size_t want = std::max(count, (size_t)h->cap_rows * h->width);
        HIP_TRY(h, hipHostMalloc((void **)&h->h_out_stage, want * sizeof(float), hipHostMallocDefault));
        h->h_out_stage_n = want;
    }
    // chunks of the DMA into pinned staging, each followed by an event; the copy out of staging of chunk c runs under
    // the DMA of chunk c+1
    const size_t chunk = std::max(kCopyChunk, (bytes / 16 + 4095) & ~(size_t)4095);
    int n = 0;
    for (size_t off = 0; off < bytes; off += chunk, ++n) {
        const size_t len = std::min(chunk, bytes - off);
        HIP_TRY(h, hipMemcpyAsync((char *)h->h_out_stage + off, (const char *)d_src + off, len, hipMemcpyDeviceToHost, h->stream));
        if (!h->ev_copy[n]) HIP_TRY(h, hipEventCreateWithFlags(&h->ev_copy[n], hipEventDisableTiming));
        HIP_TRY(h, hipEventRecord(h->ev_copy[n], h->stream));
    }
    n = 0;
    for (size_t off = 0; off < bytes; off += chunk, ++n) {
        const size_t len = std::min(chunk, bytes - off);
        HIP_TRY(h, hipEventSynchronize(h->ev_copy[n]));
        host_copy((char *)dst + off, (const char *)h->h_out_stage + off, len);
    }
    return MFX_OK;
}

int stream_front(mfx_handle *h, int wcnd)
{
    FrontParams p;
    fill_front(h, p);
    p.pcm = h->d_carry[h->cur].p;
    p.pcm_total = (int64_t)h->d_carry[h->cur].n;
    p.chunks = h->d_chunks_stream.p;
    p.n_chunks = (wcnd + h->stream_chunk_frames - 1) / h->stream_chunk_frames;
    p.row_limit = wcnd;
    p.channels = 1;
    p.pair_ok = ((h->S % 2) == 0 && (h->W % 2) == 0) ? 1 : 0; // carry-buffer frames start at multiples of S
    p.spec = h->d_spec.p;
    p.spec_pitch = h->spec_pitch;
    if (h->fast512)
        HIP_TRY(h, launch_front512(p, /*to_spectrum=*/true, /*aligned=*/(h->S % 2) == 0, h->nm16, h->stream));
    else
        HIP_TRY(h, launch_front_generic(p, /*fused=*/false, h->stream));
    h->block_wcnd = wcnd;
    return MFX_OK;
}

// move the unconsumed tail to the front of the other carry buffer (the reference copies inside
// one buffer with overlapping ranges: segmentercpu.cpp:73,92 / segmenteropencl.cpp:139,160)
int carry_tail(mfx_handle *h, int total_samples)
{
    const int other = h->cur ^ 1;
    if (h->remaining > 0) {
        const size_t bytes = sizeof(int16_t) * (size_t)h->remaining;
        const int16_t *src = h->d_carry[h->cur].p + (total_samples - h->remaining);
        if (small_block(h, bytes))
            HIP_TRY(h, launch_copy_small(h->d_carry[other].p, src, bytes, h->stream));
        else
            HIP_TRY(h, hipMemcpyAsync(h->d_carry[other].p, src, bytes, hipMemcpyDeviceToDevice, h->stream));
    }
    h->cur = other;
    return MFX_OK;
}

} // namespace

extern "C" int mfx_set_input(mfx_handle *h, const int16_t *pcm, int32_t samples, int32_t *frames_out)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (!pcm || !frames_out || samples < 0) return fail(h, MFX_ERR_ARG, "invalid argument");
    *frames_out = 0;
    if (!h->have_window) return fail(h, MFX_ERR_STATE, "set_window has not been called");
    if (samples > h->input_buffer_size) return fail(h, MFX_ERR_BUFFER_TOO_SMALL, kMsgBuffer);
    HIP_TRY(h, hipSetDevice(h->device));
    h->last_block = false; // a new stream may follow a flush (reference never resets this: DESIGN.md B7)
    h->block_applied = false;
    h->block_frames = 0;

    const int D = h->D, W = h->W, S = h->S;
    // the caller may overwrite `pcm` as soon as we return: the block goes through pinned staging (upload_block), whose
    // previous contents the stream has long consumed (get_output_data / mfx_synchronize waited for it)
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    bool direct = false;
    // `pcm` must be free for the caller on EVERY return: when the block is DMA'd straight from the caller's pinned buffer
    // (upload_block sets `direct`), any exit -- the error returns below included -- first waits for that copy
    struct DirectWait {
        mfx_handle *h;
        const bool *direct;
        ~DirectWait()
        {
            if (*direct && h->ev_copy[0]) (void)hipEventSynchronize(h->ev_copy[0]);
        }
    } direct_wait{h, &direct};

    h->last_calc_flushed = h->flushed;
    int window_count = 0, wcnd = 0;
    if (h->last_calc_flushed) {
        if (h->host_tail) {
            std::memcpy(h->h_stage, pcm, (size_t)samples * sizeof(int16_t));
            HIP_TRY(h, launch_copy_small(h->d_carry[h->cur].p, h->h_stage, (size_t)samples * sizeof(int16_t), h->stream));
        } else {
            int rcu = upload_block(h, h->d_carry[h->cur].p, pcm, (size_t)samples, &direct);
            if (rcu != MFX_OK) return rcu;
        }
        wcnd = estimated_window_count_f32(samples, W, S);
        window_count = wcnd - D;
        if (window_count <= 0) return fail(h, MFX_ERR_WINDOW_COUNT, kMsgWindow);
        const int processed = (window_count - D) * S + W - S;
        // B13: a first block of fewer than 2 D frames.  The reference guards `processed <= 0` only (segmentercpu.cpp:70-71);
        // for D < frames < 2 D with W - S > (D - window_count) S it goes on and copies its carry-over from BEFORE the start
        // of its buffer (m_tmpbuffer + samples - m_remaining_samples is negative, :72-73) -- undefined there, refused here
        // with the message the reference's own guard carries.
        if (processed <= 0 || window_count < D) return fail(h, MFX_ERR_PROCESSED, kMsgProcessed);
        int rc = stream_front(h, wcnd);
        if (rc != MFX_OK) return rc;
        h->remaining = samples - processed + W - S;
        if (h->host_tail) {
            h->stage_tail_off = (size_t)(samples - h->remaining);
        } else {
            rc = carry_tail(h, samples);
            if (rc != MFX_OK) return rc;
        }
        h->flushed = false;
        h->samples = samples;
    } else {
        if ((size_t)samples + (size_t)h->remaining > h->carry_capacity) return fail(h, MFX_ERR_BUFFER_TOO_SMALL, kMsgBuffer);
        if (h->host_tail) {
            // the pending tail moves to the front of the staging buffer (the stream is idle: no kernel is reading it), the
            // block goes behind it, and ONE copy kernel takes both to the front of the carry buffer
            if (h->stage_tail_off > 0 && h->remaining > 0)
                std::memmove(h->h_stage, h->h_stage + h->stage_tail_off, (size_t)h->remaining * sizeof(int16_t));
            h->stage_tail_off = 0;
            std::memcpy(h->h_stage + h->remaining, pcm, (size_t)samples * sizeof(int16_t));
            HIP_TRY(h, launch_copy_small(h->d_carry[h->cur].p, h->h_stage,
                                         ((size_t)h->remaining + (size_t)samples) * sizeof(int16_t), h->stream));
        } else {
            int rcu = upload_block(h, h->d_carry[h->cur].p + h->remaining, pcm, (size_t)samples, &direct);
            if (rcu != MFX_OK) return rcu;
        }
        const int total = samples + h->remaining;
        wcnd = estimated_window_count_f32(total, W, S);
        window_count = wcnd - 2 * D;
        if (window_count > 0) {
            int rc = stream_front(h, wcnd);
            if (rc != MFX_OK) return rc;
        } else
            window_count = 0;
        const int processed = window_count * S + W - S;
        h->remaining = total - processed + W - S;
        if (h->host_tail) {
            h->stage_tail_off = (size_t)(total - h->remaining);
        } else {
            int rc = carry_tail(h, total);
            if (rc != MFX_OK) return rc;
        }
        h->samples = total;
    }
    h->block_frames = window_count;
    *frames_out = window_count;
    if (direct) HIP_TRY(h, hipEventSynchronize(h->ev_copy[0])); // DMA from the caller's own (pinned) buffer: done before we return
    return MFX_OK;
}

extern "C" int mfx_flush(mfx_handle *h, int32_t *frames_out)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (!frames_out) return fail(h, MFX_ERR_ARG, "invalid argument");
    *frames_out = 0;
    if (h->last_block) return MFX_OK; // nothing to flush (mfcccpu.cpp:350-351)
    HIP_TRY(h, hipSetDevice(h->device));
    h->last_block = true;
    h->flushed = true;
    h->block_applied = false;
    h->block_frames = 0;
    const int wcnd = estimated_window_count_f32(h->remaining, h->W, h->S);
    const int window_count = wcnd - h->D;
    if (window_count <= 0) return MFX_OK;
    if (h->host_tail && h->remaining > 0) { // the tail is on the host: up it goes, to the front of the carry buffer
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (h->stage_tail_off > 0)
            std::memmove(h->h_stage, h->h_stage + h->stage_tail_off, (size_t)h->remaining * sizeof(int16_t));
        h->stage_tail_off = 0;
        HIP_TRY(h, launch_copy_small(h->d_carry[h->cur].p, h->h_stage, (size_t)h->remaining * sizeof(int16_t), h->stream));
    }
    int rc = stream_front(h, wcnd);
    if (rc != MFX_OK) return rc;
    h->block_frames = window_count;
    *frames_out = window_count;
    return MFX_OK;
}

namespace {

// (Re)build the filterbanks of a sweep and size its buffers.
int prepare_sweep(mfx_handle *h, const float *alphas, int n)
{
    const size_t wstride = (size_t)2 * h->W2, bstride = (size_t)h->nb + 2;
    const bool same = (int)h->sweep_alphas.size() == n && std::equal(alphas, alphas + n, h->sweep_alphas.begin());
    if (n > h->sweep_cap) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        HIP_TRY(h, h->d_sweep_w.alloc(wstride * n));
        HIP_TRY(h, h->d_sweep_beg.alloc(bstride * n));
        HIP_TRY(h, h->d_sweep_src.alloc((size_t)n * h->cap_rows * h->cols));
        HIP_TRY(h, h->d_sweep_blk.alloc((size_t)n * h->cap_rows * h->width));
        HIP_TRY(h, h->d_sweep_stats.alloc((size_t)3 * n * 2 * h->cols));
        HIP_TRY(h, hipMemset(h->d_sweep_stats.p, 0, (size_t)3 * n * 2 * h->cols * sizeof(float)));
        HIP_TRY(h, h->d_sweep_segs.alloc((size_t)2 * n));
        h->sweep_cap = n;
    }
    if (same && n <= h->sweep_cap && !h->sweep_alphas.empty()) return MFX_OK;
    std::vector<float> w(wstride * n);
    std::vector<int32_t> b(bstride * n);
    std::vector<MelWavePlan> plans((size_t)n);
    int rs = 4;
    for (int a = 0; a < n; ++a) {
        MelTable t;
        build_mel_table(h->nb, h->W2, h->cfg.sample_rate, h->cfg.low_freq, h->cfg.high_freq, alphas[a], t);
        for (int v : t.beg)
            if (v < 0 || v > h->W2 / 2) return fail(h, MFX_ERR_CONFIG, "mel filter edge outside [0, fft_size/2]");
        std::copy(t.weights.begin(), t.weights.end(), w.begin() + wstride * a);
        std::copy(t.beg.begin(), t.beg.end(), b.begin() + bstride * a);
        if (!build_mel_wave_plan(t, h->nb, h->W2, /*max_read_bin=*/h->W2 - 1, plans[a]))
            return fail(h, MFX_ERR_CONFIG, "mel filterbank does not fit the kernels' lane plan");
        rs = std::max(rs, plans[a].row_stride);
    }
    // one 64-lane plan per alpha, the weight rows padded to the longest plan's stride (a row's rounds lie back to back from
    // its start, so padding at the end changes nothing)
    const int rounds = plans[0].rounds;
    std::vector<float> pw((size_t)n * 64 * rs, 0.f);
    std::vector<int32_t> pst((size_t)n * 64 * rounds), pfid((size_t)n * 64 * rounds), pL((size_t)n * 8);
    for (int a = 0; a < n; ++a) {
        for (int j = 0; j < 64; ++j)
            std::copy(plans[a].w.begin() + (size_t)j * plans[a].row_stride, plans[a].w.begin() + (size_t)(j + 1) * plans[a].row_stride,
                      pw.begin() + ((size_t)a * 64 + j) * rs);
        std::copy(plans[a].start.begin(), plans[a].start.end(), pst.begin() + (size_t)a * 64 * rounds);
        std::copy(plans[a].fid.begin(), plans[a].fid.end(), pfid.begin() + (size_t)a * 64 * rounds);
        std::copy(plans[a].L, plans[a].L + 8, pL.begin() + (size_t)a * 8);
    }
    {   // the sweep's common row stride may exceed the handle's own plan's: check k_melcep's LDS here (a CONFIG error
        // at the call, not a launch failure later)
        MelcepParams probe;
        std::memset(&probe, 0, sizeof(probe));
        probe.num_banks = h->nb;
        probe.mel64_rounds = rounds;
        probe.mel64_row_stride = rs;
        probe.mag_floats = std::max(h->W2, (h->spec_pitch + 3) & ~3);
        if (melcep_lds_bytes(probe, 1) > 160 * 1024)
            return fail(h, MFX_ERR_CONFIG, "a warped mel filterbank of the sweep does not fit the kernels' LDS");
    }
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(h->d_sweep_w.p, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_sweep_beg.p, b.data(), b.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(h, upload(h->d_sweep64_w, pw));
    HIP_TRY(h, upload(h->d_sweep64_start, pst));
    HIP_TRY(h, upload(h->d_sweep64_fid, pfid));
    HIP_TRY(h, upload(h->d_sweep64_L, pL));
    h->sweep64_rs = rs;
    h->sweep_alphas.assign(alphas, alphas + n);
    return MFX_OK;
}

// k_melcep parameters that do not depend on the caller: the handle's own filterbank (sweep = false) or the sweep's tables
void fill_melcep(const mfx_handle *h, MelcepParams &mp, bool sweep)
{
    std::memset(&mp, 0, sizeof(mp));
    mp.spec_pitch = h->spec_pitch;
    mp.fft_size = h->W2;
    mp.mel_w = sweep ? h->d_sweep_w.p : h->d_mel_w.p;
    mp.mel_beg = sweep ? h->d_sweep_beg.p : h->d_mel_beg.p;
    mp.mel64_w = sweep ? h->d_sweep64_w.p : h->d_mel64_w.p;
    mp.mel64_start = sweep ? h->d_sweep64_start.p : h->d_mel64_start.p;
    mp.mel64_fid = sweep ? h->d_sweep64_fid.p : h->d_mel64_fid.p;
    mp.mel64_L = sweep ? h->d_sweep64_L.p : h->d_mel64_L.p;
    mp.mel64_rounds = h->wplan.rounds;
    mp.mel64_row_stride = sweep ? h->sweep64_rs : h->wplan.row_stride;
    mp.mag_floats = std::max(h->W2, (h->spec_pitch + 3) & ~3);
    mp.dct = h->ceps > 0 ? h->d_dct.p : nullptr;
    mp.dct_b4 = h->ceps > 0 ? h->d_dct_b4.p : nullptr;
    mp.dct_ksteps = h->dct_ksteps;
    mp.num_banks = h->nb;
    mp.dct_len = h->dl;
    mp.cols = h->cols;
    mp.n_tables = 1;
    mp.mel_w_stride = (int64_t)2 * h->W2;
    mp.mel_beg_stride = h->nb + 2;
}

// apply() for the current block: n_alpha == 0 -> the handle's alpha into d_src/d_blk (ParamBase::apply);
// n_alpha >= 1 -> every alpha of the list from the same stored spectrum, alpha a into block a of
// d_sweep_src/d_sweep_blk (the reference's alpha loop ASR_OCL.cpp:236-243 as one launch per stage).
int apply_impl(mfx_handle *h, const float *alphas, int n_alpha)
{
    HIP_TRY(h, hipSetDevice(h->device));
    const int D = h->D;
    int wcnd, wc;
    bool first = false, last = false, use_last = false;
    // the three cases of mfcccpu.cpp:371-425
    if (h->last_block) {
        wcnd = estimated_window_count_f32(h->remaining, h->W, h->S);
        wc = wcnd - D;
        last = true;
        use_last = true;
        if (wc <= 0) return MFX_OK;
    } else if (h->last_calc_flushed) {
        wcnd = estimated_window_count_f32(h->samples, h->W, h->S);
        wc = wcnd - D;
        first = true;
        if (wc <= 0) return fail(h, MFX_ERR_WINDOW_COUNT, kMsgWindow);
    } else {
        wcnd = estimated_window_count_f32(h->samples, h->W, h->S);
        wc = wcnd - 2 * D;
        if (wc <= 0) return MFX_OK;
    }
    if (wcnd > h->cap_rows) return fail(h, MFX_ERR_WINDOW_HIGH, kMsgHigh);

    const bool sweep = n_alpha > 0;
    const int n_tab = sweep ? n_alpha : 1;
    int rc = sweep ? prepare_sweep(h, alphas, n_alpha) : refresh_mel(h);
    if (rc != MFX_OK) return rc;
    float *d_src = sweep ? h->d_sweep_src.p : h->d_src.p;
    float *d_blk = sweep ? h->d_sweep_blk.p : h->d_blk.p;
    float *d_stats = sweep ? h->d_sweep_stats.p : h->d_stats_stream.p;

    // filterbank + log + DCT over all frames with context
    MelcepParams mp;
    fill_melcep(h, mp, sweep);
    mp.spec = h->d_spec.p;
    mp.n_rows = wcnd;
    mp.feat = d_src;
    mp.feat_pitch = h->cols;
    mp.n_tables = n_tab;
    mp.feat_table_stride = (int64_t)h->cap_rows * h->cols;
    HIP_TRY(h, launch_melcep(mp, h->stream));

    // static row offset as the reference reads it (mfcccpu.cpp:274,439): was_flushed() ? 0 : D.
    // With bug_compat off a flush block always reads at D (fixes B1).
    bool at_zero = h->last_calc_flushed;
    if (!h->cfg.bug_compat && h->last_block) at_zero = false;
    const int static_off = at_zero ? 0 : D;

    Segment sg; // rows with context (statics), used by the normalisation before the deltas
    std::memset(&sg, 0, sizeof(sg));
    sg.n_out = wcnd;
    Segment sd; // the block's delivered rows
    std::memset(&sd, 0, sizeof(sd));
    sd.n_out = wc;
    sd.static_off = static_off;
    if (first) { // D replicated rows in front (mfcccpu.cpp:243-248)
        sd.shift = -D;
        sd.lo = 0;
        sd.hi = wcnd - 1;
    } else if (last) { // D replicated rows behind (mfcccpu.cpp:249-254)
        sd.shift = 0;
        sd.lo = 0;
        sd.hi = wc + D - 1;
    } else {
        sd.shift = 0;
        sd.lo = 0;
        sd.hi = wcnd - 1;
    }
    const Segment *segs_ctx = nullptr, *segs_out = nullptr;
    if (sweep) { // one segment per alpha: block a of the sweep buffers
        std::vector<Segment> hs((size_t)2 * n_alpha);
        for (int a = 0; a < n_alpha; ++a) {
            hs[a] = sg;
            hs[a].src_row0 = hs[a].out_row0 = (int64_t)a * h->cap_rows;
            hs[n_alpha + a] = sd;
            hs[n_alpha + a].src_row0 = hs[n_alpha + a].out_row0 = (int64_t)a * h->cap_rows;
        }
        HIP_TRY(h, hipMemcpyAsync(h->d_sweep_segs.p, hs.data(), hs.size() * sizeof(Segment), hipMemcpyHostToDevice,
                                  h->stream));
        HIP_TRY(h, hipStreamSynchronize(h->stream)); // hs is a local
        segs_ctx = h->d_sweep_segs.p;
        segs_out = h->d_sweep_segs.p + n_alpha;
    }

    const bool norm = h->cfg.norm != MFX_NORM_NONE;
    if (norm && !h->cfg.norm_after_dyn) { // normalise statics (with context) before the deltas
        rc = run_norm(h, d_src, h->cols, 0, segs_ctx, n_tab, sweep ? nullptr : &sg, 0, d_stats, use_last, wcnd);
        if (rc != MFX_OK) return rc;
    }

    // Small blocks (round 4): the delta kernel -- the last one that touches the rows unless they are normalised after the
    // deltas -- writes them straight into the page-locked staging buffer (posted writes over the link, consecutive
    // threads on consecutive addresses), and get_output_data has nothing to launch: one kernel and one launch less per
    // call sequence (profiles/r04/stream_small_timeline.txt).  Same kernel, same values: the same bits as through d_blk.
    h->rows_in_stage = false;
    float *rows_out = d_blk;
    // (not when the rows are normalised after the deltas: the one-launch normaliser is ONE block per segment, and a
    // single CU writing 155 KB over the link takes what the copy kernel it would save takes -- measured, +- 0.5 us)
    if (!sweep && !(norm && h->cfg.norm_after_dyn) && small_block(h, (size_t)wc * h->width * sizeof(float))) {
        const size_t want = (size_t)h->cap_rows * h->width + 4;
        if (h->h_out_stage_n < want) {
            HIP_TRY(h, hipStreamSynchronize(h->stream));
            if (h->h_out_stage) (void)hipHostFree(h->h_out_stage);
            h->h_out_stage = nullptr;
            h->h_out_stage_n = 0;
            HIP_TRY(h, hipHostMalloc((void **)&h->h_out_stage, want * sizeof(float), hipHostMallocDefault));
            h->h_out_stage_n = want;
        }
        void *dev = nullptr;
        if (is_pinned_host(h->h_out_stage, &dev) && dev) {
            rows_out = (float *)dev;
            h->rows_in_stage = true;
        }
    }
    DeltaParams dp;
    std::memset(&dp, 0, sizeof(dp));
    dp.src = d_src;
    dp.src_pitch = h->cols;
    dp.out = rows_out;
    dp.out_pitch = h->width;
    dp.segs = segs_out;
    dp.n_segs = n_tab;
    dp.cols = h->cols;
    dp.l1 = h->l1;
    dp.l2 = h->l2;
    dp.tiles_per_seg_max = (wc + 63) / 64;
    dp.inline_seg = sweep ? 0 : 1;
    dp.seg0 = sd;
    HIP_TRY(h, launch_delta(dp, h->stream));

    if (norm && h->cfg.norm_after_dyn) {
        const int groups = h->width / h->cols;
        rc = run_norm(h, d_blk, h->width, 0, segs_out, n_tab, sweep ? nullptr : &sd, 0, d_stats, use_last, wc, groups,
                      (size_t)n_tab * 2 * h->cols);
        if (rc != MFX_OK) return rc;
    }
    h->block_applied = true;
    h->sweep_n = sweep ? n_alpha : 0;
    return MFX_OK;
}

} // namespace

extern "C" int mfx_apply(mfx_handle *h)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    return apply_impl(h, nullptr, 0);
}

extern "C" int mfx_apply_alphas(mfx_handle *h, const float *alphas, int32_t n_alpha)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (!alphas || n_alpha < 1 || n_alpha > 4096) return fail(h, MFX_ERR_ARG, "invalid argument");
    for (int a = 0; a < n_alpha; ++a)
        if (!(alphas[a] > 0.f)) return fail(h, MFX_ERR_ARG, "alpha must be positive");
    return apply_impl(h, alphas, n_alpha);
}

extern "C" int mfx_get_output_data_alpha(mfx_handle *h, int32_t alpha_index, float *data_out, int32_t frames)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if ((!data_out && frames > 0) || frames < 0) return fail(h, MFX_ERR_ARG, "invalid argument");
    if (alpha_index < 0 || alpha_index >= h->sweep_n) return fail(h, MFX_ERR_ARG, "alpha index outside the last sweep");
    if (frames > h->cap_rows) return fail(h, MFX_ERR_WINDOW_HIGH, kMsgHigh);
    if (frames == 0) return MFX_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    return download_rows(h, data_out, h->d_sweep_blk.p + (size_t)alpha_index * h->cap_rows * h->width, (size_t)frames * h->width);
}

extern "C" int mfx_get_output_data(mfx_handle *h, float *data_out, int32_t frames)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if ((!data_out && frames > 0) || frames < 0) return fail(h, MFX_ERR_ARG, "invalid argument");
    if (frames > h->cap_rows) return fail(h, MFX_ERR_WINDOW_HIGH, kMsgHigh);
    if (frames == 0) return MFX_OK;
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->rows_in_stage && h->block_applied) { // the delta kernel wrote the rows into page-locked memory: wait for it, copy
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        std::memcpy(data_out, h->h_out_stage, (size_t)frames * h->width * sizeof(float));
        return MFX_OK;
    }
    return download_rows(h, data_out, h->d_blk.p, (size_t)frames * h->width);
}

// ------------------------------------------------------------------------------------------------
// batch interface
// ------------------------------------------------------------------------------------------------

extern "C" int64_t mfx_batch_frames(const mfx_handle *h, int64_t samples)
{
    if (!h) return MFX_ERR_ARG;
    int64_t t = frame_count(samples, h->W, h->S);
    return t > 0 ? t : 0;
}

namespace {

// Plan of the fused front end + delta stage (k_front512<..., FUSE>).  The global chunk list is cut into
// B contiguous pieces, one per block; a piece that starts or ends inside an utterance gets a halo chunk of
// D frames on that side (both neighbours compute those statics; identical values land on the same
// scratch rows).  The block's own rows are grouped into tiles of <= 64 rows of one utterance; each tile
// names the block-local chunks whose statics it reads.
int plan_fused_delta(mfx_handle *h, const std::vector<int64_t> &T_of)
{
    h->fuse_plan = false;
    const size_t n = h->h_chunks.size();
    if (!h->fuse_delta_enabled || !h->fast512 || h->stuff256 || h->channels != 1 || h->l1 <= 0 || h->cols > 16 || h->ceps <= 0 || h->D > 16 || n == 0 ||
        n > 0x3fffffff || (h->cfg.norm != MFX_NORM_NONE && !h->cfg.norm_after_dyn))
        return MFX_OK;
    const int D = h->D;
    const int B = (int)std::min<size_t>((size_t)h->num_cus, (n + 14) / 15);
    const std::vector<int32_t> &utt_of = h->chunk_utt; // utterance of every chunk
    std::vector<Chunk> fch;
    fch.reserve(n + 2 * (size_t)B);
    std::vector<DeltaTile> tiles;
    std::vector<int32_t> coff((size_t)B + 1), toff((size_t)B + 1);
    size_t max_list = 0;
    // pieces of equal FRAME count (the tail of the chunk list holds 4-frame chunks): cut[b] = first chunk of block b
    std::vector<size_t> cut((size_t)B + 1, n);
    {
        int64_t total = 0;
        for (const Chunk &c : h->h_chunks) total += c.n_frames;
        int64_t acc = 0;
        size_t c = 0;
        for (int b = 0; b < B; ++b) {
            cut[b] = c;
            const int64_t target = total * (b + 1) / B;
            while (c < n && acc + h->h_chunks[c].n_frames <= target) acc += h->h_chunks[c++].n_frames;
            if (b + 1 == B) c = n;
        }
        cut[0] = 0;
    }
    for (int b = 0; b < B; ++b) {
        const size_t c0 = cut[b], c1 = cut[b + 1];
        coff[b] = (int32_t)fch.size();
        toff[b] = (int32_t)tiles.size();
        if (c1 <= c0) continue;
        const size_t base = fch.size();
        {   // halo in front
            const Chunk &f = h->h_chunks[c0];
            const int64_t avail = f.out_row - h->utt_row[utt_of[c0]];
            if (avail > 0) {
                const int hal = (int)std::min<int64_t>(D, avail);
                Chunk c;
                c.pcm_off = f.pcm_off - (int64_t)hal * h->S;
                c.out_row = f.out_row - hal;
                c.n_frames = hal;
                c.pad = 0;
                fch.push_back(c);
            }
        }
        const size_t own0 = fch.size() - base; // local index of the first own chunk
        for (size_t c = c0; c < c1; ++c) fch.push_back(h->h_chunks[c]);
        {   // halo behind
            const Chunk &l = h->h_chunks[c1 - 1];
            const int u = utt_of[c1 - 1];
            const int64_t end_row = l.out_row + l.n_frames;
            const int64_t avail = h->utt_row[u] + T_of[u] - end_row;
            if (avail > 0) {
                const int hal = (int)std::min<int64_t>(D, avail);
                Chunk c;
                c.pcm_off = l.pcm_off + (int64_t)l.n_frames * h->S;
                c.out_row = end_row;
                c.n_frames = hal;
                c.pad = 0;
                fch.push_back(c);
            }
        }
        const size_t cnt = fch.size() - base;
        max_list = std::max(max_list, cnt);
        // tiles over the own chunks: runs of one utterance, <= 64 rows each
        auto local_of_row = [&](int64_t r, size_t hint) -> int32_t { // block-local chunk that holds row r
            size_t k = hint;
            while (k > 0 && fch[base + k].out_row > r) --k;
            while (k + 1 < cnt && fch[base + k].out_row + fch[base + k].n_frames <= r) ++k;
            return (int32_t)k;
        };
        size_t k = own0;
        const size_t own1 = own0 + (c1 - c0);
        while (k < own1) {
            const int u = utt_of[c0 + (k - own0)];
            const int64_t r0 = fch[base + k].out_row;
            int64_t rows = 0;
            size_t k2 = k;
            while (k2 < own1 && utt_of[c0 + (k2 - own0)] == u && rows + fch[base + k2].n_frames <= 64) {
                rows += fch[base + k2].n_frames;
                ++k2;
            }
            const int64_t u0 = h->utt_row[u], u1 = u0 + T_of[u];
            DeltaTile t;
            std::memset(&t, 0, sizeof(t));
            t.out_row0 = r0;
            t.seg_row0 = u0;
            t.n_rows = (int32_t)rows;
            t.r0 = (int32_t)(r0 - u0);
            t.shift = -D;           // whole utterance: D replicated rows on both sides (as the batch Segment)
            t.lo = 0;
            t.hi = (int32_t)(T_of[u] - 1);
            t.static_off = 0;
            t.dep_lo = local_of_row(std::max(r0 - D, u0), k);
            t.dep_hi = local_of_row(std::min(r0 + rows + D, u1) - 1, k2 - 1);
            tiles.push_back(t);
            k = k2;
        }
    }
    coff[B] = (int32_t)fch.size();
    toff[B] = (int32_t)tiles.size();
    {   // one padding entry: the delta wave prefetches the descriptor after its last tile
        DeltaTile t;
        std::memset(&t, 0, sizeof(t));
        tiles.push_back(t);
    }
    const int done_words = (int)((max_list + 31) / 32) + 1;
    FrontParams probe;
    fill_front(h, probe);
    probe.dl1 = h->l1;
    probe.dl2 = h->l2;
    probe.done_words = done_words;
    if (!h->fused_ok || probe.dct_mode != 1 || front512_delta_lds_bytes(probe) > 160 * 1024) return MFX_OK;
    HIP_TRY(h, upload(h->d_fchunks, fch));
    HIP_TRY(h, upload(h->d_blk_chunk_off, coff));
    HIP_TRY(h, upload(h->d_blk_tile_off, toff));
    HIP_TRY(h, upload(h->d_tiles, tiles));
    if (!h->d_err.p) {
        HIP_TRY(h, h->d_err.alloc(1));
        HIP_TRY(h, hipMemset(h->d_err.p, 0, sizeof(int32_t)));
    }
    h->f_blocks = B;
    h->f_done_words = done_words;
    h->f_nchunks = (int32_t)fch.size();
    h->fuse_plan = true;
    return MFX_OK;
}

} // namespace

extern "C" int mfx_batch_plan(mfx_handle *h, int32_t n_utt, const int64_t *offsets, const int64_t *lengths,
                              int64_t *out_rows, int64_t *total_rows)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (n_utt < 0 || (n_utt > 0 && (!offsets || !lengths))) return fail(h, MFX_ERR_ARG, "invalid argument");
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    h->n_utt = n_utt;
    h->utt_off.assign(offsets, offsets + n_utt);
    h->utt_len.assign(lengths, lengths + n_utt);
    h->utt_row.resize(n_utt);
    h->h_chunks.clear();
    h->chunk_utt.clear();
    std::vector<Segment> segs((size_t)n_utt);
    std::vector<int64_t> T_of((size_t)n_utt);
    int64_t row = 0;
    int tiles_max = 0;
    bool aligned = (h->S % 2) == 0;
    for (int u = 0; u < n_utt; ++u) {
        if (offsets[u] < 0 || lengths[u] < 0) return fail(h, MFX_ERR_ARG, "negative utterance offset/length");
        int64_t T = frame_count(lengths[u], h->W, h->S);
        if (T < 0) T = 0;
        if (T > 0x7fffffff) return fail(h, MFX_ERR_ARG, "utterance too long");
        h->utt_row[u] = row;
        T_of[u] = T;
        if (out_rows) out_rows[u] = row;
        if (offsets[u] & 1) aligned = false;
        for (int64_t t0 = 0; t0 < T; t0 += kChunkFrames) {
            Chunk c;
            c.pcm_off = offsets[u] + t0 * h->S;
            c.out_row = row + t0;
            c.n_frames = (int32_t)std::min<int64_t>(kChunkFrames, T - t0);
            c.pad = 0;
            h->h_chunks.push_back(c);
            h->chunk_utt.push_back(u);
        }
        Segment &s = segs[u];
        std::memset(&s, 0, sizeof(s));
        s.src_row0 = row;
        s.out_row0 = row;
        s.n_out = (int32_t)T;
        s.shift = -h->D; // whole utterance: D replicated rows on both sides
        s.lo = 0;
        s.hi = (int32_t)std::max<int64_t>(T - 1, 0);
        s.static_off = 0;
        // Statistics of the normaliser (norm after dyn): the reference, fed the utterance as ONE block (its default
        // sample_limit holds ~10 minutes of audio), computes them over the T - D rows that block delivers and
        // re-uses them for the D rows of the flush (mfcccpu.cpp:377-388,395-407; normalizercpu.cpp:22-27).  That is
        // the default here too (batch_norm_stats = 0); 1 = over all T rows.  Normalisation before the deltas covers
        // the block's T rows with context in the reference as well, i.e. all rows either way.
        s.pad = (h->cfg.norm != MFX_NORM_NONE && h->cfg.norm_after_dyn && h->cfg.batch_norm_stats == 0 && T > h->D)
                    ? (int32_t)(T - h->D) : 0;
        tiles_max = std::max<int>(tiles_max, (int)((T + 63) / 64));
        row += T;
    }
    // The 512-point kernel deals chunks to the 16 waves of each block as they become free; with 16-frame
    // chunks a wave can sit idle for most of a chunk time (~34 us on C2) at the end of the launch.  The last two
    // chunks of every wave of the grid are therefore cut into 4-frame pieces (one kernel iteration each).
    // (k_front2048: 12 waves per CU, each 16-frame chunk is 8 iterations of ~10 us -- on C5 a wave sees only ~4 chunks in
    // all, so the last ONE per wave is cut, and a launch twice that long already qualifies)
    const int ts = h->cfg.tail_split;
    const bool f2048 = h->fast2048 && h->wplan32_ok; // (stereo, mono on aligned pairs, mono at any alignment: all three builds)
    if ((h->fast512 || (h->fast1024 && h->fused_ok) || f2048) && ts >= 0) {
        const size_t n = h->h_chunks.size();
        const size_t tail = std::min<size_t>(n, (size_t)(ts > 0 ? std::min(ts, 64) : f2048 ? 1 : 2) * (f2048 ? 12 : 16) * h->num_cus);
        if (n >= (f2048 ? 2 : 4) * tail) { // only when the launch is long enough for the tail to matter
            std::vector<Chunk> cut;
            std::vector<int32_t> cut_utt;
            for (size_t c = n - tail; c < n; ++c) {
                const Chunk &src = h->h_chunks[c];
                for (int f = 0; f < src.n_frames; f += 4) {
                    Chunk q = src;
                    q.pcm_off = src.pcm_off + (int64_t)f * h->S;
                    q.out_row = src.out_row + f;
                    q.n_frames = std::min(4, src.n_frames - f);
                    cut.push_back(q);
                    cut_utt.push_back(h->chunk_utt[c]);
                }
            }
            h->h_chunks.resize(n - tail);
            h->chunk_utt.resize(n - tail);
            h->h_chunks.insert(h->h_chunks.end(), cut.begin(), cut.end());
            h->chunk_utt.insert(h->chunk_utt.end(), cut_utt.begin(), cut_utt.end());
        }
    }
    h->utt_chunk0.assign((size_t)n_utt + 1, (int32_t)h->h_chunks.size());
    for (size_t c = h->h_chunks.size(); c-- > 0;) h->utt_chunk0[h->chunk_utt[c]] = (int32_t)c;
    for (int u = n_utt - 1; u >= 0; --u) // utterances without frames: empty chunk range
        if (h->utt_chunk0[u] > h->utt_chunk0[u + 1]) h->utt_chunk0[u] = h->utt_chunk0[u + 1];
    h->total_rows = row;
    h->tiles_max = tiles_max;
    h->batch_aligned = aligned;
    if (total_rows) *total_rows = row;
    HIP_TRY(h, upload(h->d_chunks, h->h_chunks));
    HIP_TRY(h, upload(h->d_segs, segs));
    if (h->cfg.norm != MFX_NORM_NONE) {
        HIP_TRY(h, h->d_stats_batch.alloc((size_t)n_utt * 3 * 2 * h->cols));
        const size_t need = norm_partial_doubles(n_utt, tiles_max * 64, h->cols);
        if (need > h->d_norm_partial.n) HIP_TRY(h, h->d_norm_partial.alloc(need));
    }
    {
        int rcf = plan_fused_delta(h, T_of);
        if (rcf != MFX_OK) return rcf;
    }
    // scratch for the compact statics (allocated here so that mfx_batch_run_device itself never allocates)
    for (int b = 0; b < (h->overlap ? 2 : 1); ++b)
        if (h->l1 > 0 && h->cols <= 16 && h->d_static16[b].n < (size_t)row * 16)
            HIP_TRY(h, h->d_static16[b].alloc((size_t)row * 16));
    return MFX_OK;
}

namespace {
// utterances [u0, u1) of the planned batch (all of them: the fused-delta and overlap modes apply)
int batch_run_range(mfx_handle *h, const int16_t *d_pcm, int64_t pcm_samples_total, float *d_out, int u0, int u1);
} // namespace

extern "C" int mfx_batch_run_device(mfx_handle *h, const int16_t *d_pcm, int64_t pcm_samples_total, float *d_out)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    return batch_run_range(h, d_pcm, pcm_samples_total, d_out, 0, h->n_utt);
}

namespace {
int batch_run_range(mfx_handle *h, const int16_t *d_pcm, int64_t pcm_samples_total, float *d_out, int u0, int u1)
{
    if (!d_pcm || !d_out || pcm_samples_total <= 0) return fail(h, MFX_ERR_ARG, "invalid argument");
    const bool whole = u0 == 0 && u1 == h->n_utt;
    const int32_t rc0 = h->utt_chunk0[u0], rc1 = h->utt_chunk0[u1]; // chunk range of the utterance range
    if (!h->have_window) return fail(h, MFX_ERR_STATE, "set_window has not been called");
    if (h->total_rows == 0) return MFX_OK;
    if (((uintptr_t)d_pcm & 3) != 0) return fail(h, MFX_ERR_ARG, "d_pcm must be 4-byte aligned");
    for (int u = u0; u < u1; ++u)
        if (h->utt_off[u] + h->utt_len[u] > pcm_samples_total)
            return fail(h, MFX_ERR_ARG, "utterance extends past the end of the PCM array");
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = refresh_mel(h);
    if (rc != MFX_OK) return rc;
    if (rc1 <= rc0) return MFX_OK;

    FrontParams p;
    fill_front(h, p);
    p.pcm = d_pcm;
    p.pcm_total = pcm_samples_total * h->channels;
    p.chunks = h->d_chunks.p + rc0;
    p.n_chunks = rc1 - rc0;
    p.row_limit = h->total_rows;
    p.feat = d_out;
    p.feat_pitch = h->width;

    // Which front end: the 512-point register kernel, else the fused wave-per-frame kernel when its
    // LDS fits, else spectrum through an HBM slab + melcep.
    const FrontKind kind = choose_front(h);
    const bool fused512 = kind == kFront512, fused1024 = kind == kFront1024, fused2048 = kind == kFront2048,
               fusedgen = kind == kFrontGenFused;
    // With deltas on, the front end writes its statics as compact 64-byte rows into a scratch buffer
    // and the delta kernel emits whole [static | d | dd] rows: every HBM write is then a full line
    // (13-float row pieces at a 156-byte pitch cost 1.5x their size in 32-byte sectors).
    const bool norm_before = h->cfg.norm != MFX_NORM_NONE && !h->cfg.norm_after_dyn;
    // Overlap (opt-in, mfx_batch_overlap): the delta/normalisation tail runs on a second stream behind an
    // event, so the memory-bound tail of batch i shares the GPU with the compute-bound front end of
    // batch i+1; the statics scratch is double buffered and the front end of batch i+2 waits for tail i.
    const int sb = (h->overlap && whole) ? (int)(h->batch_seq & 1) : 0;
    const bool via_scratch = ((fused512 && p.dct_mode == 1) || fused1024 || fused2048 || fusedgen) && h->l1 > 0 && h->cols <= 16 && !norm_before &&
                             h->d_static16[sb].n >= (size_t)h->total_rows * 16;
    // Fused delta stage: the 512-point kernel's last wave per block turns the statics into whole output
    // rows while the other 15 produce them; no separate delta launch.
    bool fuse = whole && h->fuse_plan && fused512 && via_scratch && ((uintptr_t)d_out & 15) == 0;
    if (fuse) {
        p.dl1 = h->l1;
        p.dl2 = h->l2;
        p.done_words = h->f_done_words;
        fuse = p.dct_mode == 1 && front512_delta_lds_bytes(p) <= 160 * 1024;
    }
    const bool split_tail = whole && h->overlap && via_scratch && !fuse;
    hipStream_t tail_stream = split_tail ? h->stream2 : h->stream;
    if (via_scratch) {
        p.feat = h->d_static16[sb].p;
        p.feat_pitch = 16;
    }
    if (split_tail && h->tail_pending[sb]) // tail of batch i-2 still reads this scratch buffer
        HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_tail[sb], 0));
    if (fuse) {
        p.chunks = h->d_fchunks.p;
        p.n_chunks = h->f_nchunks;
        p.blk_chunk_off = h->d_blk_chunk_off.p;
        p.blk_tile_off = h->d_blk_tile_off.p;
        p.tiles = h->d_tiles.p;
        p.out = d_out;
        p.out_pitch = h->width;
        p.n_blocks = h->f_blocks;
        p.err_flag = h->d_err.p;
        p.spec = h->d_spec.p; // unused by this kernel; a -DMFX_DSTAMPS dev build drops the delta wave's tick counts here
        ProfScope ps(h);
        HIP_TRY(h, launch_front512_delta(p, h->batch_aligned, h->nm16, h->stream));
    } else if (fused512) {
        p.spec = h->d_spec.p; // unused by the fused kernel; a -DMFX_STAMPS dev build drops its cycle sums here
        ProfScope ps(h);
        HIP_TRY(h, launch_front512(p, /*to_spectrum=*/false, h->batch_aligned, h->nm16, h->stream));
    } else if (fused1024) {
        ProfScope ps(h);
        HIP_TRY(h, launch_front1024(p, h->batch_aligned, h->nm16, h->stream, (h->cfg.engine & MFX_ENGINE_FRONT1024_12_WAVES) ? 12 : 16));
    } else if (fused2048) {
        p.spec = h->d_spec.p; // unused by the fused kernel; a -DMFX_STAMPS dev build drops its cycle sums here
        ProfScope ps(h);
        HIP_TRY(h, launch_front2048(p, h->num_cus, h->stream));
    } else if (fusedgen) {
        ProfScope ps(h);
        HIP_TRY(h, launch_front_generic(p, /*fused=*/true, h->stream));
    } else {
        // magnitudes go through an HBM slab, then melcep
        const int64_t slab_rows_max = 1 << 17;
        const int64_t slab_rows = std::min<int64_t>(h->total_rows, slab_rows_max);
        if (h->d_spec_slab.n < (size_t)slab_rows * h->spec_pitch)
            HIP_TRY(h, h->d_spec_slab.alloc((size_t)slab_rows * h->spec_pitch));
        size_t c0 = (size_t)rc0;
        const size_t nchunks = (size_t)rc1;
        while (c0 < nchunks) {
            const int64_t row0 = h->h_chunks[c0].out_row;
            size_t c1 = c0;
            int64_t rows = 0;
            while (c1 < nchunks && rows + h->h_chunks[c1].n_frames <= slab_rows) {
                rows += h->h_chunks[c1].n_frames;
                ++c1;
            }
            FrontParams q = p;
            q.chunks = h->d_chunks.p + c0;
            q.n_chunks = (int32_t)(c1 - c0);
            q.spec = h->d_spec_slab.p - row0 * (int64_t)h->spec_pitch; // rows are addressed absolutely
            q.spec_pitch = h->spec_pitch;
            {
                ProfScope ps(h);
                if (h->fast512)
                    HIP_TRY(h, launch_front512(q, /*to_spectrum=*/true, h->batch_aligned, h->nm16, h->stream));
                else
                    HIP_TRY(h, launch_front_generic(q, /*fused=*/false, h->stream));
            }
            MelcepParams mp;
            fill_melcep(h, mp, false);
            mp.spec = h->d_spec_slab.p;
            mp.n_rows = rows;
            mp.feat = p.feat + row0 * (int64_t)p.feat_pitch;
            mp.feat_pitch = p.feat_pitch;
            HIP_TRY(h, launch_melcep(mp, h->stream));
            c0 = c1;
        }
    }

    if (split_tail) {
        HIP_TRY(h, hipEventRecord(h->ev_front[sb], h->stream));
        HIP_TRY(h, hipStreamWaitEvent(h->stream2, h->ev_front[sb], 0));
    }
    struct StreamSwap { // the tail kernels below launch on h->stream: point it at the tail stream meanwhile
        mfx_handle *h;
        hipStream_t keep;
        StreamSwap(mfx_handle *hh, hipStream_t s) : h(hh), keep(hh->stream) { h->stream = s; }
        ~StreamSwap() { h->stream = keep; }
    } swap_guard(h, tail_stream);
    const bool norm = h->cfg.norm != MFX_NORM_NONE;
    if (norm && !h->cfg.norm_after_dyn) {
        rc = run_norm(h, d_out, h->width, 0, h->d_segs.p + u0, u1 - u0, nullptr, 0, h->d_stats_batch.p + (size_t)u0 * 2 * h->cols,
                      false, h->tiles_max * 64);
        if (rc != MFX_OK) return rc;
    }
    if (h->l1 > 0 && !fuse) {
        DeltaParams dp;
        std::memset(&dp, 0, sizeof(dp));
        dp.src = via_scratch ? h->d_static16[sb].p : d_out;
        dp.src_pitch = via_scratch ? 16 : h->width;
        dp.out = d_out;
        dp.out_pitch = h->width;
        dp.segs = h->d_segs.p + u0;
        dp.n_segs = u1 - u0;
        dp.cols = h->cols;
        dp.l1 = h->l1;
        dp.l2 = h->l2;
        dp.tiles_per_seg_max = h->tiles_max;
        HIP_TRY(h, launch_delta(dp, h->stream));
    }
    if (norm && h->cfg.norm_after_dyn) {
        const int groups = h->width / h->cols;
        rc = run_norm(h, d_out, h->width, 0, h->d_segs.p + u0, u1 - u0, nullptr, 0, h->d_stats_batch.p + (size_t)u0 * 2 * h->cols,
                      false, h->tiles_max * 64, groups, (size_t)h->n_utt * 2 * h->cols);
        if (rc != MFX_OK) return rc;
    }
    if (split_tail) {
        HIP_TRY(h, hipEventRecord(h->ev_tail[sb], tail_stream));
        h->tail_pending[sb] = true;
    }
    if (whole) ++h->batch_seq;
    return MFX_OK;
}
} // namespace

extern "C" void *mfx_alloc_pinned(size_t bytes)
{
    void *p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}

extern "C" void mfx_free_pinned(void *p)
{
    if (p) (void)hipHostFree(p);
}

extern "C" int mfx_batch_overlap(mfx_handle *h, int enable)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    HIP_TRY(h, hipSetDevice(h->device));
    int rc = mfx_synchronize(h);
    if (rc != MFX_OK) return rc;
    if (enable && !h->stream2) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_front[i], hipEventDisableTiming));
            HIP_TRY(h, hipEventCreateWithFlags(&h->ev_tail[i], hipEventDisableTiming));
        }
    }
    h->overlap = enable != 0;
    h->tail_pending[0] = h->tail_pending[1] = false;
    if (h->overlap && h->total_rows > 0 && h->l1 > 0 && h->cols <= 16)
        for (int b = 0; b < 2; ++b)
            if (h->d_static16[b].n < (size_t)h->total_rows * 16) HIP_TRY(h, h->d_static16[b].alloc((size_t)h->total_rows * 16));
    return MFX_OK;
}

extern "C" int mfx_batch_run_host(mfx_handle *h, const int16_t *pcm, int64_t pcm_samples_total, float *out)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (!pcm || !out || pcm_samples_total <= 0) return fail(h, MFX_ERR_ARG, "invalid argument");
    HIP_TRY(h, hipSetDevice(h->device));
    // device-side staging of the host buffers, kept by the handle and grown on demand
    const size_t n_in = (size_t)pcm_samples_total * h->channels;
    const size_t n_out = (size_t)std::max<int64_t>(h->total_rows, 1) * h->width;
    if (h->d_host_pcm.n < n_in + 8) HIP_TRY(h, h->d_host_pcm.alloc(n_in + 8));
    if (h->d_host_out.n < n_out) HIP_TRY(h, h->d_host_out.alloc(n_out));

    // Pinned caller buffers and a batch worth slicing: the utterances go through in up to 8 slices, the upload of slice
    // k + 1 and the download of slice k - 1 running beside the kernels of slice k on their own streams (PCIe is full
    // duplex: the 320 MB in and the 156 MB out of a C2 batch overlap instead of queueing up).  Utterance offsets must
    // ascend for a slice to be one contiguous piece of the PCM array; anything else takes the plain path below.
    bool ascending = true;
    for (int u = 1; u < h->n_utt && ascending; ++u) ascending = h->utt_off[u] >= h->utt_off[u - 1] + h->utt_len[u - 1];
    const int K = (int)std::min<int64_t>(8, h->n_utt / 4);
    if (K >= 2 && ascending && !h->overlap && !h->fuse_plan && n_in * sizeof(int16_t) >= ((size_t)32 << 20) &&
        is_pinned_host(pcm) && is_pinned_host(out)) {
        if (!h->stream_up) {
            HIP_TRY(h, hipStreamCreateWithFlags(&h->stream_up, hipStreamNonBlocking));
            HIP_TRY(h, hipStreamCreateWithFlags(&h->stream_dn, hipStreamNonBlocking));
            for (int i = 0; i < 16; ++i) {
                HIP_TRY(h, hipEventCreateWithFlags(&h->ev_up[i], hipEventDisableTiming));
                HIP_TRY(h, hipEventCreateWithFlags(&h->ev_run[i], hipEventDisableTiming));
            }
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        // every utterance inside the caller's array BEFORE the first copy is queued (batch_run_range only looks at the
        // slice it is given, and only after that slice's upload is in flight)
        for (int u = 0; u < h->n_utt; ++u)
            if (h->utt_off[u] < 0 || h->utt_off[u] + h->utt_len[u] > pcm_samples_total)
                return fail(h, MFX_ERR_ARG, "utterance outside the PCM array");
        const int ch = h->channels;
        // one slice; an error leaves copies in flight on three streams, which slices_done drains before returning
        auto run_slice = [&](int k) -> int {
            const int u0 = (int)((int64_t)h->n_utt * k / K), u1 = (int)((int64_t)h->n_utt * (k + 1) / K);
            // samples [s0, s1) of the array hold the slice (s0 rounded down to an even sample: 4-byte aligned pieces)
            const int64_t s0 = (k == 0 ? 0 : h->utt_off[u0]) & ~(int64_t)1;
            const int64_t s1 = std::min<int64_t>(k + 1 == K ? pcm_samples_total : h->utt_off[u1], pcm_samples_total);
            if (s1 > s0)
                HIP_TRY(h, hipMemcpyAsync(h->d_host_pcm.p + s0 * ch, pcm + s0 * ch, (size_t)(s1 - s0) * ch * sizeof(int16_t),
                                          hipMemcpyHostToDevice, h->stream_up));
            HIP_TRY(h, hipEventRecord(h->ev_up[k], h->stream_up));
            HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_up[k], 0));
            int rc = batch_run_range(h, h->d_host_pcm.p, pcm_samples_total, h->d_host_out.p, u0, u1);
            if (rc != MFX_OK) return rc;
            HIP_TRY(h, hipEventRecord(h->ev_run[k], h->stream));
            HIP_TRY(h, hipStreamWaitEvent(h->stream_dn, h->ev_run[k], 0));
            const int64_t r0 = h->utt_row[u0], r1 = u1 < h->n_utt ? h->utt_row[u1] : h->total_rows;
            if (r1 > r0)
                HIP_TRY(h, hipMemcpyAsync(out + r0 * h->width, h->d_host_out.p + r0 * h->width,
                                          (size_t)(r1 - r0) * h->width * sizeof(float), hipMemcpyDeviceToHost, h->stream_dn));
            return MFX_OK;
        };
        for (int k = 0; k < K; ++k) {
            const int rc = run_slice(k);
            if (rc != MFX_OK) { // nothing may still read `pcm` or write `out` once we have returned
                (void)hipStreamSynchronize(h->stream_up);
                (void)hipStreamSynchronize(h->stream);
                (void)hipStreamSynchronize(h->stream_dn);
                return rc;
            }
        }
        HIP_TRY(h, hipStreamSynchronize(h->stream_dn));
        return mfx_synchronize(h);
    }

    HIP_TRY(h, hipMemcpyAsync(h->d_host_pcm.p, pcm, n_in * sizeof(int16_t), hipMemcpyHostToDevice, h->stream));
    int rc = mfx_batch_run_device(h, h->d_host_pcm.p, pcm_samples_total, h->d_host_out.p);
    if (rc != MFX_OK) {
        (void)hipStreamSynchronize(h->stream);
        return rc;
    }
    if (h->stream2) HIP_TRY(h, hipStreamSynchronize(h->stream2)); // overlapped tail, if any
    if (h->total_rows > 0)
        HIP_TRY(h, hipMemcpyAsync(out, h->d_host_out.p, (size_t)h->total_rows * h->width * sizeof(float),
                                  hipMemcpyDeviceToHost, h->stream));
    return mfx_synchronize(h);
}

// ------------------------------------------------------------------------------------------------
// test taps
// ------------------------------------------------------------------------------------------------

extern "C" int64_t mfx_debug_read(mfx_handle *h, int kind, void *dst, int64_t dst_bytes)
{
    if (!h) return MFX_ERR_ARG;
    if (h->planning) return fail(h, MFX_ERR_DEVICE, kMsgPlanning);
    if (!dst) return fail(h, MFX_ERR_ARG, "invalid argument");
    if (hipSetDevice(h->device) != hipSuccess) return MFX_ERR_DEVICE;
    const void *src = nullptr;
    int64_t count = 0, esz = 4;
    switch (kind) {
    case 0:
        if (refresh_mel(h) != MFX_OK) return MFX_ERR_DEVICE;
        src = h->d_mel_w.p;
        count = 2 * (int64_t)h->W2;
        break;
    case 1:
        if (refresh_mel(h) != MFX_OK) return MFX_ERR_DEVICE;
        src = h->d_mel_beg.p;
        count = h->nb + 2;
        break;
    case 2:
        src = h->d_dct.p;
        count = h->ceps > 0 ? (int64_t)h->nb * h->dl : 0;
        break;
    case 3:
        src = h->d_spec.p;
        count = (int64_t)h->block_wcnd * h->spec_pitch;
        break;
    case 4: // raw head of the spectrum buffer (dev builds park in-kernel stamps there)
        src = h->d_spec.p;
        count = std::min<int64_t>(dst_bytes / 4, (int64_t)h->d_spec.n);
        break;
    case 5: // normaliser statistics of the last streaming apply(): [groups][2][cols] = (mean, multiplier) per column
            // group (static, delta, delta-delta when normalising after the deltas; statics only before)
        src = h->d_stats_stream.p;
        count = h->cfg.norm == MFX_NORM_NONE ? 0 : (int64_t)(h->cfg.norm_after_dyn ? h->width / h->cols : 1) * 2 * h->cols;
        break;
    case 6: // normaliser statistics of the last batch run: [groups][n_utt][2][cols]
        src = h->d_stats_batch.p;
        count = h->cfg.norm == MFX_NORM_NONE ? 0
                                             : (int64_t)(h->cfg.norm_after_dyn ? h->width / h->cols : 1) * h->n_utt * 2 * h->cols;
        break;
    default:
        return MFX_ERR_ARG;
    }
    if (count * esz > dst_bytes) return MFX_ERR_ARG;
    if (count == 0) return 0;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return MFX_ERR_DEVICE;
    if (hipMemcpy(dst, src, (size_t)(count * esz), hipMemcpyDeviceToHost) != hipSuccess) return MFX_ERR_DEVICE;
    return count;
}

// ------------------------------------------------------------------------------------------------
// host-side table builders (no device)
// ------------------------------------------------------------------------------------------------

extern "C" int mfx_host_mel_table(int32_t num_banks, int32_t fft_size, float sample_rate, float low_freq,
                                  float high_freq, float alpha, float *weights, int32_t *beg)
{
    if (num_banks <= 0 || fft_size <= 0 || !weights || !beg) return MFX_ERR_ARG;
    MelTable t;
    build_mel_table(num_banks, fft_size, sample_rate, low_freq, high_freq, alpha, t);
    std::memcpy(weights, t.weights.data(), sizeof(float) * t.weights.size());
    std::memcpy(beg, t.beg.data(), sizeof(int32_t) * t.beg.size());
    return MFX_OK;
}

// Lane plan of the mel walk (lanes = 16: the 512-point kernel's MelLanePlan, max_read_bin 479; lanes = 64: the
// long-transform kernel's MelWavePlan).  Returns the number of rounds (<= 8), or an error.  Outputs (any may be NULL to
// query): L[8] bins per lane and round, *row_stride, start / fid [rounds][lanes], w [lanes][row_stride].
extern "C" int mfx_host_mel_lane_plan(int32_t lanes, int32_t num_banks, int32_t fft_size, const float *weights,
                                      const int32_t *beg, int32_t max_read_bin, int32_t *L, int32_t *row_stride,
                                      int32_t *start, int32_t *fid, float *w, int64_t w_cap)
{
    if ((lanes != 16 && lanes != 32 && lanes != 64) || num_banks <= 0 || fft_size <= 0 || !weights || !beg) return MFX_ERR_ARG;
    MelTable t;
    t.weights.assign(weights, weights + (size_t)2 * fft_size);
    t.beg.assign(beg, beg + num_banks + 2);
    for (int v : t.beg)
        if (v < 0 || v > fft_size / 2) return MFX_ERR_ARG;
    int rounds = 0, rs = 0;
    const int *Ls = nullptr;
    const std::vector<int32_t> *st = nullptr, *fd = nullptr;
    const std::vector<float> *ww = nullptr;
    MelLanePlan p16;
    MelWavePlan p64;
    if (lanes == 16) {
        // (a 1024-point table on 16 lanes is k_front1024's plan: starts at multiples of 4 bins)
        if (!build_mel_lane_plan(t, num_banks, fft_size, max_read_bin, p16, fft_size == 1024 ? 4 : 2)) return MFX_ERR_CONFIG;
        rounds = p16.rounds, rs = p16.row_stride, Ls = p16.L, st = &p16.start, fd = &p16.fid, ww = &p16.w;
    } else {
        if (!build_mel_wave_plan(t, num_banks, fft_size, max_read_bin, p64, lanes)) return MFX_ERR_CONFIG;
        rounds = p64.rounds, rs = p64.row_stride, Ls = p64.L, st = &p64.start, fd = &p64.fid, ww = &p64.w;
    }
    if (L) std::memcpy(L, Ls, sizeof(int32_t) * 8);
    if (row_stride) *row_stride = rs;
    if (start) std::memcpy(start, st->data(), sizeof(int32_t) * st->size());
    if (fid) std::memcpy(fid, fd->data(), sizeof(int32_t) * fd->size());
    if (w) {
        if ((int64_t)ww->size() > w_cap) return MFX_ERR_ARG;
        std::memcpy(w, ww->data(), sizeof(float) * ww->size());
    }
    return rounds;
}

// Operands of the DCT on the matrix pipe: out[(tile * ksteps + j) * 64 + lane] (build_dct_mfma_operands); returns
// tiles * ksteps * 64, or the size needed when out is NULL.
extern "C" int64_t mfx_host_dct_mfma_operands(int32_t num_banks, int32_t dct_len, const float *matrix, float *out,
                                              int64_t out_cap, int32_t *tiles, int32_t *ksteps)
{
    if (num_banks <= 0 || dct_len <= 0 || !matrix) return MFX_ERR_ARG;
    std::vector<float> m(matrix, matrix + (size_t)num_banks * dct_len), ob;
    int tl = 0, ks = 0;
    build_dct_mfma_operands(m, num_banks, dct_len, tl, ks, ob);
    if (tiles) *tiles = tl;
    if (ksteps) *ksteps = ks;
    if (out) {
        if ((int64_t)ob.size() > out_cap) return MFX_ERR_ARG;
        std::memcpy(out, ob.data(), sizeof(float) * ob.size());
    }
    return (int64_t)ob.size();
}

extern "C" int mfx_host_dct_matrix(int32_t num_banks, int32_t ceps_len, int32_t want_c0, float lift_coef,
                                   float *matrix)
{
    if (num_banks <= 0 || ceps_len <= 0 || !matrix) return MFX_ERR_ARG;
    std::vector<float> m;
    build_dct_matrix(num_banks, ceps_len, want_c0 != 0, lift_coef, m);
    std::memcpy(matrix, m.data(), sizeof(float) * m.size());
    return MFX_OK;
}

extern "C" int64_t mfx_host_frame_count(int64_t samples, int32_t window_size, int32_t shift)
{
    if (window_size <= 0 || shift <= 0) return MFX_ERR_ARG;
    return frame_count(samples, window_size, shift);
}
