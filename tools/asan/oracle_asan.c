/* Sanitizer run of the CPU checker (TEST INFRASTRUCTURE): oracle/mfcc_oracle.c built with
 * -fsanitize=address,undefined and driven through the reference call sequence (ASR_OCL.cpp:149-301) over the
 * shapes the tests use: uneven blocks, flush after one block, every dyn / norm mode, short files, the long
 * transforms.  Exits non-zero on any sanitizer report (halt_on_error) or API error.  Built by `make -C oracle asan`. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../oracle/mfcc_oracle.h"

static unsigned rng = 12345u;
static int rnd(void)
{
    rng = rng * 1664525u + 1013904223u;
    return (int)(rng >> 8);
}

int main(void)
{
    int cases = 0;
    static const int shapes[][6] = {
        /* W, S, banks, ceps, samples, block */
        {400, 160, 26, 13, 114000, 32000}, {400, 160, 40, 13, 16000, 16000}, {400, 160, 15, 12, 81000, 10000000},
        {1024, 160, 80, 13, 40000, 9000},  {1102, 441, 128, 40, 60000, 20000}, {256, 64, 20, 0, 5000, 1777},
        {400, 160, 26, 13, 1500, 1500},    {300, 77, 23, 9, 30011, 4099},
    };
    for (unsigned s = 0; s < sizeof(shapes) / sizeof(shapes[0]); ++s)
        for (int dyn = 0; dyn <= 2; ++dyn)
            for (int norm = 0; norm <= 3; ++norm)
                for (int fft_mode = 0; fft_mode <= 1; ++fft_mode) {
                    const int W = shapes[s][0], S = shapes[s][1], n = shapes[s][4], block = shapes[s][5];
                    orc_config cfg;
                    cfg.input_buffer_size = block < n ? block : n + 1000;
                    cfg.window_size = W;
                    cfg.shift = S;
                    cfg.num_banks = shapes[s][2];
                    cfg.sample_rate = W > 1024 ? 44100.f : 16000.f;
                    cfg.low_freq = 64.f;
                    cfg.high_freq = cfg.sample_rate / 2;
                    cfg.ceps_len = shapes[s][3];
                    cfg.want_c0 = s & 1;
                    cfg.lift_coef = 22.f;
                    cfg.norm = norm;
                    cfg.dyn = dyn;
                    cfg.delta_l1 = 1 + (int)(s % 3);
                    cfg.delta_l2 = 1 + (int)((s + 1) % 3);
                    cfg.norm_after_dyn = (s >> 1) & 1;
                    cfg.fft_mode = fft_mode;
                    float *window = (float *)malloc(sizeof(float) * (size_t)W);
                    for (int i = 0; i < W; ++i) window[i] = (float)(0.56 - 0.46 * cos(2.0 * M_PI * i / W)) / 32768.f;
                    short *pcm = (short *)malloc(sizeof(short) * (size_t)n);
                    for (int i = 0; i < n; ++i) pcm[i] = (short)(rnd() % 20000 - 10000);
                    const int width = orc_output_width(cfg.num_banks, cfg.ceps_len, cfg.want_c0, cfg.dyn);
                    const int frames = orc_ewc(n, W, S);
                    float *out = (float *)malloc(sizeof(float) * (size_t)(frames > 0 ? frames : 1) * (size_t)width);
                    for (int compat = 0; compat <= 1; ++compat) {
                        const int got = orc_run_utterance(&cfg, window, 1.0f, compat, pcm, n, cfg.input_buffer_size, out);
                        /* short files are refused with the reference's window-count error; everything else must run */
                        if (got < 0 && got != ORC_ERR_WINDOW_COUNT) {
                            fprintf(stderr, "shape %u dyn %d norm %d: error %d\n", s, dyn, norm, got);
                            return 1;
                        }
                        ++cases;
                    }
                    free(out);
                    free(pcm);
                    free(window);
                }
    printf("oracle_asan: %d runs clean\n", cases);
    return 0;
}
