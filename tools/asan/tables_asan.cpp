// Sanitizer run of the host table builders (csrc/mfx_tables.cpp) under -fsanitize=address,undefined: mel table, DCT
// matrix, twiddles, the 16-lane mel plan of the 512-point kernel and the 64- and 32-lane wave plans, over a grid of
// configurations (bank counts, transform sizes, sample rates, band edges, VTLN warps).  Built by `make -C csrc asan`.
#include <cstdio>
#include <vector>

#include "../../asr-featext-opencl_amd/csrc/mfx_tables.h"

int main()
{
    int n = 0;
    const int ffts[] = {64, 128, 256, 512, 1024, 2048, 4096};
    const int banks[] = {1, 2, 3, 15, 16, 17, 26, 40, 41, 64, 80, 128};
    const float srs[] = {8000.f, 16000.f, 44100.f};
    const float alphas[] = {0.8f, 1.0f, 1.2f};
    for (int fft : ffts)
        for (int nb : banks)
            for (float sr : srs)
                for (float alpha : alphas)
                    for (int lo = 0; lo < 2; ++lo) {
                        mfx::MelTable t;
                        mfx::build_mel_table(nb, fft, sr, lo ? 300.f : 0.f, lo ? sr / 2 - 100.f : sr / 2, alpha, t);
                        bool edges_ok = true;
                        for (int v : t.beg) edges_ok = edges_ok && v >= 0 && v <= fft / 2;
                        if (!edges_ok) continue; // the product refuses such a configuration (mfx_create: MFX_ERR_CONFIG)
                        if (fft == 512) {
                            mfx::MelLanePlan lp;
                            (void)mfx::build_mel_lane_plan(t, nb, fft, 511 - 32, lp);
                        }
                        mfx::MelWavePlan wp;
                        (void)mfx::build_mel_wave_plan(t, nb, fft, fft - 1, wp, 64);
                        if (fft == 2048) (void)mfx::build_mel_wave_plan(t, nb, fft, 1039, wp, 32);
                        for (int ceps : {0, 1, 12, 13, 40}) {
                            if (ceps == 0) continue;
                            std::vector<float> m, mt;
                            mfx::build_dct_matrix(nb, ceps, (nb + ceps) & 1, 22.f, m);
                            int stride = 0, nb_pad = 0;
                            mfx::build_dct_transposed(m, nb, ceps + ((nb + ceps) & 1), stride, nb_pad, mt);
                        }
                        ++n;
                    }
    std::vector<float> tw;
    for (int fft : ffts) mfx::build_twiddles(fft, fft / 2 + 1, tw);
    for (long s : {0L, 1L, 399L, 400L, 160000L, 57600000L, 1L << 31})
        for (int W : {400, 1024}) (void)mfx::frame_count(s, W, 160);
    std::printf("tables_asan: %d configurations clean\n", n);
    return 0;
}
