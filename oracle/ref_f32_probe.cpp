// ref_f32_probe.cpp -- which overloads do UNQUALIFIED log() / abs() on a float pick in this translation unit?
// (TEST INFRASTRUCTURE ONLY; oracle/Makefile `ref_f32` compiles it with and without its binding flags and refuses to go
// on unless the answer flips: "8 2" = the C double log and int abs(int), which is how plain g++ compiles
// mfcccpu.cpp:21-22,37,203,212 and normalizercpu.cpp:66; "4 2.7" = the float overloads the reference's own toolchain picks.)
// Same include set as the reference files it stands for (mfcccpu.cpp:1-8, normalizercpu.cpp:1-4).
#include <cmath>
#include <algorithm>
#include <cstdio>

int main()
{
    float f = 1.0f, g = -2.7f;
    std::printf("%d %g\n", (int)sizeof(log(f)), (double)abs(g));
    return 0;
}
