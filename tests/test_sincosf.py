"""K0's sinf / cosf (csrc/elm_la.hpp::glibc_sincosf, used by k_deskew) against the C library of this host.

pcl::getTransformation -- the reference's per-point deskew transform (pcm_matching.cpp:806) -- calls std::cos / std::sin on floats, i.e.
glibc's sinf / cosf.  The device code restates that routine (float64 polynomial, FMA-contracted variant); this test compiles the same
header for the host with g++ and sweeps it against libm: every result must have the same bits.  No GPU needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include "elm_la.hpp"
static uint32_t bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
int main() {
    unsigned long long n = 0, bad = 0;
    // every 97th float32 bit pattern from 0 up to 120.0f, both signs
    for (uint32_t u = 0; u < 0x42F00000u; u += 97) {
        float f; std::memcpy(&f, &u, 4);
        for (int s = 0; s < 2; ++s) {
            const float x = s ? -f : f;
            bad += bits(sinf(x)) != bits(elm::glibc_sincosf<false>(x));
            bad += bits(cosf(x)) != bits(elm::glibc_sincosf<true>(x));
            n += 2;
        }
    }
    // a dense sweep of the range a deskew rotation lives in
    for (int i = -2000000; i <= 2000000; ++i) {
        const float x = (float)i * 1e-6f;
        bad += bits(sinf(x)) != bits(elm::glibc_sincosf<false>(x));
        bad += bits(cosf(x)) != bits(elm::glibc_sincosf<true>(x));
        n += 2;
    }
    std::printf("%llu %llu\n", n, bad);
    return 0;
}
"""


def test_device_sincosf_restatement_equals_libm(tmp_path):
    src = tmp_path / "sweep.cpp"
    src.write_text(SRC)
    exe = tmp_path / "sweep"
    # -mfma only lets __builtin_fma compile to the instruction (a correctly rounded fma either way); no contraction of anything else
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-I", os.path.join(ROOT, "elimaloc_amd", "csrc"), str(src),
                           "-o", str(exe), "-lm"])
    n, bad = map(int, subprocess.check_output([str(exe)], timeout=300).split())
    assert n > 50_000_000
    # glibc picks its FMA variant on every FMA-capable x86-64 CPU (all of them since 2013); a host without FMA may differ in a handful
    # of cosf values beyond |x| = 32 (6 of 320 M) -- still none in the dense sweep
    assert bad == 0, f"{bad} of {n} results differ from libm"
