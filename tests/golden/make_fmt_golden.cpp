// Generates tests/golden/fmt_float.tsv with fmt (header-only, the copy bundled with torch: 12.1.0).
// BUILD CONTAINER ONLY.  g++ -std=c++17 -DFMT_HEADER_ONLY -I<torch>/include make_fmt_golden.cpp
// The reference prints distances with fmt "{}" on float (src/emitrect.cpp:79-106); its pinned fmt
// version is unknown (empty submodule), so 12.1.0 is the closest available authority.
#include <fmt/format.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <limits>
#include <random>
#include <set>

int main() {
    std::set<uint32_t> bits;
    auto add = [&](float f) { uint32_t u; std::memcpy(&u, &f, 4); bits.insert(u); };
    for (int k = 0; k <= 1024; ++k) add(float(k) / 1024.f);
    for (int k = 0; k <= 1000; ++k) add(float(k) / 1000.f);
    for (int k = 0; k <= 4096; k += 7) add(float(k) / 4096.f);
    for (int e = -45; e <= 38; ++e) { add(std::pow(10.f, float(e))); add(3.f * std::pow(10.f, float(e))); add(1.2345678f * std::pow(10.f, float(e))); }
    for (int e = -149; e <= 127; e += 3) add(std::ldexp(1.f, e));
    add(std::numeric_limits<float>::infinity()); add(std::numeric_limits<float>::max()); add(std::numeric_limits<float>::min());
    add(std::numeric_limits<float>::denorm_min()); add(0.f); add(-0.f); add(1.f); add(16777216.f); add(1e15f); add(1e16f); add(9.9999e15f);
    add(123456.7f); add(0.0001f); add(0.00001f); add(0.00009999f); add(99999.99f); add(-1.5f); add(-0.001f);
    // mash distances for k = 31 and k = 21 over neq/1024
    for (int k : {21, 31}) for (int e = 1; e <= 1024; ++e) { float s = float(e) / 1024.f; add(float(std::log(2. * s / (1. + s)) * (-1. / k))); }
    std::mt19937 rng(7);
    for (int i = 0; i < 4000; ++i) { uint32_t u = rng(); float f; std::memcpy(&f, &u, 4); if (std::isnan(f)) continue; add(f); }
    for (int i = 0; i < 2000; ++i) { add(float(rng() % 100000000u) * 1e-3f); add(float(rng() % 1000000u)); }
    for (uint32_t u : bits) { float f; std::memcpy(&f, &u, 4); std::printf("%08x\t%s\n", u, fmt::format("{}", f).c_str()); }
    return 0;
}
