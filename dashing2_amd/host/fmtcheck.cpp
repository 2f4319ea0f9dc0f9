// fmtcheck <table> [fmt-compat]: reads "<hexbits>\t<expected>" lines, checks d2h::format_float; used by tests/test_host.py
#include "fmtfloat.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main(int argc, char **argv) {
    if (argc < 2) return 2;
    if (argc > 2 && !d2h::set_fmt_compat(std::atoi(argv[2]))) return 2;
    std::FILE *f = std::fopen(argv[1], "r");
    if (!f) return 2;
    char line[256]; int bad = 0, n = 0;
    while (std::fgets(line, sizeof line, f)) {
        unsigned u; char exp[128];
        if (std::sscanf(line, "%x\t%127s", &u, exp) != 2) continue;
        float v; std::memcpy(&v, &u, 4);
        char out[64]; out[d2h::format_float(v, out)] = 0; ++n;
        if (std::strcmp(out, exp)) { if (bad < 10) std::printf("MISMATCH %08x got %s expected %s\n", u, out, exp); ++bad; }
    }
    std::printf("%d checked, %d bad\n", n, bad);
    return bad != 0;
}
