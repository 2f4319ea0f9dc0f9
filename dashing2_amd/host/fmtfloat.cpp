#include "fmtfloat.h"
#include <charconv>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace d2h {

// fmt < 11 prints every floating type in fixed notation up to 1e16 (exp_upper = 16); fmt >= 11 made the bound
// digits10 + 1 of the type, 7 for float.  dashing2 v2.1.x (2021-23) vendored fmt as an unpinned submodule
// (.gitmodules:7-9): releases 8-10 of that period => 16 is the default here; `--fmt-compat 11` selects 7.
static int g_exp_upper = 16;
bool set_fmt_compat(int fmt_major) {
    if (fmt_major == 10) { g_exp_upper = 16; return true; }
    if (fmt_major == 11) { g_exp_upper = 7; return true; }
    return false;
}
int fmt_compat() { return g_exp_upper == 16 ? 10 : 11; }

size_t format_float(float v, char *out) {
    char *p = out;
    if (std::signbit(v)) { *p++ = '-'; v = -v; }
    if (std::isinf(v)) { std::memcpy(p, "inf", 3); return size_t(p - out) + 3; }
    if (std::isnan(v)) { std::memcpy(p, "nan", 3); return size_t(p - out) + 3; }
    if (v == 0.f) { *p++ = '0'; return size_t(p - out); }
    // shortest round-trip digits in scientific form: d[.ddd]e[+-]XX
    char sci[32];
    auto res = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);
    char digits[16];
    int nd = 0;
    const char *q = sci;
    for (; q < res.ptr && *q != 'e'; ++q)
        if (*q != '.') digits[nd++] = *q;
    int exp10 = 0;                                        // exponent of the first digit
    {
        const char *e = q + 1;
        const bool neg = (*e == '-');
        if (*e == '-' || *e == '+') ++e;
        std::from_chars(e, res.ptr, exp10);
        if (neg) exp10 = -exp10;
    }
    while (nd > 1 && digits[nd - 1] == '0') --nd;         // (to_chars never pads, defensive)
    if (exp10 >= -4 && exp10 < g_exp_upper) {
        if (exp10 >= nd - 1) {                            // integer: digits then zeros
            std::memcpy(p, digits, nd); p += nd;
            for (int i = 0; i < exp10 - (nd - 1); ++i) *p++ = '0';
        } else if (exp10 >= 0) {                          // point inside the digits
            std::memcpy(p, digits, exp10 + 1); p += exp10 + 1;
            *p++ = '.';
            std::memcpy(p, digits + exp10 + 1, nd - exp10 - 1); p += nd - exp10 - 1;
        } else {                                          // 0.000ddd
            *p++ = '0'; *p++ = '.';
            for (int i = 0; i < -exp10 - 1; ++i) *p++ = '0';
            std::memcpy(p, digits, nd); p += nd;
        }
        return size_t(p - out);
    }
    *p++ = digits[0];
    if (nd > 1) { *p++ = '.'; std::memcpy(p, digits + 1, nd - 1); p += nd - 1; }
    *p++ = 'e';
    int ae = exp10;
    if (exp10 < 0) { *p++ = '-'; ae = -exp10; } else *p++ = '+';
    if (ae >= 100) { *p++ = char('0' + ae / 100); ae %= 100; }
    *p++ = char('0' + ae / 10);
    *p++ = char('0' + ae % 10);
    return size_t(p - out);
}

}  // namespace d2h
