// fmtfloat.h -- text form of a float exactly as fmt's "{}" prints it (reference output path:
// src/emitrect.cpp:79-106 `of.print("\t{}", float)`).  Shortest round-trip digits; fixed notation
// when the decimal exponent is in [-4, exp_upper), else d[.ddd]e±XX.
// exp_upper = 16 for every floating type in fmt < 11; fmt >= 11 uses numeric_limits<T>::digits10 + 1
// (7 for float), so only values >= 1e7 differ (`12345678` vs `1.2345678e+07`).  The reference's fmt
// is an unpinned submodule; a 2.1.x-era dashing2 predates fmt 11, so the fmt < 11 layout is the
// default and `--fmt-compat 11` (CLI) / set_fmt_compat(11) selects the newer one.  Goldens for both:
// tests/golden/fmt_float.tsv (fmt 12.1.0 itself) and tests/golden/fmt10_float.tsv (derived by rule).
#pragma once
#include <cstddef>

namespace d2h {
constexpr int FMT_MAX_FLOAT_CHARS = 48;
// writes the text (no terminator) at out, returns its length
size_t format_float(float v, char *out);
bool set_fmt_compat(int fmt_major);   // 10 (default) or 11; false for anything else
int  fmt_compat();
}  // namespace d2h
