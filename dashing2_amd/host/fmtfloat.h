// fmtfloat.h -- text form of a float exactly as fmt's "{}" prints it (reference output path:
// src/emitrect.cpp:79-106 `of.print("\t{}", float)`).  Shortest round-trip digits; fixed notation
// when the decimal exponent is in [-4, exp_upper), else d[.ddd]e±XX.
// exp_upper = 7 for float in fmt >= 11 (numeric_limits<float>::digits10 + 1); older fmt used 16
// for every type.  The reference's pinned fmt is unknown (empty submodule): default = fmt 12.1.0
// behaviour (the goldens in tests/golden/fmt_float.tsv), override with D2_FMT_EXP_UPPER=16.
#pragma once
#include <cstddef>

namespace d2h {
constexpr int FMT_MAX_FLOAT_CHARS = 48;
// writes the text (no terminator) at out, returns its length
size_t format_float(float v, char *out);
void set_exp_upper(int e);
}  // namespace d2h
