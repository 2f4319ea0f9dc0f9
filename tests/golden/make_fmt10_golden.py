#!/usr/bin/env python3
"""tests/golden/fmt10_float.tsv: the float text of fmt < 11 ("{}" of a float), derived from fmt_float.tsv --
the table fmt 12.1.0 itself produced (make_golden.py) -- by the ONE rule that differs between the generations:
fmt >= 11 switches to exponent notation at 10^(digits10 + 1) = 1e7 for float, fmt < 11 at 1e16 for every type
(fmt/format.h write_float: `exp_lower = -4, exp_upper = 16`; use_exp_format = exp < exp_lower || exp >= exp_upper).
The shortest round-trip DIGITS are the same in both (dragonbox since fmt 7.1), so a value fmt 12 printed as
d.ddde+XX with 7 <= XX < 16 becomes its digits followed by zeros.  Pure string rewriting: independent of the
product's formatter and of the oracle's.  A few values above 1e7 are appended so that both branches are exercised."""
import os
import struct

here = os.path.dirname(os.path.abspath(__file__))


def fixed_from_sci(txt):
    neg = txt.startswith("-")
    body = txt[1:] if neg else txt
    if "e+" not in body:
        return txt
    mant, exp = body.split("e+")
    e = int(exp)
    if not (7 <= e < 16):
        return txt
    digits = mant.replace(".", "")
    assert len(digits) - 1 <= e, txt            # a float has <= 9 significant digits: always an integer here
    return ("-" if neg else "") + digits + "0" * (e - (len(digits) - 1))


def main():
    rows = []
    for line in open(os.path.join(here, "fmt_float.tsv")):
        bits, txt = line.rstrip("\n").split("\t")
        rows.append((bits, fixed_from_sci(txt)))
    have = {b for b, _ in rows}
    # extra values >= 1e7, written out by hand from their exact binary values (floats >= 2^24 are integers; the shortest
    # round-trip digit strings below were checked with repr(numpy.float32))
    extra = {
        1.0e7: "10000000", 12345678.0: "12345678", 16777216.0: "16777216", 33554432.0: "33554432", 1.0e8: "100000000",
        123456792.0: "123456790", 4.0e9: "4000000000", 9.99999986991104e14: "1000000000000000", 1.00000003e16: "1e+16",
        7.2057594e16: "7.2057594e+16", 3.4028235e38: "3.4028235e+38",
    }
    for v, txt in extra.items():
        b = "%08x" % struct.unpack("<I", struct.pack("<f", v))[0]
        if b not in have:
            rows.append((b, txt))
            have.add(b)
    with open(os.path.join(here, "fmt10_float.tsv"), "w") as f:
        for b, t in rows:
            f.write(f"{b}\t{t}\n")
    print(len(rows), "rows")


if __name__ == "__main__":
    main()
