"""Oracle-side text rendering (TEST INFRASTRUCTURE ONLY): what the reference's emit_rectangular
prints (src/emitrect.cpp:136-151,172-187) given the float matrix, with fmt's "{}" float layout
(shortest round-trip digits; fixed notation for decimal exponents in [-4, exp_upper), else d.ddde±XX).
exp_upper = 16 in fmt < 11 (the default here: a 2.1.x-era dashing2 predates fmt 11), 7 for float in
fmt >= 11.  Pinned against fmt 12.1.0 itself by tests/golden/fmt_float.tsv (EXP_UPPER = 7) and against
the fmt < 11 table derived from it by rule, tests/golden/fmt10_float.tsv."""
import numpy as np

EXP_UPPER = 16     # module default: fmt < 11.  fmt >= 11: numeric_limits<float>::digits10 + 1 = 7


def fmt_float(x, exp_upper=None):
    exp_upper = EXP_UPPER if exp_upper is None else exp_upper
    x = np.float32(x)
    if np.isnan(x):
        return "nan"
    s = "-" if np.signbit(x) else ""
    x = abs(x)
    if np.isinf(x):
        return s + "inf"
    if x == 0:
        return s + "0"
    sci = np.format_float_scientific(x, unique=True, trim="-", exp_digits=1)   # e.g. 1.2345e-5
    mant, exp = sci.split("e")
    e = int(exp)
    digits = mant.replace(".", "")
    nd = len(digits)
    if -4 <= e < exp_upper:
        if e >= nd - 1:
            return s + digits + "0" * (e - (nd - 1))
        if e >= 0:
            return s + digits[:e + 1] + "." + digits[e + 1:]
        return s + "0." + "0" * (-e - 1) + digits
    out = digits[0] + ("." + digits[1:] if nd > 1 else "")
    ae = abs(e)
    return s + out + ("e-" if e < 0 else "e+") + (f"{ae:02d}" if ae < 100 else str(ae))


def padded(name):
    return name + " " * max(0, 9 - len(name))


def render_symmetric(names, condensed, phylip, options_string=None):
    """PHYLIP (--phylip) or the default TSV for the condensed upper triangle."""
    n = len(names)
    out = []
    if phylip:
        out.append(f"{n}\n")
    else:
        out.append("#Dashing2 Symmetric pairwise Output\n")
        out.append(f"#Dashing2Options: {options_string}\n")
        out.append("#Sources" + "".join("\t" + x for x in names) + "\n")
    idx = 0
    for i in range(n):
        row = padded(names[i])
        if not phylip:
            row += "\t-" * (i + 1)
        for _ in range(i + 1, n):
            row += "\t" + fmt_float(condensed[idx])
            idx += 1
        out.append(row + "\n")
    return "".join(out)


def render_rect(names, row_names, mat, label, options_string):
    out = [f"#Dashing2 {label} Output\n", f"#Dashing2Options: {options_string}\n",
           "#Sources" + "".join("\t" + x for x in names) + "\n"]
    for i, rn in enumerate(row_names):
        out.append(padded(rn) + "".join("\t" + fmt_float(v) for v in mat[i]) + "\n")
    return "".join(out)
