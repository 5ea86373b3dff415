#!/usr/bin/env python3
"""Extracts the EVALUATION ORDER of the reference's generated 5-point polynomial code into a compact
term table (runs only where /root/reference exists; the table it writes is committed).

The reference builds Nister's 10 x 20 constraint matrix and the degree-10 determinant polynomial with
machine-generated straight-line code (/root/reference/src/estimators/essential_matrix_poly.h and
essential_matrix_coeffs.h, included at essential_matrix.cc:76-103).  Floating-point results depend on
the order of its sums and products, and that order follows no rule one could restate: it is data.  This
script parses the two headers as C++ would evaluate them (a sum of products, both left-associative) and
writes `dagsfm_amd/csrc/fivept_terms.tbl`:

    A <index> <term> <term> ...        a[index] of A.data()   (10 x 20, column-major: row = index % 10)
    C <index> <term> <term> ...        coeffs(index)
    term   = <sign><factor>*<factor>...      sign '+' / '-' applied to the finished product
    factor = e<k> | s<k> | u<k>   e[k], e2[k] = e[k]*e[k], e3[k] = e2[k]*e[k]   (E.data(), 9 x 4 column-major)
           | b<k>                 b[k] of B.data()   (13 x 3 column-major)
           | k<value>             a literal (0.5, 1.5, 3.)

tools/expand_fivept.py turns the table back into straight-line code for the oracle (C++) and the device
(HIP) at build time; tests/test_fivept_reference_order.py compares that code bit for bit with the
reference's own headers compiled behind a shim (oracle/_ref).
"""
import os
import re
import sys

REF = "/root/reference/src/estimators"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dagsfm_amd", "csrc", "fivept_terms.tbl")

FACTOR = re.compile(r"(e2|e3|e|b)\[(\d+)\]$")


def body(path):
    text = open(path).read()
    text = text[text.index("\n{") + 2:text.rindex("}")]
    text = re.sub(r"//[^\n]*", "", text)
    return text


def parse_expr(expr):
    """sum of products, no parentheses; returns [(sign, [factor, ...]), ...] in source order"""
    toks = re.findall(r"[-+*]|[a-z0-9]+\[\d+\]|\d+\.\d*|\d+", expr.replace(" ", "").replace("\n", ""))
    assert "".join(toks) == expr.replace(" ", "").replace("\n", ""), expr[:80]
    terms, sign, cur, expect_factor = [], "+", [], True
    for t in toks:
        if t in "+-" and expect_factor and not cur:
            sign = "-" if (t == "-") != (sign == "-") else "+"   # unary sign of the leading factor
            continue
        if t in "+-":
            terms.append((sign, cur))
            sign, cur, expect_factor = t, [], True
            continue
        if t == "*":
            expect_factor = True
            continue
        m = FACTOR.match(t)
        if m:
            cur.append({"e": "e", "e2": "s", "e3": "u", "b": "b"}[m.group(1)] + m.group(2))
        else:
            cur.append("k" + repr(float(t)))
        expect_factor = False
    terms.append((sign, cur))
    return terms


def main():
    lines = []
    poly = body(os.path.join(REF, "essential_matrix_poly.h"))
    # drop the preamble (pointer set-up and the e2 / e3 loop): statements of interest start with a[
    stmts = re.findall(r"\ba\[(\d+)\]\s*=([^;]*);", poly)
    assert len(stmts) == 200 and sorted(int(i) for i, _ in stmts) == list(range(200))
    nterms = 0
    for idx, expr in stmts:   # keep the source order of the statements
        terms = parse_expr(expr)
        nterms += len(terms)
        lines.append("A %s " % idx + " ".join(s + "*".join(f) for s, f in terms))
    coef = body(os.path.join(REF, "essential_matrix_coeffs.h"))
    stmts = re.findall(r"\bcoeffs\((\d+)\)\s*=([^;]*);", coef)
    assert len(stmts) == 11
    for idx, expr in stmts:
        terms = parse_expr(expr)
        nterms += len(terms)
        lines.append("C %s " % idx + " ".join(s + "*".join(f) for s, f in terms))
    header = [
        "# Evaluation order of the reference's generated 5-point polynomial code (data, not code):",
        "# /root/reference/src/estimators/essential_matrix_poly.h (A) and essential_matrix_coeffs.h (C).",
        "# Written by tools/gen_fivept_tables.py; format documented there.  %d targets, %d terms." % (len(lines), nterms),
    ]
    with open(OUT, "w") as f:
        f.write("\n".join(header + lines) + "\n")
    print("wrote %s: %d targets, %d terms" % (os.path.normpath(OUT), len(lines), nterms))


if __name__ == "__main__":
    sys.exit(main())
