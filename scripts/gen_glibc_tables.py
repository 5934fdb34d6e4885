#!/usr/bin/env python3
"""Generate criteria3d_amd/csrc/sf3d_glibcmath_tables.h: the data of the elementary functions the REFERENCE build calls -
glibc 2.35's `log`, `exp`, `pow` (x86-64, the variants the dynamic loader selects on a CPU with FMA + AVX2) and `cbrt` -
for the bit-faithful device routines of sf3d_glibcmath.inc.

Why the data comes out of the C library and not out of mpmath alone: the routines are the table designs of Szabolcs Nagy
(ARM "optimized routines", adopted by glibc 2.28): x = 2^k z, 128 pieces of z, a polynomial in r = z / c - 1.  Which c a piece
of `log` uses was chosen by a search over 2^29 candidates per piece, and the polynomials are minimax fits - neither can be re-derived
from the mathematics, and a single different last bit anywhere gives results that are merely "as accurate", not IDENTICAL to
the library's.  Identical is the point (the config-5 kink window amplifies a last-ulp difference of log / pow to
7.7e-4), so the 128 centres of `log` and the 27 polynomial coefficients (log 5 + 11, pow 7, exp 4) are READ from the installed
library; everything else - pow's whole table, exp's whole table, every log c - is DERIVED here with mpmath at 200 bits and required to
agree with what the library holds:

  log      logc_i                 == RN(-log(invc_i))                         (128 entries; invc_i read)
  pow      invc_i == 1 / centre of piece i rounded to a multiple of 2^-7 (pieces below 1) / 2^-8 (from 1 up);
           logc_i == -log(invc_i) rounded to a multiple of 2^-43,  logctail_i == RN(-log(invc_i) - logc_i)
  exp      tab[2 i + 1]           == bits(RN(2^(i/128))) - (i << 45),  tab[2 i] == bits(RN(2^(i/128) / H - 1)), H = RN(2^(i/128))
  ln2hi + ln2lo == ln 2 to 2^-90;  InvLn2N == RN(128 / ln 2);  NegLn2hiN + NegLn2loN == -ln 2 / 128 to 2^-100
  cbrt     factor[] == { 1 / RN(2^(2/3)), 1 / RN(2^(1/3)), 1, RN(2^(1/3)), RN(2^(2/3)) } (the quotients rounded once more, as the compiler folds them)

The tables are located in the library's .rodata by their leading constants (no addresses are hard-wired), so the script works on
any glibc 2.28+ x86-64 libm that carries this design; the committed header was generated from Ubuntu GLIBC 2.35-0ubuntu3.11,
the C library of the image the reference oracle (oracle/_ref) is built and run in.  glibc is LGPL-2.1-or-later; the numbers are
facts about its behaviour that a bit-compatible implementation has to share.

Run:  python scripts/gen_glibc_tables.py [path/to/libm.so.6]   (rewrites the header; the result is committed)
"""
from __future__ import annotations

import struct
import sys
from pathlib import Path

import mpmath as mp

mp.mp.prec = 200
ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "criteria3d_amd" / "csrc" / "sf3d_glibcmath_tables.h"
LIBM = Path("/lib/x86_64-linux-gnu/libm.so.6")
N = 128


def bits(x: float) -> int:
    return struct.unpack("<Q", struct.pack("<d", x))[0]


def from_bits(u: int) -> float:
    return struct.unpack("<d", struct.pack("<Q", u & 0xFFFFFFFFFFFFFFFF))[0]


def rn(x) -> float:
    return float(mp.mpf(x))          # mpmath rounds to nearest even on conversion


class Rodata:
    def __init__(self, path: Path):
        self.blob = path.read_bytes()
        b = self.blob
        assert b[:4] == b"\x7fELF" and b[4] == 2 and b[5] == 1, "not a little-endian ELF64"
        shoff, = struct.unpack_from("<Q", b, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", b, 0x3A)
        secs = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + k * shentsize) for k in range(shnum)]
        stroff = secs[shstrndx][4]
        self.lo = self.hi = None
        for s in secs:
            name = b[stroff + s[0]:b.index(b"\0", stroff + s[0])].decode()
            if name == ".rodata":
                self.lo, self.hi = s[4], s[4] + s[5]
        assert self.lo is not None, "no .rodata"

    def find_all(self, *leading: float):
        """file offsets of the 8-aligned places in .rodata that start with the given doubles"""
        pat = b"".join(struct.pack("<d", v) for v in leading)
        pos, hits = self.lo, []
        while True:
            pos = self.blob.find(pat, pos, self.hi)
            if pos < 0:
                return hits
            if pos % 8 == 0:
                hits.append(pos)
            pos += 8

    def find(self, *leading: float, then=None) -> int:
        """the one such place (whose next double satisfies `then`)"""
        hits = [h for h in self.find_all(*leading) if then is None or then(self.doubles(h + 8 * len(leading), 1)[0])]
        assert len(hits) == 1, (leading, hits)
        return hits[0]

    def doubles(self, off: int, n: int):
        return list(struct.unpack_from(f"<{n}d", self.blob, off))

    def words(self, off: int, n: int):
        return list(struct.unpack_from(f"<{n}Q", self.blob, off))


LN2HI, LN2LO = float.fromhex("0x1.62e42fefa3800p-1"), float.fromhex("0x1.ef35793c76730p-45")


def read_log(ro: Rodata):
    """struct log_data { double ln2hi, ln2lo, poly[5], poly1[11]; struct { double invc, logc; } tab[128]; ... } (sysdeps/ieee754/dbl-64/math_config.h)"""
    off = ro.find(LN2HI, LN2LO, then=lambda a0: a0 < -0.5)            # log: A[0] = -0.5 - 2^-53 ; pow: exactly -0.5
    poly = ro.doubles(off + 16, 5)
    poly1 = ro.doubles(off + 56, 11)
    tab = ro.doubles(off + 144, 2 * N)
    invc, logc = tab[0::2], tab[1::2]
    assert poly1[0] == -0.5
    for i in range(N):
        if invc[i] == 1.0:
            assert logc[i] == 0.0
            continue
        assert logc[i] == rn(-mp.log(mp.mpf(invc[i]))), ("log", i)
    assert abs(mp.mpf(LN2HI) + mp.mpf(LN2LO) - mp.log(2)) < mp.mpf(2) ** -90
    return poly, poly1, invc, logc


def read_pow_log(ro: Rodata):
    """struct pow_log_data { double ln2hi, ln2lo, poly[7]; struct { double invc, pad, logc, logctail; } tab[128]; }"""
    off = ro.find(LN2HI, LN2LO, then=lambda a0: a0 == -0.5)
    poly = ro.doubles(off + 16, 7)
    tab = ro.doubles(off + 72, 4 * N)
    invc, pad, logc, tail = tab[0::4], tab[1::4], tab[2::4], tab[3::4]
    assert all(p == 0.0 for p in pad)
    # pow's table centres ARE derivable: piece i covers the doubles with bits [OFF + i 2^45, OFF + (i + 1) 2^45), OFF = bits(0x1.69555p-1);
    # invc_i = 1 / centre rounded to a multiple of 2^-7 for the pieces below 1 and of 2^-8 from 1 up - so few bits that z * invc - 1 is exact
    off = 0x3FE6955500000000
    for i in range(N):
        lo, hi = from_bits(off + (i << 45)), from_bits(off + ((i + 1) << 45))
        q = 7 if hi <= 1.0 else 8
        derived = float(mp.nint(mp.mpf(2) ** q / ((mp.mpf(lo) + mp.mpf(hi)) / 2))) / 2 ** q
        assert derived == invc[i], ("pow_log invc", i, derived, invc[i])
    for i in range(N):
        t = -mp.log(mp.mpf(invc[i]))
        if invc[i] == 1.0:
            assert logc[i] == 0.0 and tail[i] == 0.0
            continue
        assert abs(mp.mpf(logc[i]) + mp.mpf(tail[i]) - t) < mp.mpf(2) ** -97, ("pow_log", i)
        assert logc[i] == float(mp.nint(t * 2 ** 43)) / 2 ** 43, ("pow_log logc", i)          # a multiple of 2^-43: k * ln2hi + logc is exact
        assert tail[i] == rn(t - mp.mpf(logc[i])), ("pow_log tail", i)
    return poly, invc, logc, tail


def read_exp(ro: Rodata):
    """struct exp_data { double invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5]; uint64_t tab[2 * 128]; }"""
    inv = rn(mp.mpf(N) / mp.log(2))
    shift = float.fromhex("0x1.8p52")
    off = ro.find(inv, shift)
    head = ro.doubles(off, 8)
    tab = ro.words(off + 112, 2 * N)
    assert head[0] == inv and head[1] == shift
    assert abs(mp.mpf(head[2]) + mp.mpf(head[3]) + mp.log(2) / N) < mp.mpf(2) ** -100
    for i in range(N):
        v = mp.power(2, mp.mpf(i) / N)
        h = rn(v)
        assert tab[2 * i + 1] == (bits(h) - (i << 45)) & 0xFFFFFFFFFFFFFFFF, ("exp", i)
        assert tab[2 * i] == bits(rn((v - mp.mpf(h)) / mp.mpf(h))), ("exp tail", i)       # relative: scale * (1 + tail + ...)
    return head, tab


def read_cbrt(ro: Rodata):
    """s_cbrt.c: static const double factor[5] = { 2^(-2/3), 2^(-1/3), 1, 2^(1/3), 2^(2/3) } and the degree-6 seed polynomial"""
    c2, c4 = rn(mp.power(2, mp.mpf(1) / 3)), rn(mp.power(2, mp.mpf(2) / 3))       # the source's CBRT2 and SQR_CBRT2 literals as doubles
    factor = [1.0 / c4, 1.0 / c2, 1.0, c2, c4]                                     # { 1.0 / SQR_CBRT2, 1.0 / CBRT2, 1.0, CBRT2, SQR_CBRT2 }: the two quotients rounded as doubles
    want = [0.145263899385486377, 0.784932344976639262, 1.83469277483613086, 2.44693122563534430, 2.11499494167371287, 1.50819193781584896,
            0.354895765043919860]                                            # the literals of the published source, as doubles
    # the table occurs once per precision (float, double, long double variants); the double routine's is the one followed by the
    # seed polynomial's constants in the order the compiled code uses them (highest degree first)
    hits = [h for h in ro.find_all(*factor) if ro.doubles(h + 40, 7) == want]
    assert len(hits) == 1, hits
    c = ro.doubles(hits[0] + 40, 7)
    return factor, c[::-1]                                                   # c[0] + c[1] xm ... with alternating signs applied in the routine


def fmt(v: float) -> str:
    return float(v).hex()


def main(libm: Path = LIBM):
    ro = Rodata(libm)
    lpoly, lpoly1, linvc, llogc = read_log(ro)
    ppoly, pinvc, plogc, ptail = read_pow_log(ro)
    ehead, etab = read_exp(ro)
    factor, cseed = read_cbrt(ro)
    L = []
    L.append("/* GENERATED by scripts/gen_glibc_tables.py - do not edit.  Data of glibc 2.35's log / exp / pow (FMA variants) and cbrt: the 128 table")
    L.append(' * centres and the minimax polynomials are read from the installed libm, every derived entry is re-computed with mpmath and checked.')
    L.append(" * Provenance of what is READ (155 constants: the 128 centres 1 / c of log and the 27 polynomial coefficients of log, pow's log, exp and")
    L.append(" * cbrt): glibc 2.35, sysdeps/ieee754/dbl-64/e_log_data.c, e_pow_log_data.c, e_exp_data.c and s_cbrt.c - the first three are glibc's copies")
    L.append(" * of the Arm Optimized Routines' math/log_data.c, pow_log_data.c, exp_data.c (Szabolcs Nagy, Arm Ltd, 2018; upstream MIT OR")
    L.append(" * Apache-2.0 WITH LLVM-exception), s_cbrt.c is glibc's own; glibc distributes all of them under LGPL-2.1-or-later.  They are numerical")
    L.append(" * constants of published algorithms, reproduced here so that the kernels return the library's bits (DESIGN.md 4); everything else in this")
    L.append(' * file is derived from the mathematics (mpmath, 200 bits) and only CHECKED against the library. */')
    L.append("#ifndef SF3D_GLIBCMATH_TABLES_H")
    L.append("#define SF3D_GLIBCMATH_TABLES_H")
    L.append(f"#define SF3D_GL_LN2HI {fmt(LN2HI)}")
    L.append(f"#define SF3D_GL_LN2LO {fmt(LN2LO)}")
    L.append("/* log: A[0..4] of the table path, B[0..10] of the path around 1 */")
    L.append("#define SF3D_GL_LOG_A { " + ", ".join(fmt(v) for v in lpoly) + " }")
    L.append("#define SF3D_GL_LOG_B { " + ", ".join(fmt(v) for v in lpoly1) + " }")
    L.append("#define SF3D_GL_LOG_TABLE { \\")
    for i in range(N):
        L.append(f"    {{ {fmt(linvc[i])}, {fmt(llogc[i])} }}, \\")
    L.append("}")
    L.append("/* pow: A[0..6] of log_inline, table { invc, logc, logctail } */")
    L.append("#define SF3D_GL_POW_A { " + ", ".join(fmt(v) for v in ppoly) + " }")
    L.append("#define SF3D_GL_POWLOG_TABLE { \\")
    for i in range(N):
        L.append(f"    {{ {fmt(pinvc[i])}, {fmt(plogc[i])}, {fmt(ptail[i])} }}, \\")
    L.append("}")
    L.append("/* exp (and the second half of pow) */")
    L.append(f"#define SF3D_GL_EXP_INVLN2N {fmt(ehead[0])}")
    L.append(f"#define SF3D_GL_EXP_SHIFT {fmt(ehead[1])}")
    L.append(f"#define SF3D_GL_EXP_NEGLN2HIN {fmt(ehead[2])}")
    L.append(f"#define SF3D_GL_EXP_NEGLN2LON {fmt(ehead[3])}")
    L.append("#define SF3D_GL_EXP_C { " + ", ".join(fmt(v) for v in ehead[4:8]) + " }   /* C2 .. C5 */")
    L.append("#define SF3D_GL_EXP_TABLE { \\")
    for i in range(N):
        L.append(f"    {{ 0x{etab[2 * i]:016x}ull, 0x{etab[2 * i + 1]:016x}ull }}, \\")
    L.append("}")
    L.append("/* cbrt: 2^(k/3), k = -2 .. 2, and the seed polynomial c0 + xm (c1 - xm (c2 - xm (c3 - xm (c4 - xm (c5 - c6 xm))))) */")
    L.append("#define SF3D_GL_CBRT_FACTOR { " + ", ".join(fmt(v) for v in factor) + " }")
    L.append("#define SF3D_GL_CBRT_C { " + ", ".join(fmt(v) for v in cseed) + " }")
    L.append("#endif")
    OUT.write_text("\n".join(L) + "\n")
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main(Path(sys.argv[1]) if len(sys.argv) > 1 else LIBM)
