"""Generator of tools/experiments/fft_roots_odd.inc: cos / sin (2 pi k / P), k = 0..P-1, correctly rounded (mpmath, 50 digits), for
the odd prime radices of the native mixed-radix Fourier rows (tools/experiments/fft_native.h: bfly_odd_stream).  The index is a compile-time
constant wherever the table is used, so the entries end up as literal operands of the butterflies.

Usage: python tools/gen_fft_roots_odd.py > tools/experiments/fft_roots_odd.inc"""
import mpmath

mpmath.mp.dps = 50
PRIMES = (7, 11, 13, 17, 19, 23, 29, 31, 37, 41, 43, 47)

print("// cos / sin (2 pi k / P), k = 0..P-1, for the odd prime radices P, correctly rounded (tools/gen_fft_roots_odd.py; mpmath, 50 digits)")
for P in PRIMES:
    c = [mpmath.cos(2 * mpmath.pi * k / P) for k in range(P)]
    s = [mpmath.sin(2 * mpmath.pi * k / P) for k in range(P)]
    print(f"template <> struct OddRoots<{P}> {{")
    print("    static constexpr double c[%d] = {%s};" % (P, ", ".join(repr(float(v)) for v in c)))
    print("    static constexpr double s[%d] = {%s};" % (P, ", ".join(repr(float(v)) for v in s)))
    print("};")
