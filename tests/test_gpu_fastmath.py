"""-m gpu: the kernels replace the compiler's scaled IEEE expansions of 1/x, sqrt(x) and 1/sqrt(x) (two roundings) by
short sequences (v_rcp + one Newton step; x*v_rsq corrected by its exact residual) on the operand range
[2^-100, 2^100].  They are claimed to be EXACT there; this test checks every binary32 bit pattern on the device."""
import pytest

pytestmark = pytest.mark.gpu


def test_short_rcp_sqrt_sequences_are_exact_for_all_2_pow_32_operands():
    import srz
    ctx = srz.Context(0)
    n_fast, bad_rcp, bad_sqrt, bad_rsqrt = ctx.verify_fastmath()
    ctx.close()
    assert n_fast == 2 * 201 * (1 << 23)      # both signs, exponents 27..227
    assert (bad_rcp, bad_sqrt, bad_rsqrt) == (0, 0, 0)


def test_division_by_reciprocal_matches_ieee_division():
    """div_by_rcp (Markstein's correction with the exact reciprocal) is what the scalar-tail shaders use for texel / 255 and
    intensity / distance; theorem + 4e9 pseudo-random in-range pairs + every texel value on the device."""
    import srz
    ctx = srz.Context(0)
    n, bad_pairs, bad_texel = ctx.verify_fastdiv()
    ctx.close()
    assert n == 8192 * 256 * 2048
    assert (bad_pairs, bad_texel) == (0, 0)


@pytest.mark.parametrize("p", [7.5, 2.5, 0.3, 33.3, 149.99, 1000.7, 4095.5])
def test_optimistic_pow_equals_the_correctly_rounded_one_wherever_it_does_not_flag(p):
    """pow_fast (the FAST k_shade builds for non-integer exponents: exp2(p log2 x) in binary64, ~45 operations) against pow_cr
    (binary64 pow rounded once) for EVERY binary32 x in [2^-40, 1 + 16 ulp]: wherever its rounding-safety flag is clear the
    results must be bit-identical; flagged operands (their tile goes to the generic build) must stay rare"""
    import srz
    ctx = srz.Context(0)
    n, bad, flagged, flagged_small = ctx.verify_fastpow(p)
    ctx.close()
    assert n == 0x3f800010 - 0x2b800000 + 1
    assert bad == 0, (p, bad)
    # ambiguous roundings: within 32 + 4 p binary64 ulps of a binary32 rounding boundary = 2 (32 + 4 p) / 2^29 of the normal results
    # (the band of results 2^-151 .. 2^-120 is flagged wholesale and counted apart)
    assert flagged < 100 + n * 4 * (32 + 4 * p) / 2 ** 29, (p, flagged, flagged_small, n)


def test_heron_distance_equals_the_binary64_square_root_wherever_it_does_not_flag():
    """len2d_fast (the scalar Blinn-Phong's attenuation distance in the FAST k_shade builds: two binary64 Heron steps from the
    binary32 root) against len2d_f64 (binary64 sqrt rounded once, src/Shader.cpp:516-523 of the reference) on 4.3e9 pseudo-random
    pairs: bit-identical wherever the rounding-safety flag is clear; flagged pairs (their tile goes to the generic build) stay at
    the 1e-7 level; Pythagorean pairs whose root is exactly a binary32 rounding boundary are flagged"""
    import srz
    ctx = srz.Context(0)
    n, bad, flagged, flagged_pyth, flagged_fewbits = ctx.verify_fastlen()
    ctx.close()
    print("verify_fastlen", n, bad, flagged, flagged_pyth, flagged_fewbits)
    assert n == 8192 * 256 * 2048
    assert bad == 0, (bad, flagged, flagged_pyth, flagged_fewbits)
    assert flagged < 1000 + n * 1e-6, (flagged, n)   # expected: 2 * 8 / 2^29 = 3e-8 of the random pairs (half of all pairs)
    assert flagged_pyth > 1000, flagged_pyth       # roots m^2 + n^2 > 2^24 (odd) are ties: they must be noticed
