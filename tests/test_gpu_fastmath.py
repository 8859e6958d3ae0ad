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
