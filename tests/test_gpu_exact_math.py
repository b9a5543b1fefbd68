"""The frame kernel's short exact forms (rwr_device.h: to_non_linear_depth_fast, normalize3_fast) must
return the very bits of the shader expressions they replace (compute.wgsl:78-80, :162).  The library
checks them on the GPU itself: every float of the depth form's domain, and 2^30 pseudo-random vectors
over the whole domain of the normalize form."""
import pytest


@pytest.mark.gpu
def test_short_exact_forms_match_the_ieee_expressions_bit_for_bit(rwr, gpu_ctx):
    depth_n, depth_bad, vec_n, vec_bad = gpu_ctx.selftest_exact_math(normalize_count=1 << 30, seed=7)
    # positive normal floats below 2^126: biased exponents 1..252, 2^23 mantissas each
    assert depth_n == 252 * (1 << 23)
    assert depth_bad == 0
    assert vec_n > 0.7 * (1 << 30)      # most generated vectors are in the domain
    assert vec_bad == 0
