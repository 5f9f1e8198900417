"""Block-scaled GEMMs on gfx950's scaled MFMA (placeholder until the HIP path lands in this file)."""


def mx_linear_or_none(input, weight, bias, input_scale, weight_scale, block_size, input_code, weight_code):
    return None


def mx_matmul_or_none(a, b, a_scale, b_scale, block_size, a_code, b_code):
    return None
