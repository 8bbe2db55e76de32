"""CPU property test of the filter margin of the slab tiles (rows longer than 256 values, prefhetch_amd/csrc/flat_wide16.hpp).

k_l2_wide16 tests  (|x|^2 + |y|^2)(1 - m) - 2 acc <= tau  on the accumulator of bf16 products summed in fp32, where the exact path (k_wide_fixup, and
every other path) tests  |x|^2 + |y|^2 - 2 chain <= tau  on the k-ordered fp32 fmaf chain.  The filter may never drop what the chain keeps, i.e.

    2 |acc - chain|  <=  m (|x|^2 + |y|^2),      m = 2.1 x 2^-8 + d x 2^-20   (operands rounded to bf16)
                                                 m = d x 2^-20                (operands exactly representable: 8-bit data)

Here the two sides are emulated in numpy -- bf16 by round-to-nearest-even on the bit pattern, the matrix pipe's accumulation in three different
summation orders (its real order is not documented: the bound must hold for any), with round-to-nearest AND with truncation after every addition --
and the inequality is checked on Gaussian, 8-bit, wide-dynamic-range and worst-case (every value halfway between two bf16 numbers, x parallel to y)
data.  No GPU, no oracle: arithmetic only."""
import numpy as np
import pytest


def bf16_rne(v):
    b = np.ascontiguousarray(v, np.float32).view(np.uint32).astype(np.uint64)
    r = ((b + 0x7FFF + ((b >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def chain_f32(x, y):
    """acc = fmaf(x[k], y[k], acc), k ascending: product and sum exact in a double (48 + 24 bits), one rounding to fp32 per step"""
    acc = np.zeros(x.shape[0], np.float32)
    for k in range(x.shape[1]):
        acc = (x[:, k].astype(np.float64) * y[:, k].astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
    return acc


def trunc32(v64):
    """a double rounded TOWARDS ZERO to fp32 (a matrix pipe that truncated would do this after every addition)"""
    f = v64.astype(np.float32)
    over = np.abs(f.astype(np.float64)) > np.abs(v64)
    return np.where(over, np.nextafter(f, np.float32(0)), f).astype(np.float32)


def acc_pipe(xh, yh, order, truncate):
    """sum of the (exact) bf16 x bf16 products in fp32, one rounding per addition, in a given order"""
    prod = xh.astype(np.float64) * yh.astype(np.float64)             # exact: 8 x 8 significant bits
    rnd = trunc32 if truncate else (lambda v: v.astype(np.float32))
    n, d = prod.shape
    if order == "sequential":
        acc = np.zeros(n, np.float32)
        for k in range(d):
            acc = rnd(acc.astype(np.float64) + prod[:, k])
        return acc
    if order == "blocks16":                                            # 16 products summed first (one instruction's depth), then added to the accumulator
        acc = np.zeros(n, np.float32)
        for k0 in range(0, d, 16):
            part = np.zeros(n, np.float32)
            for k in range(k0, min(k0 + 16, d)):
                part = rnd(part.astype(np.float64) + prod[:, k])
            acc = rnd(acc.astype(np.float64) + part.astype(np.float64))
        return acc
    vals = [rnd(prod[:, k]) for k in range(d)]                        # pairwise tree
    while len(vals) > 1:
        nxt = [rnd(vals[i].astype(np.float64) + vals[i + 1].astype(np.float64)) for i in range(0, len(vals) - 1, 2)]
        if len(vals) % 2:
            nxt.append(vals[-1])
        vals = nxt
    return vals[0]


def norms(v):
    acc = np.zeros(v.shape[0], np.float32)
    for k in range(v.shape[1]):
        acc = (v[:, k].astype(np.float64) ** 2 + acc.astype(np.float64)).astype(np.float32)
    return acc.astype(np.float64)


LAWS = ["gauss", "u8", "range", "halfway_parallel", "signs"]


@pytest.mark.parametrize("d", [257, 512, 1024])
@pytest.mark.parametrize("law", LAWS)
def test_slab_tile_margin_covers_operand_rounding_and_both_accumulations(d, law):
    rng = np.random.default_rng(d * 31 + len(law))
    n = 160
    if law == "gauss":
        x, y = rng.standard_normal((n, d)), rng.standard_normal((n, d))
    elif law == "u8":
        x, y = rng.integers(0, 256, (n, d)), rng.integers(0, 256, (n, d))
    elif law == "range":                                              # magnitudes over twelve binades, mixed signs
        x = rng.standard_normal((n, d)) * 2.0 ** rng.integers(-6, 7, (n, d))
        y = rng.standard_normal((n, d)) * 2.0 ** rng.integers(-6, 7, (n, d))
    elif law == "halfway_parallel":                                   # +-(1 + 2^-8) c: halfway between two bf16 values, rounded to even (down): every product loses the full rounding; y = x
        x = (rng.integers(0, 2, (n, d)) * 2 - 1) * 4.0 * (1.0 + 2.0 ** -8)
        y = x.copy()
    else:                                                             # all products of one sign (no cancellation in the sums), values halfway again
        x = np.abs(rng.standard_normal((n, d))) + 1.0
        x = bf16_rne(x.astype(np.float32)).astype(np.float64) * (1.0 + 2.0 ** -8)
        y = x[rng.permutation(n)]
    x, y = np.ascontiguousarray(x, np.float32), np.ascontiguousarray(y, np.float32)
    xh, yh = bf16_rne(x), bf16_rne(y)
    exact_operands = bool((xh == x).all() and (yh == y).all())
    assert exact_operands == (law == "u8")
    m = d * 2.0 ** -20 + (0.0 if exact_operands else 2.1 * 2.0 ** -8)
    chain = chain_f32(x, y).astype(np.float64)
    bound = m * (norms(x) + norms(y))
    for order in ("sequential", "blocks16", "pairwise"):
        for truncate in (False, True):
            acc = acc_pipe(xh, yh, order, truncate).astype(np.float64)
            err = 2.0 * np.abs(acc - chain)
            assert (err <= bound).all(), (law, d, order, truncate, float((err / bound).max()))
    # and the margin is not vacuous: on inexact data it is within a small factor of an error that does occur
    if law == "halfway_parallel":
        acc = acc_pipe(xh, yh, "sequential", False).astype(np.float64)
        assert (2.0 * np.abs(acc - chain) >= 0.8 * bound).any()
