#!/usr/bin/env python3
"""Round-2 verdict, item 7 (bounded experiment, nothing of it ships): "fp32 by split bf16" for
conv1d_2 - how exact is x.w when both operands are cut into three bf16 pieces (hi + mid + lo,
round-to-nearest-even each) and six of the nine partial products are kept (hh, hm, mh, hl, lh,
mm), accumulated in fp32 - against the fp64 oracle, beside plain fp32, for the direct convolution
and for the Winograd F(4,3) form the kernel uses (pieces taken of the TRANSFORMED operands).
CPU arithmetic (NumPy emulates the roundings; the order of the fp32 sums differs from the MFMA's,
which moves the last digit, not the picture).  Input: stage A's activations of the golden windows.
Usage: python tools/split_bf16_error.py"""
import json
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from deepbinner_amd.model_format import ModelWeights        # noqa: E402


def bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def split3(x):
    x = x.astype(np.float32)
    hi = bf16(x)
    mid = bf16(x - hi)
    lo = bf16(x - hi - mid)
    return hi, mid, lo


def gemm_split(a, b, products):
    """a [M,K] . b [K,N] from bf16 pieces, fp32 accumulation; products: pairs of piece indices."""
    pa, pb = split3(a), split3(b)
    out = np.zeros((a.shape[0], b.shape[1]), dtype=np.float32)
    for i, j in products:
        out += pa[i].astype(np.float32) @ pb[j].astype(np.float32)
    return out


SIX = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
THREE = [(0, 0), (0, 1), (1, 0)]


def main():
    gold = np.load(os.path.join(REPO, 'tests', 'golden', 'stages_EXP-NBD103_read_starts.npz'))
    w, _ = ModelWeights.load(os.path.join(REPO, 'deepbinner_amd', 'models',
                                          'EXP-NBD103_read_starts.dbw'))
    kernel, bias = w.convs[1]                   # conv1d_2: [3][48][48], [48]
    x = gold['A'][:4].astype(np.float64)        # [n][512][48]: conv2's input
    n, L, C = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (0, 0)))
    # im2col for the direct form: rows = positions, K = 3 taps x 48 channels
    a = np.concatenate([xp[:, t:t + L, :] for t in range(3)], axis=2).reshape(n * L, 3 * C)
    b = kernel.astype(np.float64).reshape(3 * C, -1)
    want = a @ b
    scale = float(np.abs(want).max())
    out = {'layer': 'conv1d_2 (48 -> 48, k = 3) on stage A of %d golden windows' % n,
           'output_scale_max_abs': scale}

    def err(y):
        return {'max_abs': float(np.abs(y - want).max()),
                'max_abs_over_scale': float(np.abs(y - want).max() / scale),
                'rms_over_scale': float(np.sqrt(np.mean((y - want) ** 2)) / scale)}

    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    out['direct, fp32'] = err((a32 @ b32).astype(np.float64))
    out['direct, bf16 x 3 pieces, 6 products'] = err(gemm_split(a32, b32, SIX).astype(np.float64))
    out['direct, bf16 x 2 pieces, 3 products'] = err(gemm_split(a32, b32, THREE).astype(np.float64))
    out['direct, bf16 x 1'] = err(gemm_split(a32, b32, [(0, 0)]).astype(np.float64))

    # Winograd F(4,3): U = B^T d (per quad), V = G g, M = U . V over the 48 channels, y = A^T M
    BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
                   [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
                  [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], dtype=np.float64)
    AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0],
                   [0, 1, -1, 8, -8, 1]], dtype=np.float64)
    quads = L // 4
    d = np.stack([xp[:, 4 * j:4 * j + 6, :] for j in range(quads)], axis=1)     # [n][q][6][C]
    V = np.einsum('xt,tio->xio', G, kernel.astype(np.float64))                  # [6][C][O]

    def wino(matmul):
        U = np.einsum('xr,nqrc->xnqc', BT, d.astype(np.float32).astype(np.float64))
        U32 = U.astype(np.float32)               # the kernel forms U in fp32
        M = np.stack([matmul(U32[xi].reshape(-1, C), V[xi].astype(np.float32))
                      for xi in range(6)]).astype(np.float64)                    # [6][n*q][O]
        y = np.einsum('rx,xmo->mro', AT, M)                                      # [n*q][4][O]
        return y.reshape(n, L, -1).reshape(n * L, -1)

    out['F(4,3), fp32'] = err(wino(lambda u, v: u @ v))
    out['F(4,3), bf16 x 3 pieces, 6 products'] = err(wino(lambda u, v: gemm_split(u, v, SIX)))
    out['F(4,3), bf16 x 2 pieces, 3 products'] = err(wino(lambda u, v: gemm_split(u, v, THREE)))
    out['note'] = ('tests/test_gpu_parity.py::test_stage_activations allows 2e-5 x scale per '
                   'stage against the fp64 oracle')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
