#!/usr/bin/env python3
"""Golden vectors of the cross-fit Whittaker baseline, produced by the
REFERENCE's own backend compiled where it lies:

    make -C oracle ref          # builds oracle/_ref/libbaseline_ref.so from rocco/native/baseline_backend.c
    python tests/golden/make_golden_baseline.py

Writes tests/golden/baseline_vectors.npz: inputs (counts-like and centred matrices, penalties from the
reference's block-size rule, rocco/inference.py:65-76) and the outputs of
rocco_crossfit_whittaker_baseline_matrix_f64.  Only data is written -- no reference source.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libbaseline_ref.so"))
dp = ctypes.POINTER(ctypes.c_double)
fn = lib.rocco_crossfit_whittaker_baseline_matrix_f64
fn.argtypes = [dp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, dp]
fn.restype = ctypes.c_int


def whittaker_lambda(block):  # rocco/inference.py:65-76 (arithmetic restated)
    block = int(max(3, block))
    if block % 2 == 0:
        block += 1
    w_hat = float(block) * 0.15915494
    return float(7.0 * (w_hat**4))


rng = np.random.default_rng(2024)
out = {}
names = []
for n in (24, 25, 26, 40, 101, 257, 1000, 5000):
    for block in ((101,) if n >= 5000 else (3, 25, 101)):
        for K in (1, 3):
            lam = whittaker_lambda(min(block, n))
            # log2(count + 1) minus the column mean, with exact zeros and a few large spikes: the shape
            # of what rocco/inference.py:302-340 hands to the baseline
            counts = rng.poisson(3.0, size=(K, n)).astype(np.float64)
            counts[:, rng.integers(0, n, size=max(1, n // 50))] += rng.integers(20, 200)
            m = np.log2(counts + 1.0)
            m = np.ascontiguousarray(m - m.mean(axis=0, keepdims=True))
            m[rng.random(m.shape) < 0.03] = 0.0
            res = np.empty_like(m)
            rc = fn(m.ctypes.data_as(dp), K, n, lam, res.ctypes.data_as(dp))
            assert rc == 0
            name = f"n{n}_b{block}_k{K}"
            names.append(name)
            out[f"{name}_matrix"], out[f"{name}_lambda"], out[f"{name}_baseline"] = m, np.float64(lam), res
out["names"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "baseline_vectors.npz"), **out)
print(len(names), "cases written")
