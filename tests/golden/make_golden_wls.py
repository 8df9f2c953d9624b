#!/usr/bin/env python3
"""Golden vectors of the centred-WLS scoring backend, produced by the
REFERENCE's own backend compiled where it lies:

    make -C oracle ref          # builds oracle/_ref/libwls_ref.so from rocco/native/wls_backend.c
    python tests/golden/make_golden_wls.py

Writes tests/golden/wls_vectors.npz: centred matrices and the six output tracks (+ degrees of freedom,
resolved window) of rocco_score_centered_wls_f64.  Only data is written -- no reference source.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libwls_ref.so"))
dp = ctypes.POINTER(ctypes.c_double)
fn = lib.rocco_score_centered_wls_f64
fn.argtypes = [dp, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_int,
               ctypes.c_int, ctypes.c_double, dp, dp, dp, dp, dp, dp, dp, ctypes.POINTER(ctypes.c_int)]
fn.restype = ctypes.c_int

rng = np.random.default_rng(4242)
out, names = {}, []
for n in (3, 4, 30, 31, 64, 257, 1000, 5000):
    for K in (1, 3):
        for variant in ("plain", "ties", "min_effect"):
            counts = rng.poisson(3.0, size=(K, n)).astype(np.float64)
            counts[:, rng.integers(0, n, size=max(1, n // 40))] += rng.integers(20, 200)
            m = np.log2(counts + 1.0)
            m = m - np.median(m, axis=1, keepdims=True)
            if variant != "ties":
                m = m + rng.normal(0.0, 0.05, size=m.shape)  # "ties" keeps the lattice of log2(count + 1) values
            m = np.ascontiguousarray(m)
            params = dict(lower_bound_z=1.0, prior_df=5.0, min_effect=0.25 if variant == "min_effect" else 0.0,
                          use_min_effect=1 if variant == "min_effect" else 0, spatial_window=31,
                          precision_floor_ratio=0.01)
            tracks = [np.empty(n) for _ in range(6)]
            df, win = ctypes.c_double(), ctypes.c_int()
            rc = fn(m.ctypes.data_as(dp), K, n, params["lower_bound_z"], params["prior_df"], params["min_effect"],
                    params["use_min_effect"], params["spatial_window"], params["precision_floor_ratio"],
                    *[t.ctypes.data_as(dp) for t in tracks], ctypes.byref(df), ctypes.byref(win))
            assert rc == 0
            name = f"n{n}_k{K}_{variant}"
            names.append(name)
            out[f"{name}_matrix"] = m
            out[f"{name}_params"] = np.array([params["lower_bound_z"], params["prior_df"], params["min_effect"],
                                              params["use_min_effect"], params["spatial_window"],
                                              params["precision_floor_ratio"]])
            out[f"{name}_tracks"] = np.stack(tracks)  # mean, raw, prior, moderated variance, se, scores
            out[f"{name}_df_window"] = np.array([df.value, win.value])
out["names"] = np.array(names)
np.savez_compressed(os.path.join(HERE, "wls_vectors.npz"), **out)
print(len(names), "cases written")
