"""Closed-form ("procedural") tensors shared by the golden-vector generator and the tests.

The goldens under ``tests/golden/`` store only small inputs and the reference's
outputs.  Weights are regenerated on every box from (name, shape) with the rules
below, so no large file is committed and nothing of the reference travels.
numpy's PCG64 ``Generator`` stream is what both sides draw from; a float64
checksum of a few tensors is stored in each golden file to catch a numpy whose
stream differs.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np


def _amp_offset(name: str, shape: Tuple[int, ...]) -> Tuple[float, float]:
    leaf = name.rsplit(".", 1)[-1]
    lname = name.lower()
    is_norm = ("norm" in lname) or (".ln" in lname) or lname.startswith("ln") or ("layernorm" in lname)
    if is_norm and leaf == "weight":
        return 0.10, 1.0
    if is_norm and leaf == "bias":
        return 0.05, 0.0
    if leaf == "bias" or leaf.endswith("_bias"):
        return 0.02, 0.0
    if "embedding" in lname or leaf in ("class_token", "cls_token", "pos_embedding"):
        return 0.05, 0.0
    if ("W_query" in name) or ("W_key" in name):
        # head projections see L2-normalised inputs (|x| ~ 1/sqrt(D)); a large
        # amplitude keeps the 16x16 softmax far from uniform so the test bites.
        return 3.0, 0.0
    if "W_value" in name:
        return 1.5, 0.0
    if len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        return 1.0 / np.sqrt(fan_in), 0.0
    return 0.02, 0.0


def proc_tensor(name: str, shape: Iterable[int]) -> np.ndarray:
    """float32 tensor that depends only on (name, shape)."""
    shape = tuple(int(s) for s in shape)
    amp, off = _amp_offset(name, shape)
    rng = np.random.default_rng(zlib.crc32(name.encode("utf-8")))
    return np.asarray(rng.standard_normal(shape, dtype=np.float32) * np.float32(amp) + np.float32(off),
                      dtype=np.float32).reshape(shape)


def proc_state(shapes: Dict[str, Tuple[int, ...]]) -> Dict[str, np.ndarray]:
    return {k: proc_tensor(k, v) for k, v in shapes.items()}


def proc_input(tag: str, shape: Iterable[int], scale: float = 1.0) -> np.ndarray:
    rng = np.random.default_rng(zlib.crc32(("input:" + tag).encode("utf-8")))
    return (rng.standard_normal(tuple(shape), dtype=np.float32) * np.float32(scale)).astype(np.float32)


def checksum(arrs: Iterable[np.ndarray]) -> float:
    s = 0.0
    for a in arrs:
        a64 = np.asarray(a, dtype=np.float64).ravel()
        s += float(np.dot(a64, np.cos(np.arange(a64.size, dtype=np.float64) * 0.001)))
    return s


def synth_captions(n: int, seq_len: int, seed: int, vocab_lo: int = 1000, vocab_hi: int = 30522,
                   cls_id: int = 101, sep_id: int = 102, pad_id: int = 0, min_len: int = 8):
    """Synthetic captions of SURVEY.md section 8(d): [CLS] ids... [SEP] pad, lengths U[min_len, seq_len]."""
    rng = np.random.default_rng(seed)
    ids = np.full((n, seq_len), pad_id, dtype=np.int64)
    mask = np.zeros((n, seq_len), dtype=np.int64)
    lens = rng.integers(min(min_len, seq_len), seq_len + 1, size=n)
    for i, L in enumerate(lens):
        ids[i, 0] = cls_id
        if L > 2:
            ids[i, 1:L - 1] = rng.integers(vocab_lo, vocab_hi, size=L - 2)
        ids[i, L - 1] = sep_id
        mask[i, :L] = 1
    return ids, mask


def counter_uniform(seed: int, idx: np.ndarray) -> np.ndarray:
    """Host mirror of the device's counter-based uniform (csrc/common.h::mmrca_uniform, splitmix64): lets tests rebuild
    the exact dropout masks of the kernels."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + np.uint64(0x9E3779B97F4A7C15) * (idx + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / 16777216.0)
