"""Counter-based synthetic weights, noise buffers and inputs.

No pretrained CosyVoice3 checkpoint is reachable offline, so every parity and
benchmark run uses weights drawn from this generator (SURVEY §8c "Synthetic
weights").  Each element is a pure function of (tensor name, flat index), so

* the reference modules (tests/golden/mint_goldens.py), the CPU oracle
  (oracle/) and the HIP engine are filled with identical values without
  shipping 859 M parameters;
* any prefix / row subset of a huge tensor (SineGen2's 7.2 M x 9 noise table,
  the 151 936-row text embedding) can be produced without the rest.

Matrix-shaped weights are by default rounded to bf16-representable fp32 values: the
engine stores them as bf16 without loss, so engine-vs-oracle differences measure the
kernels, not a quantisation step.  A real checkpoint (`llm.pt` is an fp32 state dict,
cli/model.py:65-73) is NOT bf16-representable: `unrounded_weights()` switches the
rounding off (and gives weight-norm g a general scale), which is what the `*_fp32w`
fixtures and the engine's exact-weights mode are held to.
"""
from __future__ import annotations

import re
from typing import Dict, Iterable, Optional, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLD = np.uint64(0x9E3779B97F4A7C15)
_C1 = np.uint64(0xBF58476D1CE4E5B9)
_C2 = np.uint64(0x94D049BB133111EB)

_ROUND_BF16 = True          # matrix-shaped weights are rounded to bf16-representable values (see unrounded_weights)


def weight_rounding() -> bool:
    return _ROUND_BF16


def set_weight_rounding(on: bool) -> bool:
    """Switch the bf16 rounding of matrix-shaped weights; returns the previous setting."""
    global _ROUND_BF16
    prev, _ROUND_BF16 = _ROUND_BF16, bool(on)
    return prev


class unrounded_weights:
    """`with synth.unrounded_weights():` - general fp32 weights, as a real checkpoint holds them."""

    def __enter__(self):
        self.prev = set_weight_rounding(False)
        return self

    def __exit__(self, *exc):
        set_weight_rounding(self.prev)
        return False


def name_seed(name: str) -> int:
    """FNV-1a 64-bit of the tensor name."""
    h = 0xCBF29CE484222325
    for b in name.encode():
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def _hash(seed: int, idx: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser of seed + (idx+1)*golden; idx uint64 array."""
    with np.errstate(over="ignore"):
        z = (idx + np.uint64(1)) * _GOLD + np.uint64(seed)
        z = (z ^ (z >> np.uint64(30))) * _C1
        z = (z ^ (z >> np.uint64(27))) * _C2
        z = z ^ (z >> np.uint64(31))
    return z


def u01(name: str, n: int, start: int = 0) -> np.ndarray:
    """n uniform [0,1) float64 values for flat indices start..start+n."""
    seed = name_seed(name)
    out = np.empty(n, dtype=np.float64)
    step = 1 << 24
    for s in range(0, n, step):
        e = min(n, s + step)
        idx = np.arange(start + s, start + e, dtype=np.uint64)
        out[s:e] = (_hash(seed, idx) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))
    return out


def uniform(name: str, shape, lo: float, hi: float, start: int = 0) -> np.ndarray:
    n = int(np.prod(shape))
    return (lo + (hi - lo) * u01(name, n, start)).astype(np.float32).reshape(shape)


def normal(name: str, shape, mean: float = 0.0, std: float = 1.0, start: int = 0) -> np.ndarray:
    """Box-Muller on the (2i, 2i+1) uniforms of element i."""
    n = int(np.prod(shape))
    u = u01(name, 2 * n, 2 * start).reshape(n, 2)
    r = np.sqrt(-2.0 * np.log(1.0 - u[:, 0]))
    z = r * np.cos(2.0 * np.pi * u[:, 1])
    return (mean + std * z).astype(np.float32).reshape(shape)


def randint(name: str, shape, lo: int, hi: int) -> np.ndarray:
    """Integers in [lo, hi)."""
    n = int(np.prod(shape))
    return (lo + np.floor(u01(name, n) * (hi - lo))).astype(np.int32).reshape(shape)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round fp32 to the nearest bf16-representable fp32 (ties to even)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)
    return r.view(np.float32).reshape(x.shape)


def to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """fp32 -> uint16 bf16 bit patterns (round to nearest even)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = (u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)
    return r.astype(np.uint16).reshape(x.shape)


# (regex, kind, param).  First match wins.  kinds:
#   w      : uniform(-a, a), a = gain*sqrt(3/fan_in), rounded to bf16
#   b      : uniform(-p, p)
#   one    : 1 + uniform(-p, p)
#   range  : uniform(p[0], p[1])
#   emb    : uniform(-p, p) rounded to bf16
_RULES = [
    # ---- HiFT
    (r"activations[12]\.\d+\.alpha$", "range", (0.5, 1.5)),
    (r"^conv_post\.bias$", "postbias", None),
    (r"^conv_post\..*original1$", "w", 0.5),
    (r"^(source_)?resblocks\.\d+\.convs2\..*original1$", "w", 0.35),
    (r"^(source_)?resblocks\.\d+\.convs1\..*original1$", "w", 0.7),
    (r"^source_downs\.\d+\.weight$", "w", 1.0),
    (r"^f0_predictor\.classifier\.weight$", "wf32", 60.0),
    (r"^f0_predictor\.classifier\.bias$", "range", (40.0, 60.0)),
    (r"^f0_predictor\..*original1$", "wf32", 1.3),
    (r"^m_source\.l_linear\.weight$", "wf32", 3.0),
    # ---- LLM
    (r"^llm_decoder\.weight$", "w", 4.0),
    (r"embed_tokens\.weight$|^speech_embedding\.weight$|lm_head\.weight$", "emb", 0.2),
    (r"layernorm\.weight$|model\.norm\.weight$", "one", 0.1),
    # ---- flow
    (r"^input_embedding\.weight$", "emb", 1.0),
    (r"attn_norm\.linear\.weight$|norm_out\.linear\.weight$", "w", 1.0),
    (r"proj_out\.weight$", "w", 1.0),
    # ---- generic
    (r"original1$|\.weight$", "w", 1.0),
    (r"\.bias$", "b", 0.05),
]
_RULES = [(re.compile(p), k, a) for p, k, a in _RULES]


def tensor(name: str, shape: Tuple[int, ...], rows: Optional[Iterable[int]] = None) -> np.ndarray:
    """Synthetic fp32 value of a named weight (not for weight-norm g, see state_dict)."""
    shape = tuple(int(s) for s in shape)
    for rx, kind, arg in _RULES:
        if rx.search(name):
            break
    else:
        raise KeyError(f"no synth rule for {name}")
    if rows is not None:
        rows = list(rows)
        row_elems = int(np.prod(shape[1:])) if len(shape) > 1 else 1
        parts = [_gen(name, kind, arg, (1,) + shape[1:], shape, start=r * row_elems) for r in rows]
        return np.concatenate(parts, axis=0)
    return _gen(name, kind, arg, shape, shape, 0)


def _gen(name, kind, arg, shape, full_shape, start):
    if kind in ("w", "wf32"):
        fan_in = int(np.prod(full_shape[1:])) if len(full_shape) > 1 else full_shape[0]
        a = float(arg) * np.sqrt(3.0 / fan_in)
        x = uniform(name, shape, -a, a, start)
        return bf16_round(x) if kind == "w" and _ROUND_BF16 else x
    if kind == "emb":
        x = uniform(name, shape, -arg, arg, start)
        return bf16_round(x) if _ROUND_BF16 else x
    if kind == "b":
        return uniform(name, shape, -arg, arg, start)
    if kind == "one":
        return (1.0 + uniform(name, shape, -arg, arg, start)).astype(np.float32)
    if kind == "range":
        return uniform(name, shape, arg[0], arg[1], start)
    if kind == "postbias":
        # magnitude channels (first n_fft/2+1) biased negative so exp() stays
        # well below the 1e2 clip and the waveform is not pinned at +-0.99
        x = uniform(name, shape, -0.05, 0.05, start)
        half = full_shape[0] // 2
        x[:half] -= 1.0
        return x
    raise ValueError(kind)


def weight_norm_g(v: np.ndarray) -> np.ndarray:
    """g for a weight-normed conv: per-output-channel ||v|| times a power of two
    (1 or 1/2), so g*v/||v|| is v or v/2 up to one rounding: the folded weight
    stays bf16-representable and the fold is still exercised."""
    co = v.shape[0]
    nrm = np.sqrt((v.astype(np.float64).reshape(co, -1) ** 2).sum(axis=1))
    s = np.where(np.arange(co) % 3 == 0, 0.5, 1.0)
    if not _ROUND_BF16:
        s = s * (0.8 + 0.4 * ((np.arange(co) * 7) % 11) / 10.0)        # a general scale: the fold is not a power of two
    return (nrm * s).astype(np.float32).reshape(co, 1, 1)


def state_dict(manifest: Dict[str, Tuple[int, ...]], skip: Iterable[str] = ()) -> Dict[str, np.ndarray]:
    """All tensors of a manifest (spec.py) as fp32 numpy arrays."""
    skip = tuple(skip)
    out: Dict[str, np.ndarray] = {}
    for name, shape in manifest.items():
        if any(s in name for s in skip):
            continue
        if name.endswith("parametrizations.weight.original0"):
            continue
        out[name] = tensor(name, shape)
    for name in manifest:
        if name.endswith("parametrizations.weight.original0") and not any(s in name for s in skip):
            out[name] = weight_norm_g(out[name[:-1] + "1"])
    if "llm.model.lm_head.weight" in out and "llm.model.model.embed_tokens.weight" in out:
        out["llm.model.lm_head.weight"] = out["llm.model.model.embed_tokens.weight"]  # tied
    return out


# ---- non-checkpointed noise buffers of the reference -------

def flow_rand_noise(n_frames: int, mel: int = 80, total: int = 15000) -> np.ndarray:
    """Stand-in for CausalConditionalCFM.rand_noise[:, :, :n_frames] (1, mel, n).
    Laid out (mel, total) so a frame prefix of each row is a contiguous index range."""
    rows = [normal("flow.rand_noise", (n_frames,), start=c * total) for c in range(mel)]
    return np.stack(rows, axis=0)[None]


def hift_rand_ini(harm: int = 9) -> np.ndarray:
    x = uniform("hift.rand_ini", (1, harm), 0.0, 1.0)
    x[:, 0] = 0.0
    return x


def hift_sine_noise(n_samples: int, harm: int = 9) -> np.ndarray:
    """Stand-in for SineGen2.sine_waves[:, :n_samples] (1, n, harm), uniform [0,1)."""
    return uniform("hift.sine_waves", (1, n_samples, harm), 0.0, 1.0)


# ---- the same generator on torch (GPU when given): bit-identical values, ~100x faster for the
# ---- 0.86 G parameters of the full model.  tests/test_oracle_golden.py::test_synth_torch_matches_numpy
def _s64(c: int) -> int:
    return c - (1 << 64) if c >= (1 << 63) else c


def _lsr(z, k: int):
    return (z >> k) & ((1 << (64 - k)) - 1)


def _u01_torch(name: str, n: int, device):
    import torch
    seed = _s64(name_seed(name))
    out = torch.empty(n, dtype=torch.float64, device=device)
    step = 1 << 24
    for s in range(0, n, step):
        e = min(n, s + step)
        idx = torch.arange(s, e, dtype=torch.int64, device=device)
        z = (idx + 1) * _s64(0x9E3779B97F4A7C15) + seed
        z = (z ^ _lsr(z, 30)) * _s64(0xBF58476D1CE4E5B9)
        z = (z ^ _lsr(z, 27)) * _s64(0x94D049BB133111EB)
        z = z ^ _lsr(z, 31)
        out[s:e] = _lsr(z, 11).to(torch.float64) * (1.0 / (1 << 53))
    return out


def _bf16_round_torch(x):
    import torch
    u = x.contiguous().view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    r = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return r.to(torch.int32).view(torch.float32)


_LIB_FILL_MIN = 1 << 16        # tensors from this size on are filled by the library's kernel on a GPU device; smaller ones by numpy + a copy


def _tensor_device(name: str, shape, device):
    """`tensor` on a GPU without a torch kernel: the library's counter-based fill (csrc/runtime.hip:synth_uniform_k, bit-identical
    to the numpy form) for large tensors, numpy + one host-to-device copy for small ones.  None when the library is not there."""
    import torch
    shape = tuple(int(s) for s in shape)
    n = int(np.prod(shape))
    for rx, kind, arg in _RULES:
        if rx.search(name):
            break
    else:
        raise KeyError(f"no synth rule for {name}")
    if n < _LIB_FILL_MIN or kind == "postbias":
        return torch.from_numpy(tensor(name, shape)).to(device)
    import ctypes as C
    from . import _lib
    if kind in ("w", "wf32"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = float(arg) * np.sqrt(3.0 / fan_in)
        lo, hi, flags = -a, a, (1 if kind == "w" and _ROUND_BF16 else 0)
    elif kind == "emb":
        lo, hi, flags = -arg, arg, (1 if _ROUND_BF16 else 0)
    elif kind == "b":
        lo, hi, flags = -arg, arg, 0
    elif kind == "one":
        lo, hi, flags = -arg, arg, 2
    elif kind == "range":
        lo, hi, flags = arg[0], arg[1], 0
    else:
        raise ValueError(kind)
    out = torch.empty(shape, dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        _lib.check(_lib.lib().fy_synth_uniform(out.data_ptr(), n, name_seed(name), 0, float(lo), float(hi), flags,
                                               C.c_void_p(torch.cuda.current_stream(out.device).cuda_stream)))
    return out


def tensor_torch(name: str, shape, device="cpu"):
    """torch twin of `tensor` (whole tensors only)."""
    import torch
    shape = tuple(int(s) for s in shape)
    for rx, kind, arg in _RULES:
        if rx.search(name):
            break
    else:
        raise KeyError(f"no synth rule for {name}")
    n = int(np.prod(shape))

    def uni(lo, hi):
        return (lo + (hi - lo) * _u01_torch(name, n, device)).to(torch.float32).reshape(shape)
    if kind in ("w", "wf32"):
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        a = float(arg) * np.sqrt(3.0 / fan_in)
        x = uni(-a, a)
        return _bf16_round_torch(x).reshape(shape) if kind == "w" and _ROUND_BF16 else x
    if kind == "emb":
        return _bf16_round_torch(uni(-arg, arg)).reshape(shape) if _ROUND_BF16 else uni(-arg, arg)
    if kind == "b":
        return uni(-arg, arg)
    if kind == "one":
        return (1.0 + uni(-arg, arg)).to(torch.float32)
    if kind == "range":
        return uni(arg[0], arg[1])
    if kind == "postbias":
        x = uni(-0.05, 0.05)
        x[: shape[0] // 2] -= 1.0
        return x
    raise ValueError(kind)


def _have_lib() -> bool:
    import os
    from . import _lib
    return os.path.exists(_lib.LIB_PATH)


def state_dict_torch(manifest, device="cpu", skip: Iterable[str] = ()):
    """torch twin of `state_dict`: name -> fp32 tensor on `device`."""
    import torch
    skip = tuple(skip)
    out = {}
    on_gpu = torch.device(device).type == "cuda" and _have_lib()
    for name, shape in manifest.items():
        if any(s in name for s in skip) or name.endswith("parametrizations.weight.original0"):
            continue
        if on_gpu and (name[:-1] + "0") in manifest and name.endswith("parametrizations.weight.original1"):
            continue                                   # a weight-normed pair: below, on the host (g needs the norm of v)
        out[name] = _tensor_device(name, shape, device) if on_gpu else tensor_torch(name, shape, device)
    for name in manifest:
        if name.endswith("parametrizations.weight.original0") and not any(s in name for s in skip):
            if on_gpu:
                v = tensor(name[:-1] + "1", manifest[name[:-1] + "1"])
                out[name[:-1] + "1"] = torch.from_numpy(v).to(device)
                out[name] = torch.from_numpy(weight_norm_g(v)).to(device)
                continue
            v = out[name[:-1] + "1"]
            co = v.shape[0]
            nrm = torch.sqrt((v.to(torch.float64).reshape(co, -1) ** 2).sum(dim=1))
            s = torch.where(torch.arange(co, device=v.device) % 3 == 0, 0.5, 1.0).to(torch.float64)
            if not _ROUND_BF16:
                s = s * (0.8 + 0.4 * ((torch.arange(co, device=v.device) * 7) % 11).to(torch.float64) / 10.0)
            out[name] = (nrm * s).to(torch.float32).reshape(co, 1, 1)
    if "llm.model.lm_head.weight" in out and "llm.model.model.embed_tokens.weight" in out:
        out["llm.model.lm_head.weight"] = out["llm.model.model.embed_tokens.weight"]
    return out
