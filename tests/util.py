"""Helpers shared by the parity tests."""
import numpy as np
import torch


def check_summary(g, name, t, rtol=0.0, atol=0.0):
    """Compare tensor `t` with a golden entry written by make_golden.put():
    either the full tensor or strided samples + float64 checksums."""
    if name in g:
        ref = torch.from_numpy(g[name])
        assert tuple(ref.shape) == tuple(t.shape), (name, ref.shape, t.shape)
        if rtol == 0 and atol == 0:
            assert torch.equal(ref, t.to(ref.dtype)), name
        else:
            torch.testing.assert_close(t.to(ref.dtype), ref, rtol=rtol, atol=atol)
        return
    shape = tuple(int(v) for v in g[name + ".shape"])
    assert shape == tuple(t.shape), (name, shape, tuple(t.shape))
    flat = t.detach().double().reshape(-1).numpy()
    step = int(g[name + ".step"])
    sample = g[name + ".sample"].astype(np.float64)
    if rtol == 0 and atol == 0:
        assert np.array_equal(flat[::step], sample), name
        assert flat.sum() == float(g[name + ".sum"]), name
        assert np.abs(flat).sum() == float(g[name + ".abssum"]), name
    else:
        np.testing.assert_allclose(flat[::step], sample, rtol=rtol, atol=atol)
        scale = float(g[name + ".abssum"])
        assert abs(flat.sum() - float(g[name + ".sum"])) <= rtol * scale + atol * flat.size


def golden_indices(g, tag):
    """Full reference gather-index tensor (C,G,G,G) of a reprojection case, or
    None when only samples were stored."""
    key = tag + ".idx_delta16"
    if tag + ".idx" in g:
        return torch.from_numpy(g[tag + ".idx"]).long()
    if key not in g:
        return None
    return torch.from_numpy(np.cumsum(g[key].astype(np.int64), axis=-1))


def index_plane_hashes(idx):
    """Position-sensitive 64-bit hash of every camera's (G,G,G) gather-index plane:
    sum_i idx[i] * (i * 0x9E3779B97F4A7C15 + 1) mod 2^64.  A single wrong index anywhere changes
    its camera's hash; used where the full field is too large to commit (cfg5: 14 M indices)."""
    a = np.ascontiguousarray(idx.detach().cpu().numpy()).astype(np.uint64)
    a = a.reshape(a.shape[0], -1)
    with np.errstate(over="ignore"):
        w = np.arange(a.shape[1], dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1)
        return [int((row * w).sum(dtype=np.uint64)) for row in a]


def same_cpu_as_golden():
    """True when this machine's CPU model is the one the fixtures were made on
    (torch's CPU kernels, hence the oracle's last bits, depend on the ISA)."""
    import json
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    env = json.load(open(os.path.join(here, "golden", "environment.json")))
    model = "unknown"
    for line in open("/proc/cpuinfo"):
        if line.startswith("model name"):
            model = line.split(":", 1)[1].strip()
            break
    return model == env["cpu"]
