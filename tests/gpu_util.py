"""Helpers for the -m gpu parity tests (HIP path vs the CPU oracle)."""
import json
import os

import torch

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
OUT = os.path.join(ROOT, "gpurun_out")


def report(name, **metrics):
    """Append one JSON line of measured errors to gpurun_out/parity.jsonl."""
    os.makedirs(OUT, exist_ok=True)
    clean = {k: (float(v) if isinstance(v, (int, float)) or hasattr(v, "item") else v)
             for k, v in metrics.items()}
    with open(os.path.join(OUT, "parity.jsonl"), "a") as f:
        f.write(json.dumps(dict(test=name, **clean)) + "\n")


def rel_err(a, b):
    """max |a-b| / max |b| (both moved to CPU float64)."""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def max_err(a, b):
    return (a.detach().cpu().double() - b.detach().cpu().double()).abs().max().item()


def cuda(x):
    return x.to("cuda").contiguous()
