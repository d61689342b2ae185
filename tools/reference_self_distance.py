"""How far does the reference restatement move against ITSELF when torch runs its float32 convolutions through the native
backend instead of oneDNN?  (CPU only; the scale the parity figures of DESIGN.md section 1 are read against.)
    python tools/reference_self_distance.py cfg3 cfg3_medium cfg3_large default_medium_320"""
import sys, time
sys.path.insert(0, __import__("os").getcwd())
import torch
from jarvis_hybridnet_amd import synthetic as S
from oracle import hybridnet_oracle as O
from tests import cases
for CASE in sys.argv[1:]:
    c = cases.PREDICTOR_CASES[CASE]; inp = cases.predictor_inputs(CASE)
    res = []
    for mk in (True, False):
        torch.backends.mkldnn.enabled = mk
        t0 = time.time()
        with torch.no_grad():
            rp, rc = O.predictor3d_forward(inp["sd_center"], inp["sd_hybrid"], inp["imgs"], inp["cam"], inp["intr"], inp["dist"],
                                           center_size=c["center_size"], bbox=c["bbox"], roi_cube_size=c["roi"],
                                           grid_spacing=c["spacing"], mean=S.MEAN, std=S.STD, chunk=5,
                                           center_model=c.get("size", "small"), kp_model=c.get("size", "small"))
        res.append(rp); print(CASE, "mkldnn" if mk else "native", round(time.time() - t0, 1), "s", flush=True)
    torch.backends.mkldnn.enabled = True
    d = (res[0] - res[1]).abs().max(dim=-1)[0][0]
    print(CASE, "reference oneDNN vs native convolutions: max %.3g mm, median %.3g mm" % (float(d.max()), float(d.median())))
