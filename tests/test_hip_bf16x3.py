"""GPU: the labelled reduced-precision mode `bf16x3` (jh_set_precision; BASELINE configs[1] is
worded "bf16", the reference's own fast path is half precision: jarvis3D.py:93,107,122).  V2V's
3x3x3 stride-1 convolutions run on the bf16 matrix cores with every fp32 operand split into two
bf16 terms (csrc/conv3d_bf16x3.hip).  The fp32 mode stays the default and the parity mode; this
file measures what the split costs in accuracy and holds it to the SAME bars: 3D keypoints within
1e-3 mm of the reference fixtures, integer paths exact."""
import pytest
import torch

from tests import cases
from tests.gpu_util import cuda, max_err, rel_err, report

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16x3():
    from jarvis_hybridnet_amd import _native as N
    prev = N.set_precision("bf16x3")
    yield N
    N.set_precision(prev)


@pytest.mark.parametrize("cin,cout,D,H,W,norm_act", [(46, 46, 16, 16, 16, 1), (92, 92, 8, 12, 20, 1),
                                                     (6, 23, 5, 9, 11, -1), (60, 60, 12, 12, 48, 1),
                                                     (120, 120, 6, 10, 14, 1), (46, 46, 32, 32, 32, -1),
                                                     (80, 80, 6, 8, 18, 1), (100, 100, 5, 6, 16, 1),
                                                     (72, 72, 4, 8, 16, -1), (136, 136, 4, 4, 16, 1)])
def test_conv3d_bf16x3_vs_torch(cin, cout, D, H, W, norm_act, bf16x3):
    """One 3x3x3 stride-1 convolution (+ the fused InstanceNorm + ReLU of the network) against torch
    fp32 on the CPU, incl. volumes that are not multiples of the 4 x 4 x 16 tile, channel counts
    that need two column-block groups, and counts whose 16-channel blocks do not fill the groups (80 and 72 -> 5
    blocks as 3 x 2, 100 -> 7 as 4 x 2, 136 -> 9 as 3 x 3: V2V has 2J and 4J channels, any J must build)."""
    from tests.test_hip_ops import _conv
    g = torch.Generator().manual_seed(cin + D)
    x = torch.randn(2, cin, D, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y, ref = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=norm_act)
    e = rel_err(y, ref)
    bf16x3.set_precision("f32")
    y32, _ = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=norm_act)
    bf16x3.set_precision("bf16x3")
    e32 = rel_err(y32, ref)
    report("conv3d_bf16x3", cin=cin, cout=cout, shape=[D, H, W], rel=e, rel_fp32_kernel=e32)
    assert not torch.equal(y, y32), "the precision mode had no effect"
    # 3-term split: ~2^-16 per product, averaged over K = 27 cin terms (plain bf16 would sit at ~3e-3)
    assert e < 5e-5


XCONV = [  # nd, k, stride, cin, cout, spatial, norm_act   (the layer shapes of the three model sizes + ragged ones)
    (3, 3, 2, 23, 46, (24, 24, 24), 1), (3, 3, 2, 30, 60, (10, 14, 36), -1), (3, 3, 2, 12, 24, (8, 8, 8), 1),
    (2, 3, 1, 16, 16, (32, 48), -1), (2, 3, 1, 40, 240, (16, 16), 2), (2, 3, 1, 32, 32, (20, 36), 2),
    (2, 3, 2, 8, 48, (40, 40), 2), (2, 3, 2, 24, 144, (32, 32), 2), (2, 3, 2, 16, 96, (18, 34), -1),
    (2, 5, 2, 16, 96, (32, 32), -1), (2, 5, 2, 24, 144, (20, 28), 2), (2, 5, 2, 40, 240, (16, 16), 2),
]


@pytest.mark.parametrize("nd,k,stride,cin,cout,shape,norm_act", XCONV)
def test_conv_bf16x3_generic_vs_torch(nd, k, stride, cin, cout, shape, norm_act, bf16x3):
    """The generic split-bf16 convolution (csrc/conv_bf16x3.h: 3D k3 s2, 2D k3 s1 / k3 s2 / k5 s2; stride 2
    reads a de-interleaved patch) against torch fp32, incl. sizes that are not multiples of the tile."""
    from tests.test_hip_ops import _conv
    bf16x3.set_precision("bf16x3_wide")          # (the 2D forms belong to the wide level; the fixture restores)
    g = torch.Generator().manual_seed(k * 10 + cin + shape[-1])
    x = torch.randn(2, cin, *shape, generator=g)
    w = torch.randn(cout, cin, *([k] * nd), generator=g) / (cin * k ** nd) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1 if nd == 3 else None
    y, ref = _conv(nd, 0, k, stride, k // 2, cin, cout, x, w, b, norm_act=norm_act)
    e = rel_err(y, ref)
    bf16x3.set_precision("f32")
    y32, _ = _conv(nd, 0, k, stride, k // 2, cin, cout, x, w, b, norm_act=norm_act)
    report("conv_bf16x3_generic", nd=nd, k=k, stride=stride, cin=cin, cout=cout, rel=e, rel_fp32_kernel=rel_err(y32, ref))
    assert not torch.equal(y, y32), "the precision mode had no effect"
    assert e < 5e-5


@pytest.mark.parametrize("cin,cout,H,W,n", [(64, 23, 64, 64, 12), (64, 30, 20, 24, 2), (88, 23, 12, 12, 2),
                                               (64, 12, 9, 21, 3), (160, 23, 8, 16, 2)])
def test_deconv2d_bf16x3_vs_torch(cin, cout, H, W, n, bf16x3):
    """The keypoint head's ConvTranspose2d(k4, s2, p1) (no bias, no norm after it) in bf16x3 mode
    (csrc/deconv4_bf16x3.hip: one parity per wave from one staged patch) against torch; asymmetric
    random weights catch tap / parity swaps, ragged sizes the tile edges."""
    from tests.test_hip_ops import _conv
    g = torch.Generator().manual_seed(8 + cin + H)
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5
    y, ref = _conv(2, 1, 4, 2, 1, cin, cout, x, w, None, norm_act=-1)
    e = rel_err(y, ref)
    bf16x3.set_precision("f32")
    y32, _ = _conv(2, 1, 4, 2, 1, cin, cout, x, w, None, norm_act=-1)
    bf16x3.set_precision("bf16x3")
    report("deconv2d_bf16x3", cin=cin, cout=cout, h=H, w=W, rel=e, rel_fp32_kernel=rel_err(y32, ref))
    assert not torch.equal(y, y32), "the precision mode had no effect"
    assert e < 5e-5


@pytest.mark.parametrize("tag", ["j23_g48", "j23_g64"])
def test_v2v_bf16x3(tag, golden, bf16x3):
    """V2VNet + soft-argmax tail in bf16x3 mode against the oracle and the reference's golden points."""
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd.hybridnet.v2vnet import V2VNet
    from oracle import hybridnet_oracle as O
    N = bf16x3
    J, G, wseed, xseed = cases.V2V_CASES[tag]
    sd = S.v2v_weights(J, wseed)
    x = cases.v2v_input(J, G, xseed)
    center = torch.tensor([[35, -58, 549]], dtype=torch.int32)
    with torch.no_grad():
        ref = O.v2v_forward(sd, x)
        _, rpts, rconf = O.softargmax_tail(ref, center, G * 2, 2)
    net = V2VNet(J, J)
    net.load_state_dict(sd, strict=True)
    out = net(cuda(x))
    Gh = G // 2
    pts = torch.empty((1, J, 3), device="cuda")
    conf = torch.empty((1, J), device="cuda")
    ws = N.workspace(N.lib().jh_softargmax_workspace_bytes(1, J, Gh), "cuda")
    N.check(N.lib().jh_softargmax(out.data_ptr(), 1, J, Gh, 2.0, float(G * 2), cuda(center).data_ptr(), None,
                                  pts.data_ptr(), conf.data_ptr(), ws.data_ptr(), ws.numel(), N.stream()))
    torch.cuda.synchronize()
    e, ep, ec = rel_err(out, ref), max_err(pts, rpts), max_err(conf, rconf)
    eg = float((pts.cpu() - torch.from_numpy(golden("v2v")[tag + ".points"])).abs().max())
    report("v2v_bf16x3", tag=tag, rel=e, points_mm=ep, conf=ec, points_mm_vs_fixture=eg)
    assert e < 2e-4 and ep < 1e-3 and eg < 1e-3 and ec < 1e-4


@pytest.mark.parametrize("J", [17, 20, 25])
def test_v2v_bf16x3_any_joint_count(J, bf16x3):
    """Joint counts whose 4J channels are 5 or 7 sixteen-channel blocks (J = 17, 20 -> 68, 80; J = 25 -> 100): the
    split-bf16 3x3x3 kernel pads the last column-block group instead of refusing the layer."""
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd.hybridnet.v2vnet import V2VNet
    from oracle import hybridnet_oracle as O
    sd = S.v2v_weights(J, 90 + J)
    x = cases.v2v_input(J, 24, 91 + J)
    with torch.no_grad():
        ref = O.v2v_forward(sd, x)
    net = V2VNet(J, J)
    net.load_state_dict(sd, strict=True)
    out = net(cuda(x))
    torch.cuda.synchronize()
    e = rel_err(out, ref)
    report("v2v_bf16x3_any_J", J=J, rel=e)
    assert e < 2e-4


@pytest.mark.parametrize("tag", ["cfg2", "cfg3", "cfg5"])
def test_predictor3d_bf16x3(tag, golden, bf16x3):
    """JarvisPredictor3D.forward in bf16x3 mode vs the imported reference's output (fixtures): the
    integer path (centre argmax, crop centres, truncated centre) bit-exact, 3D keypoints within the
    north-star bar of 1e-3 mm; the measured delta is what the bench's bf16x3 line quotes."""
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    from tests.test_hip_predictor import make_cfg
    c = cases.PREDICTOR_CASES[tag]
    inp = cases.predictor_inputs(tag)
    pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
    calib = (cuda(inp["cam"]), cuda(inp["intr"]), cuda(inp["dist"]))
    pts, conf = pred(cuda(inp["imgs"]), *calib)
    torch.cuda.synchronize()
    g = golden("predictor")
    dbg = pred.native(c["H"], c["W"]).debug("cuda")
    assert torch.equal(dbg["det"][0, :, :2].cpu().long(), torch.from_numpy(g[tag + ".preds"]).reshape(c["C"], 2))
    assert torch.equal(dbg["center_hm"][0].cpu(), torch.from_numpy(g[tag + ".center_hm"]))
    assert torch.equal(dbg["center3d_int"][0].cpu(), torch.from_numpy(g[tag + ".center3d"]).int())
    ep = max_err(pts, torch.from_numpy(g[tag + ".points3D"]))
    ec = max_err(conf, torch.from_numpy(g[tag + ".confidences"]))
    bf16x3.set_precision("f32")
    p32, _ = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])(
        cuda(inp["imgs"]), *calib)
    bf16x3.set_precision("bf16x3")
    torch.cuda.synchronize()
    report("predictor3d_bf16x3", tag=tag, points_mm=ep, conf=ec, points_mm_vs_fp32_mode=max_err(pts, p32))
    assert not torch.equal(pts, p32), "the precision mode had no effect"
    assert ep < 1e-3, "3D keypoints must be within 1e-3 mm of the reference"
    assert ec < 1e-4


def test_bf16x3_seed_sweep_vs_f32_mode(bf16x3):
    """64 seeds of the cfg2 rig: bf16x3 against the fp32 mode of the same library (which the 64-seed
    sweep of tests/test_hip_predictor.py ties to the oracle).  The two modes share everything up to the
    V2V input, so validity, centre argmax, crop centres and the truncated 3D centre are the SAME kernels'
    outputs (asserted equal); what the split changes is the V2V output, i.e. the 3D keypoints."""
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import NativePredictor
    N = bf16x3
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    calib = (inp["cam"], inp["intr"], inp["dist"])
    T = 16
    kw = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center_size"], bbox=c["bbox"],
              roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
              mean=S.MEAN, std=S.STD, time_batch=T)
    dev = [cuda(t) for t in calib]
    pb = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
    N.set_precision("f32")
    pf = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
    N.set_precision("bf16x3")
    pb.set_calibration(*dev)
    pf.set_calibration(*dev)
    worst_p, worst_c, nvalid = 0.0, 0.0, 0
    for s0 in range(1000, 1064, T):
        frames = cuda(torch.stack([S.blob_frames(calib, c["W"], c["H"], c["J"], s)[0] for s in range(s0, s0 + T)]))
        ob = [t.clone() for t in pb.forward(frames)]
        db = {k: v.clone() for k, v in pb.debug("cuda").items()}
        of = [t.clone() for t in pf.forward(frames)]
        df = pf.debug("cuda")
        torch.cuda.synchronize()
        assert torch.equal(ob[2], of[2])
        for k in ("det", "center3d", "center3d_int", "center_hm"):
            assert torch.equal(db[k], df[k]), k
        ok = ob[2].bool()
        nvalid += int(ok.sum())
        if bool(ok.any()):
            worst_p = max(worst_p, (ob[0][ok] - of[0][ok]).abs().max().item())
            worst_c = max(worst_c, (ob[1][ok] - of[1][ok]).abs().max().item())
    report("bf16x3_seed_sweep", seeds=64, valid=nvalid, worst_points_mm_vs_f32_mode=worst_p, worst_conf=worst_c)
    assert nvalid >= 48 and worst_p < 1e-3 and worst_c < 1e-4


def test_predictor3d_bf16x3_wide(golden, bf16x3):
    """Experimental level bf16x3_wide (the trunk's dense 2D convolutions split as well): measured against the
    reference fixtures; it stays inside 1e-3 mm on them but without margin (7.6e-4 mm at cfg3), which is why it
    is not the labelled mode.  Integer path still exact on the fixtures (top-2 margins >= 1 %)."""
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    from tests.test_hip_predictor import make_cfg
    bf16x3.set_precision("bf16x3_wide")
    g = golden("predictor")
    worst = 0.0
    for tag in ("cfg2", "cfg3", "cfg5"):
        c = cases.PREDICTOR_CASES[tag]
        inp = cases.predictor_inputs(tag)
        pred = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"])
        pts, conf = pred(cuda(inp["imgs"]), cuda(inp["cam"]), cuda(inp["intr"]), cuda(inp["dist"]))
        torch.cuda.synchronize()
        dbg = pred.native(c["H"], c["W"]).debug("cuda")
        assert torch.equal(dbg["det"][0, :, :2].cpu().long(), torch.from_numpy(g[tag + ".preds"]).reshape(c["C"], 2))
        assert torch.equal(dbg["center3d_int"][0].cpu(), torch.from_numpy(g[tag + ".center3d"]).int())
        ep = max_err(pts, torch.from_numpy(g[tag + ".points3D"]))
        report("predictor3d_bf16x3_wide", tag=tag, points_mm=ep)
        worst = max(worst, ep)
    assert worst < 2e-3


def test_precision_is_a_property_of_the_predictor(golden):
    """jh_predictor_config.precision (ABI v4): an fp32 and a bf16x3 predictor built in EITHER order in one process,
    with the process default left at fp32, give the results of the process-wide switch; the reference's
    `trt_mode='new' | 'previous'` (jarvis3D.py:42-46, utils/paramClasses.py:21) selects bf16x3 instead of raising."""
    from jarvis_hybridnet_amd import _native as N
    from jarvis_hybridnet_amd import synthetic as S
    from jarvis_hybridnet_amd._predictor import NativePredictor
    from jarvis_hybridnet_amd.prediction.jarvis3D import JarvisPredictor3D
    from tests.test_hip_predictor import make_cfg
    assert N.get_precision() == "f32"
    c = cases.PREDICTOR_CASES["cfg2"]
    inp = cases.predictor_inputs("cfg2")
    kw = dict(num_cameras=c["C"], num_joints=c["J"], center_size=c["center_size"], bbox=c["bbox"],
              roi_cube_size=c["roi"], grid_spacing=c["spacing"], img_h=c["H"], img_w=c["W"],
              mean=S.MEAN, std=S.STD, time_batch=1)
    dev = [cuda(t) for t in (inp["cam"], inp["intr"], inp["dist"])]
    x = cuda(inp["imgs"]).unsqueeze(0).contiguous()

    def run(p):
        p.set_calibration(*dev)
        out = [t.clone() for t in p.forward(x)]
        torch.cuda.synchronize()
        return out
    b1 = NativePredictor(inp["sd_center"], inp["sd_hybrid"], precision="bf16x3", **kw)
    f1 = NativePredictor(inp["sd_center"], inp["sd_hybrid"], precision="f32", **kw)
    b2 = NativePredictor(inp["sd_center"], inp["sd_hybrid"], precision="bf16x3", **kw)
    dflt = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
    assert (b1.precision, f1.precision, b2.precision, dflt.precision) == ("bf16x3", "f32", "bf16x3", "f32")
    ob1, of1, ob2, od = run(b1), run(f1), run(b2), run(dflt)
    assert torch.equal(ob1[0], ob2[0]) and torch.equal(of1[0], od[0])      # independent of creation order
    assert not torch.equal(ob1[0], of1[0])                                 # and really different kernels
    prev = N.set_precision("bf16x3")                                       # the process-wide switch = the default only
    try:
        via_default = NativePredictor(inp["sd_center"], inp["sd_hybrid"], **kw)
        explicit_f32 = NativePredictor(inp["sd_center"], inp["sd_hybrid"], precision="f32", **kw)
    finally:
        N.set_precision(prev)
    assert via_default.precision == "bf16x3" and explicit_f32.precision == "f32"
    assert torch.equal(run(via_default)[0], ob1[0]) and torch.equal(run(explicit_f32)[0], of1[0])
    # the trt_mode seam of the reference's constructor
    g = golden("predictor")
    gold = torch.from_numpy(g["cfg2.points3D"])
    for mode in ("new", "previous"):
        jp = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"], mode)
        pts, conf = jp(cuda(inp["imgs"]), *dev)
        torch.cuda.synchronize()
        assert jp.precision == "bf16x3" and jp.native(c["H"], c["W"]).precision == "bf16x3"
        assert torch.equal(pts[0], ob1[0][0])
        assert max_err(pts, gold) < 1e-3
    off = JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"], "off")
    assert off.precision is None and torch.equal(off(cuda(inp["imgs"]), *dev)[0][0], of1[0][0])
    with pytest.raises(ValueError, match="trt_mode"):
        JarvisPredictor3D(make_cfg(c, c["center_size"]), inp["sd_center"], inp["sd_hybrid"], "fast")
