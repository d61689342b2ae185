"""GPU: single kernels through the C ABI vs the same op in PyTorch-CPU fp32.

Tolerance: these are fp32 kernels with a different summation order than the
CPU library, so the bar is float32 round-off: max abs error <= 2e-5 x the
output's max magnitude (x 10 after an InstanceNorm, which divides by sigma).
"""
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import cuda, rel_err, report

pytestmark = pytest.mark.gpu


def _conv(nd, kind, k, stride, pad, cin, cout, x, w, b, gate=None, norm_act=-1):
    from jarvis_hybridnet_amd import _native as N
    n = x.shape[0]
    if nd == 2:
        d, (h, wd) = 1, x.shape[2:]
    else:
        d, h, wd = x.shape[2:]
    xc = cuda(x)
    if kind == 0:
        ref_fn = F.conv2d if nd == 2 else F.conv3d
        ref = ref_fn(x * gate[:, :, None, None] if gate is not None else x, w, b, stride, pad)
    elif kind == 1:
        ref = F.conv_transpose2d(x, w, b, 2, 1)
    else:
        ref = F.conv_transpose3d(x, w, b, 2, 0)
    if norm_act >= 0:
        ref = F.instance_norm(ref, eps=1e-5)
        ref = [lambda v: v, F.relu, F.silu][norm_act](ref)
    y = torch.empty(ref.shape, device="cuda")
    wh, bh = w.contiguous(), (b.contiguous() if b is not None else None)
    gc = cuda(gate) if gate is not None else None
    N.check(N.lib().jh_op_conv(nd, kind, k, stride, pad, cin, cout, wh.data_ptr(),
                               bh.data_ptr() if bh is not None else None, xc.data_ptr(), n, d, h,
                               wd, N.ptr(gc), norm_act, y.data_ptr(), N.stream()))
    torch.cuda.synchronize()
    return y.cpu(), ref


CONV2D = [  # k, stride, cin, cout, H, W, bias, norm_act
    (3, 2, 3, 16, 64, 64, False, 2), (3, 1, 16, 16, 32, 48, False, -1), (3, 2, 8, 48, 40, 40, False, 2),
    (5, 2, 16, 96, 32, 32, False, -1), (5, 1, 40, 240, 12, 12, False, 2), (1, 1, 240, 56, 16, 16, False, 0),
    (1, 1, 56, 56, 4, 4, True, 0), (1, 1, 24, 56, 8, 8, True, -1), (3, 1, 40, 240, 16, 16, False, -1),
    (3, 2, 24, 144, 32, 32, False, 2), (1, 1, 56, 336, 16, 16, False, -1), (1, 1, 336, 56, 6, 10, False, 0),
    # pointwise layers on images that do not tile into 8 x 16: flattened to one row of H * W pixels, 1 x 128 tiles
    # (csrc/conv_host.hip) -- the 20 x 20 / 10 x 10 / 40 x 40 levels of the reference's default 320-pixel geometry
    (1, 1, 480, 80, 20, 20, False, 0), (1, 1, 112, 672, 20, 20, False, 2), (1, 1, 40, 88, 40, 40, True, 0),
    (1, 1, 88, 88, 10, 10, True, -1), (1, 1, 24, 88, 12, 24, True, 0), (1, 1, 16, 8, 10, 13, False, 2),
    # 3 x 3 stride-1 layers on images 17..20 wide, at most 22 high: one whole-image tile of 23 x 20 pixel slots
    (3, 1, 80, 480, 20, 20, False, 2), (3, 1, 24, 40, 22, 17, True, -1), (3, 1, 16, 16, 9, 20, False, 0),
    (3, 1, 40, 56, 23, 20, False, 0),     # (23 rows: past the whole-image tile's reach -> the 8 x 16 tiles)
    # 5 x 5 layers on rows of 33..40 pixels (8 x 40 tiles) and 3 x 3 stride-2 layers onto 17..20 x <= 22 (the 23 x 20 tile)
    (5, 1, 40, 240, 40, 40, False, 2), (5, 2, 24, 144, 80, 80, False, 2), (5, 1, 16, 32, 36, 38, True, -1),
    (5, 2, 8, 16, 70, 66, True, 0), (3, 2, 40, 240, 40, 40, False, 2), (3, 2, 16, 48, 44, 38, False, 0),
    # few-channel pointwise layers straight from registers (csrc/conv_pw_direct.hip): K <= 48, 1 / 4 / 6 column blocks
    (1, 1, 16, 8, 64, 64, False, 2), (1, 1, 48, 16, 32, 32, False, 0), (1, 1, 16, 56, 32, 32, True, 0),
    (1, 1, 24, 56, 16, 16, True, -1), (1, 1, 24, 88, 16, 32, True, 0), (1, 1, 40, 88, 20, 20, True, 0),
    (1, 1, 32, 16, 64, 32, False, 2), (1, 1, 8, 8, 40, 40, True, 2),
]


@pytest.mark.parametrize("k,stride,cin,cout,H,W,bias,norm_act", CONV2D)
def test_conv2d(k, stride, cin, cout, H, W, bias, norm_act):
    g = torch.Generator().manual_seed(k * 100 + cin)
    x = torch.randn(2, cin, H, W, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=g) if bias else None
    y, ref = _conv(2, 0, k, stride, k // 2, cin, cout, x, w, b, norm_act=norm_act)
    e = rel_err(y, ref)
    report("conv2d", k=k, stride=stride, cin=cin, cout=cout, rel=e)
    assert e < (2e-4 if norm_act >= 0 else 2e-5)


def test_conv2d_gate():
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 48, 20, 20, generator=g)
    w = torch.randn(16, 48, 1, 1, generator=g) / 7
    gate = torch.rand(3, 48, generator=g)
    y, ref = _conv(2, 0, 1, 1, 0, 48, 16, x, w, None, gate=gate)
    e = rel_err(y, ref)
    report("conv2d_gate", rel=e)
    assert e < 2e-5


@pytest.mark.parametrize("case", [
    (64, 23, 24, 24, False, -1),     # the keypoint head: swapped-operand epilogue (no statistics)
    (64, 23, 24, 24, True, 2),       # + bias, fused statistics -> InstanceNorm + SiLU after it
    (32, 8, 20, 12, True, -1),       # one column block per parity, ragged tiles
    (16, 23, 8, 16, False, 0),       # 16-channel passes
    (88, 23, 12, 12, False, -1),     # no instantiation of the fused kernel: four-phase general path
    (64, 23, 64, 64, False, 2, 12),  # more workgroups than CUs (a store-data hazard of an earlier epilogue
    (64, 23, 64, 64, False, -1, 12), # only showed with two workgroups resident per CU)
])
def test_deconv2d_k4s2p1(case):
    cin, cout, H, W, bias, norm_act = case[:6]
    n = case[6] if len(case) > 6 else 2
    """ConvTranspose2d(k4, s2, p1): csrc/deconv4.hip (all four output parities from one staged patch)
    and the four-phase form of conv_mfma.h, against torch."""
    g = torch.Generator().manual_seed(8 + cin + H)
    x = torch.randn(n, cin, H, W, generator=g)
    w = torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5    # asymmetric: catches tap/phase swaps
    b = torch.randn(cout, generator=g) * 0.3 if bias else None
    y, ref = _conv(2, 1, 4, 2, 1, cin, cout, x, w, b, norm_act=norm_act)
    e = rel_err(y, ref)
    report("deconv2d", cin=cin, cout=cout, h=H, norm_act=norm_act, rel=e)
    assert e < (2e-4 if norm_act >= 0 else 2e-5)


CONV3D = [  # k, stride, pad, cin, cout, G, norm_act
    (3, 1, 1, 46, 46, 16, 1), (3, 2, 1, 23, 46, 24, 1), (2, 2, 0, 46, 92, 16, 1), (3, 1, 1, 92, 92, 8, -1),
    (1, 1, 0, 46, 23, 12, -1), (3, 1, 1, 6, 6, 10, -1),
]


@pytest.mark.parametrize("k,stride,pad,cin,cout,G,norm_act", CONV3D)
def test_conv3d(k, stride, pad, cin, cout, G, norm_act):
    g = torch.Generator().manual_seed(k * 10 + cin)
    x = torch.randn(2, cin, G, G, G, generator=g)
    w = torch.randn(cout, cin, k, k, k, generator=g) / (cin * k ** 3) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y, ref = _conv(3, 0, k, stride, pad, cin, cout, x, w, b, norm_act=norm_act)
    e = rel_err(y, ref)
    report("conv3d", k=k, stride=stride, cin=cin, cout=cout, rel=e)
    assert e < (2e-4 if norm_act >= 0 else 2e-5)


@pytest.mark.parametrize("cin,cout,D,H,W", [(46, 46, 16, 16, 16), (92, 92, 8, 12, 20), (6, 23, 5, 9, 11)])
def test_conv3d_winograd_equals_direct(cin, cout, D, H, W, monkeypatch):
    """The 3x3x3 stride-1 convs run as Winograd F(2x2,3x3) x direct z by default
    (csrc/conv3d_wino.hip); JH_WINO=0 selects the direct MFMA kernel.  Both must match
    torch, on volumes that are not multiples of the 4 x 8 x 8 workgroup tile too."""
    g = torch.Generator().manual_seed(cin + D)
    x = torch.randn(2, cin, D, H, W, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y_w, ref = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=1)
    monkeypatch.setenv("JH_WINO_PW", "0")          # the one-role kernel (default: persistent form,
    y_q, _ = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=1)   # which falls back to it here)
    monkeypatch.delenv("JH_WINO_PW")
    assert rel_err(y_q, ref) < 2e-4
    monkeypatch.setenv("JH_WINO", "0")
    y_d, _ = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=1)
    ew, ed = rel_err(y_w, ref), rel_err(y_d, ref)
    report("conv3d_winograd", cin=cin, cout=cout, rel_winograd=ew, rel_direct=ed)
    assert ew < 2e-4 and ed < 2e-4
    assert not torch.equal(y_w, y_d), "JH_WINO had no effect"


def test_deconv3d_k2s2():
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 92, 8, 8, 8, generator=g)
    w = torch.randn(92, 46, 2, 2, 2, generator=g) / 10
    b = torch.randn(46, generator=g) * 0.1
    y, ref = _conv(3, 2, 2, 2, 0, 92, 46, x, w, b, norm_act=1)
    e = rel_err(y, ref)
    report("deconv3d", rel=e)
    assert e < 2e-4


@pytest.mark.parametrize("k,c,H,W,norm_act", [(3, 56, 32, 32, -1), (5, 240, 16, 16, 2), (3, 88, 8, 8, -1),
                                              (5, 336, 12, 20, 2), (5, 480, 20, 20, 2), (3, 88, 18, 20, -1),
                                              (5, 96, 24, 20, 2)])
def test_depthwise(k, c, H, W, norm_act):
    from jarvis_hybridnet_amd import _native as N
    g = torch.Generator().manual_seed(k + c)
    x = torch.randn(2, c, H, W, generator=g)
    w = torch.randn(c, 1, k, k, generator=g) / k
    ref = F.conv2d(x, w, None, 1, k // 2, 1, c)
    if norm_act >= 0:
        ref = F.silu(F.instance_norm(ref, eps=1e-5))
    xc, y = cuda(x), torch.empty(ref.shape, device="cuda")
    N.check(N.lib().jh_op_depthwise(k, c, w.contiguous().data_ptr(), xc.data_ptr(), 2, H, W,
                                    norm_act, y.data_ptr(), N.stream()))
    torch.cuda.synchronize()
    e = rel_err(y, ref)
    report("depthwise", k=k, c=c, rel=e)
    assert e < 2e-5 * (10 if norm_act >= 0 else 1)


@pytest.mark.parametrize("k,c,H,W", [(5, 240, 16, 16), (5, 672, 16, 16), (3, 88, 8, 8), (5, 336, 12, 14), (5, 528, 16, 16),
                                     (5, 480, 20, 20), (3, 96, 17, 19), (5, 672, 20, 18)])   # (17 .. 20: the 20 x 20 tile)
def test_depthwise_fused_se_pool(k, c, H, W):
    """One-tile images: the depthwise launch also delivers sum_p SiLU(InstanceNorm(y)) per (image, channel) -- the
    squeeze-excite pooling of MBConvBlock.forward (efficientnet.py:100-107) without a second pass over y."""
    from jarvis_hybridnet_amd import _native as N
    g = torch.Generator().manual_seed(k + c + H)
    x = torch.randn(3, c, H, W, generator=g)
    w = torch.randn(c, 1, k, k, generator=g) / k
    ref = F.conv2d(x, w, None, 1, k // 2, 1, c)
    ref_pool = F.silu(F.instance_norm(ref, eps=1e-5)).sum((2, 3))
    xc, y = cuda(x), torch.empty(ref.shape, device="cuda")
    pool = torch.empty((3, c), device="cuda")
    N.check(N.lib().jh_op_depthwise_pool(k, c, w.contiguous().data_ptr(), xc.data_ptr(), 3, H, W, y.data_ptr(),
                                         pool.data_ptr(), N.stream()))
    torch.cuda.synchronize()
    e, ep = rel_err(y, ref), rel_err(pool, ref_pool)
    report("depthwise_pool", k=k, c=c, rel=e, pool_rel=ep)
    assert e < 2e-5 and ep < 1e-4


@pytest.mark.parametrize("cin,cout,N,G,in_norm", [(46, 46, 5, 32, 1), (92, 92, 20, 16, 0), (60, 60, 3, 48, 1)])
def test_conv3d_winograd_persistent(cin, cout, N, G, in_norm, monkeypatch):
    """The persistent wave-specialised Winograd kernel (csrc/conv3d_wino_pw.hip, the default for
    launches with at least two tiles per CU) against torch, on the V2V layer shapes: 46->46 @ 32^3
    and 92->92 @ 16^3 (J = 23), 60->60 @ 48^3 (J = 30: two column-block groups).  Odd image counts
    give the workgroups unequal tile lists."""
    g = torch.Generator().manual_seed(cin + N)
    x = torch.randn(N, cin, G, G, G, generator=g)
    w = torch.randn(cout, cin, 3, 3, 3, generator=g) / (cin * 27) ** 0.5
    b = torch.randn(cout, generator=g) * 0.1
    y, ref = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=1 if in_norm else -1)
    monkeypatch.setenv("JH_WINO_PW", "0")
    y0, _ = _conv(3, 0, 3, 1, 1, cin, cout, x, w, b, norm_act=1 if in_norm else -1)
    e, e0 = rel_err(y, ref), rel_err(y0, ref)
    report("conv3d_winograd_persistent", cin=cin, cout=cout, n=N, g=G, rel=e, rel_one_role=e0)
    assert e < 2e-4 and e0 < 2e-4
    # the two kernels implement the same arithmetic in the same order: a launch may pick either
    # (by its tile count) without changing a bit of the result
    assert torch.equal(y, y0)



NODE_CASES = [  # (C, Cout, H, W, modes, act, n): the node shapes of the small / medium pyramids
    (56, 56, 64, 64, (0, 1), 2, 3),         # top-down P3: same level + nearest x2 of P4
    (56, 56, 32, 32, (0, 0, 3), 2, 3),      # bottom-up P4: two same-level inputs + 2x2 max-pool of P3
    (56, 56, 8, 8, (0, 3), 2, 3),           # P7 of the bottom-up pass
    (56, 64, 64, 64, (0, 1, 2), 0, 3),      # head: P3 + x2 P4 + x4 P5 -> first_conv (no activation)
    (88, 88, 32, 32, (0, 0, 3), 2, 3),      # medium model: wider pyramid (chunked halo path)
    # >= 2048 strips per launch: the stand-alone operator takes the row-streaming form (csrc/bifpn_rows.hip)
    (56, 56, 64, 64, (0, 1), 2, 64),        # P3 top-down, 16-row segments
    (56, 64, 64, 64, (0, 1, 2), 0, 64),     # head (three inputs, no activation, 64 output channels)
    (56, 56, 32, 32, (0, 1), 2, 256),       # P4 top-down, 8-row segments
    (56, 56, 48, 32, (0, 1), 2, 192),       # height that is not a power of two
    (56, 56, 32, 32, (0, 0, 0), 2, 256),    # P4 bottom-up when P3's node has written its pooled output: three same-level inputs
    # the medium model's 88-channel pyramid in the row-streaming form (22 channel quads x 2 pixel slots per wave,
    # 6 output column blocks, one wave per SIMD)
    (88, 88, 64, 64, (0, 1), 2, 64),        # P3 top-down
    (88, 88, 64, 64, (0, 1, 2), 0, 64),     # head: first_conv 88 -> 88, no activation
    (88, 88, 32, 32, (0, 1), 2, 256),       # P4 top-down
    (88, 88, 32, 32, (0, 0, 0), 2, 256),    # P4 bottom-up with three same-level inputs
    (88, 88, 48, 32, (0, 1), 2, 192),       # ragged height
    (88, 88, 16, 16, (0, 1), 2, 2048),      # P5 top-down of the medium model: one 16-pixel strip per image
    # the large model's 160-channel pyramid: workgroup row-streaming form (csrc/bifpn_rows_wg.hip: ten waves per
    # strip, one output column block and one 16-channel slice of the ring each)
    (160, 160, 32, 32, (0, 0, 3), 2, 3),    # tile form (small launch)
    (160, 160, 64, 64, (0, 1), 2, 64),      # P3 top-down
    (160, 160, 64, 64, (0, 1, 2), 0, 64),   # three inputs x1 / x2 / x4, no activation
    (160, 160, 32, 32, (0, 1), 2, 256),     # P4 top-down
    (160, 160, 32, 32, (0, 0, 0), 2, 256),  # P4 bottom-up with three same-level inputs
    (160, 160, 48, 32, (0, 1), 2, 192),     # ragged height
    (160, 160, 16, 16, (0, 1), 2, 2048),    # P5 top-down: one strip per image
    # ... and the levels narrower than a strip / not a multiple of 16 pixels (pixels past the row end masked)
    (160, 160, 8, 8, (0, 1), 2, 2048),      # P6 top-down
    (160, 160, 8, 8, (0, 0, 0), 2, 2048),   # P6 bottom-up with P5's pooled output
    (160, 160, 4, 4, (0, 0), 2, 4096),      # P7 bottom-up: two same-level inputs (P7_in, pooled P6)
    (160, 160, 2, 2, (0, 0), 2, 4096),      # ... of a 128-pixel crop
    (160, 160, 24, 24, (0, 1), 2, 512),     # 192-pixel crops: P4 is 24 wide (one full and one half strip)
    (160, 160, 24, 40, (0, 1, 2), 0, 384),  # three strips, the last one half empty; x1 / x2 / x4 inputs
    (56, 56, 4, 4, (0, 0), 2, 3),           # the tile kernel's two-same-input variant (fallback of the above)
    # 88 channels on level sizes of the reference's DEFAULT 320-pixel geometry (round 6; the stand-alone operator has no
    # time-batch class, so the ragged levels take the tile kernel here -- the workgroup row form on them is exercised by
    # the default_medium_320 fixture in its T = 8 class)
    (88, 88, 40, 40, (0, 1), 2, 20),        # P4 top-down, 2.5 tiles wide
    (88, 88, 20, 20, (0, 0, 3), 2, 20),     # P5 bottom-up
    (88, 88, 80, 80, (0, 1), 2, 43),        # P3 top-down: 40-row segments (80 % 32 != 0); 430 items: one-wave form
    (88, 88, 80, 80, (0, 1), 2, 44),        # ... 440 items = 110 workgroups of four pairs: the pair form
    (88, 88, 32, 32, (0, 1), 2, 255),       # aligned level, item count no multiple of 4
]


@pytest.mark.parametrize("C,Cout,H,W,modes,act,n", NODE_CASES)
def test_bifpn_node(C, Cout, H, W, modes, act, n):
    """One fused BiFPN node (csrc/bifpn_node.hip) against torch: InstanceNorm of every raw input
    applied on load, fast-normalised weighted fusion with nearest up-sampling / 2x2 max-pooling
    of the neighbour levels, SiLU, depthwise 3x3, pointwise 1x1 + bias -- the fusion expressions of
    jarvis/efficienttrack/model.py:301-353 (+ :119-126 for the head) followed by
    SeparableConvBlock.forward (:223-232) without its trailing InstanceNorm."""
    import ctypes
    from jarvis_hybridnet_amd import _native as N
    g = torch.Generator().manual_seed(C + H + len(modes))
    shape = {0: (H, W), 1: (H // 2, W // 2), 2: (H // 4, W // 4), 3: (H * 2, W * 2)}
    xs = [torch.randn(n, C, *shape[m], generator=g) * (1.0 + i) + 0.3 * i for i, m in enumerate(modes)]
    wts = torch.rand(len(modes), generator=g) + 0.2
    wts = wts / (wts.sum() + 1e-4)
    dw = torch.randn(C, 1, 3, 3, generator=g) / 3
    pw = torch.randn(Cout, C, generator=g) / C ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    fused = 0
    for x, m, wk in zip(xs, modes, wts):
        v = F.instance_norm(x, eps=1e-5)
        if m == 1:
            v = F.interpolate(v, scale_factor=2, mode="nearest")
        elif m == 2:
            v = F.interpolate(v, scale_factor=4, mode="nearest")
        elif m == 3:
            v = F.max_pool2d(v, 2, 2)
        fused = fused + wk * v
    if act == 2:
        fused = F.silu(fused)
    ref = F.conv2d(F.conv2d(fused, dw, None, 1, 1, 1, C), pw[:, :, None, None], bias)
    dev = [cuda(x) for x in xs] + [None] * (3 - len(xs))
    y = torch.empty(ref.shape, device="cuda")
    m3 = (ctypes.c_int * 3)(*(list(modes) + [0] * (3 - len(modes))))
    w3 = (ctypes.c_float * 3)(*([float(v) for v in wts] + [0.0] * (3 - len(modes))))
    N.check(N.lib().jh_op_bifpn_node(len(modes), m3, w3, act, n, C, Cout, H, W, N.ptr(dev[0]), N.ptr(dev[1]),
                                     N.ptr(dev[2]), dw.contiguous().data_ptr(), pw.contiguous().data_ptr(),
                                     bias.contiguous().data_ptr(), y.data_ptr(), N.stream()))
    torch.cuda.synchronize()
    e = rel_err(y, ref)
    report("bifpn_node", c=C, cout=Cout, h=H, modes=str(modes), n=n, rel=e)
    assert e < 2e-5
