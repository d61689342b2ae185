"""CPU oracle for the JARVIS-HybridNet multi-view inference hot path.

TEST INFRASTRUCTURE ONLY.  This module is a functional PyTorch-CPU restatement
of the reference's algorithm.  It may be imported by `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` and by
nothing else: the product path (package `jarvis-hybridnet_amd`) runs
hand-written HIP kernels and fails loudly when its extension is missing.

Parity pin: every function below is checked bit-for-bit against the imported
reference inside the build container by `tests/golden/make_golden.py`, which
also writes the committed golden vectors under `tests/golden/*.npz`;
`tests/test_oracle_golden.py` re-checks the oracle against those vectors on
any machine.  Third-party arithmetic that the reference itself does not pin
(torchvision's tensor `resize`, PyTorch's own conv / instance_norm / SVD
kernels) is restated as the torch 2.10 CPU ops named in SURVEY.md section 8c.

Everything works directly on a flat state dict (`{key: tensor}`) in the
reference's `.pth` key layout, so reference checkpoints load unchanged.
Each function cites the reference lines it restates (paths relative to the
upstream repository root).
"""
import math

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Architecture tables
# --------------------------------------------------------------------------

# jarvis/efficienttrack/model.py:34-51
MODEL_SIZES = {
    "small": dict(coef=0, fpn_w=56, fpn_cells=3, head_w=64),
    "medium": dict(coef=1, fpn_w=88, fpn_cells=4, head_w=88),
    "large": dict(coef=3, fpn_w=160, fpn_cells=6, head_w=160),
}
# jarvis/efficienttrack/utils.py:152-155  (width, depth) multipliers
_COEFS = {0: (0.5, 0.5), 1: (1.0, 1.0), 3: (1.1, 1.2)}
# jarvis/efficienttrack/utils.py:267-272  (k, repeats, in, out, expand, stride)
_STAGES = [(3, 1, 32, 16, 1, 1), (3, 2, 16, 24, 6, 2), (5, 2, 24, 40, 6, 2),
           (3, 3, 40, 80, 6, 2), (5, 3, 80, 112, 6, 1), (5, 4, 112, 192, 6, 2),
           (3, 1, 192, 320, 6, 1)]


def _round_filters(f, width, divisor=8):
    # jarvis/efficienttrack/utils.py:76-96
    f = f * width
    nf = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if nf < 0.9 * f:
        nf += divisor
    return int(nf)


def backbone_blocks(model_size):
    """Per-block table of the truncated EfficientNet trunk.

    Returns (stem_channels, blocks, tap_indices); each block is a dict with
    stage, k, stride, cin, cout, expand, mid (= cin*expand), squeeze.
    Restates efficientnet.py:156-173 (block construction),
    efficienttrack/model.py:515-533 (truncation + feature taps).
    """
    width, depth = _COEFS[MODEL_SIZES[model_size]["coef"]]
    blocks = []
    for stage, (k, rep, cin, cout, e, s) in enumerate(_STAGES):
        cin, cout = _round_filters(cin, width), _round_filters(cout, width)
        rep = int(math.ceil(depth * rep))
        for r in range(rep):
            b_in, b_s = (cin, s) if r == 0 else (cout, 1)
            blocks.append(dict(stage=stage, k=k, stride=b_s, cin=b_in,
                               cout=cout, expand=e, mid=b_in * e,
                               squeeze=max(1, int(b_in * 0.25))))
    # save_idxs: the first stride-2 block is ignored, every later stride-2
    # block marks "tap the block before me"; trunk is cut before the last one.
    flags, first, last = [], True, 0
    for i, b in enumerate(blocks):
        if first and b["stride"] == 2:
            first = False
            flags.append(False)
        else:
            flags.append(b["stride"] == 2)
            if b["stride"] == 2:
                last = i - 1
    kept = blocks[:last + 1]
    taps = [i for i in range(len(kept)) if flags[i + 1]]
    return _round_filters(32, width), kept, taps


def efficienttrack_state_spec(model_size, num_joints):
    """Ordered (key, shape) list of EfficientTrackBackbone.state_dict()."""
    ms = MODEL_SIZES[model_size]
    W, R, Fh = ms["fpn_w"], ms["fpn_cells"], ms["head_w"]
    stem, blocks, taps = backbone_blocks(model_size)
    cc = [blocks[t]["cout"] for t in taps]
    spec = [("weights_cat", (3,))]
    for i in range(R):
        p = "bifpn.%d." % i
        spec += [(p + n, (2,)) for n in ("p6_w1", "p5_w1", "p4_w1", "p3_w1")]
        spec += [(p + n, (3,)) for n in ("p4_w2", "p5_w2", "p6_w2")]
        spec += [(p + "p7_w2", (2,))]
        for n in ("conv6_up", "conv5_up", "conv4_up", "conv3_up", "conv4_down",
                  "conv5_down", "conv6_down", "conv7_down"):
            spec += [(p + n + ".depthwise_conv.weight", (W, 1, 3, 3)),
                     (p + n + ".pointwise_conv.weight", (W, W, 1, 1)),
                     (p + n + ".pointwise_conv.bias", (W,))]
        if i == 0:
            for n, c in (("p5_down_channel", cc[2]), ("p4_down_channel", cc[1]),
                         ("p3_down_channel", cc[0]), ("p5_to_p6", cc[2]),
                         ("p4_down_channel_2", cc[1]),
                         ("p5_down_channel_2", cc[2])):
                spec += [(p + n + ".0.weight", (W, c, 1, 1)),
                         (p + n + ".0.bias", (W,))]
    spec.append(("backbone_net.model._conv_stem.weight", (stem, 3, 3, 3)))
    for i, b in enumerate(blocks):
        p = "backbone_net.model._blocks.%d." % i
        if b["expand"] != 1:
            spec.append((p + "_expand_conv.weight", (b["mid"], b["cin"], 1, 1)))
        if b["stage"] < 4:
            spec.append((p + "_depthwise_conv.weight",
                         (b["mid"], b["cin"], b["k"], b["k"])))
        else:
            spec.append((p + "_depthwise_conv.weight",
                         (b["mid"], 1, b["k"], b["k"])))
        spec += [(p + "_se_reduce.weight", (b["squeeze"], b["mid"], 1, 1)),
                 (p + "_se_reduce.bias", (b["squeeze"],)),
                 (p + "_se_expand.weight", (b["mid"], b["squeeze"], 1, 1)),
                 (p + "_se_expand.bias", (b["mid"],)),
                 (p + "_project_conv.weight", (b["cout"], b["mid"], 1, 1))]
    spec += [("first_conv.depthwise_conv.weight", (W, 1, 3, 3)),
             ("first_conv.pointwise_conv.weight", (Fh, W, 1, 1)),
             ("first_conv.pointwise_conv.bias", (Fh,)),
             ("deconv1.weight", (Fh, num_joints, 4, 4)),
             ("final_conv1.weight", (num_joints, Fh, 3, 3)),
             ("final_conv2.weight", (num_joints, Fh, 1, 1))]
    return spec


def v2v_state_spec(num_joints):
    """Ordered (key, shape) list of V2VNet.state_dict() (v2vnet.py:86-96)."""
    J = num_joints
    spec = []

    def conv(name, co, ci, k):
        spec.extend([(name + ".weight", (co, ci, k, k, k)),
                     (name + ".bias", (co,))])

    def res(name, c):
        conv(name + ".res_branch.0", c, c, 3)
        conv(name + ".res_branch.3", c, c, 3)

    conv("front_layers.0.block.0", 2 * J, J, 3)
    res("front_layers.1", 2 * J)
    conv("encoder_decoder.encoder_pool1.block.0", 4 * J, 2 * J, 2)
    res("encoder_decoder.mid_res", 4 * J)
    # ConvTranspose3d weight layout is (in, out, k, k, k)
    spec.extend([("encoder_decoder.decoder_upsample1.block.0.weight",
                  (4 * J, 2 * J, 2, 2, 2)),
                 ("encoder_decoder.decoder_upsample1.block.0.bias", (2 * J,))])
    res("encoder_decoder.decoder_res1", 2 * J)
    res("encoder_decoder.skip_res1", 2 * J)
    conv("output_layer", J, 2 * J, 1)
    return spec


def hybridnet_state_spec(model_size, num_joints):
    """HybridNetBackbone.state_dict(): effTrack.* then v2vNet.*
    (hybridnet/model.py:28-41; reproLayer owns no parameters)."""
    return ([("effTrack." + k, s)
             for k, s in efficienttrack_state_spec(model_size, num_joints)] +
            [("v2vNet." + k, s) for k, s in v2v_state_spec(num_joints)])


# --------------------------------------------------------------------------
# EfficientTrack 2D network
# --------------------------------------------------------------------------

def _inorm(x):
    # InstanceNorm{2,3}d defaults: affine=False, no running stats, eps=1e-5
    return F.instance_norm(x, eps=1e-5)


def _mbconv(sd, p, b, x):
    """MBConvBlock.forward, efficientnet.py:90-123."""
    inp = x
    if b["stage"] < 4:
        # "fused" path: one dense kxk conv; _expand_conv is never executed
        x = F.conv2d(x, sd[p + "_depthwise_conv.weight"], None, b["stride"],
                     b["k"] // 2)
    else:
        if b["expand"] != 1:
            x = F.conv2d(x, sd[p + "_expand_conv.weight"])
        x = F.conv2d(x, sd[p + "_depthwise_conv.weight"], None, b["stride"],
                     b["k"] // 2, 1, b["mid"])
    x = F.silu(_inorm(x))
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.conv2d(s, sd[p + "_se_reduce.weight"], sd[p + "_se_reduce.bias"])
    s = F.silu(s)
    s = F.conv2d(s, sd[p + "_se_expand.weight"], sd[p + "_se_expand.bias"])
    x = torch.sigmoid(s) * x
    x = _inorm(F.conv2d(x, sd[p + "_project_conv.weight"]))
    if b["stride"] == 1 and b["cin"] == b["cout"]:
        x = x + inp
    return x


def backbone_forward(sd, x, model_size, prefix=""):
    """EfficientNet feature wrapper, efficienttrack/model.py:535-548."""
    _, blocks, taps = backbone_blocks(model_size)
    p = prefix + "backbone_net.model."
    x = F.silu(_inorm(F.conv2d(x, sd[p + "_conv_stem.weight"], None, 2, 1)))
    feats = []
    for i, b in enumerate(blocks):
        x = _mbconv(sd, p + "_blocks.%d." % i, b, x)
        if i in taps:
            feats.append(x)
    return feats


def _sepconv(sd, p, x, act=False):
    """SeparableConvBlock.forward, efficienttrack/model.py:223-232."""
    w = sd[p + "depthwise_conv.weight"]
    x = F.conv2d(x, w, None, 1, 1, 1, w.shape[0])
    x = F.conv2d(x, sd[p + "pointwise_conv.weight"],
                 sd[p + "pointwise_conv.bias"])
    x = _inorm(x)
    return F.silu(x) if act else x


def _fuse_w(param, eps=1e-4):
    # fast-normalised fusion, e.g. efficienttrack/model.py:309-311
    w = F.relu(param)
    return w / (torch.sum(w, dim=0) + eps)


def _up2(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")


def _pool2(x):
    return F.max_pool2d(x, 2, 2)


def bifpn_forward(sd, p, feats, first):
    """BiFPN_first.forward (model.py:446-504) / BiFPN.forward (:301-353)."""
    def lateral(name, x):
        return _inorm(F.conv2d(x, sd[p + name + ".0.weight"],
                               sd[p + name + ".0.bias"]))
    if first:
        p3, p4, p5 = feats
        p6_in = _pool2(lateral("p5_to_p6", p5))
        p7_in = _pool2(p6_in)
        p3_in = lateral("p3_down_channel", p3)
        p4_in = lateral("p4_down_channel", p4)
        p5_in = lateral("p5_down_channel", p5)
    else:
        p3_in, p4_in, p5_in, p6_in, p7_in = feats

    w = _fuse_w(sd[p + "p6_w1"])
    p6_up = _sepconv(sd, p + "conv6_up.",
                     F.silu(w[0] * p6_in + w[1] * _up2(p7_in)))
    w = _fuse_w(sd[p + "p5_w1"])
    p5_up = _sepconv(sd, p + "conv5_up.",
                     F.silu(w[0] * p5_in + w[1] * _up2(p6_up)))
    w = _fuse_w(sd[p + "p4_w1"])
    p4_up = _sepconv(sd, p + "conv4_up.",
                     F.silu(w[0] * p4_in + w[1] * _up2(p5_up)))
    w = _fuse_w(sd[p + "p3_w1"])
    p3_out = _sepconv(sd, p + "conv3_up.",
                      F.silu(w[0] * p3_in + w[1] * _up2(p4_up)))
    if first:
        p4_in = lateral("p4_down_channel_2", p4)
        p5_in = lateral("p5_down_channel_2", p5)
    w = _fuse_w(sd[p + "p4_w2"])
    p4_out = _sepconv(sd, p + "conv4_down.", F.silu(
        w[0] * p4_in + w[1] * p4_up + w[2] * _pool2(p3_out)))
    w = _fuse_w(sd[p + "p5_w2"])
    p5_out = _sepconv(sd, p + "conv5_down.", F.silu(
        w[0] * p5_in + w[1] * p5_up + w[2] * _pool2(p4_out)))
    w = _fuse_w(sd[p + "p6_w2"])
    p6_out = _sepconv(sd, p + "conv6_down.", F.silu(
        w[0] * p6_in + w[1] * p6_up + w[2] * _pool2(p5_out)))
    w = _fuse_w(sd[p + "p7_w2"])
    p7_out = _sepconv(sd, p + "conv7_down.", F.silu(
        w[0] * p7_in + w[1] * _pool2(p6_out)))
    return p3_out, p4_out, p5_out, p6_out, p7_out


def efficienttrack_forward(sd, x, model_size, prefix="", want_res1=True):
    """EfficientTrackBackbone.forward, efficienttrack/model.py:114-130.

    Returns (res1, res2); res1 (the `final_conv1` branch) is dead on the
    inference path and can be skipped with want_res1=False.
    """
    feats = backbone_forward(sd, x, model_size, prefix)
    for i in range(MODEL_SIZES[model_size]["fpn_cells"]):
        feats = bifpn_forward(sd, prefix + "bifpn.%d." % i, feats, i == 0)
    x3 = F.interpolate(feats[2], scale_factor=4, mode="nearest")
    x2 = F.interpolate(feats[1], scale_factor=2, mode="nearest")
    w = F.softplus(sd[prefix + "weights_cat"])
    w = w / (torch.sum(w, dim=0) + 0.0001)
    x1 = w[0] * feats[0] + w[1] * x2 + w[2] * x3
    # first_conv = SeparableConvBlock(W, F, True): the positional True is
    # `norm`, so there is NO activation here (model.py:86-88 vs :191-192)
    mid = _sepconv(sd, prefix + "first_conv.", x1, act=False)
    res2 = F.conv_transpose2d(mid, sd[prefix + "deconv1.weight"], None, 2, 1)
    res1 = (F.conv2d(mid, sd[prefix + "final_conv1.weight"], None, 1, 1)
            if want_res1 else None)
    return res1, res2


# --------------------------------------------------------------------------
# V2V 3D network
# --------------------------------------------------------------------------

def _res3d(sd, p, x):
    """Res3DBlock.forward, hybridnet/v2vnet.py:27-43 (dropout = identity)."""
    r = F.conv3d(x, sd[p + "res_branch.0.weight"], sd[p + "res_branch.0.bias"],
                 1, 1)
    r = F.relu(_inorm(r))
    r = F.conv3d(r, sd[p + "res_branch.3.weight"], sd[p + "res_branch.3.bias"],
                 1, 1)
    return F.relu(_inorm(r) + x)


def v2v_forward(sd, x, prefix=""):
    """V2VNet.forward, hybridnet/v2vnet.py:98-102 with :64-83 inlined."""
    p = prefix
    x = F.conv3d(x, sd[p + "front_layers.0.block.0.weight"],
                 sd[p + "front_layers.0.block.0.bias"], 2, 1)
    x = F.relu(_inorm(x))
    x = _res3d(sd, p + "front_layers.1.", x)
    e = p + "encoder_decoder."
    skip = _res3d(sd, e + "skip_res1.", x)
    x = F.conv3d(x, sd[e + "encoder_pool1.block.0.weight"],
                 sd[e + "encoder_pool1.block.0.bias"], 2, 0)
    x = F.relu(_inorm(x))
    x = _res3d(sd, e + "mid_res.", x)
    x = F.conv_transpose3d(x, sd[e + "decoder_upsample1.block.0.weight"],
                           sd[e + "decoder_upsample1.block.0.bias"], 2, 0)
    x = F.relu(_inorm(x))
    x = _res3d(sd, e + "decoder_res1.", x)
    x = x + skip
    return F.conv3d(x, sd[p + "output_layer.weight"],
                    sd[p + "output_layer.bias"])


# --------------------------------------------------------------------------
# Reprojection layer
# --------------------------------------------------------------------------

def reprojection_grid(roi_cube_size, grid_spacing):
    """Coarse voxel grid in mm, hybridnet/repro_layer.py:18-36."""
    G = int(roi_cube_size / grid_spacing)
    Gh = int(G / 2)
    half = int(G / 2 / 2)
    ax = torch.arange(Gh, dtype=torch.float32) - half
    ii, jj, kk = torch.meshgrid(ax, ax, ax, indexing="ij")
    return torch.stack([ii, jj, kk], dim=3) * grid_spacing * 2


def reprojection_indices(grid, cam_m, intr, dist, center_hm, hs, G):
    """ReprojectionLayer.reprojectPoints, hybridnet/repro_layer.py:40-85.

    grid (Gh,Gh,Gh,3) mm incl. centre; cam_m (C,4,3); intr (C,3,3);
    dist (C,1,5); center_hm (C,2) int.  Returns int64 (C,G,G,G) plus the
    fine u/v coordinate fields (C,G,G,G) for diagnostics.
    """
    C = cam_m.shape[0]
    Gh = grid.shape[0]
    K = intr.permute(1, 2, 0)
    D = dist.permute(1, 2, 0)
    chm = center_hm.permute(1, 0)
    ones = torch.ones([Gh, Gh, Gh, 1])
    x = torch.cat((grid, ones), 3)
    part = torch.matmul(x.view(1, -1, 4), cam_m).view(-1, Gh, Gh, Gh, 3)
    part = part.permute(1, 2, 3, 4, 0)
    v1 = part[:, :, :, 0] / part[:, :, :, 2] - K[2, 0]
    v2 = part[:, :, :, 1] / part[:, :, :, 2] - K[2, 1]
    r2 = torch.square(v1 / K[0, 0]) + torch.square(v2 / K[1, 1])
    dd = 1 + (D[0, 0] + D[0, 1] * r2) * r2
    v1 = v1 * dd + K[2, 0]
    v2 = v2 * dd + K[2, 1]
    v1 = torch.clamp(v1, chm[0] - (hs - 1), chm[0] + hs - 2) - chm[0] + hs - 1
    v2 = torch.clamp(v2, chm[1] - (hs - 1), chm[1] + hs - 2) - chm[1] + hs - 1
    v1 = F.interpolate(v1.permute(3, 0, 1, 2).view(1, -1, Gh, Gh, Gh),
                       size=(G, G, G), mode="trilinear").view(C, G, G, G)
    v2 = F.interpolate(v2.permute(3, 0, 1, 2).view(1, -1, Gh, Gh, Gh),
                       size=(G, G, G), mode="trilinear").view(C, G, G, G)
    idx = ((v2 / 2).int() * hs + (v1 / 2).int()).long()
    return idx, v1, v2


def reprojection_forward(heatmaps_padded, center3d, center_hm, cam_m, intr,
                         dist, roi_cube_size, grid_spacing, chunk=None,
                         return_idx=False, idx_override=None):
    """ReprojectionLayer.forward, hybridnet/repro_layer.py:88-119.

    heatmaps_padded (1,C,J,hs,hs); center3d (1,3); center_hm (1,C,2);
    cam_m (1,C,4,3); intr (1,C,3,3); dist (1,C,1,5) -> (1,J,G,G,G).
    The reference materialises the (J, C*G^3) gather; `chunk` lets the oracle
    do it per joint range with identical arithmetic (mean over the camera
    axis of the same gathered values) to bound memory.
    `idx_override` (C,G,G,G) int64 replaces the index field of
    repro_layer.py:82-83 (checker use: see `tail_with_indices`).
    """
    hm = heatmaps_padded[0].transpose(0, 1)  # (J,C,hs,hs)
    J, C, hs = hm.shape[0], hm.shape[1], hm.shape[2]
    G = int(roi_cube_size / grid_spacing)
    grid = reprojection_grid(roi_cube_size, grid_spacing) + center3d[0]
    if idx_override is None:
        idx, _, _ = reprojection_indices(grid, cam_m[0], intr[0], dist[0],
                                         center_hm[0], hs, G)
    else:
        idx = idx_override.long()
    off = torch.arange(0, hs * hs * C, hs * hs)
    flat_idx = (idx.flatten(1).transpose(1, 0) + off).transpose(1, 0).flatten()
    flat_hm = hm.flatten(1)
    if chunk is None:
        out = torch.mean(torch.index_select(flat_hm, 1, flat_idx)
                         .view(J, C, G, G, G), dim=1)
    else:
        out = torch.cat([
            torch.mean(torch.index_select(flat_hm[j:j + chunk], 1, flat_idx)
                       .view(-1, C, G, G, G), dim=1)
            for j in range(0, J, chunk)], 0)
    out = out.unsqueeze(0)
    return (out, idx) if return_idx else out


# --------------------------------------------------------------------------
# HybridNet backbone + soft-argmax tail
# --------------------------------------------------------------------------

def softargmax_tail(v2v_out, center3d, roi_cube_size, grid_spacing):
    """hybridnet/model.py:73-88: softplus, soft-argmax, confidences, mm."""
    Gh = v2v_out.shape[2]
    ax = torch.arange(Gh)
    xx, yy, zz = torch.meshgrid(ax, ax, ax, indexing="ij")
    h = F.softplus(v2v_out)
    norm = torch.sum(h, dim=[2, 3, 4])
    x = torch.sum(torch.mul(h, xx), dim=[2, 3, 4]) / norm
    y = torch.sum(torch.mul(h, yy), dim=[2, 3, 4]) / norm
    z = torch.sum(torch.mul(h, zz), dim=[2, 3, 4]) / norm
    pts = torch.stack([x, y, z], dim=2)
    conf = torch.clamp(torch.max(h.view(*h.shape[:2], -1), dim=2)[0],
                       max=255.) / 255.
    gs = torch.tensor(grid_spacing)
    roi = torch.tensor(roi_cube_size)
    pts = (pts.transpose(0, 1) * gs * 2 - roi / 2. + center3d).transpose(0, 1)
    return F.softplus(h), pts, conf


def hybridnet_forward(sd, model_size, roi_cube_size, grid_spacing, imgs,
                      center_hm, center3d, cam_m, intr, dist, chunk=None):
    """HybridNetBackbone.forward, hybridnet/model.py:53-90.

    imgs (b,C,3,B,B) normalised crops.  Returns the reference's 4-tuple
    (heatmap_final, heatmaps_padded, points3D, confidences).
    """
    b = imgs.shape[0]
    hm = efficienttrack_forward(sd, imgs.reshape(-1, *imgs.shape[2:]),
                                model_size, "effTrack.", want_res1=False)[1]
    hm = hm.reshape(b, -1, hm.shape[1], hm.shape[2], hm.shape[3])
    hm_pad = F.pad(hm, [1, 1, 1, 1], mode="constant", value=0.)
    vol = reprojection_forward(hm_pad, center3d, center_hm, cam_m, intr, dist,
                               roi_cube_size, grid_spacing, chunk=chunk)
    out = v2v_forward(sd, vol / 255., "v2vNet.")
    final, pts, conf = softargmax_tail(out, center3d, roi_cube_size,
                                       grid_spacing)
    return final, hm_pad, pts, conf


def tail_with_indices(sd, heatmaps_padded, idx, center3d, roi_cube_size,
                      grid_spacing, chunk=None):
    """hybridnet/model.py:67-88 from the gather on, with the gather index field
    GIVEN instead of computed (repro_layer.py:88-107 unchanged).

    Checker use only.  torch's CPU kernels differ in the last bit between CPU
    models, so the reference's OWN index field (repro_layer.py:82-83: a
    truncation of u/2, v/2) differs in a handful of voxels between hosts --
    wherever u/2 or v/2 lies within an ulp of an integer.  The HIP path's
    indices are proven equal to the fixture host's (tests/test_hip_stages.py::
    test_reprojection); feeding them here removes that host dependence, so the
    comparison against an oracle run on ANY host can hold the strict bar.
    heatmaps_padded (1,C,J,hs,hs), idx (C,G,G,G), center3d (1,3) int."""
    vol = reprojection_forward(heatmaps_padded, center3d, None, None, None,
                               None, roi_cube_size, grid_spacing, chunk=chunk,
                               idx_override=idx)
    out = v2v_forward(sd, vol / 255., "v2vNet.")
    _, pts, conf = softargmax_tail(out, center3d, roi_cube_size, grid_spacing)
    return pts, conf


def index_flip_report(idx_a, idx_b, u, v, hs):
    """Which voxels two index fields (C,G,G,G) disagree on, and how close this
    host's u/2 resp. v/2 (fine coordinate fields of `reprojection_indices`)
    is to an integer there: a flip caused by last-bit differences of the
    floating-point path sits within a few ulp of a truncation boundary."""
    bad = (idx_a != idx_b).nonzero()
    rows = []
    for c, i, j, k in bad.tolist():
        hu, hv = float(u[c, i, j, k]) / 2, float(v[c, i, j, k]) / 2
        du, dv = abs(hu - round(hu)), abs(hv - round(hv))
        a, b = int(idx_a[c, i, j, k]), int(idx_b[c, i, j, k])
        moved_u = (a % hs) != (b % hs)
        moved_v = (a // hs) != (b // hs)
        # the distance that matters: of the coordinate(s) whose truncation differs
        d = max(du if moved_u else 0.0, dv if moved_v else 0.0)
        # one ulp of the halved coordinate (float32)
        ulp = 2.0 ** (math.floor(math.log2(max(abs(hu if moved_u else hv), 1e-30))) - 23)
        rows.append(dict(voxel=[c, i, j, k], half_u=hu, half_v=hv,
                         dist_to_integer=d, ulp=ulp, dist_in_ulp=d / ulp))
    return rows


def host_parity(sd_hybrid, inter, idx_hip, pts_hip, pts_host, calib, roi_cube_size,
                grid_spacing, bbox, chunk=None):
    """The HIP path's 3D keypoints against the oracle RUN ON THIS HOST, free of the host's
    index flips: `inter` = the intermediates of `predictor3d_forward` on the same frame set,
    `idx_hip` (C,G,G,G) the HIP path's gather indices for the oracle's own heat maps and centres,
    `pts_host` the oracle's unmodified result.  Returns the raw distance, the number of index
    flips, where they sit relative to the truncation boundary, and the distance to the oracle
    re-run from the gather on with the HIP indices (the figure held to the 1e-3 mm bar)."""
    cam, intr, dist = calib
    G = int(roi_cube_size / grid_spacing)
    hs = bbox // 2 + 2
    c3i = inter["center3d"].int()[None]
    grid = reprojection_grid(roi_cube_size, grid_spacing) + c3i[0]
    ridx, u, v = reprojection_indices(grid, cam, intr, dist, inter["center_hm"], hs, G)
    idx_hip = idx_hip.reshape(ridx.shape).long()
    flips = int((idx_hip != ridx).sum())
    out = dict(raw_mm=(pts_hip - pts_host).abs().max().item(), flips=flips, of=int(ridx.numel()))
    if flips == 0:
        out["same_indices_mm"] = out["raw_mm"]
        out["flip_voxels"] = []
        return out
    with torch.no_grad():
        pts_same, _ = tail_with_indices(sd_hybrid, inter["heatmaps_padded"], idx_hip, c3i,
                                        roi_cube_size, grid_spacing, chunk=chunk)
    out["same_indices_mm"] = (pts_hip - pts_same).abs().max().item()
    out["flip_voxels"] = index_flip_report(idx_hip, ridx, u, v, hs)[:32]
    return out


# --------------------------------------------------------------------------
# Camera geometry (ReprojectionTool)
# --------------------------------------------------------------------------

def reproject_point(point3d, cam_m, intr, dist):
    """ReprojectionTool.reprojectPoint, utils/reprojection.py:49-66.
    point3d (1,3) -> (C,2) distorted pixel coordinates."""
    ones = torch.ones([point3d.shape[0], 1])
    p = torch.cat((point3d, ones), 1).unsqueeze(0)
    pr = torch.matmul(p, cam_m).permute(1, 2, 0)
    pr[:, 0] = pr[:, 0] / pr[:, 2] - intr[:, 2, 0]
    pr[:, 1] = pr[:, 1] / pr[:, 2] - intr[:, 2, 1]
    r2 = (torch.square(pr[:, 0] / intr[:, 0, 0]) +
          torch.square(pr[:, 1] / intr[:, 1, 1]))
    dd = 1 + (dist[:, 0, 0] + dist[:, 0, 1] * r2) * r2
    pr[:, 0] = pr[:, 0] * dd + intr[:, 2, 0]
    pr[:, 1] = pr[:, 1] * dd + intr[:, 2, 1]
    return pr[:, :2].permute(0, 2, 1).squeeze()


def reconstruct_point(points, maxvals, cam_m, intr, dist):
    """ReprojectionTool.reconstructPoint, utils/reprojection.py:69-90.
    points (2,C) pixels (not modified here), maxvals (C,1,1) -> (3,) mm."""
    P = cam_m.permute(0, 2, 1)
    u = points[0] - intr[:, 2, 0]
    v = points[1] - intr[:, 2, 1]
    r2 = torch.square(u / intr[:, 0, 0]) + torch.square(v / intr[:, 1, 1])
    dd = 1 + (dist[:, 0, 0] + dist[:, 0, 1] * r2) * r2
    u = u / dd + intr[:, 2, 0]
    v = v / dd + intr[:, 2, 1]
    pts = torch.stack([u, v], 0)
    A = (torch.bmm(pts.permute(1, 0).reshape(pts.shape[1], 2, 1),
                   P[:, 2].reshape(P.shape[0], 1, 4)) - P[:, 0:2])
    A = A * maxvals
    _, _, vh = torch.linalg.svd(A.flatten(0, 1))
    X = vh.transpose(0, 1)[:, -1]
    X = X / X[-1]
    return X[0:3]


# --------------------------------------------------------------------------
# Full predictor
# --------------------------------------------------------------------------

def predictor3d_forward(sd_center, sd_hybrid, imgs, cam_m, intr, dist, *,
                        center_size, bbox, roi_cube_size, grid_spacing,
                        mean, std, center_model="small", kp_model="small",
                        chunk=None, intermediates=None):
    """JarvisPredictor3D.forward, prediction/jarvis3D.py:129-190.

    imgs (C,3,H,W) RGB in [0,1].  Returns (points3D (1,J,3), confidences
    (1,J)) or (None, None) when fewer than two cameras see the subject.
    `intermediates`, when a dict, receives every integer-path tensor.
    """
    C = imgs.shape[0]
    hw = int(bbox / 2)
    mean_t = torch.tensor(mean).view(3, 1, 1)
    std_t = torch.tensor(std).view(3, 1, 1)
    img_size = torch.tensor([imgs.shape[3], imgs.shape[2]])
    scale = torch.tensor([imgs.shape[3] / float(center_size),
                          imgs.shape[2] / float(center_size)]).float()
    small = F.interpolate(imgs, size=[center_size, center_size],
                          mode="bilinear", align_corners=False)
    small = (small - mean_t) / std_t
    hm = efficienttrack_forward(sd_center, small, center_model,
                                want_res1=False)[1]
    flat = hm.view(hm.shape[0], hm.shape[1], -1)
    m = flat.argmax(2).view(flat.shape[0], flat.shape[1], 1)
    preds = torch.cat((m % hm.shape[2], m // hm.shape[3]), dim=2)
    maxvals = flat.gather(2, m)
    n_detect = torch.numel(maxvals[maxvals > 50])
    maxvals = maxvals / 255.
    if intermediates is not None:
        intermediates.update(preds=preds.clone(), maxvals=maxvals.clone(),
                             n_detect=n_detect, center_heatmap=hm)
    if n_detect < 2:
        return None, None
    center3d = reconstruct_point(
        (preds.reshape(C, 2) * (scale * 2)).transpose(0, 1), maxvals,
        cam_m, intr, dist)
    chm = reproject_point(center3d.unsqueeze(0), cam_m, intr, dist).int()
    chm[:, 0] = torch.clamp(chm[:, 0], hw, img_size[0] - hw)
    chm[:, 1] = torch.clamp(chm[:, 1], hw, img_size[1] - hw)
    crops = torch.zeros((C, 3, bbox, bbox))
    for i in range(C):
        cx, cy = int(chm[i, 0]), int(chm[i, 1])
        crops[i] = imgs[i, :, cy - hw:cy + hw, cx - hw:cx + hw]
    crops = (crops - mean_t) / std_t
    if intermediates is not None:
        intermediates.update(center3d=center3d.clone(), center_hm=chm.clone())
    final, hm_pad, pts, conf = hybridnet_forward(
        sd_hybrid, kp_model, roi_cube_size, grid_spacing, crops.unsqueeze(0),
        chm.unsqueeze(0), center3d.int().unsqueeze(0), cam_m.unsqueeze(0),
        intr.unsqueeze(0), dist.unsqueeze(0), chunk=chunk)
    if intermediates is not None:
        intermediates.update(heatmaps_padded=hm_pad, heatmap_final=final)
    return pts, conf


def predictor2d_forward(sd_center, sd_kp, img, *, center_size, bbox, mean, std,
                        center_model="small", kp_model="small",
                        intermediates=None):
    """JarvisPredictor2D.forward, prediction/jarvis2D.py:102-155.

    img (1,3,H,W) RGB in [0,1] -> (points2D (J,2) int64 pixels, confidences
    (J,)) or (None, None) when the centre heatmap maximum is <= 40.
    """
    hw = int(bbox / 2)
    mean_t = torch.tensor(mean).view(3, 1, 1)
    std_t = torch.tensor(std).view(3, 1, 1)
    img_size = torch.tensor([img.shape[3], img.shape[2]])
    scale = torch.tensor([img_size[0] / float(center_size),
                          img_size[1] / float(center_size)]).float()
    small = F.interpolate(img, size=[center_size, center_size],
                          mode="bilinear", align_corners=False)
    small = (small - mean_t) / std_t
    hm = efficienttrack_forward(sd_center, small, center_model,
                                want_res1=False)[1]
    flat = hm.view(hm.shape[0], hm.shape[1], -1)
    m = flat.argmax(2).view(flat.shape[0], flat.shape[1], 1)
    maxval = flat.gather(2, m).squeeze()
    if intermediates is not None:
        intermediates.update(center_heatmap=hm, maxval=maxval.clone())
    if not maxval > 40:
        return None, None
    chm = torch.cat((m % hm.shape[2], m // hm.shape[3]),
                    dim=2).squeeze() * scale * 2
    chm = chm.int()
    chm[0] = torch.clamp(chm[0], hw, img_size[0] - hw - 1)
    chm[1] = torch.clamp(chm[1], hw, img_size[1] - hw - 1)
    cx, cy = int(chm[0]), int(chm[1])
    crop = img[:, :, cy - hw:cy + hw, cx - hw:cx + hw]
    crop = (crop - mean_t) / std_t
    kh = efficienttrack_forward(sd_kp, crop, kp_model, want_res1=False)[1]
    kflat = kh.view(kh.shape[0], kh.shape[1], -1)
    km = kflat.argmax(2).view(kflat.shape[0], kflat.shape[1], 1)
    pts = torch.cat((km % kh.shape[2], km // kh.shape[3]), dim=2).squeeze() * 2
    conf = kflat.gather(2, km).squeeze()
    conf = torch.clamp(conf, max=255.) / 255.
    pts = pts + chm - hw
    if intermediates is not None:
        intermediates.update(center_hm=chm.clone(), kp_heatmap=kh)
    return pts, conf
