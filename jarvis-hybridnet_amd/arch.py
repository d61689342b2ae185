"""Architecture tables of the networks on the hot path (product side).

The same numbers drive (a) the parameter trees of the Python modules, so that
reference `.pth` state dicts load unchanged, and (b) the C++ plan builder in
`csrc/` (which re-derives them from `model_size` and is cross-checked against
this file by tests/test_state_spec.py).

Reference: jarvis/efficienttrack/model.py:34-51 (sizes),
jarvis/efficienttrack/utils.py:76-112,152-155,267-272 (scaling rules and
stage strings), jarvis/efficienttrack/efficientnet.py:29-88 (block layout),
jarvis/hybridnet/v2vnet.py:86-96 (V2V layout).
"""
import math

SIZES = {
    "small": {"width": 0.5, "depth": 0.5, "fpn": 56, "cells": 3, "head": 64},
    "medium": {"width": 1.0, "depth": 1.0, "fpn": 88, "cells": 4, "head": 88},
    "large": {"width": 1.1, "depth": 1.2, "fpn": 160, "cells": 6, "head": 160},
}
SIZE_IDS = {"small": 0, "medium": 1, "large": 2}

# kernel, repeats, in, out, expand, stride  (base EfficientNet stages)
STAGES = ((3, 1, 32, 16, 1, 1), (3, 2, 16, 24, 6, 2), (5, 2, 24, 40, 6, 2),
          (3, 3, 40, 80, 6, 2), (5, 3, 80, 112, 6, 1), (5, 4, 112, 192, 6, 2),
          (3, 1, 192, 320, 6, 1))

BIFPN_SEPCONVS = ("conv6_up", "conv5_up", "conv4_up", "conv3_up",
                  "conv4_down", "conv5_down", "conv6_down", "conv7_down")


def scale_channels(c, width):
    c = c * width
    out = max(8, int(c + 4) // 8 * 8)
    if out < 0.9 * c:
        out += 8
    return int(out)


def trunk(model_size):
    """(stem_channels, [block dicts], [tap block indices]) of the trunk after
    it has been cut behind the last stride-16 block."""
    s = SIZES[model_size]
    blocks = []
    for stage, (k, rep, cin, cout, e, stride) in enumerate(STAGES):
        cin, cout = scale_channels(cin, s["width"]), scale_channels(cout, s["width"])
        for r in range(int(math.ceil(s["depth"] * rep))):
            ci = cin if r == 0 else cout
            blocks.append({"stage": stage, "k": k, "cin": ci, "cout": cout,
                           "stride": stride if r == 0 else 1, "expand": e,
                           "mid": ci * e, "squeeze": max(1, int(ci * 0.25)),
                           "dense": stage < 4})
    s2 = [i for i, b in enumerate(blocks) if b["stride"] == 2]
    # taps sit right before the 2nd, 3rd and 4th resolution drop
    taps = [i - 1 for i in s2[1:4]]
    return scale_channels(32, s["width"]), blocks[:taps[-1] + 1], taps


def efficienttrack_params(model_size, num_joints):
    s = SIZES[model_size]
    W, Fh = s["fpn"], s["head"]
    stem, blocks, taps = trunk(model_size)
    tapc = [blocks[t]["cout"] for t in taps]
    out = [("weights_cat", (3,))]
    for cell in range(s["cells"]):
        p = "bifpn.%d." % cell
        for lvl in (6, 5, 4, 3):
            out.append((p + "p%d_w1" % lvl, (2,)))
        for lvl in (4, 5, 6):
            out.append((p + "p%d_w2" % lvl, (3,)))
        out.append((p + "p7_w2", (2,)))
        for name in BIFPN_SEPCONVS:
            out.append((p + name + ".depthwise_conv.weight", (W, 1, 3, 3)))
            out.append((p + name + ".pointwise_conv.weight", (W, W, 1, 1)))
            out.append((p + name + ".pointwise_conv.bias", (W,)))
        if cell == 0:
            for name, c in (("p5_down_channel", tapc[2]), ("p4_down_channel", tapc[1]),
                            ("p3_down_channel", tapc[0]), ("p5_to_p6", tapc[2]),
                            ("p4_down_channel_2", tapc[1]), ("p5_down_channel_2", tapc[2])):
                out.append((p + name + ".0.weight", (W, c, 1, 1)))
                out.append((p + name + ".0.bias", (W,)))
    out.append(("backbone_net.model._conv_stem.weight", (stem, 3, 3, 3)))
    for i, b in enumerate(blocks):
        p = "backbone_net.model._blocks.%d." % i
        if b["expand"] != 1:
            out.append((p + "_expand_conv.weight", (b["mid"], b["cin"], 1, 1)))
        dw_in = b["cin"] if b["dense"] else 1
        out.append((p + "_depthwise_conv.weight", (b["mid"], dw_in, b["k"], b["k"])))
        out.append((p + "_se_reduce.weight", (b["squeeze"], b["mid"], 1, 1)))
        out.append((p + "_se_reduce.bias", (b["squeeze"],)))
        out.append((p + "_se_expand.weight", (b["mid"], b["squeeze"], 1, 1)))
        out.append((p + "_se_expand.bias", (b["mid"],)))
        out.append((p + "_project_conv.weight", (b["cout"], b["mid"], 1, 1)))
    out += [("first_conv.depthwise_conv.weight", (W, 1, 3, 3)),
            ("first_conv.pointwise_conv.weight", (Fh, W, 1, 1)),
            ("first_conv.pointwise_conv.bias", (Fh,)),
            ("deconv1.weight", (Fh, num_joints, 4, 4)),
            ("final_conv1.weight", (num_joints, Fh, 3, 3)),
            ("final_conv2.weight", (num_joints, Fh, 1, 1))]
    return out


def v2v_params(num_joints):
    J = num_joints
    out = []

    def c3(name, co, ci, k):
        out.append((name + ".weight", (co, ci, k, k, k)))
        out.append((name + ".bias", (co,)))

    def res(name, c):
        c3(name + ".res_branch.0", c, c, 3)
        c3(name + ".res_branch.3", c, c, 3)

    c3("front_layers.0.block.0", 2 * J, J, 3)
    res("front_layers.1", 2 * J)
    c3("encoder_decoder.encoder_pool1.block.0", 4 * J, 2 * J, 2)
    res("encoder_decoder.mid_res", 4 * J)
    out.append(("encoder_decoder.decoder_upsample1.block.0.weight", (4 * J, 2 * J, 2, 2, 2)))
    out.append(("encoder_decoder.decoder_upsample1.block.0.bias", (2 * J,)))
    res("encoder_decoder.decoder_res1", 2 * J)
    res("encoder_decoder.skip_res1", 2 * J)
    c3("output_layer", J, 2 * J, 1)
    return out


def hybridnet_params(model_size, num_joints):
    return ([("effTrack." + k, s) for k, s in efficienttrack_params(model_size, num_joints)]
            + [("v2vNet." + k, s) for k, s in v2v_params(num_joints)])
