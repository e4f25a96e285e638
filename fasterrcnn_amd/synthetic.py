"""
Synthetic VOC2007-shaped workload (no dataset or trained checkpoint exists offline): calibrated
random VGG-16 Faster R-CNN weights, preprocessed 3x600x1000 images and ground-truth boxes, all
reproducible from integer seeds with torch's CPU generator (same torch build here and on the GPU
box, so tests, bench and the golden-vector generator see identical tensors).

Recipe: He-normal conv/linear weights, zero biases; five layers are then
scaled by fixed constants so the pipeline is exercised non-degenerately (feature map std ~1,
objectness logits std ~1, RPN deltas std ~0.3, class logits std ~3, box deltas std ~1).  The
constants were measured once with the imported reference (oracle/calibrate.py prints them) and
are frozen here so no calibration forward is needed at run time.
"""
import math

import numpy as np
import torch as t

BGR_MEANS = (103.939, 116.779, 123.680)   # models/vgg16.py:146 (reference)

# layer -> multiplier applied on top of He-normal init (oracle/calibrate.py, seed 1234, image seed 0)
CALIBRATION = {
    "_stage1_feature_extractor._block5_conv3.weight": 0.0176625132,
    "_stage2_region_proposal_network._rpn_class.weight": 0.919417192,
    "_stage2_region_proposal_network._rpn_boxes.weight": 0.19239781,
    "_stage3_detector_network._classifier.weight": 1.26394965,
    "_stage3_detector_network._regressor.weight": 0.481527901,
}

_VGG_CONVS = [("_block1_conv1", 3, 64), ("_block1_conv2", 64, 64), ("_block2_conv1", 64, 128),
              ("_block2_conv2", 128, 128), ("_block3_conv1", 128, 256), ("_block3_conv2", 256, 256),
              ("_block3_conv3", 256, 256), ("_block4_conv1", 256, 512), ("_block4_conv2", 512, 512),
              ("_block4_conv3", 512, 512), ("_block5_conv1", 512, 512), ("_block5_conv2", 512, 512),
              ("_block5_conv3", 512, 512)]


def vgg16_state_dict(seed=1234, num_classes=21, calibration=None):
    """state_dict (CPU float32) with the reference's key names."""
    cal = CALIBRATION if calibration is None else calibration
    g = t.Generator().manual_seed(int(seed))
    sd = {}

    def he(key, shape, fan_in):
        w = t.randn(shape, generator=g, dtype=t.float32) * math.sqrt(2.0 / fan_in)
        sd[key + ".weight"] = w * float(cal.get(key + ".weight", 1.0))
        sd[key + ".bias"] = t.zeros(shape[0], dtype=t.float32)

    for name, cin, cout in _VGG_CONVS:
        he("_stage1_feature_extractor." + name, (cout, cin, 3, 3), cin * 9)
    he("_stage2_region_proposal_network._rpn_conv1", (512, 512, 3, 3), 512 * 9)
    he("_stage2_region_proposal_network._rpn_class", (9, 512, 1, 1), 512)
    he("_stage2_region_proposal_network._rpn_boxes", (36, 512, 1, 1), 512)
    he("_stage3_detector_network._pool_to_feature_vector._fc1", (4096, 512 * 7 * 7), 512 * 7 * 7)
    he("_stage3_detector_network._pool_to_feature_vector._fc2", (4096, 4096), 4096)
    he("_stage3_detector_network._classifier", (num_classes, 4096), 4096)
    he("_stage3_detector_network._regressor", ((num_classes - 1) * 4, 4096), 4096)
    return sd


RESNET_BLOCKS = {"ResNet50": (3, 4, 6, 3), "ResNet101": (3, 4, 23, 3), "ResNet152": (3, 8, 36, 3)}
RGB_MEANS = (0.485, 0.456, 0.406)      # models/resnet.py:141 (reference)
RGB_STDS = (0.229, 0.224, 0.225)

# head multipliers on top of He-normal init (oracle/make_golden.py --calibrate-resnet, seed 1234, image seed 0)
RESNET_CALIBRATION = {
    "_stage2_region_proposal_network._rpn_conv1.weight": 0.613980047,
    "_stage2_region_proposal_network._rpn_class.weight": 0.461671139,
    "_stage2_region_proposal_network._rpn_boxes.weight": 0.166079862,
    "_stage3_detector_network._classifier.weight": 0.996291455,
    "_stage3_detector_network._regressor.weight": 0.27933594,
}


def resnet_state_dict(seed=1234, architecture="ResNet50", num_classes=21, calibration=None):
    """
    state_dict (CPU float32) of FasterRCNNModel over a ResNet backbone with the reference's key
    names.  He-normal convolutions; BatchNorm statistics and affine parameters are random but
    benign (gamma of every block's last BN is small so the residual stream stays bounded), which
    exercises the BN folding non-trivially.
    """
    cal = RESNET_CALIBRATION if calibration is None else calibration
    g = t.Generator().manual_seed(int(seed))
    sd = {}

    def conv(key, cout, cin, k):
        sd[key + ".weight"] = t.randn((cout, cin, k, k), generator=g, dtype=t.float32) * math.sqrt(2.0 / (cin * k * k))

    def bn(key, c, gamma_lo, gamma_hi):
        sd[key + ".weight"] = t.rand((c,), generator=g) * (gamma_hi - gamma_lo) + gamma_lo
        sd[key + ".bias"] = t.randn((c,), generator=g) * 0.1
        sd[key + ".running_mean"] = t.randn((c,), generator=g) * 0.1
        sd[key + ".running_var"] = t.rand((c,), generator=g) + 0.5
        sd[key + ".num_batches_tracked"] = t.tensor(0, dtype=t.long)

    fe = "_stage1_feature_extractor._feature_extractor."
    conv(fe + "0", 64, 3, 7)
    bn(fe + "1", 64, 0.8, 1.2)
    inplanes = 64
    blocks = RESNET_BLOCKS[architecture]
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks)):
        prefix = (fe + "%d." % (4 + li)) if li < 3 else "_stage3_detector_network._pool_to_feature_vector._layer4."
        for b in range(n):
            p = prefix + "%d." % b
            conv(p + "conv1", planes, inplanes, 1); bn(p + "bn1", planes, 0.8, 1.2)
            conv(p + "conv2", planes, planes, 3);   bn(p + "bn2", planes, 0.8, 1.2)
            # the residual stream grows with depth: ResNet-152's 50 blocks get a smaller last-BN gamma so that its feature map
            # stays in the range the (ResNet-50-calibrated) heads were tuned for (std ~6 instead of ~30, where the RPN saturates)
            g3 = (0.12, 0.24) if architecture == "ResNet152" else (0.2, 0.4)
            conv(p + "conv3", planes * 4, planes, 1); bn(p + "bn3", planes * 4, g3[0], g3[1])
            if b == 0:
                conv(p + "downsample.0", planes * 4, inplanes, 1); bn(p + "downsample.1", planes * 4, 0.6, 1.0)
            inplanes = planes * 4

    def he(key, shape, fan_in):
        sd[key + ".weight"] = t.randn(shape, generator=g, dtype=t.float32) * math.sqrt(2.0 / fan_in) * float(cal.get(key + ".weight", 1.0))
        sd[key + ".bias"] = t.zeros(shape[0], dtype=t.float32)

    he("_stage2_region_proposal_network._rpn_conv1", (1024, 1024, 3, 3), 1024 * 9)
    he("_stage2_region_proposal_network._rpn_class", (9, 1024, 1, 1), 1024)
    he("_stage2_region_proposal_network._rpn_boxes", (36, 1024, 1, 1), 1024)
    he("_stage3_detector_network._classifier", (num_classes, 2048), 2048)
    he("_stage3_detector_network._regressor", ((num_classes - 1) * 4, 2048), 2048)
    return sd


def image_rgb(seed, height=600, width=1000):
    """Image preprocessed the ResNet way (models/resnet.py:141): RGB, /255, ImageNet mean/std."""
    g = t.Generator().manual_seed(1000003 * int(seed) + 17)
    lh, lw = max(2, int(round(height / 31.6))), max(2, int(round(width / 31.25)))
    low = t.rand((1, 3, lh, lw), generator=g, dtype=t.float32)
    up = t.nn.functional.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)[0]
    means = t.tensor(RGB_MEANS, dtype=t.float32).reshape(3, 1, 1)
    stds = t.tensor(RGB_STDS, dtype=t.float32).reshape(3, 1, 1)
    return ((up - means) / stds).contiguous()


def image(seed, height=600, width=1000):
    """
    One preprocessed image, float32 (3, height, width): low-resolution uniform noise, bilinearly
    upsampled, x255, minus the BGR ImageNet means (what datasets/image.py:43-57 would produce for
    the VGG-16 backbone).
    """
    g = t.Generator().manual_seed(1000003 * int(seed) + 17)
    lh, lw = max(2, int(round(height / 31.6))), max(2, int(round(width / 31.25)))
    low = t.rand((1, 3, lh, lw), generator=g, dtype=t.float32)
    up = t.nn.functional.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)[0] * 255.0
    means = t.tensor(BGR_MEANS, dtype=t.float32).reshape(3, 1, 1)
    return (up - means).contiguous()


def image_u8(seed, height=375, width=625):
    """A DECODED image as imageio / PIL hand it to datasets/image.py: uint8 (height, width, 3) RGB, the same low-resolution noise field as
    image() / image_rgb().  375 x 625 is a VOC-sized frame that load_image's 600-pixel minimum side scales to exactly 600 x 1000."""
    g = t.Generator().manual_seed(1000003 * int(seed) + 17)
    lh, lw = max(2, int(round(height / 19.75))), max(2, int(round(width / 19.53)))
    low = t.rand((1, 3, lh, lw), generator=g, dtype=t.float32)
    up = t.nn.functional.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)[0] * 255.0
    return up.round().clamp(0, 255).to(t.uint8).permute(1, 2, 0).contiguous()


def ground_truth(seed, height=600, width=1000, num_classes=21):
    """
    1-5 ground-truth boxes for image `seed`: list of (class_index, (y1, x1, y2, x2) float32 array),
    classes uniform in 1..num_classes-1, sides >= 32 px.
    """
    rng = np.random.RandomState(7919 * int(seed) + 3)
    out = []
    for _ in range(int(rng.randint(1, 6))):
        cls = int(rng.randint(1, num_classes))
        h = float(rng.uniform(32, height * 0.8))
        w = float(rng.uniform(32, width * 0.8))
        y1 = float(rng.uniform(0, height - h))
        x1 = float(rng.uniform(0, width - w))
        out.append((cls, np.array([y1, x1, y1 + h, x1 + w], dtype=np.float32)))
    return out


# ---- stress recipes (round 5: VERDICT r4 "what's missing" 3) ------------------------------------------------------------------------------
# Everything above is ONE weight recipe (He-normal, frozen multipliers) and ONE image recipe (smooth upsampled noise).  The f32x3
# arithmetic of the HIP path scales its operands by data statistics (per tile / per filter row / per tensor), so the held-out parity
# claim is repeated on inputs built to hurt exactly that: heavy-tailed weights with outlier output channels, images with hard edges,
# saturated blocks and flat black regions, and a network one of whose activation channels sits 2^12 above the tensor's median.
# tests/golden/stress/ holds the imported reference's outputs and the float64 truth for them (oracle/make_stress.py).
STRESS_KINDS = ("heavy", "edges", "heavy_edges", "outlier")

# head / feature multipliers measured by oracle/make_stress.py --calibrate with the imported reference (same targets as CALIBRATION /
# RESNET_CALIBRATION: feature map std 1, objectness logits std 1, RPN deltas std 0.3, class logits std 3, box deltas std 1), frozen here
STRESS_CALIBRATION = {
    ("VGG16", "heavy", 7001): {
        "_stage1_feature_extractor._block5_conv3.weight": 0.000142300062,
        "_stage2_region_proposal_network._rpn_class.weight": 0.907114254,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.216679272,
        "_stage3_detector_network._classifier.weight": 1.11351381,
        "_stage3_detector_network._regressor.weight": 0.470355192,
    },
    ("VGG16", "heavy", 7002): {
        "_stage1_feature_extractor._block5_conv3.weight": 0.00164024396,
        "_stage2_region_proposal_network._rpn_class.weight": 0.802515284,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.162345209,
        "_stage3_detector_network._classifier.weight": 1.60085872,
        "_stage3_detector_network._regressor.weight": 0.461532045,
    },
    ("VGG16", "outlier", 7003): {
        "_stage1_feature_extractor._block5_conv3.weight": 4.0644685e-05,
        "_stage2_region_proposal_network._rpn_class.weight": 0.565443829,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.178230513,
        "_stage3_detector_network._classifier.weight": 1.53737983,
        "_stage3_detector_network._regressor.weight": 0.510813817,
    },
    ("VGG16", "outlier", 7004): {
        "_stage1_feature_extractor._block5_conv3.weight": 2.95892116e-05,
        "_stage2_region_proposal_network._rpn_class.weight": 0.642786937,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.17912604,
        "_stage3_detector_network._classifier.weight": 1.73594005,
        "_stage3_detector_network._regressor.weight": 0.443077818,
    },
    ("ResNet50", "heavy", 7101): {
        "_stage2_region_proposal_network._rpn_conv1.weight": 0.613042401,
        "_stage2_region_proposal_network._rpn_class.weight": 0.711188039,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.165805606,
        "_stage3_detector_network._classifier.weight": 0.845735559,
        "_stage3_detector_network._regressor.weight": 0.346658095,
    },
    ("ResNet50", "outlier", 7103): {
        "_stage2_region_proposal_network._rpn_conv1.weight": 0.00854886377,
        "_stage2_region_proposal_network._rpn_class.weight": 0.663629708,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.179113533,
        "_stage3_detector_network._classifier.weight": 0.015837789,
        "_stage3_detector_network._regressor.weight": 0.00438118444,
    },
    ("ResNet50", "outlier", 7104): {
        "_stage2_region_proposal_network._rpn_conv1.weight": 0.0140314262,
        "_stage2_region_proposal_network._rpn_class.weight": 0.644335767,
        "_stage2_region_proposal_network._rpn_boxes.weight": 0.200332335,
        "_stage3_detector_network._classifier.weight": 0.0252646431,
        "_stage3_detector_network._regressor.weight": 0.00656980969,
    },
}


def _student_t3(shape, g):
    """Student-t, 3 degrees of freedom, scaled to unit variance (heavy tails: the largest of 10^6 draws is ~50-100 sigma)."""
    z = t.randn(shape, generator=g, dtype=t.float32)
    chi = t.zeros(shape, dtype=t.float32)
    for _ in range(3):
        chi += t.randn(shape, generator=g, dtype=t.float32) ** 2
    return z / t.sqrt(chi / 3.0) / math.sqrt(3.0)


def _lognormal_gain(n, g, sigma=0.5):
    """per-output-channel gains exp(sigma N(0,1)), mean square 1"""
    return t.exp(sigma * t.randn((n,), generator=g, dtype=t.float32)) / math.exp(sigma * sigma)


def stress_weights_kind(kind):
    """the weight recipe of a stress kind: 'he' (the standard recipe), 'heavy' or 'outlier'"""
    return {"heavy": "heavy", "edges": "he", "heavy_edges": "heavy", "outlier": "outlier"}[kind]


def stress_vgg16_state_dict(seed, kind, num_classes=21, calibration=None):
    """VGG-16 Faster R-CNN weights for stress kind `kind` (reference key names).
    heavy:   Student-t(3) convolution / fc2 weights, log-normal per-output-channel gains on every layer, two output channels x64 in
             conv2_2, conv3_3 and conv4_2;
    outlier: the standard He-normal recipe with ONE output channel of conv3_2 x4096 (its activations sit 2^12 above the tensor's median:
             the input of conv3_3 and, through it, everything behind);
    he:      vgg16_state_dict(seed) itself."""
    wk = stress_weights_kind(kind)
    cal = STRESS_CALIBRATION.get(("VGG16", wk, int(seed)), {}) if calibration is None else calibration
    if wk == "he":
        return vgg16_state_dict(seed, num_classes)
    g = t.Generator().manual_seed(int(seed))
    sd = {}

    def layer(key, shape, fan_in, heavy_tail):
        if wk == "heavy":
            w = (_student_t3(shape, g) if heavy_tail else t.randn(shape, generator=g, dtype=t.float32)) * math.sqrt(2.0 / fan_in)
            w = w * _lognormal_gain(shape[0], g).reshape((-1,) + (1,) * (len(shape) - 1))
        else:
            w = t.randn(shape, generator=g, dtype=t.float32) * math.sqrt(2.0 / fan_in)
        sd[key + ".weight"] = w
        sd[key + ".bias"] = t.zeros(shape[0], dtype=t.float32)

    for name, cin, cout in _VGG_CONVS:
        layer("_stage1_feature_extractor." + name, (cout, cin, 3, 3), cin * 9, True)
    layer("_stage2_region_proposal_network._rpn_conv1", (512, 512, 3, 3), 512 * 9, True)
    layer("_stage2_region_proposal_network._rpn_class", (9, 512, 1, 1), 512, False)
    layer("_stage2_region_proposal_network._rpn_boxes", (36, 512, 1, 1), 512, False)
    layer("_stage3_detector_network._pool_to_feature_vector._fc1", (4096, 512 * 7 * 7), 512 * 7 * 7, False)
    layer("_stage3_detector_network._pool_to_feature_vector._fc2", (4096, 4096), 4096, True)
    layer("_stage3_detector_network._classifier", (num_classes, 4096), 4096, False)
    layer("_stage3_detector_network._regressor", ((num_classes - 1) * 4, 4096), 4096, False)
    if wk == "heavy":
        for name in ("_block2_conv2", "_block3_conv3", "_block4_conv2"):
            w = sd["_stage1_feature_extractor." + name + ".weight"]
            for c in t.randperm(w.shape[0], generator=g)[:2].tolist():
                w[c] *= 64.0
    else:
        w = sd["_stage1_feature_extractor._block3_conv2.weight"]
        w[int(t.randint(0, w.shape[0], (1,), generator=g))] *= 4096.0
    for k, v in cal.items():
        sd[k] = sd[k] * float(v)
    return sd


def stress_resnet_state_dict(seed, kind, architecture="ResNet50", num_classes=21, calibration=None):
    """ResNet Faster R-CNN weights for stress kind `kind`.
    heavy:   Student-t(3) convolution weights; the BatchNorm gammas get log-normal gains and two channels of layer1's last bn3 (the
             residual stream every later block reads) x64;
    outlier: the standard recipe with ONE channel of layer1's last bn3 x4096 (gamma and beta): a residual-stream channel 2^12 above
             the tensor's median -- the per-TENSOR operand scale of conv_gather_x3_kernel sees it in every layer2 / layer3 convolution;
    he:      resnet_state_dict(seed) itself."""
    wk = stress_weights_kind(kind)
    cal = STRESS_CALIBRATION.get((architecture, wk, int(seed)), {}) if calibration is None else calibration
    if wk == "he":
        return resnet_state_dict(seed, architecture, num_classes)
    sd = resnet_state_dict(seed, architecture, num_classes, calibration={k: 1.0 for k in RESNET_CALIBRATION})
    g = t.Generator().manual_seed(7 * int(seed) + 1)
    last = "_stage1_feature_extractor._feature_extractor.4.%d.bn3." % (RESNET_BLOCKS[architecture][0] - 1)
    if wk == "heavy":
        for k in sorted(sd):
            if k.endswith(".weight") and sd[k].dim() == 4 and "_rpn_class" not in k and "_rpn_boxes" not in k:
                std = float(sd[k].std())
                sd[k] = _student_t3(tuple(sd[k].shape), g) * std
            elif k.endswith(".weight") and sd[k].dim() == 1 and (".bn" in k or "downsample.1" in k or k.endswith("_feature_extractor.1.weight")):
                sd[k] = sd[k] * _lognormal_gain(sd[k].shape[0], g)
        for c in t.randperm(sd[last + "weight"].shape[0], generator=g)[:2].tolist():
            sd[last + "weight"][c] *= 64.0
            sd[last + "bias"][c] *= 64.0
    else:
        c = int(t.randint(0, sd[last + "weight"].shape[0], (1,), generator=g))
        sd[last + "weight"][c] *= 4096.0
        sd[last + "bias"][c] = sd[last + "bias"][c].abs() * 4096.0
    for k, v in cal.items():
        sd[k] = sd[k] * float(v)
    return sd


def stress_frame_u8(seed, kind, height=600, width=1000):
    """uint8 (3, height, width) frame for stress kind `kind`: 'edges' / 'heavy_edges' / 'outlier' = piecewise-constant rectangles with hard
    edges over a smooth-noise background, saturated (255) blocks, flat black (0) regions and one-pixel lines; 'heavy' = the smooth
    noise of image().  The flat regions are at most 160 px in one dimension -- below the 196-px receptive field of conv5_3 / the RPN
    trunk, so no two anchors see bit-identical inputs (exact score ties would test the sort's tie rule, not the arithmetic) -- while
    conv1_1 ... conv3_3 (receptive fields 3 ... 40 px) run over truly constant regions."""
    g = t.Generator().manual_seed(1000003 * int(seed) + 29)
    lh, lw = max(2, int(round(height / 31.6))), max(2, int(round(width / 31.25)))
    low = t.rand((1, 3, lh, lw), generator=g, dtype=t.float32)
    img = (t.nn.functional.interpolate(low, size=(height, width), mode="bilinear", align_corners=False)[0] * 255.0).round().clamp(0, 255)
    if kind == "heavy":
        return img.to(t.uint8)
    n = 36
    ys = t.randint(0, height, (n,), generator=g).tolist()
    xs = t.randint(0, width, (n,), generator=g).tolist()
    hs = t.randint(8, 161, (n,), generator=g).tolist()
    ws = t.randint(8, 401, (n,), generator=g).tolist()
    mode = t.randint(0, 4, (n,), generator=g).tolist()          # 0 black, 1 white, 2 one saturated channel, 3 random flat colour
    col = t.randint(0, 256, (n, 3), generator=g).to(t.float32)
    for i in range(n):
        y0, x0 = ys[i], xs[i]
        y1, x1 = min(height, y0 + hs[i]), min(width, x0 + ws[i])
        if mode[i] == 0:
            c = t.zeros(3)
        elif mode[i] == 1:
            c = t.full((3,), 255.0)
        elif mode[i] == 2:
            c = t.zeros(3); c[i % 3] = 255.0
        else:
            c = col[i]
        img[:, y0:y1, x0:x1] = c.reshape(3, 1, 1)
    for i in range(6):                                          # one-pixel lines, alternately white and black
        v = 255.0 if i % 2 == 0 else 0.0
        if i < 3:
            img[:, ys[i]:ys[i] + 1, :] = v
        else:
            img[:, :, xs[i]:xs[i] + 1] = v
    return img.to(t.uint8)


def stress_image(seed, kind, height=600, width=1000):
    """stress_frame_u8 preprocessed the VGG-16 way (BGR channel order is immaterial for a synthetic frame): float32 minus the BGR means"""
    means = t.tensor(BGR_MEANS, dtype=t.float32).reshape(3, 1, 1)
    return (stress_frame_u8(seed, kind, height, width).to(t.float32) - means).contiguous()


def stress_image_rgb(seed, kind, height=600, width=1000):
    """stress_frame_u8 preprocessed the ResNet way: /255, ImageNet mean / std"""
    means = t.tensor(RGB_MEANS, dtype=t.float32).reshape(3, 1, 1)
    stds = t.tensor(RGB_STDS, dtype=t.float32).reshape(3, 1, 1)
    return ((stress_frame_u8(seed, kind, height, width).to(t.float32) / 255.0 - means) / stds).contiguous()
