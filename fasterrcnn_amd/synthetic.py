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
