import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def sd_cpu():
    """The calibrated synthetic VGG-16 Faster R-CNN state_dict (CPU), seed 1234."""
    from fasterrcnn_amd import synthetic
    return synthetic.vgg16_state_dict(1234)


@pytest.fixture(scope="session")
def gpu_model(sd_cpu):
    """FasterRCNNModel with the synthetic weights on cuda:0 (GPU tests only)."""
    import torch
    from fasterrcnn_amd.models.faster_rcnn import FasterRCNNModel
    from fasterrcnn_amd.models.vgg16 import VGG16Backbone
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    model = FasterRCNNModel(num_classes=21, backbone=VGG16Backbone(dropout_probability=0.0))
    model.load_state_dict(sd_cpu, strict=True)
    return model.cuda().eval()
