"""The C-ABI library loads, exports every symbol include/frcnn_hip.h declares, and validates
arguments without touching a GPU (no compute calls here)."""
import ctypes as C
import os
import re

from fasterrcnn_amd import _native as nv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "frcnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frcnn_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(nv.SYMBOLS) == syms


def test_library_exports_every_header_symbol():
    lib = nv.lib()
    raw = C.CDLL(nv.LIB_PATH)
    for name in header_symbols():
        assert hasattr(raw, name), name
    assert lib.frcnn_abi_version() == nv.ABI_VERSION
    assert lib.frcnn_error_string(0) == b"ok"
    assert lib.frcnn_error_string(-1) == b"invalid argument"


def test_argument_validation_without_gpu():
    lib = nv.lib()
    assert lib.frcnn_anchors(600, 1000, 37, 62, 16, None, None, None) == -1
    assert lib.frcnn_conv3x3_nhwc(None, None, None, None, 8, 8, 16, 64, 0, None, 0, None) == -1
    assert lib.frcnn_linear(None, 16, None, None, None, 16, 1, 1, 16, 0, None, 0, None) == -1
    assert lib.frcnn_roi_pool(None, 1, 1, 4, None, None, 1, 7, 0.0625, None, None) == -1
    assert lib.frcnn_detections(None, None, None, None, 300, 21, 600, 1000, 0.05, 0.3, None, None, None) == -1
    assert lib.frcnn_conv_nhwc_x3g(None, None, None, None, None, 1, 8, 8, 32, 64, 1, 1, 0, 0, None, None, None, None, 0, None) == -1
    assert lib.frcnn_conv_nhwc_x3g_tickets(None, None, None, None, None, 1, 8, 8, 32, 64, 1, 1, 0, 0, None, None, None, None, 0, None, None) == -1
    assert lib.frcnn_pack_conv_x3g_weights(None, None, None, 1, 64, 64, None) == -1
    assert lib.frcnn_tensor_absmax(None, 16, None, None) == -1
    handle = C.c_void_p()
    assert lib.frcnn_ctx_create(C.byref(handle), 4, 4, 300) == -1          # image too small
    assert lib.frcnn_ctx_create(C.byref(handle), 600, 1000, 100000) == -1  # too many rois
    assert lib.frcnn_ctx_bytes(None) == 0
    lib.frcnn_ctx_destroy(None)                                            # must be a no-op


def test_linear_workspace_plan_is_deterministic():
    lib = nv.lib()
    # fc1 of one image's 300 RoIs: all rows in one block tile, 32 column blocks, split-K 8
    assert lib.frcnn_linear_workspace_bytes(300, 4096, 25088) == 8 * 300 * 4096 * 4
    assert lib.frcnn_linear_workspace_bytes(300, 4096, 4096) == 8 * 300 * 4096 * 4
    assert lib.frcnn_linear_workspace_bytes(1, 1, 16) == 0
    # conv: the 37x62 block-5 / RPN-trunk layers split 8-way, the 600x1000 layer does not
    assert lib.frcnn_conv3x3_workspace_bytes(37, 62, 512, 512) == 8 * 37 * 62 * 512 * 4
    assert lib.frcnn_conv3x3_workspace_bytes(600, 1000, 64, 64) == 0


def test_library_load_puts_torchs_hip_runtime_first():
    """libfrcnn_hip.so must bind to the HIP runtime torch bundles (one runtime per process): loading it imports torch first."""
    import subprocess
    import sys
    code = ("import sys; from fasterrcnn_amd import _native as nv; assert 'torch' not in sys.modules; nv.lib(); "
            "assert 'torch' in sys.modules; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_winograd_layer_rule_matches_binding():
    lib = nv.lib()
    for cin in (3, 16, 64, 120, 128, 256, 512, 1024):
        for cout in (64, 128, 200, 256, 512, 1024):
            assert bool(lib.frcnn_conv3x3_uses_winograd(cin, cout)) == nv.uses_winograd(cin, cout), (cin, cout)
    for width in (64, 128, 256, 512):
        for stride in (1, 2):
            assert bool(lib.frcnn_resnet_block_uses_winograd(width, stride)) == nv.resnet_block_uses_winograd(width, stride)
    assert nv.resnet_block_uses_winograd(512, 1) and not nv.resnet_block_uses_winograd(512, 2) and nv.resnet_block_uses_winograd(256, 1) \
        and not nv.resnet_block_uses_winograd(128, 1)
    for n_maps in (1, 300):
        for width in (32, 64, 128, 256, 512):
            for stride in (1, 2):
                assert bool(lib.frcnn_resnet_block_uses_winograd_fused(n_maps, width, stride)) == \
                    nv.resnet_block_uses_winograd_fused(n_maps, width, stride)
    assert nv.resnet_block_uses_winograd_fused(1, 64, 1) and not nv.resnet_block_uses_winograd_fused(300, 512, 1) \
        and not nv.resnet_block_uses_winograd_fused(1, 256, 2)
    # three-launch form (round 1; still used for ResNet's per-RoI maps and by the train step): conv3_1 .. conv5_3 and the RPN trunk
    assert nv.uses_winograd(128, 256) and nv.uses_winograd(512, 512) and not nv.uses_winograd(128, 128) and not nv.uses_winograd(64, 128)
