"""
Checkpoint I/O, mirroring pytorch/FasterRCNN/state.py:178-288: load
trained weights into the HIP-backed FasterRCNNModel from the file formats the reference reads.

  1. the reference's own checkpoints  {"epoch": int, "model_state_dict": {...}}  (state.py:259-264,
     written by __main__.py:195-198): the state_dict keys of this package are identical, so these
     load as-is (VGG-16 and ResNet alike);
  2. Caffe VGG-16 `.pth` files (torchvision-style keys features.N / classifier.N, state.py:178-219);
  3. the author's Keras VGG-16 `.h5` files (state.py:61-176) when h5py is installed.

One deliberate difference: state.py:197-198 maps fc1/fc2 to `_stage3_detector_network._fc1/_fc2`,
which are not the model's keys (the live ones are `..._pool_to_feature_vector._fc1/_fc2`), so the
reference's strict load raises after copying the conv layers and the FC weights are silently lost
(the exception is printed and swallowed, :267-272).  Here the FC layers are mapped to the live keys
and the partial state is loaded non-strictly, with the untouched keys reported.
"""
import torch as t

_CAFFE_VGG16 = {
    "features.0": "_stage1_feature_extractor._block1_conv1", "features.2": "_stage1_feature_extractor._block1_conv2",
    "features.5": "_stage1_feature_extractor._block2_conv1", "features.7": "_stage1_feature_extractor._block2_conv2",
    "features.10": "_stage1_feature_extractor._block3_conv1", "features.12": "_stage1_feature_extractor._block3_conv2",
    "features.14": "_stage1_feature_extractor._block3_conv3", "features.17": "_stage1_feature_extractor._block4_conv1",
    "features.19": "_stage1_feature_extractor._block4_conv2", "features.21": "_stage1_feature_extractor._block4_conv3",
    "features.24": "_stage1_feature_extractor._block5_conv1", "features.26": "_stage1_feature_extractor._block5_conv2",
    "features.28": "_stage1_feature_extractor._block5_conv3",
    "classifier.0": "_stage3_detector_network._pool_to_feature_vector._fc1",
    "classifier.3": "_stage3_detector_network._pool_to_feature_vector._fc2",
}


def vgg16_state_from_caffe(caffe):
    """torchvision/Caffe VGG-16 state dict -> partial FasterRCNNModel state dict (conv blocks + fc1/fc2)."""
    state, missing = {}, []
    for src, dst in _CAFFE_VGG16.items():
        if (src + ".weight") in caffe and (src + ".bias") in caffe:
            state[dst + ".weight"] = caffe[src + ".weight"]
            state[dst + ".bias"] = caffe[src + ".bias"]
        else:
            missing.append(src)
    if len(missing) == len(_CAFFE_VGG16):
        raise ValueError("not a Caffe VGG-16 model")
    return state, missing


def vgg16_state_from_keras(filepath):
    """The author's Keras VGG-16 h5 layout (state.py:61-176); needs h5py."""
    import h5py
    import numpy as np
    state, missing = {}, []

    def layer(f, name):
        group = f["model_weights/" + name]
        for key in group:
            if key.startswith("conv") or key.startswith("dense"):
                w = np.array(group[key]["kernel:0"]).astype(np.float32)
                b = np.array(group[key]["bias:0"]).astype(np.float32)
                return t.from_numpy(w), t.from_numpy(b)
        return None, None

    with h5py.File(filepath, "r") as f:
        for block, convs in ((1, 2), (2, 2), (3, 3), (4, 3), (5, 3)):
            for i in range(1, convs + 1):
                name = "block%d_conv%d" % (block, i)
                w, b = layer(f, name)
                if w is None:
                    missing.append(name)
                    continue
                key = "_stage1_feature_extractor._" + name
                state[key + ".weight"] = w.permute(3, 2, 0, 1).contiguous()     # Keras (kh,kw,in,out) -> OIHW
                state[key + ".bias"] = b
        w, b = layer(f, "fc1")
        if w is not None:
            # Keras flattens (7,7,512) channels-last; PyTorch expects (512,7,7) (state.py:150-160)
            w = w.reshape(7, 7, 512, 4096).permute(2, 0, 1, 3).reshape(-1, 4096).permute(1, 0).contiguous()
            state["_stage3_detector_network._pool_to_feature_vector._fc1.weight"] = w
            state["_stage3_detector_network._pool_to_feature_vector._fc1.bias"] = b
        else:
            missing.append("fc1")
        w, b = layer(f, "fc2")
        if w is not None:
            state["_stage3_detector_network._pool_to_feature_vector._fc2.weight"] = w.permute(1, 0).contiguous()
            state["_stage3_detector_network._pool_to_feature_vector._fc2.bias"] = b
        else:
            missing.append("fc2")
    return state, missing


def load(model, filepath):
    """
    Loads weights from `filepath` into `model` (state.py:221-272).  Returns the list of model keys that
    the file did not provide (empty for a complete checkpoint).
    """
    state, partial = None, False
    try:
        state, missing = vgg16_state_from_keras(filepath)
        partial = True
        print("Loaded initial VGG-16 layer weights from Keras model '%s'" % filepath)
    except Exception:
        state = None
    blob = None
    if state is None:
        blob = t.load(filepath, map_location="cpu")
        if isinstance(blob, dict) and "model_state_dict" not in blob:
            try:
                state, missing = vgg16_state_from_caffe(blob)
                partial = True
                print("Loaded initial VGG-16 layer weights from Caffe model '%s'" % filepath)
            except ValueError:
                state = None
    if state is None:
        if not isinstance(blob, dict) or "model_state_dict" not in blob:
            raise KeyError("Model state file '%s' is missing top-level key 'model_state_dict'" % filepath)
        state = blob["model_state_dict"]
    result = model.load_state_dict(state, strict=not partial)
    not_loaded = list(result.missing_keys) if partial else []
    if partial and result.unexpected_keys:
        raise KeyError("unexpected keys in '%s': %s" % (filepath, result.unexpected_keys[:4]))
    print("Loaded initial weights from '%s'%s" % (filepath, (" (%d model tensors keep their initialisation)" % len(not_loaded)) if not_loaded else ""))
    return not_loaded


def save(model, filepath, epoch=0):
    """The reference's checkpoint format (__main__.py:195-198, 212-214)."""
    t.save({"epoch": epoch, "model_state_dict": model.state_dict()}, filepath)


class BestWeightsTracker:
    """state.py:274-288."""
    def __init__(self, filepath):
        self._filepath = filepath
        self._best_state = None
        self._best_mAP = 0

    def on_epoch_end(self, model, epoch, mAP):
        if mAP > self._best_mAP:
            self._best_mAP = mAP
            self._best_state = {"epoch": epoch, "model_state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}}

    def save_best_weights(self, model):
        if self._best_state is not None:
            t.save(self._best_state, self._filepath)
            print("Saved best model weights (Mean Average Precision = %1.2f%%) to '%s'" % (self._best_mAP, self._filepath))
