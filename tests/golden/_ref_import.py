"""Import shim for the upstream reference (THIS container only).

The reference package cannot be imported whole here (graphviz / peft / wandb /
evaluate / timm / torchvision are absent and one vendored file needs a symbol
removed from the installed transformers), so the hot-path modules are imported
piecewise.  Used ONLY by gen_golden.py to emit the
fixtures committed next to it; nothing under tests/, bench.py or the package
imports this at run time on the GPU box (/root/reference does not exist there).
"""
import importlib
import os
import sys
import tempfile
import types

REF_ROOT = "/root/reference"
REF_PKG = os.path.join(REF_ROOT, "src", "quantized_training")


def _write(path, text):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(text)


def _stub_third_party():
    d = tempfile.mkdtemp(prefix="qt_ref_stubs_")
    _write(os.path.join(d, "graphviz", "__init__.py"), "class Digraph: pass\n")
    _write(os.path.join(d, "wandb", "__init__.py"), "run = None\n")
    _write(os.path.join(d, "peft", "__init__.py"), "")
    _write(os.path.join(d, "peft", "tuners", "__init__.py"), "")
    _write(os.path.join(d, "peft", "tuners", "lora.py"),
           "import torch\nclass Linear(torch.nn.Linear): pass\n")
    _write(os.path.join(d, "peft", "utils", "__init__.py"), "")
    _write(os.path.join(d, "peft", "utils", "other.py"),
           "def transpose(w, f):\n    return w.T if f else w\n")
    sys.path.insert(0, d)


def load_reference(with_quantize=False):
    """Returns a namespace with the reference hot-path modules."""
    if not os.path.isdir(REF_PKG):
        raise RuntimeError("reference checkout not present: " + REF_PKG)
    for k in [k for k in sys.modules if k == "quantized_training" or k.startswith("quantized_training.")]:
        del sys.modules[k]
    _stub_third_party()
    pkg = types.ModuleType("quantized_training")
    pkg.__path__ = [REF_PKG]
    sys.modules["quantized_training"] = pkg
    q = importlib.import_module("quantized_training.quantizer.quantizer")
    pkg.per_tensor_symmetric = q.QScheme.PER_TENSOR_SYMMETRIC
    pkg.per_channel_symmetric = q.QScheme.PER_CHANNEL_SYMMETRIC
    pkg.microscaling = q.QScheme.MICROSCALING
    pkg.group_wise_affine = q.QScheme.GROUP_WISE_AFFINE
    ns = types.SimpleNamespace(pkg=pkg, quantizer=q)
    ns.fake_quantize = importlib.import_module("quantized_training.fake_quantize")
    ns.decomposed = importlib.import_module("quantized_training.decomposed")
    ns.fp8 = importlib.import_module("quantized_training.fp8")
    ns.posit = importlib.import_module("quantized_training.posit")
    ns.qconfig = importlib.import_module("quantized_training.qconfig")
    ns.training_args = importlib.import_module("quantized_training.training_args")
    if with_quantize:
        import torch
        import torch.nn as nn
        mods = types.ModuleType("quantized_training.modules")
        mods.__path__ = [os.path.join(REF_PKG, "modules")]
        sys.modules["quantized_training.modules"] = mods
        sm = importlib.import_module("quantized_training.modules.softmax")
        mods.Softmax = sm.Softmax
        mods.modeling_bert = types.ModuleType("quantized_training.modules.modeling_bert")
        mods.modeling_mobilebert = types.ModuleType("quantized_training.modules.modeling_mobilebert")
        qz = types.ModuleType("quantized_training.modules.quantizable")
        qz.__path__ = [os.path.join(REF_PKG, "modules", "quantizable")]
        sys.modules["quantized_training.modules.quantizable"] = qz
        fm = importlib.import_module("quantized_training.modules.quantizable.functional_modules")
        qz.AddFunctional, qz.MulFunctional, qz.MatmulFunctional = (
            fm.AddFunctional, fm.MulFunctional, fm.MatmulFunctional)
        mods.quantizable = qz
        nnqat = importlib.import_module("quantized_training.modules.qat")
        mods.qat = nnqat
        qm = types.ModuleType("quantized_training.quantization_mappings")
        qm.DEFAULT_QAT_MODULE_MAPPINGS = {nn.Linear: nnqat.Linear}
        qm.TRANSFORMER_MODULE_MAPPINGS = {}
        qm.QCONFIG_PROPAGATE_MODULE_CLASS_LIST = {
            "activation": [nn.ReLU, nn.GELU, nn.Softmax],
            "gemm": [nn.Conv1d, nn.Conv2d, nn.Conv3d, nn.Linear, qz.MatmulFunctional],
            "layernorm": [nn.LayerNorm],
            "residual": [qz.AddFunctional],
            "scaling": [qz.MulFunctional],
        }
        sys.modules["quantized_training.quantization_mappings"] = qm
        ns.quantize = importlib.import_module("quantized_training.quantize")
        try:      # PT2E: the accelerator package must be imported first (circular import otherwise)
            importlib.import_module("quantized_training.codegen")
            ns.quantize_pt2e = importlib.import_module("quantized_training.quantize_pt2e")
        except Exception as e:  # noqa: BLE001
            ns.quantize_pt2e = None
            ns.pt2e_error = e
        ns.functional_modules = fm
        ns.nnqat = nnqat
    return ns
