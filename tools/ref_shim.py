"""Import shim for the upstream reference (survey container only).

Loads ``sylber.model.sylber`` and ``sylber.utils.segment_utils`` from the read-only
checkout at /root/reference WITHOUT executing ``sylber/__init__.py`` (which pulls in
resynthesis dependencies that are not installed) and with ``torchaudio`` stubbed
(only used for file I/O, sylber/model/sylber.py:83-85).  Contains no reference code.
Never imported by the product, by tests marked gpu, by bench.py or by smoke():
/root/reference does not exist on the GPU box.
"""
import importlib
import os
import sys
import tempfile
import types

REFERENCE_ROOT = os.environ.get("SYLBER_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "sylber"))


def load():
    """Returns (ref_model_module, ref_segment_utils_module, hubert_config_dir)."""
    if not available():
        raise RuntimeError("reference checkout not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True  # the reference dir is read-only
    import transformers  # noqa: F401  (must be imported before the torchaudio stub)
    from transformers import HubertConfig, HubertModel, BertModel, BertConfig  # noqa: F401

    if "sylber" not in sys.modules:
        pkg = types.ModuleType("sylber")
        pkg.__path__ = [os.path.join(REFERENCE_ROOT, "sylber")]
        sys.modules["sylber"] = pkg
    if "torchaudio" not in sys.modules:
        ta = types.ModuleType("torchaudio")
        ta.load = None
        ta.transforms = types.SimpleNamespace(Resample=None)
        sys.modules["torchaudio"] = ta
    ref = importlib.import_module("sylber.model.sylber")
    seg_utils = importlib.import_module("sylber.utils.segment_utils")
    cfg_dir = os.path.join(tempfile.gettempdir(), "hubert-base-ls960-offline")
    if not os.path.exists(os.path.join(cfg_dir, "config.json")):
        HubertConfig().save_pretrained(cfg_dir)
    return ref, seg_utils, cfg_dir
