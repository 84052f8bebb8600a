"""CPU-side checks of the C-ABI boundary: libawt.so loads without a GPU and exports every symbol include/awt.h
declares; the Python binding covers all of them; the product path refuses to run without a GPU."""
import ctypes
import os

import pytest
import torch

from mlx8_ws_audio_transformer_amd import _lib


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(_lib.LIB_PATH):
        from mlx8_ws_audio_transformer_amd.build import build
        build(verbose=False)
    return _lib.lib()


def test_header_symbols_are_exported_and_bound(built):
    names = _lib.declared_symbols()
    assert len(names) >= 20
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/awt.h but not exported by libawt.so"
        assert n in _lib._SIGNATURES, f"{n} has no ctypes signature in _lib.py"


def test_version_and_workspace_queries_need_no_gpu(built):
    assert b"gfx950" in built.awt_version()
    assert built.awt_logmel_workspace_bytes(64) >= 256
    assert built.awt_op_linear_workspace_bytes(128, 128, 64) > 0


def test_no_cpu_fallback():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.ctx()
    from mlx8_ws_audio_transformer_amd.feature_extraction import WhisperFeatureExtractor
    fe = WhisperFeatureExtractor()
    with pytest.raises(ValueError, match="16000"):
        fe([0.0] * 100, sampling_rate=8000)
    with pytest.raises(RuntimeError):
        fe([0.0] * 100, sampling_rate=16000)


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(_lib.HERE)
    pkg = os.path.join(root, "mlx8-ws-audio-transformer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
