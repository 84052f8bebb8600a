"""Import alias for the package directory `mlx8-ws-audio-transformer_amd/`.

The repository layout names the package directory with hyphens, which Python cannot import by
name; this alias points `__path__` at that directory and runs its `__init__.py` in this namespace,
so `import mlx8_ws_audio_transformer_amd as awt` and `awt.<submodule>` resolve to the real files.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "mlx8-ws-audio-transformer_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
