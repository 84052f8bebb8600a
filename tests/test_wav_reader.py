"""urbansound.read_wav: the RIFF/WAVE flavours UrbanSound8K ships as plain samples.  CPU only."""
import struct
import wave

import numpy as np
import pytest

from mlx8_ws_audio_transformer_amd import urbansound


def _write_pcm(path, x_int, rate, width):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(x_int.shape[1]); w.setsampwidth(width); w.setframerate(rate)
        if width == 3:
            b = (x_int.astype(np.int32) & 0xFFFFFF).astype("<u4").view(np.uint8).reshape(-1, 4)[:, :3]
            w.writeframes(b.tobytes())
        elif width == 1:
            w.writeframes(x_int.astype(np.uint8).tobytes())
        else:
            w.writeframes(x_int.astype({2: "<i2", 4: "<i4"}[width]).tobytes())


def _write_raw(path, fmt_body, data):
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + b"LIST" + struct.pack("<I", 3) + b"abc\0" \
        + b"data" + struct.pack("<I", len(data)) + data
    path.write_bytes(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_integer_pcm_widths(tmp_path):
    rng = np.random.default_rng(0)
    x16 = rng.integers(-32768, 32767, size=(1000, 2))
    _write_pcm(tmp_path / "a.wav", x16, 44100, 2)
    s, sr = urbansound.read_wav(str(tmp_path / "a.wav"))
    assert sr == 44100 and s.dtype.is_floating_point is False and tuple(s.shape) == (1000, 2)
    np.testing.assert_array_equal(s.numpy(), x16)
    x24 = rng.integers(-(1 << 23), (1 << 23) - 1, size=(500, 1))
    _write_pcm(tmp_path / "b.wav", x24, 8000, 3)
    s, sr = urbansound.read_wav(str(tmp_path / "b.wav"))
    assert sr == 8000
    np.testing.assert_array_equal(s.numpy(), (x24 / float(1 << 23)).astype(np.float32))
    x8 = rng.integers(0, 255, size=(300, 1))
    _write_pcm(tmp_path / "c.wav", x8, 22050, 1)
    np.testing.assert_array_equal(urbansound.read_wav(str(tmp_path / "c.wav"))[0].numpy(), ((x8 - 128.0) / 128.0).astype(np.float32))
    x32 = rng.integers(-(1 << 31), (1 << 31) - 1, size=(200, 2))
    _write_pcm(tmp_path / "d.wav", x32, 48000, 4)
    np.testing.assert_allclose(urbansound.read_wav(str(tmp_path / "d.wav"))[0].numpy(), x32 / float(1 << 31), atol=1e-7)


def test_float_and_extensible_and_odd_chunks(tmp_path):
    x = np.random.default_rng(1).standard_normal((400, 2)).astype("<f4")
    _write_raw(tmp_path / "f.wav", struct.pack("<HHIIHH", 3, 2, 48000, 48000 * 8, 8, 32), x.tobytes())
    s, sr = urbansound.read_wav(str(tmp_path / "f.wav"))
    assert sr == 48000
    np.testing.assert_array_equal(s.numpy(), x)
    x16 = np.arange(-300, 300, dtype="<i2").reshape(-1, 1)
    ext = struct.pack("<HHIIHH", 0xFFFE, 1, 16000, 32000, 2, 16) + struct.pack("<HHI", 22, 16, 4) + struct.pack("<H", 1) + b"\0" * 14
    _write_raw(tmp_path / "e.wav", ext, x16.tobytes())
    s, sr = urbansound.read_wav(str(tmp_path / "e.wav"))
    assert sr == 16000
    np.testing.assert_array_equal(s.numpy(), x16)


def test_unsupported_and_malformed(tmp_path):
    _write_raw(tmp_path / "adpcm.wav", struct.pack("<HHIIHH", 2, 1, 22050, 11100, 256, 4), b"\0" * 512)
    with pytest.raises(ValueError):
        urbansound.read_wav(str(tmp_path / "adpcm.wav"))
    (tmp_path / "junk.wav").write_bytes(b"not a wav file at all")
    with pytest.raises(ValueError):
        urbansound.read_wav(str(tmp_path / "junk.wav"))
