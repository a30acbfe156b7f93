"""tools/oalsfx_wav: the WAV command-line program (counterpart of the reference's oalsfxpp_test)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from harness import OracleApi, ROOT
from oalsfxpp_amd import desc

TOOL = os.path.join(ROOT, "tools", "oalsfx_wav")


def write_wav(path, rate, channels, bits, data):
    """data: integer array [frames][channels] already in the file's sample format."""
    raw = data.astype("<i2" if bits == 16 else "u1").tobytes()
    hdr = b"RIFF" + struct.pack("<I", 36 + len(raw)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, channels, rate, rate * channels * bits // 8,
                                                                                   channels * bits // 8, bits)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(raw)) + raw)


def run(*args, stdin=None):
    return subprocess.run([TOOL, *args], capture_output=True, text=True, input=stdin)


def test_usage_and_file_errors(tmp_path):
    assert os.path.exists(TOOL), "build it with __graft_entry__.build()"
    r = run()
    assert r.returncode == 1 and "Usage:" in r.stdout
    r = run(str(tmp_path / "missing.wav"), str(tmp_path / "o.wav"), "echo")
    assert r.returncode == 2 and "Failed to open a file" in r.stdout
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"RIFX" + b"\0" * 64)
    r = run(str(bad), str(tmp_path / "o.wav"), "echo")
    assert r.returncode == 2 and "Not a WAV stream." in r.stdout
    f32 = tmp_path / "f32.wav"
    write_wav(f32, 48000, 2, 16, np.zeros((16, 2)))
    raw = bytearray(f32.read_bytes())
    raw[20:22] = struct.pack("<H", 3)  # IEEE float tag
    f32.write_bytes(bytes(raw))
    r = run(str(f32), str(tmp_path / "o.wav"), "echo")
    assert r.returncode == 2 and "Expected a PCM codec." in r.stdout


def expected_s16(fmt, rate, effect_type, x):
    api = OracleApi(fmt, rate, 1)
    api.set_effect_type(0, effect_type)
    api.apply_changes()
    y = np.concatenate([api.mix(x[i:i + 2048]) for i in range(0, len(x), 2048)]).reshape(-1)
    # scale so that nothing clips, truncate toward zero (reference src/oalsfxpp_test.cpp:602-636)
    lo, hi = np.float32(-1.0), np.float32(1.0)
    for v in y:  # the reference's scan is not symmetric (else-if), keep its exact shape
        if v < lo:
            lo = v
        elif v > hi:
            hi = v
    scale = np.float32(1.0) / max(hi, -lo)
    return np.trunc((scale * y) * np.float32(32767.0)).astype(np.int16)


@pytest.mark.gpu
@pytest.mark.parametrize("case", [("eax_reverb", desc.EAX_REVERB, 2, 16, 44100, None), ("8", desc.ECHO, 1, 8, 22050, None),
                                  (None, desc.FLANGER, 2, 16, 48000, "nonsense\n10\n")])
def test_wav_round_trip_matches_oracle(tmp_path, case):
    arg, etype, channels, bits, rate, stdin = case
    rng = np.random.default_rng(5)
    frames = 5000
    if bits == 16:
        data = rng.integers(-30000, 30000, size=(frames, channels))
        x = (data.astype(np.int16).astype(np.float32) / np.float32(32768.0))
    else:
        data = rng.integers(0, 256, size=(frames, channels))
        x = ((data.astype(np.int32) - 128).astype(np.float32) / np.float32(128.0))
    src, dst = str(tmp_path / "in.wav"), str(tmp_path / "out.wav")
    write_wav(src, rate, channels, bits, data)
    r = run(src, dst, *( [arg] if arg else [] ), stdin=stdin)
    assert r.returncode == 0, r.stdout + r.stderr
    if stdin:
        assert r.stdout.count("Enter effect number: ") == 2 and "10. Flanger" in r.stdout
    out = open(dst, "rb").read()
    assert out[:4] == b"RIFF" and out[8:16] == b"WAVEfmt " and struct.unpack("<HHI", out[20:28]) == (1, channels, rate)
    assert struct.unpack("<H", out[34:36]) == (16,)
    got = np.frombuffer(out[44:], dtype="<i2")
    want = expected_s16(desc.FMT_MONO if channels == 1 else desc.FMT_STEREO, rate, etype, x.astype(np.float32))
    assert got.size == want.size and np.array_equal(got, want), f"{np.count_nonzero(got != want)} samples differ"
