"""Test helper: write an 8-bit stereo IQ.wav, optionally with extra RIFF chunks in front of (or behind)
the ``data`` chunk, the way SDRSharp (``auxi``) and audio editors (``LIST``) lay recordings out."""
import struct


def write_iq_wav(path, raw_u8_n2, rate, before=(), after=(), bits=8, channels=2):
    data = bytes(raw_u8_n2.tobytes())

    def chunk(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")
    fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * channels * bits // 8, channels * bits // 8, bits)
    body = b"WAVE" + chunk(b"fmt ", fmt)
    for cid, payload in before:
        body += chunk(cid, payload)
    body += chunk(b"data", data)
    for cid, payload in after:
        body += chunk(cid, payload)
    with open(path, "wb") as fh:
        fh.write(b"RIFF" + struct.pack("<I", len(body)) + body)
