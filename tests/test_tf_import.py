"""TensorFlow-free SavedModel variable reader (aquaticgymenv_amd/tf_import.py): round trip through a tensor
bundle written by this test (same on-disk format: SSTable index + raw data shard), and -- only where the
reference tree is mounted -- the reference's own two checkpoints against the committed fixture."""
import os
import struct

import numpy as np
import pytest

from aquaticgymenv_amd.tf_import import read_checkpoint, dense_stack, _TABLE_MAGIC


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _block(entries, restart_interval=2):
    """LevelDB table block with prefix-compressed keys and restart points"""
    body, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(body))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        body += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _entry(dtype, shape, offset, size):
    dims = b"".join(b"\x12" + _vi(len(d)) + d for d in (b"\x08" + _vi(s) for s in shape))
    return b"\x08" + _vi(dtype) + b"\x12" + _vi(len(dims)) + dims + b"\x20" + _vi(offset) + b"\x28" + _vi(size) + \
        b"\x35" + struct.pack("<I", 0xDEADBEEF)            # crc32c (fixed32, ignored by the reader)


def _write_bundle(path, tensors):
    os.makedirs(path, exist_ok=True)
    data, entries = bytearray(), [(b"", b"\x08\x01")]        # header entry under the empty key
    for key in sorted(tensors):
        arr = np.asarray(tensors[key])
        dt = {np.dtype(np.float32): 1, np.dtype(np.int64): 9}[arr.dtype]
        entries.append((key.encode(), _entry(dt, arr.shape, len(data), arr.nbytes)))
        data += arr.tobytes()
    half = len(entries) // 2
    blocks = [_block(entries[:half]), _block(entries[half:])]
    out, handles = bytearray(), []
    for b, last in zip(blocks, (entries[half - 1][0], entries[-1][0])):
        handles.append((last, _vi(len(out)) + _vi(len(b))))
        out += b + b"\x00" + struct.pack("<I", 0)           # type = uncompressed, crc
    meta_off = len(out)
    meta = _block([])
    out += meta + b"\x00" + struct.pack("<I", 0)
    idx_off = len(out)
    idx = _block(handles, restart_interval=1)
    out += idx + b"\x00" + struct.pack("<I", 0)
    footer = _vi(meta_off) + _vi(len(meta)) + _vi(idx_off) + _vi(len(idx))
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    open(os.path.join(path, "variables.index"), "wb").write(out)
    open(os.path.join(path, "variables.data-00000-of-00001"), "wb").write(data)


def test_round_trip(tmp_path):
    rng = np.random.RandomState(0)
    want = {}
    for i, (a, b) in enumerate(((5, 64), (64, 64), (64, 3))):
        want["layer_with_weights-%d/kernel/.ATTRIBUTES/VARIABLE_VALUE" % i] = rng.randn(a, b).astype(np.float32)
        want["layer_with_weights-%d/bias/.ATTRIBUTES/VARIABLE_VALUE" % i] = rng.randn(b).astype(np.float32)
        want["layer_with_weights-%d/kernel/.OPTIMIZER_SLOT/optimizer/m/.ATTRIBUTES/VARIABLE_VALUE" % i] = rng.randn(a, b).astype(np.float32)
    want["optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE"] = np.array(123456, dtype=np.int64)
    _write_bundle(str(tmp_path), want)
    got = read_checkpoint(str(tmp_path))
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].dtype == want[k].dtype and np.array_equal(got[k], want[k])
    layers = dense_stack(got)
    assert [k.shape for k, _ in layers] == [(5, 64), (64, 64), (64, 3)]
    with pytest.raises(ValueError):
        open(tmp_path / "variables.index", "ab").write(b"x")
        read_checkpoint(str(tmp_path))


@pytest.mark.skipif(not os.path.isdir("/root/reference/example_policies"), reason="reference tree not mounted")
def test_reference_checkpoints_equal_the_fixture():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dqn_policies.npz"))
    for tag, sub in (("no_obs", "example_no_obs/models/model-00030"), ("with_obs", "example_with_obs/models/model-00032")):
        layers = dense_stack(read_checkpoint(os.path.join("/root/reference/example_policies", sub, "variables")))
        assert len(layers) == 3
        for li, (k, b) in enumerate(layers):
            assert np.array_equal(k, z["%s_kernel%d" % (tag, li)]) and np.array_equal(b, z["%s_bias%d" % (tag, li)])
    assert abs(z["no_obs_published_success"].mean() - 0.938) < 1e-9 and abs(z["with_obs_published_success"].mean() - 0.667) < 1e-9
