"""TensorFlow-free reader of a Keras SavedModel's variables (tensor-bundle checkpoint), for replaying the
reference's trained DQN policies (example_policies/*/models/model-000NN, written by
main/impl/dqn.py:316-321 with ``tf.keras.Model.save``) on the batched environment.

TensorFlow is not installable where this build runs, and nothing of TensorFlow is needed to read the
two files that hold the numbers:

  variables.index             an SSTable (LevelDB table format): sorted key -> BundleEntryProto
  variables.data-00000-of-00001   the raw little-endian tensor bytes

Only what those files use is implemented: uncompressed blocks, prefix-compressed keys with restart
points, the 48-byte footer, and the BundleEntryProto fields dtype (1), shape (2), shard_id (3),
offset (4), size (5).  Formats: leveldb/doc/table_format.md, tensorflow/core/protobuf/tensor_bundle.proto.
"""
import os
import struct

import numpy as np

_TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}     # tensorflow DataType enum


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _block_entries(buf, offset, size):
    """yield (key, value) of one table block (contents only; the 5-byte trailer follows it)."""
    if buf[offset + size] != 0:
        raise ValueError("compressed SSTable blocks are not supported (block type %d)" % buf[offset + size])
    block = buf[offset:offset + size]
    n_restarts = struct.unpack_from("<I", block, size - 4)[0]
    limit = size - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        value_len, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + value_len])
        pos += value_len


def _parse_entry(value):
    """BundleEntryProto -> dict(dtype, shape, shard_id, offset, size)"""
    out = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0}
    pos = 0
    while pos < len(value):
        tag, pos = _varint(value, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            v, pos = _varint(value, pos)
            if field == 1:
                out["dtype"] = v
            elif field == 3:
                out["shard_id"] = v
            elif field == 4:
                out["offset"] = v
            elif field == 5:
                out["size"] = v
        elif wire == 2:
            n, pos = _varint(value, pos)
            sub = value[pos:pos + n]
            pos += n
            if field == 2:                      # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1; }
                sp = 0
                while sp < len(sub):
                    t2, sp = _varint(sub, sp)
                    if t2 >> 3 == 2 and t2 & 7 == 2:
                        m, sp = _varint(sub, sp)
                        dim = sub[sp:sp + m]
                        sp += m
                        dp, size = 0, 0
                        while dp < len(dim):
                            t3, dp = _varint(dim, dp)
                            if t3 & 7 == 0:
                                v3, dp = _varint(dim, dp)
                                if t3 >> 3 == 1:
                                    size = v3
                            elif t3 & 7 == 2:
                                k, dp = _varint(dim, dp)
                                dp += k
                            else:
                                raise ValueError("unexpected wire type in TensorShapeProto.Dim")
                        out["shape"].append(size)
                    elif t2 & 7 == 0:
                        _, sp = _varint(sub, sp)
                    elif t2 & 7 == 2:
                        m, sp = _varint(sub, sp)
                        sp += m
                    else:
                        raise ValueError("unexpected wire type in TensorShapeProto")
        elif wire == 5:
            pos += 4
        elif wire == 1:
            pos += 8
        else:
            raise ValueError("unexpected wire type %d in BundleEntryProto" % wire)
    return out


def read_checkpoint(variables_dir):
    """-> {key: ndarray} for every tensor of variables.index / variables.data-00000-of-00001."""
    index = open(os.path.join(variables_dir, "variables.index"), "rb").read()
    if struct.unpack_from("<Q", index, len(index) - 8)[0] != _TABLE_MAGIC:
        raise ValueError("not an SSTable (bad magic)")
    footer = index[-48:]
    pos = 0
    _, pos = _varint(footer, pos)       # metaindex handle
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    data = open(os.path.join(variables_dir, "variables.data-00000-of-00001"), "rb").read()
    tensors = {}
    for _, handle in _block_entries(index, idx_off, idx_size):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        for key, value in _block_entries(index, off, size):
            if key == b"":              # BundleHeaderProto
                continue
            e = _parse_entry(value)
            if e["dtype"] not in _DTYPES:
                continue                # strings (object-graph proto) and other non-numeric entries
            dt = np.dtype(_DTYPES[e["dtype"]])
            arr = np.frombuffer(data, dtype=dt.newbyteorder("<"), count=e["size"] // dt.itemsize, offset=e["offset"])
            tensors[key.decode("utf-8")] = arr.reshape(e["shape"]).astype(dt)
    return tensors


def dense_stack(tensors):
    """Keras functional model of main/impl/dqn.py:300-313 -> [(kernel [in, out], bias [out]), ...] in layer order."""
    layers = []
    i = 0
    while True:
        k = "layer_with_weights-%d/kernel/.ATTRIBUTES/VARIABLE_VALUE" % i
        b = "layer_with_weights-%d/bias/.ATTRIBUTES/VARIABLE_VALUE" % i
        if k not in tensors:
            break
        layers.append((tensors[k], tensors[b]))
        i += 1
    if not layers:
        raise ValueError("no layer_with_weights-*/kernel entries found: %s" % sorted(tensors)[:8])
    return layers


class GreedyQPolicy(object):
    """argmax_a Q(s, a) of the reference's 5 -> 64 -> 64 -> 3 ReLU MLP (main/testing/test_dqn.py:14-22), batched
    on the device: input is the NORMALISED observation [N, 5] (main/impl/utils.py:15-33; BatchedAqua's fused
    obs_norm epilogue produces it), output int64 actions [N].  Plain library GEMMs (torch -> rocBLAS/hipBLASLt)."""

    def __init__(self, layers, device):
        import torch
        self.torch = torch
        self.weights = [(torch.as_tensor(np.ascontiguousarray(k), dtype=torch.float32, device=device),
                         torch.as_tensor(np.ascontiguousarray(b), dtype=torch.float32, device=device)) for k, b in layers]

    def q_values(self, obs_norm):
        x = obs_norm
        for li, (k, b) in enumerate(self.weights):
            x = self.torch.addmm(b, x, k)
            if li + 1 < len(self.weights):
                x = self.torch.relu(x)
        return x

    def __call__(self, obs_norm):
        return self.q_values(obs_norm).argmax(dim=1)
