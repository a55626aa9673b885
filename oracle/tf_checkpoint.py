"""Independent pure-Python writer / reader of TensorFlow V2 checkpoints -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
Restates the same published formats as video_prediction_amd/checkpoint.py (tensor_bundle.proto, LevelDB table format) but
shares no code with it: it prefix-compresses keys with a restart interval of 16 like LevelDB's BlockBuilder (the product
writer restarts at every key), so the product reader is exercised on shared-prefix blocks and multi-block tables.
PARITY UNPINNED (no TensorFlow, no published checkpoint offline)."""
import struct

import numpy as np

from .tfrecord import masked_crc32c

_MAGIC = 0xdb4775248b80fb57
_IDS = {np.dtype(np.float32): 1, np.dtype(np.float64): 2, np.dtype(np.int32): 3, np.dtype(np.int64): 9}


def _v(x):
    out = bytearray()
    while x >= 0x80:
        out.append((x & 0x7f) | 0x80)
        x >>= 7
    out.append(x)
    return bytes(out)


def _msg(field, payload):
    return _v(field << 3 | 2) + _v(len(payload)) + payload


def _build_block(entries, interval=16):
    body, restarts, last = bytearray(), [], b''
    for n, (k, val) in enumerate(entries):
        shared = 0
        if n % interval == 0:
            restarts.append(len(body))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        body += _v(shared) + _v(len(k) - shared) + _v(len(val)) + k[shared:] + val
        last = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        body += struct.pack('<I', r)
    return bytes(body + struct.pack('<I', len(restarts)))


def write(prefix, tensors, entries_per_block=5):
    data = bytearray()
    items = [(b'', _v(1 << 3) + _v(1) + _msg(3, _v(1 << 3) + _v(1)))]
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        raw = a.tobytes()
        shape = b''.join(_msg(2, _v(1 << 3) + _v(int(d))) for d in a.shape)
        e = _v(1 << 3) + _v(_IDS[a.dtype]) + _msg(2, shape) + _v(3 << 3) + _v(0)
        e += _v(4 << 3) + _v(len(data)) + _v(5 << 3) + _v(len(raw)) + _v(6 << 3 | 5) + struct.pack('<I', masked_crc32c(raw))
        items.append((name.encode(), e))
        data += raw
    out = bytearray()

    def emit(block):
        off = len(out)
        out.extend(block + b'\0' + struct.pack('<I', masked_crc32c(block + b'\0')))
        return _v(off) + _v(len(block))
    index = []
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        index.append((chunk[-1][0], emit(_build_block(chunk, interval=3))))
    footer = emit(_build_block([])) + emit(_build_block(index, interval=1))
    out.extend(footer + b'\0' * (40 - len(footer)) + struct.pack('<Q', _MAGIC))
    open(prefix + '.index', 'wb').write(out)
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
