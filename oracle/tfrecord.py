"""Pure-Python restatement of the two wire formats behind the reference's input pipeline -- TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): used to WRITE fixture files for tests and to cross-check the C++ reader (libsavp_io.so).

* TFRecord framing (tf.data.TFRecordDataset, base_dataset.py:135; tensorflow/core/lib/io/record_writer.cc of the un-vendored
  tensorflow-gpu>=1.9.0): uint64 length, uint32 masked crc32c(length), data, uint32 masked crc32c(data), little endian;
  mask(c) = ((c >> 15) | (c << 17)) + 0xa282ead8.  CRC-32C known answers: RFC 3720 B.4.
* tf.train.Example (tensorflow/core/example/{example,feature}.proto): Example{features=1} / Features{map feature=1} /
  Feature{bytes_list=1 | float_list=2 | int64_list=3}, lists packed.
* slice_sequences (base_dataset.py:189-229) for a given t_start.
PARITY UNPINNED: no TensorFlow here, the reference ships no record fixtures; the CRC is pinned by the RFC vectors.
"""
import struct


def crc32c(data):
    c = 0xffffffff
    for b in bytes(data):
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
    return c ^ 0xffffffff


def masked_crc32c(data):
    c = crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xffffffff


def write_records(path, records):
    with open(path, 'wb') as f:
        for r in records:
            hdr = struct.pack('<Q', len(r))
            f.write(hdr + struct.pack('<I', masked_crc32c(hdr)) + r + struct.pack('<I', masked_crc32c(r)))


def read_records(path):
    out = []
    with open(path, 'rb') as f:
        while True:
            hdr = f.read(12)
            if not hdr:
                return out
            n, c = struct.unpack('<QI', hdr)
            assert masked_crc32c(hdr[:8]) == c, 'corrupt length'
            r = f.read(n)
            (c2,) = struct.unpack('<I', f.read(4))
            assert masked_crc32c(r) == c2, 'corrupt data'
            out.append(r)


def _varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def encode_example(features):
    """features: {name: bytes | [bytes, ...] | list of float | list of int (tag with ('int64', [...]))}."""
    body = b''
    for name in sorted(features):
        v = features[name]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if isinstance(v, tuple) and v[0] == 'int64':
            feat = _ld(3, _ld(1, b''.join(_varint(x & 0xffffffffffffffff) for x in v[1])))
        elif len(v) and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b''.join(_ld(1, bytes(x)) for x in v))
        else:
            feat = _ld(2, _ld(1, struct.pack('<%df' % len(v), *v)))
        body += _ld(1, _ld(1, name.encode()) + _ld(2, feat))
    return _ld(1, body)


def slice_times(example_len, sequence_length, frame_skip, t_start):
    """base_dataset.py:213-214: (state-like frame indices, action-like step indices)."""
    fs1 = frame_skip + 1
    state = list(range(t_start, t_start + (sequence_length - 1) * fs1 + 1, fs1))
    action = list(range(t_start, t_start + (sequence_length - 1) * fs1))
    assert state[-1] < example_len
    return state, action
