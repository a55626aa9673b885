"""TensorFlow checkpoint (V2 "tensor bundle") import / export by variable name, without TensorFlow.

The reference restores through tf.train.Saver (video_prediction/utils/tf_utils.py:528-559, models/base_model.py:229-247,
savp_model.py:848-855); the on-disk format belongs to the un-vendored tensorflow-gpu>=1.9.0 and is restated here from its
published definition (tensorflow/core/util/tensor_bundle/tensor_bundle.{h,cc}, tensorflow/core/lib/io/{format,table,block}.cc,
tensorflow/core/protobuf/tensor_bundle.proto):

  <prefix>.index                 an SSTable (LevelDB table format): key "" -> BundleHeaderProto, key <variable name> ->
                                 BundleEntryProto {dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (masked)}
  <prefix>.data-00000-of-00001   the tensors' bytes (little endian, row major) at [offset, offset + size)
  checkpoint                     text proto naming the latest prefix (tf.train.latest_checkpoint)

SSTable: data blocks of prefix-compressed entries (varint shared | varint non_shared | varint value_len | key delta | value)
followed by the restart array, each block trailed by 1 compression byte (0 none, 1 snappy) + masked crc32c; an index block
maps separator keys to block handles (varint offset, varint size); 48-byte footer = metaindex handle, index handle, padding,
magic 0xdb4775248b80fb57.

PARITY UNPINNED: no TensorFlow and no published checkpoint is available offline; the reader is exercised on files produced by
an independent writer (oracle/tf_checkpoint.py) and vice versa.
"""
import os
import re
import struct
from collections import OrderedDict

import numpy as np

from . import io as sio

MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}      # types.proto DataType
DTYPE_IDS = {np.dtype(v): k for k, v in DTYPES.items()}


def _varint(buf, i):
    v = s = 0
    while True:
        c = buf[i]
        i += 1
        v |= (c & 0x7f) << s
        s += 7
        if not c & 0x80:
            return v, i


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _snappy_decompress(data):
    """Raw snappy block format (length varint, then literal / copy elements)."""
    n, i = _varint(data, 0)
    out = bytearray()
    while i < len(data):
        tag = data[i]
        i += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(data[i:i + nb], 'little')
                i += nb
            ln += 1
            out += data[i:i + ln]
            i += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | data[i]
            i += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = data[i] | (data[i + 1] << 8)
            i += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(data[i:i + 4], 'little')
            i += 4
        for _ in range(ln):                      # overlapping copies are allowed
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('corrupt snappy block')
    return bytes(out)


def _read_block(buf, offset, size):
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    (crc,) = struct.unpack('<I', buf[offset + size + 1:offset + size + 5])
    if sio.masked_crc32c(raw + bytes([ctype])) != crc:
        raise ValueError('checkpoint index: block checksum mismatch')
    if ctype == 1:
        raw = _snappy_decompress(raw)
    elif ctype != 0:
        raise ValueError('checkpoint index: unknown block compression %d' % ctype)
    (nrestarts,) = struct.unpack('<I', raw[-4:])
    end = len(raw) - 4 - 4 * nrestarts
    entries, i, key = [], 0, b''
    while i < end:
        shared, i = _varint(raw, i)
        non_shared, i = _varint(raw, i)
        vlen, i = _varint(raw, i)
        key = key[:shared] + raw[i:i + non_shared]
        i += non_shared
        entries.append((key, raw[i:i + vlen]))
        i += vlen
    return entries


def read_table(path):
    """All (key, value) pairs of an SSTable file, in key order."""
    buf = open(path, 'rb').read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != MAGIC:
        raise ValueError('%s is not a TensorFlow checkpoint index (bad magic)' % path)
    footer = buf[-48:]
    _, i = _varint(footer, 0)
    _, i = _varint(footer, i)                    # metaindex handle (unused)
    ioff, i = _varint(footer, i)
    isize, i = _varint(footer, i)
    out = []
    for _, handle in _read_block(buf, ioff, isize):
        off, j = _varint(handle, 0)
        size, j = _varint(handle, j)
        out.extend(_read_block(buf, off, size))
    return out


def _parse_entry(val):
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    i = 0
    while i < len(val):
        key, i = _varint(val, i)
        f, w = key >> 3, key & 7
        if w == 0:
            v, i = _varint(val, i)
            if f == 1: e['dtype'] = v
            elif f == 3: e['shard_id'] = v
            elif f == 4: e['offset'] = v
            elif f == 5: e['size'] = v
        elif w == 5:
            if f == 6:
                (e['crc32c'],) = struct.unpack('<I', val[i:i + 4])
            i += 4
        elif w == 1:
            i += 8
        elif w == 2:
            n, i = _varint(val, i)
            sub = val[i:i + n]
            i += n
            if f == 2:                           # TensorShapeProto { repeated Dim dim = 2 { int64 size = 1 } }
                j = 0
                while j < len(sub):
                    k2, j = _varint(sub, j)
                    if k2 & 7 == 2:
                        m, j = _varint(sub, j)
                        dim = sub[j:j + m]
                        j += m
                        if k2 >> 3 == 2:
                            size, q = 0, 0
                            while q < len(dim):
                                k3, q = _varint(dim, q)
                                if k3 & 7 == 0:
                                    v3, q = _varint(dim, q)
                                    if k3 >> 3 == 1: size = v3
                                else:
                                    m3, q = _varint(dim, q)
                                    q += m3
                            e['shape'].append(size)
                    else:
                        _, j = _varint(sub, j)
            elif f == 7:
                e['sliced'] = True
        else:
            raise ValueError('corrupt BundleEntryProto')
    return e


def latest_checkpoint(path):
    """tf.train.latest_checkpoint: a directory -> the prefix named by its `checkpoint` state file; a prefix stays itself."""
    if os.path.isdir(path):
        state = os.path.join(path, 'checkpoint')
        if not os.path.exists(state):
            raise FileNotFoundError('no checkpoint state file in %s' % path)
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if not m:
            raise ValueError('malformed checkpoint state file %s' % state)
        p = m.group(1)
        return p if os.path.isabs(p) else os.path.join(path, p)
    return path


def read_checkpoint(path, names=None):
    """{variable name: numpy array} of a V2 checkpoint (directory or prefix); `names` restricts the tensors that are loaded."""
    prefix = latest_checkpoint(path)
    entries = read_table(prefix + '.index')
    shards = {}
    num_shards = 1
    out = OrderedDict()
    for key, val in entries:
        if key == b'':
            i = 0
            while i < len(val):                  # BundleHeaderProto { int32 num_shards = 1; ... }
                k, i = _varint(val, i)
                if k & 7 == 0:
                    v, i = _varint(val, i)
                    if k >> 3 == 1: num_shards = v
                elif k & 7 == 2:
                    n, i = _varint(val, i)
                    i += n
                else:
                    break
            continue
        name = key.decode()
        if names is not None and name not in names:
            continue
        e = _parse_entry(val)
        if e['sliced']:
            raise NotImplementedError('partitioned variable %s (tensor slices) is not supported' % name)
        if e['dtype'] not in DTYPES:
            continue                             # strings etc.: nothing the model restores
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), dtype=np.uint8, mode='r')
        raw = bytes(shards[sid][e['offset']:e['offset'] + e['size']])
        if e['crc32c'] is not None and sio.masked_crc32c(raw) != e['crc32c'] and sio.crc32c(raw) != e['crc32c']:
            raise ValueError('checkpoint tensor %s: checksum mismatch' % name)
        out[name] = np.frombuffer(raw, dtype=DTYPES[e['dtype']]).reshape(e['shape']).copy()
    return out


def variable_names(path):
    return [k.decode() for k, _ in read_table(latest_checkpoint(path) + '.index') if k]


# ---- writer --------------------------------------------------------------------------------------------------------------
def _ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def _block(entries):
    body, restarts = bytearray(), []
    for k, v in entries:                         # restart at every entry (shared = 0): valid, uncompressed keys
        restarts.append(len(body))
        body += _enc_varint(0) + _enc_varint(len(k)) + _enc_varint(len(v)) + k + v
    body += b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))
    return bytes(body)


def write_checkpoint(prefix, tensors, update_state=True):
    """Write {name: array} as a single-shard V2 checkpoint at `prefix` (+ the directory's `checkpoint` state file)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    data = bytearray()
    header = _enc_varint(1 << 3) + _enc_varint(1) + _ld(3, _enc_varint(1 << 3) + _enc_varint(1))     # num_shards=1, version.producer=1
    items = [(b'', header)]
    for name in sorted(tensors):
        a = np.asarray(tensors[name])
        if a.dtype not in DTYPE_IDS:
            raise TypeError('unsupported dtype %s for %s' % (a.dtype, name))
        raw = a.tobytes()
        shape = b''.join(_ld(2, _enc_varint(1 << 3) + _enc_varint(d)) for d in a.shape)
        entry = _enc_varint(1 << 3) + _enc_varint(DTYPE_IDS[a.dtype]) + _ld(2, shape)
        if len(data):
            entry += _enc_varint(4 << 3) + _enc_varint(len(data))
        entry += _enc_varint(5 << 3) + _enc_varint(len(raw)) + _enc_varint((6 << 3) | 5) + struct.pack('<I', sio.masked_crc32c(raw))
        items.append((name.encode(), entry))
        data += raw
    out = bytearray()

    def put(block):
        off = len(out)
        out.extend(block + b'\x00' + struct.pack('<I', sio.masked_crc32c(block + b'\x00')))
        return _enc_varint(off) + _enc_varint(len(block))
    index = []
    for i in range(0, len(items), 64):           # data blocks of 64 entries; the index key is the block's last key
        chunk = items[i:i + 64]
        index.append((chunk[-1][0], put(_block(chunk))))
    meta = put(_block([]))
    idx = put(_block(index))
    footer = meta + idx
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC)
    out.extend(footer)
    with open(prefix + '.index', 'wb') as f:
        f.write(out)
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(data)
    if update_state:
        base = os.path.basename(prefix)
        with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), 'checkpoint'), 'w') as f:
            f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


def restore_values(checkpoints, wanted, mapping=None, skip_global_step=None, log=None, optional=()):
    """get_checkpoint_restore_saver semantics (tf_utils.py:528-559): for every wanted variable name look up mapping(name, names
    in the checkpoint); restore what both sides have, report the rest.  Several checkpoints may each hold a subset
    (base_model.py:231-236; global_step is skipped automatically then).  Names in `optional` (optimizer slots) are not reported
    when absent.  Returns {name: array}."""
    if not isinstance(checkpoints, (list, tuple)):
        checkpoints = [checkpoints]
    skip_global_step = len(checkpoints) > 1 if skip_global_step is None else skip_global_step
    mapping = mapping or (lambda name, _names: name.split(':')[0])
    log = log or (lambda *_: None)
    out = OrderedDict()
    for ck in checkpoints:
        names = set(variable_names(ck))
        lookup = {mapping(w, names): w for w in wanted}
        if skip_global_step:
            lookup.pop('global_step', None)
        both = {k: w for k, w in lookup.items() if k in names}
        vals = read_checkpoint(ck, set(both))
        for k, w in both.items():
            out[w] = vals[k]
        missing = sorted(w for k, w in lookup.items() if k not in names and w not in optional)
        unused = sorted(n for n in names if n not in lookup and not (skip_global_step and n == 'global_step'))
        if missing:
            log('variables that were not restored because they are not in the checkpoint:', missing)
        if unused:
            log('checkpoint variables that were not used for restoring:', unused)
    return out
