"""ctypes binding of libsavp_io.so (include/savp_io.h): the C++ TFRecord / tf.train.Example reader and the batched,
prefetching video pipeline that replace tf.data for the BAIR / softmotion record layout."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libsavp_io.so')

c_i32, c_i64, c_u64, c_vp, c_cp = ctypes.c_int32, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_char_p
ERRORS = {-1: 'invalid argument', -2: 'I/O error', -3: 'corrupt record', -4: 'end of data', -5: 'feature not found'}
EOF_CODE = -4


class SavpVideoPipelineArgs(ctypes.Structure):
    _fields_ = [
        ('filenames', ctypes.POINTER(c_cp)), ('num_files', c_i32), ('image_key_fmt', c_cp), ('example_frames', c_i32),
        ('height', c_i32), ('width', c_i32), ('channels', c_i32), ('sequence_length', c_i32), ('frame_skip', c_i32),
        ('time_shift', c_i32), ('batch_size', c_i32), ('shuffle', c_i32), ('shuffle_buffer', c_i32), ('num_epochs', c_i32),
        ('seed', c_u64), ('prefetch_batches', c_i32),
        ('float_keys_fmt', ctypes.POINTER(c_cp)), ('float_dims', ctypes.POINTER(c_i32)),
        ('float_per_frame_minus', ctypes.POINTER(c_i32)), ('num_float_keys', c_i32), ('var_len', c_i32),
    ]


_lib = None


def get():
    """The loaded library; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError('libsavp_io.so is missing: build it with __graft_entry__.build()')
        L = ctypes.CDLL(LIB_PATH)
        P = ctypes.POINTER
        L.savp_io_crc32c.argtypes, L.savp_io_crc32c.restype = [c_vp, c_u64], ctypes.c_uint32
        L.savp_io_masked_crc32c.argtypes, L.savp_io_masked_crc32c.restype = [c_vp, c_u64], ctypes.c_uint32
        L.savp_tfr_open.argtypes, L.savp_tfr_open.restype = [c_cp, c_i64, P(c_vp)], c_i32
        L.savp_tfr_next.argtypes, L.savp_tfr_next.restype = [c_vp, P(c_vp), P(c_u64)], c_i32
        L.savp_tfr_close.argtypes, L.savp_tfr_close.restype = [c_vp], None
        L.savp_example_feature.argtypes = [c_vp, c_u64, c_cp, c_i32, P(c_i32), P(c_vp), P(c_u64)]
        L.savp_example_feature.restype = c_i32
        L.savp_example_floats.argtypes, L.savp_example_floats.restype = [c_vp, c_u64, c_cp, c_vp, c_i64], c_i32
        L.savp_example_int64.argtypes, L.savp_example_int64.restype = [c_vp, c_u64, c_cp, c_i32, P(c_i64)], c_i32
        L.savp_pipeline_create.argtypes, L.savp_pipeline_create.restype = [P(SavpVideoPipelineArgs), P(c_vp)], c_i32
        L.savp_pipeline_next.argtypes, L.savp_pipeline_next.restype = [c_vp, c_vp, P(c_vp)], c_i32
        L.savp_pipeline_error.argtypes, L.savp_pipeline_error.restype = [c_vp], c_cp
        L.savp_pipeline_destroy.argtypes, L.savp_pipeline_destroy.restype = [c_vp], None
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed: %s (%d)' % (what, ERRORS.get(rc, 'error'), rc))


def crc32c(data):
    data = bytes(data)
    return int(get().savp_io_crc32c(data, len(data)))


def masked_crc32c(data):
    data = bytes(data)
    return int(get().savp_io_masked_crc32c(data, len(data)))


def read_records(path, buffer_bytes=8 << 20):
    """All records of one TFRecord file (CRC-verified) as bytes objects."""
    L = get()
    h = c_vp()
    check(L.savp_tfr_open(path.encode(), buffer_bytes, ctypes.byref(h)), 'savp_tfr_open(%s)' % path)
    out = []
    try:
        while True:
            p, n = c_vp(), c_u64()
            rc = L.savp_tfr_next(h, ctypes.byref(p), ctypes.byref(n))
            if rc == EOF_CODE:
                return out
            check(rc, 'savp_tfr_next(%s)' % path)
            out.append(ctypes.string_at(p, n.value))
    finally:
        L.savp_tfr_close(h)


def example_feature(example, name, index=0):
    """(kind, payload) of one feature of a serialized tf.train.Example: kind 1 -> the index-th bytes value,
    kind 2 -> list of floats, kind 3 -> number of int64 values."""
    import struct
    L = get()
    kind, p, n = c_i32(), c_vp(), c_u64()
    check(L.savp_example_feature(example, len(example), name.encode(), index, ctypes.byref(kind), ctypes.byref(p), ctypes.byref(n)),
          'feature %s' % name)
    if kind.value == 1:
        return 1, ctypes.string_at(p, n.value)
    if kind.value == 2:
        return 2, list(struct.unpack('<%df' % n.value, ctypes.string_at(p, 4 * n.value))) if n.value else []
    return kind.value, n.value


def example_int64(example, name, index=0):
    """The index-th value of an int64_list feature of a serialized tf.train.Example."""
    v = c_i64()
    check(get().savp_example_int64(example, len(example), name.encode(), index, ctypes.byref(v)), 'int64 feature %s' % name)
    return int(v.value)


class VideoPipeline(object):
    """Batched, shuffling, prefetching reader (one C++ thread): next() fills caller buffers with uint8 frames
    [B, T, H, W, C] and the optional float features."""

    def __init__(self, filenames, image_key_fmt, example_frames, image_shape, sequence_length, batch_size, frame_skip=0,
                 time_shift=0, shuffle=False, shuffle_buffer=1024, num_epochs=1, seed=0, prefetch_batches=2, float_keys=(),
                 var_len=False):
        import numpy as np
        self._np = np
        L = get()
        files = [f.encode() for f in filenames]
        self._files = (c_cp * len(files))(*files)
        a = SavpVideoPipelineArgs()
        a.filenames, a.num_files = self._files, len(files)
        a.image_key_fmt = image_key_fmt.encode()
        a.example_frames = example_frames
        a.var_len = int(bool(var_len))
        a.height, a.width, a.channels = image_shape
        a.sequence_length, a.frame_skip, a.time_shift, a.batch_size = sequence_length, frame_skip, time_shift, batch_size
        a.shuffle, a.shuffle_buffer, a.num_epochs, a.seed, a.prefetch_batches = int(shuffle), shuffle_buffer, num_epochs or 0, seed, prefetch_batches
        self.float_keys = list(float_keys)                 # [(fmt, dim, minus)]
        if self.float_keys:
            self._fk = (c_cp * len(self.float_keys))(*[k[0].encode() for k in self.float_keys])
            self._fd = (c_i32 * len(self.float_keys))(*[k[1] for k in self.float_keys])
            self._fm = (c_i32 * len(self.float_keys))(*[k[2] for k in self.float_keys])
            a.float_keys_fmt, a.float_dims, a.float_per_frame_minus, a.num_float_keys = self._fk, self._fd, self._fm, len(self.float_keys)
        self._args = a
        self.shape = (batch_size, sequence_length) + tuple(image_shape)
        self.frame_skip = frame_skip
        self._h = c_vp()
        check(L.savp_pipeline_create(ctypes.byref(a), ctypes.byref(self._h)), 'savp_pipeline_create')

    def float_shape(self, k):
        fmt, dim, minus = self.float_keys[k]
        B, T = self.shape[:2]
        return (B, T, dim) if minus == 0 else (B, T - 1, dim * (self.frame_skip + 1))

    def next(self, images=None, floats=None):
        """Returns (images uint8 [B,T,H,W,C], [float arrays]) or None at the end of the data.  `images` / `floats` may be
        caller-provided numpy arrays (e.g. views of pinned memory)."""
        np = self._np
        if images is None:
            images = np.empty(self.shape, dtype=np.uint8)
        if floats is None:
            floats = [np.empty(self.float_shape(k), dtype=np.float32) for k in range(len(self.float_keys))]
        fp = (c_vp * max(1, len(floats)))(*[f.ctypes.data for f in floats]) if floats else None
        rc = get().savp_pipeline_next(self._h, images.ctypes.data, fp)
        if rc == EOF_CODE:
            return None
        if rc != 0:
            raise RuntimeError('input pipeline: %s (%s, %d)' % (get().savp_pipeline_error(self._h).decode(), ERRORS.get(rc, 'error'), rc))
        return images, floats

    def close(self):
        if self._h:
            get().savp_pipeline_destroy(self._h)
            self._h = c_vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
