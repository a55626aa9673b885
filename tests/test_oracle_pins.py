"""More known-answer pins of the CPU oracle (round 3; `PARITY UNPINNED` stays true -- TensorFlow cannot run here -- but these remove
the largest surfaces that rested on restatement alone): the video discriminator's explicitly padded, strided conv3d against naive
loops (models/networks.py:76-99, ops.py:764-777), one spectral-norm step by hand and its gradient THROUGH sigma / u' / v by finite
differences (ops.py:1020-1049), CRC-32C / masked CRC / varint known answers from the published specifications (RFC 3720 B.4,
tensorflow/core/lib/hash/crc32c.h, protobuf encoding guide), and a TensorFlow V2 checkpoint assembled byte by byte in this file
(tensor_bundle.proto + LevelDB table format) read by the product reader."""
import os
import struct

import numpy as np
import pytest
import torch

from oracle import ops as O
from oracle import tf_ops as TF
from oracle import tfrecord as R


# ---- conv3d: tf.pad([[0,0],[1,1],[1,1],[1,1],[0,0]]) + VALID, strides 1 / (1,2,2) / (2,2,2) --------------------------------------
@pytest.mark.parametrize('k,strides', [(3, (1, 1, 1)), (4, (1, 2, 2)), (4, (2, 2, 2))])
def test_video_discriminator_conv3d_padding_and_strides_against_naive_loops(k, strides):
    rng = np.random.default_rng(0)
    N, D, H, W, Ci, Co = 2, 5, 6, 8, 3, 4
    x = rng.standard_normal((N, D, H, W, Ci))
    w = rng.standard_normal((k, k, k, Ci, Co))
    b = rng.standard_normal(Co)
    xp = np.zeros((N, D + 2, H + 2, W + 2, Ci))
    xp[:, 1:-1, 1:-1, 1:-1] = x
    od, oh, ow = [(n + 2 - k) // s + 1 for n, s in zip((D, H, W), strides)]
    want = np.zeros((N, od, oh, ow, Co))
    for n in range(N):
        for d in range(od):
            for i in range(oh):
                for j in range(ow):
                    patch = xp[n, d * strides[0]:d * strides[0] + k, i * strides[1]:i * strides[1] + k, j * strides[2]:j * strides[2] + k]
                    want[n, d, i, j] = np.tensordot(patch, w, axes=([0, 1, 2, 3], [0, 1, 2, 3])) + b      # cross-correlation, no flip
    paddings = [[0, 0], [1, 1], [1, 1], [1, 1], [0, 0]]
    got = O.conv3d(TF.pad_constant(torch.tensor(x), paddings), torch.tensor(w), torch.tensor(b), strides=strides, padding='VALID')
    assert got.shape == want.shape
    assert np.allclose(got.numpy(), want, atol=1e-12)


# ---- spectral norm --------------------------------------------------------------------------------------------------------------------
def test_spectral_norm_one_power_iteration_by_hand():
    """W = [[3, 0], [0, 1], [0, 0]] as a [K=3, C=2] matrix, u = [0.6, 0.8]: v = norm(u W^T) = norm([1.8, 0.8, 0]),
    u' = norm(v W), sigma = v W u'^T -- every number below worked out with a pocket calculator, not with the oracle."""
    W = torch.tensor([[3.0, 0.0], [0.0, 1.0], [0.0, 0.0]], dtype=torch.float64).reshape(3, 1, 2)     # any leading shape, C last
    u = torch.tensor([[0.6, 0.8]], dtype=torch.float64)
    W_bar, u_new = O.spectral_normed_weight(W, u)
    nv = (1.8 ** 2 + 0.8 ** 2) ** 0.5                       # |u W^T| = 1.96977...
    v = np.array([1.8 / nv, 0.8 / nv, 0.0])
    vw = np.array([3 * v[0], v[1]])                         # v W
    nu = (vw ** 2).sum() ** 0.5
    u1 = vw / nu
    sigma = float(vw @ u1)                                  # = |v W|
    assert abs(sigma - nu) < 1e-15 and abs(sigma - 2.7713557) < 1e-6        # sqrt((3*1.8/1.969772)^2 + (0.8/1.969772)^2)
    assert np.allclose(u_new.numpy(), u1[None], atol=1e-12)
    assert np.allclose(W_bar.reshape(3, 2).numpy(), W.reshape(3, 2).numpy() / sigma, atol=1e-12)
    # a rank-one matrix a b^T has its only singular value |a||b| after ONE iteration from any u not orthogonal to b
    a, b = torch.tensor([1.0, -2.0, 2.0], dtype=torch.float64), torch.tensor([0.5, 1.5], dtype=torch.float64)
    Wb, _ = O.spectral_normed_weight(torch.outer(a, b), torch.tensor([[1.0, 0.2]], dtype=torch.float64))
    assert abs(float(torch.linalg.matrix_norm(Wb, ord=2)) - 1.0) < 1e-12


def test_spectral_norm_gradient_flows_through_sigma_u_and_v_like_tf():
    """The reference has no stop_gradient around the power iteration (ops.py:1034-1043): d/dW of sum(G * W_bar(W)) is the TOTAL
    derivative, including the paths through v, u' and sigma.  Central finite differences of the same scalar (fp64) pin autograd's
    answer; a restatement that detached u' or v would fail this."""
    rng = np.random.default_rng(3)
    W = torch.tensor(rng.standard_normal((2, 2, 3, 4)), dtype=torch.float64, requires_grad=True)
    u = torch.tensor(rng.standard_normal((1, 4)), dtype=torch.float64)
    G = torch.tensor(rng.standard_normal((2, 2, 3, 4)), dtype=torch.float64)

    def f(Wt):
        return float((O.spectral_normed_weight(Wt, u)[0] * G).sum())
    (O.spectral_normed_weight(W, u)[0] * G).sum().backward()
    fd = torch.zeros_like(W)
    base = W.detach().clone()
    h = 1e-6
    flat = base.reshape(-1)
    for i in range(flat.numel()):
        p, m = flat.clone(), flat.clone()
        p[i] += h
        m[i] -= h
        fd.reshape(-1)[i] = (f(p.reshape(W.shape)) - f(m.reshape(W.shape))) / (2 * h)
    assert float((W.grad - fd).abs().max()) < 1e-7 * max(1.0, float(fd.abs().max()))
    # ... and it differs from the "sigma is a constant" shortcut G / sigma by a lot more than that
    with torch.no_grad():
        Wr = base.reshape(-1, 4)
        v = torch.nn.functional.normalize(u @ Wr.t(), dim=1)
        u1 = torch.nn.functional.normalize(v @ Wr, dim=1)
        sigma = float(v @ Wr @ u1.t())
    assert float((W.grad - G / sigma).abs().max()) > 1e-2


# ---- CRC-32C / varint / TFRecord framing -------------------------------------------------------------------------------------------------
def _crc32c_bitwise(data):
    """Third, table-less implementation (reflected polynomial 0x82F63B78) used only to build the hand-made bundle below."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 & -(crc & 1))
    return crc ^ 0xFFFFFFFF


def _mask(crc):
    return (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF          # tensorflow/core/lib/hash/crc32c.h: Mask()


RFC3720 = [(bytes(32), 0x8A9136AA), (b'\xff' * 32, 0x62A8AB43), (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C),
           (b'123456789', 0xE3069283)]


def test_crc32c_known_answers_rfc3720_in_oracle_product_and_local_implementations():
    from video_prediction_amd import io as sio
    for data, want in RFC3720:
        assert R.crc32c(data) == want
        assert sio.crc32c(data) == want
        assert _crc32c_bitwise(data) == want
        assert R.masked_crc32c(data) == _mask(want) == sio.masked_crc32c(data)
    assert _mask(0xE3069283) == ((0xE3069283 >> 15 | (0xE3069283 << 17 & 0xFFFFFFFF)) + 0xA282EAD8) & 0xFFFFFFFF


def test_varint_known_answers_from_the_protobuf_encoding_guide():
    from video_prediction_amd import checkpoint as CK
    for value, enc in ((0, b'\x00'), (1, b'\x01'), (127, b'\x7f'), (128, b'\x80\x01'), (150, b'\x96\x01'), (300, b'\xac\x02'),
                       (16384, b'\x80\x80\x01'), (2 ** 32 - 1, b'\xff\xff\xff\xff\x0f')):
        assert R._varint(value) == enc
        assert CK._enc_varint(value) == enc
        assert CK._varint(enc + b'\x55', 0) == (value, len(enc))


def test_tfrecord_framing_known_answer():
    """One record 'abc': uint64 length | masked crc of the length bytes | data | masked crc of the data (tf.io.TFRecordWriter)."""
    import tempfile
    want = struct.pack('<Q', 3) + struct.pack('<I', _mask(_crc32c_bitwise(struct.pack('<Q', 3)))) + b'abc' + struct.pack('<I', _mask(_crc32c_bitwise(b'abc')))
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, 'one.tfrecords')
        R.write_records(p, [b'abc'])
        assert open(p, 'rb').read() == want
        from video_prediction_amd import io as sio
        assert list(sio.read_records(p)) == [b'abc']


# ---- a V2 checkpoint written out byte by byte ----------------------------------------------------------------------------------------------
def test_product_reader_on_a_hand_assembled_v2_bundle(tmp_path):
    """Two float32 tensors, 'a' [2] and 'b/c' [1, 2], as TensorFlow's BundleWriter lays them out.  Every varint below is < 128, so
    each is one literal byte; protobuf field keys are (field << 3 | wire type).

    .data-00000-of-00001 : raw little-endian tensor bytes back to back ('a' at offset 0, 'b/c' at offset 8)
    .index (LevelDB table): data block { '' -> BundleHeaderProto, 'a' -> BundleEntryProto, 'b/c' -> BundleEntryProto },
                            empty metaindex block, index block { 'b/c' -> handle(data block) }, 48-byte footer."""
    from video_prediction_amd import checkpoint as CK
    a = np.array([1.5, -2.0], dtype='<f4')
    bc = np.array([[3.25, 4.0]], dtype='<f4')
    data = a.tobytes() + bc.tobytes()

    def entry(shape_dims, offset, raw):
        dims = b''.join(b'\x12\x02\x08' + bytes([d]) for d in shape_dims)         # TensorShapeProto.dim (field 2) { size (field 1) = d }
        return (b'\x08\x01'                                                       # dtype = DT_FLOAT (1)
                + b'\x12' + bytes([len(dims)]) + dims                             # shape
                + b'\x20' + bytes([offset])                                       # offset (field 4); shard_id 0 is the default, omitted
                + b'\x28' + bytes([len(raw)])                                     # size (field 5)
                + b'\x35' + struct.pack('<I', _mask(_crc32c_bitwise(raw))))       # crc32c (field 6, fixed32), masked
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'                                    # num_shards = 1; version { producer = 1 }
    items = [(b'', header), (b'a', entry([2], 0, a.tobytes())), (b'b/c', entry([1, 2], 8, bc.tobytes()))]

    def block(kvs):                                                               # every key a restart point (shared = 0)
        body, restarts = b'', []
        for k, v in kvs:
            restarts.append(len(body))
            body += b'\x00' + bytes([len(k)]) + bytes([len(v)]) + k + v
        restarts = restarts or [0]
        return body + b''.join(struct.pack('<I', r) for r in restarts) + struct.pack('<I', len(restarts))

    def trailer(blk):                                                             # compression type 0 + masked crc32c(block + type)
        return b'\x00' + struct.pack('<I', _mask(_crc32c_bitwise(blk + b'\x00')))
    d_blk, m_blk = block(items), block([])
    off_m = len(d_blk) + 5
    assert len(d_blk) < 128 and off_m + len(m_blk) + 5 < 128                      # so that the handles below are single-byte varints
    i_blk = block([(b'b/c', bytes([0, len(d_blk)]))])                              # separator key >= last key of the data block
    off_i = off_m + len(m_blk) + 5
    footer = bytes([off_m, len(m_blk)]) + bytes([off_i, len(i_blk)])
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    index = d_blk + trailer(d_blk) + m_blk + trailer(m_blk) + i_blk + trailer(i_blk) + footer
    prefix = str(tmp_path / 'model-7')
    open(prefix + '.index', 'wb').write(index)
    open(prefix + '.data-00000-of-00001', 'wb').write(data)
    got = CK.read_checkpoint(prefix)
    assert list(got) == ['a', 'b/c']
    assert got['a'].dtype == np.float32 and np.array_equal(got['a'], a) and np.array_equal(got['b/c'], bc) and got['b/c'].shape == (1, 2)
    assert CK.variable_names(prefix) == ['a', 'b/c']
    # a flipped payload byte is caught by the entry's checksum
    bad = bytearray(data)
    bad[1] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(bad))
    with pytest.raises(ValueError, match='checksum'):
        CK.read_checkpoint(prefix)
