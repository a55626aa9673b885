"""GPU parity tests proper: every HIP kernel through the C ABI vs the CPU oracle (fp64) on seeded inputs."""
import os

import pytest

pytestmark = pytest.mark.gpu


def _run(fn):
    from tests import gpu_checks
    res = fn()
    bad = gpu_checks.failures(res)
    assert not bad, 'parity failures (name, rel err, tol): %r' % bad


def test_conv_fprop_dgrad_wgrad():
    from tests import gpu_checks
    _run(gpu_checks.check_conv)


def test_conv_views_and_epilogues():
    from tests import gpu_checks
    _run(gpu_checks.check_conv_views_and_epilogues)


def test_instnorm_act():
    from tests import gpu_checks
    _run(gpu_checks.check_inorm)


def test_convlstm_gates():
    from tests import gpu_checks
    _run(gpu_checks.check_lstm)


def test_util_ops():
    from tests import gpu_checks
    _run(gpu_checks.check_util)


def test_dense_few_rows():
    from tests import gpu_checks
    _run(gpu_checks.check_dense)


def test_cdna_and_composite():
    from tests import gpu_checks
    _run(gpu_checks.check_cdna_composite)


def test_small_ops():
    from tests import gpu_checks
    _run(gpu_checks.check_small)


def test_weight_prep_and_spectral_norm():
    from tests import gpu_checks
    _run(gpu_checks.check_weight_prep)


def test_conv_bf16_mode():
    from tests import gpu_checks
    _run(gpu_checks.check_conv_bf16)


def test_conv_thin_rgb_first_layer():
    from tests import gpu_checks
    _run(gpu_checks.check_conv_thin)


def test_bf16_activation_io():
    from tests import gpu_checks
    _run(gpu_checks.check_bf16_activation_io)


def test_fused_convlstm_cell_bf16():
    from tests import gpu_checks
    _run(gpu_checks.check_conv_cell)


def test_conv_epilogue_statistics_feed_the_instance_norm():
    from tests import gpu_checks
    _run(gpu_checks.check_conv_stats_fp32)


def test_ring_kernel_weight_warmup_does_not_change_results():
    from tests import gpu_checks
    _run(gpu_checks.check_ring_weight_warmup_invisible)


def test_weight_gradient_lds_dma_staging_equals_register_staging():
    from tests import gpu_checks
    _run(gpu_checks.check_wgrad_dma_staging)


def test_wide_to_thin_3x3_convolution_vs_fp64_taps_and_general_kernels():
    from tests import gpu_checks
    _run(gpu_checks.check_wide_thin_fprop)


def test_thin_to_wide_data_gradient_of_the_mask_convolution_vs_fp64_taps_and_general_kernels():
    from tests import gpu_checks
    _run(gpu_checks.check_thin8_wide_dgrad)


def test_flow_warp_and_dna():
    from tests import gpu_checks
    _run(gpu_checks.check_warp_dna)


def test_eval_metrics_and_sampling_fold():
    from tests import gpu_checks
    _run(gpu_checks.check_metrics)


def test_input_pipeline_to_device(tmp_path):
    """TFRecords (written by the oracle) -> C++ pipeline -> uint8 over PCIe -> savp_u8_frames_to_f32: float32 images in [0,1],
    bit-equal to tf.image.convert_image_dtype's x * (1/255) (base_dataset.py:187)."""
    import numpy as np
    import torch
    from oracle import tfrecord as R
    from video_prediction_amd import kernels as K
    from video_prediction_amd.datasets import get_dataset_class
    rng = np.random.default_rng(0)
    d = tmp_path / 'test'
    d.mkdir()
    frames = rng.integers(0, 256, (6, 30, 64, 64, 3), dtype=np.uint8)
    R.write_records(str(d / 'traj_0_to_5.tfrecords'),
                    [R.encode_example({'%d/image_aux1/encoded' % t: frames[i, t].tobytes() for t in range(30)}) for i in range(6)])
    u8 = torch.from_numpy(frames[:4, :12]).cuda().contiguous()
    out = torch.empty(12, 4, 64, 64, 3, device='cuda')
    K.u8_frames_to_f32(u8, out)
    want = (frames[:4, :12].astype(np.float32) * np.float32(1.0 / 255.0)).transpose(1, 0, 2, 3, 4)
    assert np.array_equal(out.cpu().numpy(), want)
    ds = get_dataset_class('bair')(str(tmp_path), mode='test', num_epochs=1, hparams='sequence_length=12')
    it = ds.make_batch(4)
    batch = next(it)
    assert batch['images'].shape == (4, 12, 64, 64, 3)
    assert np.array_equal(batch['images'].cpu().numpy(), frames[:4, :12].astype(np.float32) * np.float32(1.0 / 255.0))
    with pytest.raises(StopIteration):                      # 6 examples, batch 4, drop_remainder
        next(it)


def test_metric_kernels_against_committed_golden():
    """HIP psnr / mse / ssim vs tests/golden/metrics_golden.npz (fp64 oracle values committed with their generating script)."""
    import os
    import numpy as np
    import torch
    from video_prediction_amd import kernels as K
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'metrics_golden.npz'))
    a, b = torch.tensor(d['a']).cuda(), torch.tensor(d['b']).cuda()
    T, B = a.shape[:2]
    mse = torch.empty(T, B, device='cuda'); psnr = torch.empty(T, B, device='cuda'); ssim = torch.empty(T, B, device='cuda')
    K.frame_mse_psnr(a, b, mse=mse, psnr=psnr)
    K.frame_ssim(a, b, ssim)
    for name, got in (('mse', mse), ('psnr', psnr), ('ssim', ssim)):
        assert np.allclose(got.cpu().numpy(), d[name], rtol=2e-5, atol=0), name


@pytest.mark.parametrize('precision', ['bf16', 'f32'])
def test_every_shipped_tuning_table_instantiation_vs_fp64(precision):
    """The (problem, tile, split-K) pairs bench.py actually launches (video_prediction_amd/tuning_gfx950_*.json, N = 32 / 464 / 928 ...
    at their own shapes), each against an fp64 tap-loop reference: rel <= 1e-2 (bf16 operands) / 2e-5 (exact fp32)."""
    from tests import gpu_checks
    res = gpu_checks.check_tuning_table(precision)
    assert len(res) >= 90
    bad = gpu_checks.failures(res)
    assert not bad, 'parity failures (name, rel err, tol): %r' % bad[:20]


def test_tiled_z_gradient_and_gapped_gate_dgrad():
    from tests import gpu_checks
    _run(gpu_checks.check_tiled_z_and_gapped_dgrad)



def test_norm_backward_sums_from_the_dgrad_epilogue():
    from tests import gpu_checks
    _run(gpu_checks.check_norm_bwd_stats_epilogue)


@pytest.mark.gpu
def test_gate_convolution_kernel_vs_fp64_and_ring_kernel():
    from tests import gpu_checks
    _run(gpu_checks.check_gate_conv_kernel)


@pytest.mark.gpu
def test_convlstm_cell_forward_as_one_launch_vs_oracle_and_two_launch_path():
    from tests import gpu_checks
    _run(gpu_checks.check_one_launch_cell)
