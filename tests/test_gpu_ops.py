"""GPU parity tests proper: every HIP kernel through the C ABI vs the CPU oracle (fp64) on seeded inputs."""
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


def test_flow_warp_and_dna():
    from tests import gpu_checks
    _run(gpu_checks.check_warp_dna)


def test_eval_metrics_and_sampling_fold():
    from tests import gpu_checks
    _run(gpu_checks.check_metrics)
