"""video_prediction_amd -- MI355X-native SAVP hot path (HIP/gfx950 kernels behind the reference's model API).

Pure-Python helpers (hparams, variable inventory) import without the native library; every compute entry point goes
through ``video_prediction_amd.lib`` which raises if ``libsavp_hip.so`` has not been built -- there is no CPU or
eager fallback.
"""
__all__ = ['hparams', 'variables']
