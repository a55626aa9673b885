"""Run a few named model-level checks and print every result line (developer aid: python tests/tools/run_checks.py where_add)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import gpu_model_checks as G  # noqa: E402


def where_add():
    res = []
    for wa in ('input', 'middle'):
        res += G.check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_where_add_' + wa, where_add=wa)
        res += G.check_train_step(B=1, T=4, nz=8, steps=1, tag='train_where_add_' + wa, where_add=wa, video_sn_vae_gan_weight=0.0,
                                  video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
        from video_prediction_amd import kernels as K
        K.set_conv_precision('bf16')
        try:
            res += [(n + '(bf16)', e, 5e-2) for n, e, t in G.check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_where_add_' + wa, where_add=wa)
                    if 'argmax' not in n]
        finally:
            K.set_conv_precision('f32')
    return res


def backgrounds():
    res = G.check_generator_forward(nz=8, B=1, T=5, tag='gen_fwd_bg_last_frames', context_frames=3, last_image_background=True,
                                    last_context_image_background=True)
    res += G.check_generator_forward(nz=0, B=1, T=5, tag='gen_fwd_bg_context_images', context_frames=3, context_images_background=True,
                                     prev_image_background=False)
    res += G.check_train_step(B=1, T=5, nz=8, steps=1, tag='train_bg_context_images', context_frames=3, context_images_background=True,
                              prev_image_background=False, video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                              vae_gan_feature_cdist_weight=0.0)
    res += G.check_train_step(B=1, T=5, nz=8, steps=1, tag='train_bg_last_frames', context_frames=3, last_image_background=True,
                              last_context_image_background=True, video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0,
                              vae_gan_feature_cdist_weight=0.0)
    return res


def mask_options():
    res = G.check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_independent_mask', dependent_mask=False)
    res += G.check_generator_forward(nz=8, B=1, T=4, tag='gen_fwd_no_scratch', generate_scratch_image=False)
    off = dict(video_sn_vae_gan_weight=0.0, video_sn_gan_weight=0.0, vae_gan_feature_cdist_weight=0.0)
    res += G.check_train_step(B=1, T=4, nz=8, steps=1, tag='train_no_scratch_independent_mask_flow', generate_scratch_image=False,
                              dependent_mask=False, transformation='flow', **off)
    res += G.check_train_step(B=1, T=4, nz=8, steps=1, tag='train_independent_mask', dependent_mask=False, **off)
    res += G.check_train_step(B=1, T=4, nz=8, steps=1, tag='train_no_scratch', generate_scratch_image=False, **off)
    return res


if __name__ == '__main__':
    bad = 0
    for n, e, t in {'where_add': where_add, 'backgrounds': backgrounds, 'mask_options': mask_options}[sys.argv[1]]():
        ok = e <= t
        bad += not ok
        print('%-4s %-70s %.3e (tol %.1e)' % ('ok' if ok else 'FAIL', n, e, t))
    sys.exit(1 if bad else 0)
