"""Default hyper-parameters of the reference model classes, restated verbatim as data.

Sources: BaseVideoPredictionModel.get_default_hparams_dict (/root/reference/video_prediction/models/base_model.py:
75-97), VideoPredictionModel.get_default_hparams_dict (:337-400), SAVPVideoPredictionModel.get_default_hparams_dict
(models/savp_model.py:779-822).  Published recipes (hparams/**/model_hparams.json) load on top of these unchanged.
"""
import itertools


def base_defaults():
    return dict(
        context_frames=-1,
        sequence_length=-1,
        repeat=1,
    )


def trainable_defaults():
    hparams = dict(
        batch_size=16,
        lr=0.001,
        end_lr=0.0,
        decay_steps=(200000, 300000),
        lr_boundaries=(0,),
        max_steps=300000,
        beta1=0.9,
        beta2=0.999,
        context_frames=-1,
        sequence_length=-1,
        clip_length=10,
        l1_weight=0.0,
        l2_weight=1.0,
        vgg_cdist_weight=0.0,
        feature_l2_weight=0.0,
        ae_l2_weight=0.0,
        state_weight=0.0,
        tv_weight=0.0,
        image_sn_gan_weight=0.0,
        image_sn_vae_gan_weight=0.0,
        images_sn_gan_weight=0.0,
        images_sn_vae_gan_weight=0.0,
        video_sn_gan_weight=0.0,
        video_sn_vae_gan_weight=0.0,
        gan_feature_l2_weight=0.0,
        gan_feature_cdist_weight=0.0,
        vae_gan_feature_l2_weight=0.0,
        vae_gan_feature_cdist_weight=0.0,
        gan_loss_type='LSGAN',
        joint_gan_optimization=False,
        kl_weight=0.0,
        kl_anneal='linear',
        kl_anneal_k=-1.0,
        kl_anneal_steps=(50000, 100000),
        z_l1_weight=0.0,
    )
    return dict(itertools.chain(base_defaults().items(), hparams.items()))


def savp_defaults():
    hparams = dict(
        l1_weight=1.0,
        l2_weight=0.0,
        n_layers=3,
        ndf=32,
        norm_layer='instance',
        use_same_discriminator=False,
        ngf=32,
        downsample_layer='conv_pool2d',
        upsample_layer='upsample_conv2d',
        activation_layer='relu',
        transformation='cdna',
        kernel_size=(5, 5),
        dilation_rate=(1, 1),
        where_add='all',
        use_tile_concat=True,
        learn_initial_state=False,
        rnn='lstm',
        conv_rnn='lstm',
        conv_rnn_norm_layer='instance',
        num_transformed_images=4,
        last_frames=1,
        prev_image_background=True,
        first_image_background=True,
        last_image_background=False,
        last_context_image_background=False,
        context_images_background=False,
        generate_scratch_image=True,
        dependent_mask=True,
        schedule_sampling='inverse_sigmoid',
        schedule_sampling_k=900.0,
        schedule_sampling_steps=(0, 100000),
        use_e_rnn=False,
        learn_prior=False,
        nz=8,
        num_samples=8,
        nef=64,
        use_rnn_z=True,
        ablation_conv_rnn_norm=False,
        ablation_rnn=False,
    )
    return dict(itertools.chain(trainable_defaults().items(), hparams.items()))


SAVP_DEPRECATED_KEYS = [  # savp_model.py:826-843
    'num_gpus', 'e_net', 'd_conditional', 'd_downsample_layer', 'd_net', 'd_use_gt_inputs',
    'acvideo_gan_weight', 'acvideo_vae_gan_weight', 'image_gan_weight', 'image_vae_gan_weight',
    'tuple_gan_weight', 'tuple_vae_gan_weight', 'gan_weight', 'vae_gan_weight', 'video_gan_weight',
    'video_vae_gan_weight',
]
