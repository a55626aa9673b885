"""Datasets with the reference's class API (video_prediction/datasets/__init__.py:9-24) on the C++ input pipeline."""
from .softmotion_dataset import SoftmotionVideoDataset
from .kth_dataset import KTHVideoDataset


def get_dataset_class(dataset):
    dataset_mappings = {
        'bair': 'SoftmotionVideoDataset',
        'softmotion': 'SoftmotionVideoDataset',
        'softmotion30_v1': 'SoftmotionVideoDataset',
        'kth': 'KTHVideoDataset',
    }
    dataset_class = dataset_mappings.get(dataset, dataset)
    dataset_class = globals().get(dataset_class)
    if dataset_class is None:
        raise ValueError('Invalid dataset %s' % dataset)
    return dataset_class
