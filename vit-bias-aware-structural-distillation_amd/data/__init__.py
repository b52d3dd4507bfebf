"""Dual-view input pipeline (reference ``src/data/datasets.py``) over local data."""
from .datasets import (create_dataloaders, create_eval_loader, build_eval_transform, dataset_info,  # noqa: F401
                       get_channel_stats, get_subset_indices, is_local_dataset)
