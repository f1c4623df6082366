"""CAMUS patient tree -> raw (frame, label map) samples (reference datasets/camus.py:41-119).

Only the host side: patient listing and split, file naming, the missing-file resample loop and the .mhd decode.  Resize,
crop, /255 and the LV/LA one-hot planes (camus.py:98-105,121-159) run on the GPU in graphecho_amd.data.
"""
import glob
import os
import random

import numpy as np

from .formats import read_mhd

RANDOM_SEED = 123          # camus.py:35


class CamusSet:
    """`dataset_path/training/<patient>/<patient>_<view>.mhd`; stage in {'train', 'valid', 'test'}.

    Sample = (frame uint8 (1, H, W), label map uint8 (H, W), mask_index 0, index).  `class_values` names the label ids
    that become one-hot planes: (1, 3) = LV, LA when seg_parts (camus.py:98-101), else every id as stored.
    """

    def __init__(self, dataset_path, input_name, condition_name, stage, seg_parts=True, train_ratio=1.0,
                 valid_ratio=0.2, rng=None):
        self.dataset_path, self.input_name, self.condition_name = dataset_path, input_name, condition_name
        self.seg_parts = seg_parts
        self.class_values = (1, 3) if seg_parts else None
        patients = [d for d in sorted(glob.glob(os.path.join(dataset_path, "training", "*")))
                    if os.path.isdir(d) and os.listdir(d)]
        random.Random(RANDOM_SEED).shuffle(patients)                     # camus.py:60
        num_train = int(len(patients) * train_ratio)
        num_valid = int(num_train * valid_ratio)
        split = {"train": patients[num_valid:num_train], "valid": patients[:num_valid // 2],
                 "test": patients[num_valid // 2:num_valid]}             # camus.py:65-67
        if stage not in split:
            raise ValueError(f"stage must be train/valid/test, got {stage!r}")
        self.data_list = split[stage]
        self.rng = rng or random.Random(RANDOM_SEED)

    def __len__(self):
        return len(self.data_list)

    def get_path(self, patient_dir):
        pid = os.path.basename(patient_dir)
        return (os.path.join(patient_dir, f"{pid}_{self.input_name}.mhd"),
                os.path.join(patient_dir, f"{pid}_{self.condition_name}.mhd"))

    def __getitem__(self, index):
        if not self.data_list:
            raise IndexError("empty CAMUS split")
        image_path, label_path = self.get_path(self.data_list[index])
        tries = 0
        while not os.path.exists(image_path):
            # camus.py:90-93 redraws with randint(0, len) -- inclusive upper bound, an IndexError waiting to happen;
            # the redraw here stays in range and gives up after one pass over the split
            tries += 1
            if tries > 4 * len(self.data_list):
                raise FileNotFoundError(f"no patient of the split has {self.input_name}.mhd")
            index = self.rng.randrange(len(self.data_list))
            image_path, label_path = self.get_path(self.data_list[index])
        frame = np.squeeze(read_mhd(image_path))                         # camus.py:110-112
        label = np.squeeze(read_mhd(label_path))
        if frame.ndim != 2 or label.shape != frame.shape:
            raise ValueError(f"{image_path}: expected matching 2-D frame and label map, got {frame.shape} / {label.shape}")
        return frame.astype(np.uint8)[None], label.astype(np.uint8), 0, index
