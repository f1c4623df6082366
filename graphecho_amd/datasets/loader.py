"""Batches of raw samples handed to the GPU formatting step (the role torch DataLoader + MONAI play in the reference,
train_camus_echo.py:146-176) -- without worker processes: decoding a .mhd/.nii slice is microseconds next to a step."""
import random

import torch


class RawBatches:
    """Iterates `dataset` in (optionally shuffled) batches; yields ([frames...], [label maps...]) as lists of uint8 device
    tensors -- lists because CAMUS / CardiacUDA frames differ in size before the resize kernel."""

    def __init__(self, dataset, batch_size, device, shuffle=False, drop_last=False, seed=0):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self.shuffle, self.drop_last = shuffle, drop_last
        self.rng = random.Random(seed)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(range(len(self.dataset)))
        if self.shuffle:
            self.rng.shuffle(order)
        for i in range(0, len(order), self.batch_size):
            idx = order[i:i + self.batch_size]
            if self.drop_last and len(idx) < self.batch_size:
                return
            frames, labels = [], []
            for j in idx:
                f, m = self.dataset[j][:2]
                frames.append(torch.from_numpy(f).unsqueeze(0).to(self.device, non_blocking=True))
                labels.append(torch.from_numpy(m).unsqueeze(0).to(self.device, non_blocking=True))
            yield frames, labels
