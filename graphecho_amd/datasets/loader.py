"""Batches of raw samples handed to the GPU formatting step (the role torch DataLoader + MONAI play in the reference,
train_camus_echo.py:146-176) -- without worker processes: decoding a .mhd/.nii slice is microseconds next to a step."""
import random

import torch


class RawBatches:
    """Iterates `dataset` in (optionally shuffled) batches; yields ([frames...], [label maps...]) as lists of uint8 device
    tensors -- lists because CAMUS / CardiacUDA frames differ in size before the resize kernel."""

    def __init__(self, dataset, batch_size, device, shuffle=False, drop_last=False, seed=0, rank=0, world=1):
        self.dataset, self.batch_size, self.device = dataset, batch_size, device
        self.shuffle, self.drop_last = shuffle, drop_last
        self.rank, self.world = rank, world      # data-parallel shard: every world-th sample of the (shared) order,
        self.rng = random.Random(seed)           # truncated to a common length like DistributedSampler(drop_last=True)

    def _shard(self, order):
        per = len(order) // self.world
        return order[self.rank:per * self.world:self.world] if self.world > 1 else order

    def __len__(self):
        n = len(self.dataset) // self.world if self.world > 1 else len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(range(len(self.dataset)))
        if self.shuffle:
            self.rng.shuffle(order)          # same seed on every rank -> same permutation, disjoint shards
        order = self._shard(order)
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
