"""Batching contract of the reference's data layer (the file-I/O half -- ITK,
NIfTI, private data -- is out of scope).

Items are dicts {"t1w": (1,*S) f32, "t2w": (1,*S) f32} in [-1, 1]
(code/GAN/GAN_final.py:386-396); a batch stacks them along dim 0.
"""
from __future__ import annotations

from typing import Dict, Iterator, List, Sequence

import torch


def collate(items: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
    """torch default_collate for this item type: stack each key along dim 0."""
    keys = items[0].keys()
    return {k: torch.stack([it[k] for it in items], dim=0) for k in keys}


class BatchLoader:
    """DataLoader(batch_size, shuffle=True) semantics of GAN_final.py:421-425:
    a fresh permutation each epoch, last partial batch kept."""

    def __init__(self, dataset, batch_size: int = 4, shuffle: bool = True, seed: int = 0, device=None):
        self.dataset, self.batch_size, self.shuffle, self.device = dataset, batch_size, shuffle, device
        self.gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self) -> Iterator[Dict[str, torch.Tensor]]:
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.gen).tolist() if self.shuffle else list(range(n))
        for i in range(0, n, self.batch_size):
            b = collate([self.dataset[j] for j in order[i:i + self.batch_size]])
            if self.device is not None:
                b = {k: v.to(self.device, non_blocking=True) for k, v in b.items()}
            yield b


class CustomDataLoader:
    """test_runs/GAN.py:204-233: sequential batches, wraps to index 0 when the
    next batch would overrun the dataset (so the tail is dropped), never stops."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size
        self.curr_index, self.n_elems = 0, len(dataset)

    def __iter__(self):
        return self

    def __next__(self):
        if self.curr_index + self.batch_size > self.n_elems:
            self.curr_index = 0
        items = [self.dataset[i] for i in range(self.curr_index, self.curr_index + self.batch_size)]
        self.curr_index += self.batch_size
        return collate(items)


class SyntheticPairs:
    """Synthetic T1/T2 pairs: uniform in [-1,1) like the output of
    ScaleIntensityRangePercentilesd(b_min=-1,b_max=1,clip=True) (GAN_final.py:386-394)."""

    def __init__(self, n: int, spatial: Sequence[int], seed: int = 1234):
        g = torch.Generator().manual_seed(seed)
        self.items: List[Dict[str, torch.Tensor]] = [
            {"t1w": torch.rand(1, *spatial, generator=g) * 2 - 1, "t2w": torch.rand(1, *spatial, generator=g) * 2 - 1}
            for _ in range(n)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]
