"""Batching of observation files: the host-side mirror of reference torbi/data/
{dataset.py:10-29, collate.py:9-45, loader.py:10-25}.

On-disk format (reference torbi/data/preprocess/core.py:50-53, torbi/core.py:471-473):
inputs are `torch.save`d float (frames, states) tensors, outputs int32 (frames,) tensors.
Entropy-based chunking (reference torbi/chunk.py) is an approximation that changes results
and is out of scope; `batch_chunks` is kept in the collate tuple for signature parity.
"""
import torch


class Dataset(torch.utils.data.Dataset):
    """One item per file: (observation (frames, states), input_file) -- dataset.py:10-29"""

    def __init__(self, input_files):
        self.input_files = input_files

    def __getitem__(self, index):
        input_file = self.input_files[index]
        observation = torch.load(input_file, map_location='cpu')
        return observation, input_file

    def __len__(self):
        return len(self.input_files)


def collate(batch):
    """Zero-pad to the longest item; returns (observation, batch_frames, batch_chunks,
    input_files) exactly as collate.py:9-33 (batch_frames is int64 there too)."""
    observations, input_files = zip(*batch)

    if isinstance(observations[0], list):
        batch_chunks = [len(obs) for obs in observations]
        observations = sum(observations, [])
    else:
        batch_chunks = [1] * len(observations)
    batch_frames = torch.tensor([obs.shape[0] for obs in observations])

    batch = len(observations)
    if batch == 0:
        raise ValueError('batch must contain at least 1 item')

    max_frames = max(observation.shape[0] for observation in observations)

    observation = torch.zeros(
        (batch, max_frames, observations[0].shape[-1]), dtype=observations[0].dtype)

    for i, obs in enumerate(observations):
        observation[i, :obs.shape[0]] = obs

    return observation, batch_frames, batch_chunks, input_files


def separate(indices, batch_chunks, batch_frames):
    """Re-join chunked items (collate.py:36-45)."""
    start = 0
    separated = []
    for chunks in batch_chunks:
        frames = batch_frames[start:start + chunks]
        separated.append(
            torch.cat([indices[start + i, :frames[i]] for i in range(0, chunks)]))
        start += chunks
    return separated


def loader(input_files, num_workers=None, collate_fn=collate, batch_size=None, pin_memory=None):
    """DataLoader over observation files in the given order (loader.py:10-25).

    Batches are collated into pinned host memory when a HIP device is present (the reference has
    `pin_memory` commented out, loader.py:22): from_probabilities then copies them with a
    non-blocking H2D that overlaps the previous batch's decode (PCIe)."""
    from . import core
    if pin_memory is None:
        pin_memory = torch.cuda.is_available()
    return torch.utils.data.DataLoader(
        Dataset(input_files),
        num_workers=core.NUM_WORKERS if num_workers is None else num_workers,
        batch_size=core.BATCH_SIZE if batch_size is None else batch_size,
        shuffle=False,
        collate_fn=collate_fn,
        pin_memory=pin_memory)
