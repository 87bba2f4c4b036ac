"""Batching of observation files: the host-side mirror of reference torbi/data/
{dataset.py:10-29, collate.py:9-45, loader.py:10-25}.

On-disk format (reference torbi/data/preprocess/core.py:50-53, torbi/core.py:471-473):
inputs are `torch.save`d float (frames, states) tensors, outputs int32 (frames,) tensors.
With `core.MIN_CHUNK_SIZE` set, a file is cut at low-entropy frames (torbi_amd/chunk.py, reference
torbi/chunk.py): its pieces become consecutive batch rows, `batch_chunks` says how many rows each file owns,
and `separate` joins the decoded rows again.
"""
import torch


class Dataset(torch.utils.data.Dataset):
    """One item per file: (observation (frames, states), input_file) -- dataset.py:10-29"""

    def __init__(self, input_files):
        self.input_files = input_files

    def __getitem__(self, index):
        input_file = self.input_files[index]
        observation = torch.load(input_file, map_location='cpu')
        from . import core
        if core.MIN_CHUNK_SIZE is not None:          # dataset.py:22-23
            from .chunk import chunk
            observation = chunk(observation)
        return observation, input_file

    def __len__(self):
        return len(self.input_files)


def collate(batch):
    """(observation, batch_frames, batch_chunks, input_files) for a list of dataset items, with the
    observation zero-padded to the longest sequence -- the tuple of reference collate.py:9-33
    (`batch_frames` is int64 there too; `from_probabilities` casts it).

    An item whose observation is a list of tensors stands for one file cut into chunks
    (torbi_amd/chunk.py): its pieces become consecutive batch rows and `batch_chunks` remembers how
    many belong together."""
    if len(batch) == 0:
        raise ValueError('batch must contain at least 1 item')
    input_files = tuple(item[1] for item in batch)
    pieces, batch_chunks = [], []
    for observation, _ in batch:
        group = observation if isinstance(observation, list) else [observation]
        batch_chunks.append(len(group))
        pieces.extend(group)
    batch_frames = torch.tensor([piece.shape[0] for piece in pieces])
    # pad_sequence zero-fills past each sequence's end, batch-major: (rows, longest, states)
    observation = torch.nn.utils.rnn.pad_sequence(pieces, batch_first=True, padding_value=0.0)
    return observation, batch_frames, batch_chunks, input_files


def separate(indices, batch_chunks, batch_frames):
    """Undo `collate`'s chunk flattening on decoded indices: one 1-D tensor per file, the valid
    frames of its rows concatenated in order (reference collate.py:36-45)."""
    lengths = [int(n) for n in batch_frames]
    rows = [indices[row, :lengths[row]] for row in range(len(lengths))]
    joined, first = [], 0
    for count in batch_chunks:
        joined.append(torch.cat(rows[first:first + count]))
        first += count
    return joined


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
