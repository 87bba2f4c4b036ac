"""Batches of observation files read straight into pinned host memory.

The reference's many-file driver loads every file with `torch.load`, pads the batch with `pad_sequence` and only
then moves it (torbi/data/dataset.py:18-20, collate.py:24-31, core.py:417-425): three passes over every byte on
the host, in Python worker processes that hand whole batches back through shared memory.  The decode of a batch
takes a few milliseconds on an MI355X, so a many-file job is bound by exactly that host path.

`FileBatches` yields the same `(observation, batch_frames, batch_chunks, input_files)` tuples as
`data.loader(...)`'s collate (zero padding included), but a file's float32 payload is `pread` from its place in the
`torch.save` container directly into its row of the pinned batch buffer: one pass over the bytes, threads instead
of processes (the reads release the GIL), the next batch assembled while the current one is copied and decoded.

Only what `torch.save` writes for a plain contiguous float32 CPU tensor is taken this way (an uncompressed zip
container: `<name>/data.pkl` + `<name>/data/<key>`); anything else -- another dtype or layout, the legacy
non-zip format, chunked decoding (`core.MIN_CHUNK_SIZE`) -- makes `open_batches` return None and the caller uses
`data.loader` like the reference.
"""
import io
import os
import pickle
import queue
import struct
import threading
import zipfile
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch


class UnsupportedFile(Exception):
    """The file is not a plain float32 tensor in torch.save's zip container."""


def _rebuild_tensor(storage, offset, size, stride, *unused):
    return ('tensor', storage, int(offset), tuple(int(n) for n in size), tuple(int(n) for n in stride))


class _LayoutUnpickler(pickle.Unpickler):
    """Reads `data.pkl` of a torch.save container WITHOUT building tensors: only where the payload lives."""

    def find_class(self, module, name):
        if module == 'torch._utils' and name == '_rebuild_tensor_v2':
            return _rebuild_tensor
        if module == 'torch' and name.endswith('Storage'):
            return name
        if module == 'collections' and name == 'OrderedDict':
            return dict
        raise UnsupportedFile(f'{module}.{name} in data.pkl')

    def persistent_load(self, pid):
        return pid          # ('storage', storage type, key, location, numel)


def payload(path):
    """(frames, states, byte offset of the float32 payload) of a `torch.save`d (frames, states) tensor."""
    try:
        with open(path, 'rb') as handle:
            archive = zipfile.ZipFile(handle)
            members = {info.filename.split('/', 1)[-1]: info for info in archive.infolist()}
            if 'byteorder' in members and archive.read(members['byteorder']).strip() != b'little':
                raise UnsupportedFile('byte order')
            record = _LayoutUnpickler(io.BytesIO(archive.read(members['data.pkl']))).load()
            if not (isinstance(record, tuple) and record and record[0] == 'tensor'):
                raise UnsupportedFile('not a tensor')
            _, storage, offset, size, stride = record
            if storage[1] != 'FloatStorage' or len(size) != 2 or stride != (size[1], 1):
                raise UnsupportedFile('not a contiguous float32 (frames, states) tensor')
            info = members['data/' + str(storage[2])]
            if info.compress_type != zipfile.ZIP_STORED:
                raise UnsupportedFile('compressed payload')
            handle.seek(info.header_offset)
            header = handle.read(30)
            if len(header) != 30 or header[:4] != b'PK\x03\x04':
                raise UnsupportedFile('local header')
            name_bytes, extra_bytes = struct.unpack('<HH', header[26:30])
            start = info.header_offset + 30 + name_bytes + extra_bytes + 4 * offset
            if 4 * (offset + size[0] * size[1]) > info.file_size:
                raise UnsupportedFile('payload shorter than the tensor')
            return size[0], size[1], start
    except (zipfile.BadZipFile, KeyError, pickle.UnpicklingError, OSError, IndexError, TypeError, struct.error) as exc:
        raise UnsupportedFile(str(exc)) from exc


def _read_into(path, start, row):
    """pread the payload at `start` into the numpy view `row` (contiguous float32)."""
    view = memoryview(row).cast('B')
    fd = os.open(path, os.O_RDONLY)
    try:
        done = 0
        while done < len(view):
            got = os.preadv(fd, [view[done:]], start + done)
            if got <= 0:
                raise OSError(f'short read from {path}')
            done += got
    finally:
        os.close(fd)


class FileBatches:
    """Iterator over `(observation, batch_frames, batch_chunks, input_files)` for consecutive groups of
    `batch_size` files; the observation is a (rows, longest, states) float32 tensor in pinned memory when a HIP
    device is present.  `threads` readers fill a batch; one batch is prepared ahead of the consumer."""

    def __init__(self, input_files, batch_size, threads=None, pin_memory=None, ahead=1):
        self.input_files = list(input_files)
        self.batch_size = int(batch_size)
        self.threads = max(1, int(threads if threads else min(32, (os.cpu_count() or 4))))
        self.pin_memory = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        self.ahead = max(1, int(ahead))

    def __len__(self):
        return (len(self.input_files) + self.batch_size - 1) // self.batch_size

    def _assemble(self, pool, files):
        try:
            layouts = list(pool.map(payload, files))
            if any(layout[1] != layouts[0][1] for layout in layouts):
                raise UnsupportedFile('files of one batch differ in their number of states')
        except UnsupportedFile:
            # a file the direct reader does not take: this batch goes the reference's way (torch.load + collate)
            from . import data
            observation, batch_frames, batch_chunks, names = data.collate(
                [(torch.load(file, map_location='cpu'), file) for file in files])
            return (observation.pin_memory() if self.pin_memory else observation), batch_frames, batch_chunks, names
        states = layouts[0][1]
        longest = max(layout[0] for layout in layouts)
        observation = torch.empty((len(files), longest, states), dtype=torch.float32, pin_memory=self.pin_memory)
        rows = observation.numpy()

        def fill(k):
            frames, _, start = layouts[k]
            if frames:
                _read_into(files[k], start, rows[k, :frames])
            rows[k, frames:] = 0.0             # collate's zero padding (collate.py:24-31)

        list(pool.map(fill, range(len(files))))
        batch_frames = torch.tensor([layout[0] for layout in layouts])
        return observation, batch_frames, [1] * len(files), tuple(files)

    def __iter__(self):
        groups = [self.input_files[k:k + self.batch_size] for k in range(0, len(self.input_files), self.batch_size)]
        ready = queue.Queue(maxsize=self.ahead)
        stop = threading.Event()

        def produce():
            try:
                with ThreadPoolExecutor(max_workers=self.threads) as pool:
                    for files in groups:
                        if stop.is_set():
                            return
                        item = self._assemble(pool, files)
                        while not stop.is_set():
                            try:
                                ready.put(item, timeout=0.1)
                                break
                            except queue.Full:
                                continue
                ready.put(None)
            except BaseException as exc:      # surfaces in the consumer
                ready.put(exc)

        worker = threading.Thread(target=produce, name='torbi-file-batches', daemon=True)
        worker.start()
        try:
            while True:
                item = ready.get()
                if item is None:
                    return
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            worker.join(timeout=5.0)


def open_batches(input_files, batch_size, threads=None):
    """A `FileBatches` over `input_files` when the direct reader applies, else None (the caller falls back to
    `data.loader`).  The first and the last file are looked at; a batch holding a file in between that does not
    fit is loaded with `torch.load` + `collate`."""
    from . import core
    if core.MIN_CHUNK_SIZE is not None or not input_files:
        return None
    try:
        payload(input_files[0])
        payload(input_files[-1])
    except UnsupportedFile:
        return None
    return FileBatches(input_files, batch_size, threads=threads)
