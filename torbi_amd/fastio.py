"""Batches of observation files read straight into pinned host memory.

The reference's many-file driver loads every file with `torch.load`, pads the batch with `pad_sequence` and only
then moves it (torbi/data/dataset.py:18-20, collate.py:24-31, core.py:417-425): three passes over every byte on
the host, in Python worker processes that hand whole batches back through shared memory.  The decode of a batch
takes a few milliseconds on an MI355X, so a many-file job is bound by exactly that host path.

`FileBatches` yields the same `(observation, batch_frames, batch_chunks, input_files)` tuples as
`data.loader(...)`'s collate (zero padding included), but a file's float32 payload is `pread` from its place in the
`torch.save` container directly into its row of the pinned batch buffer by native threads
(`torbi_cpu_read_rows`, csrc/file_rows.hpp): one pass over the bytes, outside the interpreter, the next batch
assembled while the current one is copied and decoded.

Only what `torch.save` writes for a plain contiguous float32 CPU tensor is taken this way (an uncompressed zip
container: `<name>/data.pkl` + `<name>/data/<key>`); a batch holding anything else -- another dtype or layout, the
legacy non-zip format -- is loaded with `torch.load` + `collate` like the reference does, and with chunked decoding
(`core.MIN_CHUNK_SIZE`) `open_batches` returns None and the caller uses `data.loader`.
"""
import ctypes
import io
import os
import pickle
import struct
import threading
import zipfile

import numpy as np
import torch

from . import _lib


SERIAL_READS = os.environ.get('TORBI_SERIAL_READS', '1') != '0'
LAST_TIMINGS = None        # TORBI_FILE_TIMINGS=1: per-batch reader timings of the last FileBatches that ran (tools/)


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


_layouts = {}          # data.pkl bytes -> (storage type, key, offset, size, stride): files of one shape share them


def _tensor_record(pickled):
    known = _layouts.get(pickled)
    if known is None:
        record = _LayoutUnpickler(io.BytesIO(pickled)).load()
        if not (isinstance(record, tuple) and record and record[0] == 'tensor'):
            raise UnsupportedFile('not a tensor')
        _, storage, offset, size, stride = record
        known = (storage[1], str(storage[2]), offset, size, stride)
        if len(_layouts) > 4096:
            _layouts.clear()
        _layouts[pickled] = known
    storage_type, key, offset, size, stride = known
    if storage_type != 'FloatStorage' or len(size) != 2 or stride != (size[1], 1):
        raise UnsupportedFile('not a contiguous float32 (frames, states) tensor')
    return key, offset, size


def _scan_head(head):
    """Members of a torch.save container from its first bytes alone: {name: (data offset, size)}, walking the local
    file headers up to the first storage record (torch writes data.pkl, three small records, then the payload; all
    stored).  torch's writer leaves the sizes to a data descriptor behind each record, so a record's end is where
    the next descriptor signature stands and agrees with the distance walked.  None when the walk meets anything
    else (compression, a record that does not end inside `head`)."""
    members, at = {}, 0
    while at + 30 <= len(head) and head[at:at + 4] == b'PK\x03\x04':
        flags, method = struct.unpack_from('<HH', head, at + 6)
        packed, size, name_bytes, extra_bytes = struct.unpack_from('<IIHH', head, at + 18)
        data = at + 30 + name_bytes + extra_bytes
        if method != 0 or at + 30 + name_bytes > len(head):
            return None
        name = head[at + 30:at + 30 + name_bytes].split(b'/', 1)[-1]
        if name.startswith(b'data/'):
            members[name] = (data, None)
            return members
        if flags & 8:
            # sizes follow the data: signature, crc32, packed size, size (4 bytes each, 8 in the zip64 form)
            mark = data
            while True:
                mark = head.find(b'PK\x07\x08', mark)
                if mark < 0 or mark + 24 > len(head):
                    return None
                if struct.unpack_from('<I', head, mark + 12)[0] == mark - data:
                    size, after = mark - data, mark + 16
                    break
                if struct.unpack_from('<Q', head, mark + 16)[0] == mark - data:
                    size, after = mark - data, mark + 24
                    break
                mark += 4
        else:
            if packed != size or size == 0xFFFFFFFF:
                return None
            after = data + size
        members[name] = (data, size)
        at = after
    return None


def payload_of_open_file(fd, head=None):
    """(frames, states, byte offset of the float32 payload) of the torch.save()d tensor behind `fd`: one 4 KB read
    (unless the caller has it: `head`) and a walk over the local headers.  LookupError when that does not get there
    (`payload` then goes through the central directory)."""
    if head is None:
        head = os.pread(fd, 4096, 0)
    members = _scan_head(head)
    if members is None or b'data.pkl' not in members:
        raise LookupError
    start, size = members[b'data.pkl']
    if start + size > len(head):
        raise LookupError
    if b'byteorder' in members:
        at, n = members[b'byteorder']
        if head[at:at + n].strip() != b'little':
            raise UnsupportedFile('byte order')
    key, offset, shape = _tensor_record(bytes(head[start:start + size]))
    entry = members.get(b'data/' + key.encode())
    if entry is None:
        raise LookupError
    # the storage record's size is only in the descriptor behind it; what can be checked from here is that the tensor
    # the pickle describes fits between the payload's start and the end of the file (torch.load would raise on a
    # header that promises more than the record holds; a file cut short also shows up as a failed read)
    start = entry[0] + 4 * offset
    if start + 4 * shape[0] * shape[1] > os.fstat(fd).st_size:
        raise UnsupportedFile('payload shorter than the tensor')
    return shape[0], shape[1], start


def payload(path):
    """(frames, states, byte offset of the float32 payload) of a `torch.save`d (frames, states) tensor, through
    the container's central directory."""
    try:
        with open(path, 'rb') as handle:
            archive = zipfile.ZipFile(handle)
            members = {info.filename.split('/', 1)[-1]: info for info in archive.infolist()}
            if 'byteorder' in members and archive.read(members['byteorder']).strip() != b'little':
                raise UnsupportedFile('byte order')
            key, offset, size = _tensor_record(archive.read(members['data.pkl']))
            info = members['data/' + key]
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


HEAD_BYTES = 4096


def _open_payloads(paths, gpu, threads):
    """`_open_payload` for a batch: the files are opened and their heads read by native threads in one call
    (torbi_cpu_open_heads); the interpreter only walks the headers it is handed.  [(fd, frames, states, offset)];
    on any failure every descriptor is closed again."""
    count = len(paths)
    names = [os.fsencode(os.fspath(p)) for p in paths]
    table = (ctypes.c_char_p * count)(*names)
    fds = np.full(count, -1, dtype=np.int32)
    heads = np.empty(count * HEAD_BYTES, dtype=np.uint8)
    lengths = np.zeros(count, dtype=np.int32)
    error = ctypes.c_int(0)
    code = _lib.host_open_heads(gpu)(table, count, max(1, min(int(threads), count)), HEAD_BYTES, fds.ctypes.data,
                                     heads.ctypes.data, lengths.ctypes.data, ctypes.byref(error))
    opened = []
    try:
        if code <= -100:
            raise OSError(error.value, os.strerror(error.value) if error.value else 'could not read', os.fsdecode(names[-(code + 100)]))
        _lib.check_io(code, 'open_heads')
        view = memoryview(heads)
        for k in range(count):
            fd = int(fds[k])
            head = bytes(view[k * HEAD_BYTES:k * HEAD_BYTES + int(lengths[k])])
            try:
                opened.append((fd,) + payload_of_open_file(fd, head))
            except (LookupError, struct.error, pickle.UnpicklingError):
                opened.append((fd,) + payload(paths[k]))
        return opened
    except BaseException:
        for fd in fds.tolist():
            if fd >= 0:
                os.close(fd)
        raise


def _open_payload(path):
    """(fd, frames, states, payload offset); the caller closes the descriptor."""
    fd = os.open(path, os.O_RDONLY)
    try:
        try:
            return (fd,) + payload_of_open_file(fd)
        except (LookupError, struct.error, pickle.UnpicklingError):
            return (fd,) + payload(path)
    except BaseException:
        os.close(fd)
        raise


_descriptors_reserved = 0


def _reserve_descriptors(count):
    """Grow the process's descriptor table to `count` entries in ONE step.  The kernel doubles the table on demand, and in
    a process with several threads every doubling waits for an RCU grace period while all other `open()` calls of the
    process queue up behind it: the first 1024 files of a job -- two batches held open by the assembling threads -- cost
    0.4-0.8 s of `open()` (9 ms per call instead of 4 us; profiles/r05_cold_file_job.txt).  `F_DUPFD` asks for the lowest
    free descriptor >= count, which makes the kernel expand the table once; nothing is closed or kept."""
    global _descriptors_reserved
    if count <= _descriptors_reserved:
        return
    try:
        import fcntl
        import resource
        soft, _ = resource.getrlimit(resource.RLIMIT_NOFILE)
        target = min(int(count), int(soft) - 1) if soft != resource.RLIM_INFINITY else int(count)
        if target > 64:
            fd = os.open(os.devnull, os.O_RDONLY)
            try:
                os.close(fcntl.fcntl(fd, fcntl.F_DUPFD, target))
            finally:
                os.close(fd)
        _descriptors_reserved = count
    except (OSError, ValueError, ImportError):
        _descriptors_reserved = count         # (not fatal: the table then grows on demand as before)


class FileBatches:
    """Iterator over `(observation, batch_frames, batch_chunks, input_files)` for consecutive groups of
    `batch_size` files; the observation is a (rows, longest, states) float32 tensor in pinned memory when a HIP
    device is present.  `threads` native readers fill a batch; `ahead` batches are prepared ahead of the consumer."""

    def __init__(self, input_files, batch_size, threads=None, pin_memory=None, ahead=1, gpu=None):
        # which library reads the rows: libtorbi_hip.so for a GPU job, libtorbi_cpu.so for gpu=None (no HIP runtime needed)
        self.gpu = torch.cuda.is_available() if gpu is None else bool(gpu)
        self.input_files = list(input_files)
        self.batch_size = int(batch_size)
        # (more reader threads are not better: 16-32 copy 85-97 GB/s from the page cache into pinned memory on the GPU
        # box's EPYC, 64 drop to 33 GB/s, 128 to 25 -- tools/host_bw_probe.py)
        self.threads = max(1, min(32, int(threads if threads else min(32, (os.cpu_count() or 4)))))
        self.pin_memory = torch.cuda.is_available() if pin_memory is None else bool(pin_memory)
        self.ahead = max(1, int(ahead))
        self.producers = 2
        self._read_lock = threading.Lock()
        # optional: called by the assembling thread with the finished (pinned) observation, returns what is yielded in its
        # place -- the many-file driver starts the host-to-device copy here, so that copies are queued as soon as batches
        # exist, not when the consuming thread next comes round (torbi_amd/core.py::_Staging.upload)
        self.stage = None
        # ... or, in its place: called with ((rows, longest, states), batch_frames, fill) BEFORE anything is read; calls
        # fill(address, first, k) for the pieces it wants (rows first .. first + k - 1 read to `address`, padded like
        # collate.py:24-31) and returns what is yielded (torbi_amd/core.py::_Staging.upload_rows: a ring of pinned chunks)
        self.stage_rows = None
        self._window = None
        self.timings = [] if os.environ.get('TORBI_FILE_TIMINGS') else None     # (open + headers, slab, native read, bytes)
        # descriptors of `producers + ahead` batches are open at once (+ the savers' and the interpreter's own)
        # (in a thread of its own: the one expansion waits for an RCU grace period -- 0.13 s on the GPU box's 256 logical
        # CPUs -- while the calling thread prepares the transition matrix and the job's streams; an open() that needs the
        # larger table waits for it in the kernel, every other one goes ahead)
        threading.Thread(target=_reserve_descriptors, args=(self.batch_size * (self.producers + self.ahead + 1) + 256,),
                         name='torbi-descriptor-table', daemon=True).start()
        self._workers = None
        self._groups = None

    def more_ready(self):
        """True while the batch after the one just yielded is already assembled (or the job is over).  The many-file
        driver launches a partial group instead of waiting for a full one when the reader is the slower side."""
        window = self._window
        return window is None or not window or window[0].done()

    def __len__(self):
        return (len(self.input_files) + self.batch_size - 1) // self.batch_size

    def _assemble(self, files):
        import time
        opened = []
        t0 = time.perf_counter()
        try:
            try:
                opened = _open_payloads(files, self.gpu, self.threads // self.producers)
                if any(entry[2] != opened[0][2] for entry in opened):
                    raise UnsupportedFile('files of one batch differ in their number of states')
            except UnsupportedFile:
                # a file the direct reader does not take: this batch goes the reference's way (torch.load + collate)
                from . import data
                observation, batch_frames, batch_chunks, names = data.collate(
                    [(torch.load(file, map_location='cpu'), file) for file in files])
                return (observation.pin_memory() if self.pin_memory else observation), batch_frames, batch_chunks, names
            count, states = len(files), opened[0][2]
            longest = max(entry[1] for entry in opened)
            row_bytes = 4 * longest * states
            t1 = time.perf_counter()
            if self.stage_rows is not None and self.pin_memory:
                return self._assemble_through_ring(files, opened, count, longest, states, t0, t1)
            if self.pin_memory:
                # a pinned slab of the process-wide pool (torbi_amd/slabs.py): the consumer hands it back with the event of
                # its host-to-device copy (`observation.torbi_slab`); one that never does just lets it be collected
                from . import slabs
                slab = slabs.pool(None).take(count * row_bytes, limit=4)     # two being read, one ready, one being copied
                observation = slab[:count * row_bytes].view(torch.float32).view(count, longest, states)
                observation.torbi_slab = slab
            else:
                observation = torch.empty((count, longest, states), dtype=torch.float32)
            fds = np.array([entry[0] for entry in opened], dtype=np.int32)
            frames = np.array([entry[1] for entry in opened], dtype=np.int64)
            offsets = np.array([entry[3] for entry in opened], dtype=np.int64)
            sizes = 4 * states * frames                                                      # payload bytes
            rows = observation.data_ptr() + row_bytes * np.arange(count, dtype=np.int64)     # row addresses
            zeros = row_bytes - sizes                                      # collate's zero padding (collate.py:24-31)
            error = ctypes.c_int(0)
            read_rows, _ = _lib.host_io(self.gpu)
            t2 = time.perf_counter()
            # one native read at a time, with every reader thread: two concurrent reads share the same 85 GB/s (and 64
            # threads collapse it to 33), so serialising them costs nothing and hands the first batch of a job over in
            # a third of the time; the other assembling thread does its opens and headers meanwhile
            if SERIAL_READS:
                with self._read_lock:
                    code = read_rows(fds.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, rows.ctypes.data,
                                     zeros.ctypes.data, count, max(1, min(self.threads, count)), ctypes.byref(error))
            else:
                code = read_rows(fds.ctypes.data, offsets.ctypes.data, sizes.ctypes.data, rows.ctypes.data, zeros.ctypes.data,
                                 count, max(1, min(self.threads // self.producers, count)), ctypes.byref(error))
            if code <= -100:
                raise OSError(error.value, f'could not read {files[-(code + 100)]} in full')
            _lib.check_io(code, 'read_rows')
            if self.timings is not None:
                self.timings.append((t1 - t0, t2 - t1, time.perf_counter() - t2, count * row_bytes))
            batch_frames = torch.from_numpy(frames.copy())
            if self.stage is not None:
                observation = self.stage(observation, batch_frames)
            return observation, batch_frames, [1] * count, tuple(files)
        finally:
            for entry in opened:
                os.close(entry[0])

    def _assemble_through_ring(self, files, opened, count, longest, states, t0, t1):
        """The batch goes to the device in pieces of RING_CHUNK_BYTES: rows are read into a pinned chunk of a small ring and
        copied from there (`stage_rows`: torbi_amd/core.py::_Staging.upload_rows), the chunk returns to the ring behind its
        copy.  The first job of a process pins 1.5 GB instead of three or four whole-batch slabs (2.7 GB asked, 4 GB pinned
        each: the host allocator rounds to powers of two), and a batch's copy starts when its first chunk has been read."""
        import time
        row_bytes = 4 * longest * states
        fds = np.array([entry[0] for entry in opened], dtype=np.int32)
        frames = np.array([entry[1] for entry in opened], dtype=np.int64)
        offsets = np.array([entry[3] for entry in opened], dtype=np.int64)
        sizes = 4 * states * frames
        zeros = row_bytes - sizes
        read_rows, _ = _lib.host_io(self.gpu)
        spent = [0.0]

        def fill(address, first, k):
            rows = address + row_bytes * np.arange(k, dtype=np.int64)
            part = slice(first, first + k)
            f, o, b, z = (np.ascontiguousarray(a[part]) for a in (fds, offsets, sizes, zeros))
            error = ctypes.c_int(0)
            t = time.perf_counter()
            with self._read_lock:          # one native read at a time, with every reader thread (see _assemble)
                code = read_rows(f.ctypes.data, o.ctypes.data, b.ctypes.data, rows.ctypes.data, z.ctypes.data, k,
                                 max(1, min(self.threads, k)), ctypes.byref(error))
            spent[0] += time.perf_counter() - t
            if code <= -100:
                raise OSError(error.value, f'could not read {files[first - (code + 100)]} in full')
            _lib.check_io(code, 'read_rows')

        batch_frames = torch.from_numpy(frames.copy())
        observation = self.stage_rows((count, longest, states), batch_frames, fill)
        if self.timings is not None:
            self.timings.append((t1 - t0, time.perf_counter() - t1 - spent[0], spent[0], count * row_bytes))
        return observation, batch_frames, [1] * count, tuple(files)

    def start(self):
        """Begin assembling the first batches now (the iterator would at its first `next`): the many-file driver calls this
        once `stage` / `stage_rows` are set and prepares the transition matrix meanwhile."""
        import collections
        from concurrent.futures import ThreadPoolExecutor
        if self._workers is not None:
            return
        groups = self._groups = iter([self.input_files[k:k + self.batch_size]
                                      for k in range(0, len(self.input_files), self.batch_size)])
        workers = self._workers = ThreadPoolExecutor(max_workers=self.producers, thread_name_prefix='torbi-file-batches')
        window = self._window = collections.deque()
        for _ in range(self.ahead + self.producers - 1):
            files = next(groups, None)
            if files is not None:
                window.append(workers.submit(self._assemble, files))

    def __iter__(self):
        """Batches in order, assembled by `producers` threads that take alternate batches (a batch's Python side -- 512
        opens, header reads and closes under the interpreter lock, 10-15 ms -- then hides behind the native copy of the
        other one), at most `ahead` + `producers` batches ahead of the consumer."""
        self.start()
        groups, workers, window = self._groups, self._workers, self._window
        self._groups = self._workers = None
        try:
            def submit():
                files = next(groups, None)
                if files is not None:
                    window.append(workers.submit(self._assemble, files))
            while window:
                item = window.popleft().result()         # (an exception of the worker surfaces here)
                submit()
                yield item
        finally:
            if self.timings is not None:
                global LAST_TIMINGS
                LAST_TIMINGS = list(self.timings)
            for future in window:
                future.cancel()
            window.clear()
            workers.shutdown(wait=True)                  # (a batch being assembled finishes: bounded, no consumer needed)


def open_batches(input_files, batch_size, threads=None, pin_memory=None, gpu=None):
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
    try:
        _lib.host_io(torch.cuda.is_available() if gpu is None else bool(gpu))
    except (OSError, AttributeError):
        return None            # the native reader's library is not built: the reference's loader does the job
    return FileBatches(input_files, batch_size, threads=threads, pin_memory=pin_memory, gpu=gpu)


# ---- outputs: one small torch.save container per file ------------------------------------------------------------
# The reference writes every decoded sequence with torch.save(tensor[..., :length].clone()) (torbi/core.py:466-473):
# ~0.1 ms of interpreter work per file, which on a 40 000-file job is ten times the decode.  `save_indices` writes
# the same container -- the records torch itself wrote for an int32 vector of that length (data.pkl, byteorder,
# version, ...), taken once per length from torch.save, around the new payload -- from a prebuilt image.

_images = {}           # length -> (bytearray image of the whole file, payload offset, [offsets of the payload's crc32])
_images_lock = threading.Lock()


def _container_image(length):
    """A stored-only zip image holding the records torch.save writes for a (length,) int32 CPU tensor, with sizes and
    checksums in the local headers, the payload 64-byte aligned like torch's own, and no serialization id (an
    optional record: it names one particular torch.save call)."""
    import zlib
    buffer = io.BytesIO()
    torch.save(torch.zeros((length,), dtype=torch.int32), buffer)
    archive = zipfile.ZipFile(buffer)
    records = []
    for info in archive.infolist():
        name = 'archive/' + info.filename.split('/', 1)[-1]
        if name.endswith('serialization_id'):
            continue
        records.append((name.encode(), archive.read(info)))
    image, directory, payload_at, crc_at = bytearray(), bytearray(), None, []
    for name, body in records:
        is_payload = name.startswith(b'archive/data/')
        extra = b''
        if is_payload:
            # pad with an extra field so the payload starts on a 64-byte boundary (torch's 'FB' padding field)
            start = len(image) + 30 + len(name) + 4
            pad = (-start) % 64
            extra = b'FB' + struct.pack('<H', pad) + b'Z' * pad
        crc = zlib.crc32(body) & 0xffffffff
        header_at = len(image)
        image += struct.pack('<4sHHHHHIIIHH', b'PK\x03\x04', 20, 0x0800, 0, 0, 0x21, crc, len(body), len(body),
                             len(name), len(extra)) + name + extra
        if is_payload:
            payload_at = len(image)
            crc_at.append(header_at + 14)
        image += body
        central_at = len(directory)
        directory += struct.pack('<4sHHHHHHIIIHHHHHII', b'PK\x01\x02', 20, 20, 0x0800, 0, 0, 0x21, crc, len(body),
                                 len(body), len(name), 0, 0, 0, 0, 0, header_at) + name
        if is_payload:
            crc_at.append(central_at + 16)          # relative to the directory; made absolute below
    directory_at = len(image)
    crc_at = [crc_at[0], directory_at + crc_at[1]]
    image += directory
    image += struct.pack('<4sHHHHIIH', b'PK\x05\x06', 0, 0, len(records), len(records), len(directory), directory_at, 0)
    if payload_at is None or len(records) < 2:
        raise UnsupportedFile('torch.save wrote no storage record')
    return image, payload_at, crc_at


def save_indices(indices, file):
    """Write a 1-D int32 CPU tensor so that torch.load(file) returns it (what the reference's save / save_masked
    leave on disk, torbi/core.py:466-473); anything else goes through torch.save."""
    out = _filled_image(indices)
    if out is None:
        torch.save(indices.clone(), file)
        return
    fd = os.open(os.fspath(file), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o666)
    try:
        view, done = memoryview(out), 0
        while done < len(out):
            done += os.write(fd, view[done:])
    finally:
        os.close(fd)


def _filled_image(indices):
    """The container bytes for one 1-D int32 CPU tensor (see save_indices), or None for anything else."""
    import zlib
    if not (isinstance(indices, torch.Tensor) and indices.dtype == torch.int32 and indices.dim() == 1
            and indices.device.type == 'cpu' and indices.numel() > 0):
        return None
    length = indices.numel()
    with _images_lock:
        known = _images.get(length)
        if known is None:
            if len(_images) > 2048:
                _images.clear()
            known = _images[length] = _container_image(length)
    image, payload_at, crc_at = known
    body = indices.contiguous().numpy().tobytes()
    out = bytearray(image)
    out[payload_at:payload_at + len(body)] = body
    crc = struct.pack('<I', zlib.crc32(body) & 0xffffffff)
    for at in crc_at:
        out[at:at + 4] = crc
    return out


def save_index_rows(rows, files, lengths, threads=8, gpu=None):
    """`save_indices(rows[k][:lengths[k]], files[k])` for a whole batch: the containers are put together here and
    written by native threads in one call (`torbi_cpu_write_files`)."""
    images, names = [], []
    for row, file, length in zip(rows, files, lengths):
        piece = row if length is None else row[..., :length]
        image = _filled_image(piece)
        if image is None:
            torch.save(piece.clone(), file)
            continue
        images.append(image)
        names.append(os.fsencode(os.fspath(file)))
    count = len(images)
    if not count:
        return
    paths = (ctypes.c_char_p * count)(*names)
    buffers = [(ctypes.c_char * len(image)).from_buffer(image) for image in images]
    data = (ctypes.c_void_p * count)(*[ctypes.addressof(buffer) for buffer in buffers])
    sizes = np.array([len(image) for image in images], dtype=np.int64)
    error = ctypes.c_int(0)
    _, write_files = _lib.host_io(torch.cuda.is_available() if gpu is None else bool(gpu))
    code = write_files(paths, data, sizes.ctypes.data, count, max(1, min(int(threads), count)), ctypes.byref(error))
    if code <= -100:
        raise OSError(error.value, f'could not write {os.fsdecode(names[-(code + 100)])}')
    _lib.check_io(code, 'write_files')
