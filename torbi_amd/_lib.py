"""ctypes binding of libtorbi_hip.so (C ABI declared in include/torbi_hip.h).

The library is built IN-TREE by `torbi_amd._lib.build()` (hipcc --offload-arch=gfx950) and is
the only compute path of a GPU request: there is no CPU or eager-PyTorch fallback (the CPU operator that `gpu=None`
selects lives in its own library, libtorbi_cpu.so, see build_cpu / load_cpu).  A missing
library, a missing symbol or an ABI mismatch raises at first use.
"""
import ctypes
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SOURCE = os.path.join(_HERE, 'csrc', 'torbi_hip.hip')
INCLUDE = os.path.join(ROOT, 'include')
# TORBI_HIP_LIBRARY: an alternative build of the library (tools/variants_probe.py: -D experiments)
LIBRARY = os.environ.get('TORBI_HIP_LIBRARY') or os.path.join(_HERE, 'libtorbi_hip.so')
ABI_VERSION = 15

# every symbol include/torbi_hip.h declares: name -> (restype, argtypes)
_c = ctypes
SYMBOLS = {
    'torbi_hip_abi_version': (_c.c_int, []),
    'torbi_hip_error_string': (_c.c_char_p, [_c.c_int]),
    'torbi_hip_device_count': (_c.c_int, []),
    'torbi_hip_compute_units': (_c.c_int, [_c.c_int]),
    'torbi_hip_workspace_bytes': (_c.c_size_t, [_c.c_int, _c.c_int, _c.c_int]),
    'torbi_hip_viterbi_decode': (_c.c_int, [
        _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
        _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'torbi_hip_viterbi_decode_uniform': (_c.c_int, [
        _c.c_void_p, _c.c_void_p, _c.c_float, _c.c_void_p, _c.c_void_p,
        _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'torbi_hip_viterbi_decode_uniform_probabilities': (_c.c_int, [
        _c.c_void_p, _c.c_void_p, _c.c_float, _c.c_void_p, _c.c_void_p,
        _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    'torbi_hip_viterbi_decode_profiled': (_c.c_int, [
        _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
        _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint,
        _c.POINTER(_c.c_float)]),
    'torbi_hip_viterbi_decode_batches': (_c.c_int, [
        _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint,
        _c.POINTER(_c.c_float)]),
    'torbi_hip_band_reach': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(_c.c_int),
                                        _c.POINTER(_c.c_int)]),
    'torbi_hip_band_members': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_int]),
    'torbi_hip_viterbi_decode_banded': (_c.c_int, [
        _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint,
        _c.POINTER(_c.c_float)]),
    'torbi_hip_band_reach_over': (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.POINTER(_c.c_int),
                                             _c.POINTER(_c.c_int), _c.POINTER(_c.c_float)]),
    'torbi_hip_band_members_over': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_int]),
    'torbi_hip_viterbi_decode_banded_over': (_c.c_int, [
        _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_float, _c.c_int, _c.c_void_p, _c.c_uint,
        _c.POINTER(_c.c_float)]),
    'torbi_hip_preparation_bytes': (_c.c_size_t, [_c.c_int]),
    'torbi_hip_viterbi_decode_batches_prepared': (_c.c_int, [
        _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint,
        _c.POINTER(_c.c_float), _c.c_void_p, _c.c_size_t, _c.POINTER(_c.c_int)]),
    'torbi_hip_viterbi_decode_ex': (_c.c_int, [
        _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p,
        _c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint]),
    'torbi_hip_scan_stats': (_c.c_int, [_c.c_void_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p,
                                        _c.c_int, _c.c_void_p, _c.c_uint]),
    'torbi_hip_set_forward_path': (_c.c_int, [_c.c_int]),
    'torbi_hip_forward_path': (_c.c_int, [_c.c_int, _c.c_int]),
    'torbi_hip_forward_path_on': (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_uint]),
    'torbi_hip_last_forward_kernel': (_c.c_int, [_c.c_char_p, _c.c_size_t]),
    'torbi_hip_read_posterior': (_c.c_int, [
        _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.c_void_p,
        _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_uint]),
    'torbi_hip_epsilon_clamp': (_c.c_int, [_c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p]),
    'torbi_hip_log_epsilon_clamp': (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_uint64, _c.c_int, _c.c_void_p]),
    'torbi_hip_fill_synthetic': (_c.c_int, [
        _c.c_void_p, _c.c_uint64, _c.c_uint64, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
}

MAX_BATCHES = 16        # TORBI_HIP_MAX_BATCHES


class Batch(_c.Structure):
    """torbi_hip_batch (include/torbi_hip.h): one batch of a torbi_hip_viterbi_decode_batches call."""
    _fields_ = [('observation', _c.c_void_p), ('batch_frames', _c.c_void_p), ('indices_out', _c.c_void_p),
                ('workspace', _c.c_void_p), ('workspace_bytes', _c.c_size_t), ('B', _c.c_int), ('T', _c.c_int)]


_LIB = None

# the host twin (include/torbi_cpu.h): a separate library, g++ -fopenmp, no HIP dependency
CPU_SOURCE = os.path.join(_HERE, 'csrc', 'torbi_cpu.cpp')
CPU_LIBRARY = os.path.join(_HERE, 'libtorbi_cpu.so')
CPU_ABI_VERSION = 3
_CPU_LIB = None


class TorbiHipError(RuntimeError):
    """A non-zero return code from libtorbi_hip.so."""


def hipcc():
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise FileNotFoundError('hipcc not found (set HIPCC or install ROCm under /opt/rocm)')


def build(force=False, verbose=False):
    """Compile csrc/torbi_hip.hip for gfx950 into torbi_amd/libtorbi_hip.so (in-tree)."""
    csrc = os.path.dirname(SOURCE)
    deps = [os.path.join(INCLUDE, 'torbi_hip.h')] + [
        os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith(('.hip', '.hpp', '.h'))]
    if not force and os.path.exists(LIBRARY):
        newest = max(os.path.getmtime(d) for d in deps)
        if os.path.getmtime(LIBRARY) >= newest:
            return LIBRARY
    cmd = [hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-ffp-contract=off', '-fno-slp-vectorize', '-Wno-pass-failed', '-pthread', f'-I{INCLUDE}', '-o', LIBRARY + '.tmp', SOURCE]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIBRARY + '.tmp', LIBRARY)
    global _LIB
    _LIB = None
    return LIBRARY


def load():
    """Load the library and bind every declared symbol; raises if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIBRARY):
        raise FileNotFoundError(
            f'{LIBRARY} is missing: the HIP extension has not been built. Run '
            '`python -c "import __graft_entry__ as g; g.build()"` (or torbi_amd._lib.build()). '
            'torbi_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIBRARY)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.torbi_hip_abi_version()
    if got != ABI_VERSION:
        raise RuntimeError(f'libtorbi_hip.so ABI version {got}, expected {ABI_VERSION}: rebuild')
    _LIB = lib
    return lib


def check(code, what='torbi_hip call'):
    if code != 0:
        msg = load().torbi_hip_error_string(code)
        raise TorbiHipError(f'{what} failed with code {code}: {msg.decode() if msg else "?"}')


def build_cpu(force=False, verbose=False):
    """Compile csrc/torbi_cpu.cpp into torbi_amd/libtorbi_cpu.so (in-tree; the CPU operator of gpu=None callers)."""
    deps = [CPU_SOURCE, os.path.join(INCLUDE, 'torbi_cpu.h'), os.path.join(os.path.dirname(CPU_SOURCE), 'file_rows.hpp')]
    if not force and os.path.exists(CPU_LIBRARY) and os.path.getmtime(CPU_LIBRARY) >= max(os.path.getmtime(d) for d in deps):
        return CPU_LIBRARY
    cxx = os.environ.get('CXX') or shutil.which('g++') or 'g++'
    cmd = [cxx, '-O3', '-std=c++17', '-fopenmp', '-pthread', '-fPIC', '-shared', '-ffp-contract=off', f'-I{INCLUDE}', '-o',
           CPU_LIBRARY + '.tmp', CPU_SOURCE]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    os.replace(CPU_LIBRARY + '.tmp', CPU_LIBRARY)
    global _CPU_LIB
    _CPU_LIB = None
    return CPU_LIBRARY


def load_cpu():
    """Load libtorbi_cpu.so and bind its entry points; raises if it has not been built."""
    global _CPU_LIB
    if _CPU_LIB is not None:
        return _CPU_LIB
    if not os.path.exists(CPU_LIBRARY):
        raise FileNotFoundError(f'{CPU_LIBRARY} is missing: run `python -c "import __graft_entry__ as g; g.build()"`')
    lib = ctypes.CDLL(CPU_LIBRARY)
    lib.torbi_cpu_abi_version.restype = _c.c_int
    lib.torbi_cpu_abi_version.argtypes = []
    lib.torbi_cpu_viterbi_decode.restype = _c.c_int
    lib.torbi_cpu_viterbi_decode.argtypes = [_c.c_void_p] * 5 + [_c.c_int] * 4
    # host side of the many-file job (include/torbi_cpu.h)
    lib.torbi_cpu_read_rows.restype = _c.c_int
    lib.torbi_cpu_read_rows.argtypes = [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int,
                                        _c.POINTER(_c.c_int)]
    lib.torbi_cpu_write_files.restype = _c.c_int
    lib.torbi_cpu_write_files.argtypes = [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.POINTER(_c.c_int)]
    lib.torbi_cpu_open_heads.restype = _c.c_int
    lib.torbi_cpu_open_heads.argtypes = [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_void_p, _c.c_void_p,
                                         _c.POINTER(_c.c_int)]
    if lib.torbi_cpu_abi_version() != CPU_ABI_VERSION:
        raise RuntimeError('libtorbi_cpu.so ABI mismatch: rebuild')
    _CPU_LIB = lib
    return lib


def host_io(gpu: bool = False):
    """(read_rows, write_files) of the many-file job's host side: libtorbi_cpu.so for GPU jobs and CPU jobs alike (the same
    host code, csrc/file_rows.hpp, behind include/torbi_cpu.h; `gpu` is accepted for the callers of rounds 2-4, which took
    these from libtorbi_hip.so for a GPU job -- a reader thread's first call then waited behind the HIP runtime's start-up)."""
    lib = load_cpu()
    return lib.torbi_cpu_read_rows, lib.torbi_cpu_write_files


def host_open_heads(gpu: bool = False):
    """`open_heads` of the library `host_io` takes its entry points from."""
    return load_cpu().torbi_cpu_open_heads


def check_io(code, what):
    """Return codes of the host I/O entry points (both libraries use the same numbering)."""
    if code != 0:
        raise TorbiHipError(f'{what} failed with code {code}: '
                            + ('invalid argument' if code == -1 else 'an item could not be read or written in full'))
