"""CPU oracle for the torbi Viterbi decode operator -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (torbi_amd/) never imports, links or executes anything here.

  oracle.decode(...)      the C restatement (oracle/viterbi_oracle.c), numpy in / numpy out
  oracle.ref_decode(...)  the reference's own operator compiled from /root/reference into
                          oracle/_ref/ (available when that build exists); torch in / out

Parity status: pinned (see the header of viterbi_oracle.c and tests/test_oracle.py).
"""
import ctypes
import os

import numpy as np

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF_LOADED = False


def lib():
    global _LIB
    if _LIB is None:
        path = _build.build_oracle()
        L = ctypes.CDLL(path)
        L.torbi_oracle_viterbi_decode.restype = ctypes.c_int
        L.torbi_oracle_viterbi_decode.argtypes = [
            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_void_p, ctypes.c_void_p,
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.torbi_oracle_max_threads.restype = ctypes.c_int
        L.torbi_oracle_abi_version.restype = ctypes.c_int
        _LIB = L
    return _LIB


def decode(observation, batch_frames, transition, initial, num_threads=1, mode=1,
           return_posterior=False):
    """Decode with the C restatement.

    observation (B,T,S) float32, batch_frames (B,) int32, transition (S,S) float32
    [next, prev], initial (S,) float32 -- all already in log space, exactly what the
    reference's `torbi::viterbi_decode` receives (torbi/viterbi.py:53).
    mode 0 = reference-shaped (S*S scratch pass), mode 1 = fused.  Returns int32 (B,T).
    """
    obs = np.ascontiguousarray(np.asarray(observation), dtype=np.float32)
    frames = np.ascontiguousarray(np.asarray(batch_frames), dtype=np.int32)
    trans = np.ascontiguousarray(np.asarray(transition), dtype=np.float32)
    init = np.ascontiguousarray(np.asarray(initial), dtype=np.float32)
    B, T, S = obs.shape
    assert frames.shape == (B,) and trans.shape == (S, S) and init.shape == (S,)
    out = np.empty((B, T), dtype=np.int32)
    post = np.empty((B, S), dtype=np.float32) if return_posterior else None
    rc = lib().torbi_oracle_viterbi_decode(
        obs.ctypes.data, frames.ctypes.data, trans.ctypes.data, init.ctypes.data,
        out.ctypes.data, post.ctypes.data if post is not None else None,
        B, T, S, int(num_threads), int(mode))
    if rc != 0:
        raise ValueError(f'torbi_oracle_viterbi_decode failed with code {rc}')
    return (out, post) if return_posterior else out


def max_threads():
    return lib().torbi_oracle_max_threads()


def ref_available():
    return _build.build_ref() is not None


def ref_decode(observation, batch_frames, transition, initial, num_threads=1):
    """Run the REFERENCE's compiled CPU operator (oracle/_ref).  torch tensors in/out.

    Mirrors torbi/viterbi.py:51-53 (global thread count, then the dispatcher call).
    """
    global _REF_LOADED
    import torch
    if not _REF_LOADED:
        path = _build.build_ref()
        if path is None:
            raise RuntimeError('oracle/_ref is not built and /root/reference is absent')
        torch.ops.load_library(path)
        _REF_LOADED = True
    torch.set_num_threads(int(num_threads))
    return torch.ops.torbi.viterbi_decode(
        torch.as_tensor(observation, dtype=torch.float32).contiguous(),
        torch.as_tensor(batch_frames, dtype=torch.int32).contiguous(),
        torch.as_tensor(transition, dtype=torch.float32).contiguous(),
        torch.as_tensor(initial, dtype=torch.float32).contiguous())
