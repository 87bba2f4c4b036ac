"""Optional PyTorch-dispatcher registration: `torch.ops.torbi.viterbi_decode` on HIP devices and on the CPU.

The reference defines the operator schema in torbi/csrc/ops.cpp:16-18 and registers its CUDA
implementation with TORCH_LIBRARY_IMPL (torbi/csrc/cuda/viterbi.cu:365-367); torbi/viterbi.py:53
calls it through the dispatcher.  `register()` binds the same schema's CUDA key (= HIP on
PyTorch-ROCm) to the C-ABI decode, so code written against `torch.ops.torbi.viterbi_decode` runs
unchanged.  This is the stub of INTEGRATION.md, shipped.  The CPU key (reference: torbi/csrc/viterbi.cpp:237-239)
is bound to the operator's host twin (include/torbi_cpu.h), so the dispatcher picks by tensor device as upstream.
"""
import torch

from . import viterbi

SCHEMA = ('viterbi_decode(Tensor observation, Tensor batch_frames, Tensor transition, '
          'Tensor initial) -> Tensor')
_LIBRARY = None


def _viterbi_decode_hip(observation, batch_frames, transition, initial):
    return viterbi.decode(observation, batch_frames, transition, initial)


def _viterbi_decode_cpu(observation, batch_frames, transition, initial):
    # the thread count is torch's, as upstream sets it before calling the operator (torbi/viterbi.py:51-52)
    return viterbi.decode_cpu(observation, batch_frames, transition, initial, num_threads=torch.get_num_threads())


def register():
    """Idempotent; returns torch.ops.torbi.viterbi_decode."""
    global _LIBRARY
    if _LIBRARY is None:
        library = torch.library.Library('torbi', 'FRAGMENT')
        try:
            library.define(SCHEMA)
        except RuntimeError:
            pass      # schema already present (e.g. the reference's own extension is loaded)
        library.impl('viterbi_decode', _viterbi_decode_hip, 'CUDA')
        library.impl('viterbi_decode', _viterbi_decode_cpu, 'CPU')
        _LIBRARY = library
    return torch.ops.torbi.viterbi_decode
