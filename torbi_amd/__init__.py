"""torbi_amd -- MI355X-native batched Viterbi decoding behind torbi's API.

Drop-in for ONE path of maxrmorrison/torbi: from_probabilities / from_file* -> decode ->
viterbi_decode (reference torbi/core.py:110-368, torbi/viterbi.py:5-53,
torbi/csrc/ops.cpp:17), computed by hand-written gfx950 HIP kernels reached through the C
ABI in include/torbi_hip.h.  `gpu=None` selects, like upstream, the CPU operator: its twin behind
include/torbi_cpu.h (`decode_cpu`), an explicit entry point -- the HIP path has no CPU fallback.
"""
from .viterbi import decode, decode_batches, decode_cpu, decode_uniform, workspace_bytes, set_forward_path, forward_path
from .core import (BATCH_SIZE, NUM_WORKERS, from_probabilities, from_file, from_file_to_file,
                   from_files_to_files, from_dataloader, save, save_masked)
from . import data
from .chunk import chunk
from . import synth
from . import distributed
from . import timer
from .pipeline import DecodePipeline
from .state import reset as reset_path_state
from .core import release_job_memory

__all__ = ['decode', 'decode_batches', 'decode_cpu', 'chunk', 'decode_uniform', 'workspace_bytes', 'set_forward_path', 'forward_path', 'from_probabilities', 'from_file', 'from_file_to_file',
           'from_files_to_files', 'from_dataloader', 'save', 'save_masked', 'data', 'synth',
           'distributed', 'DecodePipeline', 'BATCH_SIZE', 'NUM_WORKERS', 'reset_path_state', 'release_job_memory', 'timer']
