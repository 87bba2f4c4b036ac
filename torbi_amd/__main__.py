"""`python -m torbi_amd`: decode observation files to index files on an MI355X (--gpu N) or on the CPU.

Flag names and meanings are those of the reference CLI (torbi/__main__.py:16-49) so scripts written
for it keep working; the decode itself is torbi_amd.from_files_to_files.
"""
import argparse
import pathlib
import sys

import torbi_amd

# flag -> argparse keywords.  Paths are pathlib.Path like upstream; unknown flags are ignored like upstream.
FLAGS = {
    'input_files': dict(type=pathlib.Path, nargs='+', required=True,
                        help='torch.save()d (frames, states) observation tensors, one per sequence'),
    'output_files': dict(type=pathlib.Path, nargs='+', required=True,
                         help='where the int32 (frames,) index tensors go, same order as --input_files'),
    'transition_file': dict(type=pathlib.Path, default=None,
                            help='(states, states) transition matrix indexed [next, prev]; uniform when omitted '
                                 '(decoded by the O(states)-per-frame kernel)'),
    'initial_file': dict(type=pathlib.Path, default=None,
                         help='(states,) initial distribution; uniform when omitted'),
    'log_probs': dict(action='store_true', help='the files already hold natural-log probabilities'),
    'gpu': dict(type=int, default=None, help='HIP device index (default: the CPU operator, like upstream)'),
    'num_threads': dict(type=int, default=1, help='worker threads of the CPU operator (no --gpu); ignored on a HIP device'),
}


def main(argv=None):
    parser = argparse.ArgumentParser(prog='python -m torbi_amd', description=__doc__.splitlines()[0])
    for name, keywords in FLAGS.items():
        parser.add_argument(f'--{name}', **keywords)
    options, _ = parser.parse_known_args(argv)
    torbi_amd.from_files_to_files(**vars(options))
    return 0


if __name__ == '__main__':
    sys.exit(main())
