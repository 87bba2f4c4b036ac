"""Command line: same flags as reference torbi/__main__.py:12-53."""
import argparse
from pathlib import Path

import torbi_amd


def parse_args():
    parser = argparse.ArgumentParser(
        description='Viterbi-decode categorical distribution files on an MI355X')
    parser.add_argument('--input_files', type=Path, nargs='+', required=True,
                        help='Time-varying categorical distribution files')
    parser.add_argument('--output_files', type=Path, nargs='+', required=True,
                        help='Files to save decoded indices')
    parser.add_argument('--transition_file', type=Path,
                        help='Categorical transition matrix file; defaults to uniform')
    parser.add_argument('--initial_file', type=Path,
                        help='Categorical initial distribution file; defaults to uniform')
    parser.add_argument('--log_probs', action='store_true',
                        help='Whether inputs are in (natural) log space')
    parser.add_argument('--gpu', type=int, help='GPU index to use for decoding')
    parser.add_argument('--num_threads', type=int, default=1,
                        help='Ignored (CPU thread count in the reference)')
    return parser.parse_known_args()[0]


if __name__ == '__main__':
    torbi_amd.from_files_to_files(**vars(parse_args()))
