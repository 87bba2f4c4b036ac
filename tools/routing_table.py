"""The routing table of DESIGN.md section 4, straight from the library (torbi_hip_forward_path_on; 256 compute units
are assumed where no HIP device is visible, which is what an MI355X has).  tests/test_host_cpu.py asserts that DESIGN.md
carries exactly this text.
    python tools/routing_table.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

BATCHES = (1, 3, 4, 8, 16, 17, 512, 2049)
STATES = (256, 1440, 4096)          # (up to 192 states AUTO answers `small` whatever the batch)
PATHS = ('auto', 'dense', 'pruned', 'resident', 'cluster', 'held')


def table() -> str:
    from torbi_amd import viterbi
    lines = ['| B | S | ' + ' | '.join(PATHS) + ' |', '|---|---|' + '---|' * len(PATHS)]
    for S in STATES:
        for B in BATCHES:
            lines.append(f'| {B} | {S} | ' + ' | '.join(viterbi.forward_path(B, S, path=p) for p in PATHS) + ' |')
    return '\n'.join(lines)


if __name__ == '__main__':
    print(table())
