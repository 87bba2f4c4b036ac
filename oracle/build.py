"""Build recipes for the test oracle (TEST INFRASTRUCTURE, not product code).

    python oracle/build.py            # builds both targets that are buildable here

Targets
  oracle/libviterbi_oracle.so   gcc build of oracle/viterbi_oracle.c (the C restatement)
  oracle/_ref/libtorbi_ref.so   g++ build of the REFERENCE's own CPU operator, compiled from
                                the sources where they lie under /root/reference
                                (torbi/csrc/ops.cpp + torbi/csrc/viterbi.cpp; flags from
                                the reference's setup.py:60-65, "-O3 -fopenmp").  Nothing is
                                copied into this repository; only the .so lands in
                                oracle/_ref/ (git-ignored, but it travels with gpurun).  The
                                operator needs ATen/libtorch, which the image's own PyTorch
                                provides -- no stand-in headers or libraries are written.
                                Skipped silently when /root/reference is absent (GPU box).
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = '/root/reference'
ORACLE_SO = os.path.join(HERE, 'libviterbi_oracle.so')
REF_DIR = os.path.join(HERE, '_ref')
REF_SO = os.path.join(REF_DIR, 'libtorbi_ref.so')


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_oracle(force=False):
    src = os.path.join(HERE, 'viterbi_oracle.c')
    if force or _stale(ORACLE_SO, [src]):
        cmd = ['gcc', '-O3', '-fopenmp', '-ffp-contract=off', '-shared', '-fPIC',
               '-o', ORACLE_SO, src]
        subprocess.check_call(cmd)
    return ORACLE_SO


def build_ref(force=False):
    """Compile the reference CPU operator in place; returns the .so path or None."""
    srcs = [os.path.join(REFERENCE, 'torbi', 'csrc', 'ops.cpp'),
            os.path.join(REFERENCE, 'torbi', 'csrc', 'viterbi.cpp')]
    if not all(os.path.exists(s) for s in srcs):
        return REF_SO if os.path.exists(REF_SO) else None
    if not (force or _stale(REF_SO, srcs)):
        return REF_SO
    import torch
    from torch.utils import cpp_extension
    os.makedirs(REF_DIR, exist_ok=True)
    inc = []
    for p in cpp_extension.include_paths():
        inc += ['-isystem', p]
    inc += ['-isystem', sysconfig.get_paths()['include']]
    libdir = os.path.join(os.path.dirname(torch.__file__), 'lib')
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cmd = (['g++', '-std=c++17', '-O3', '-fopenmp', '-fPIC', '-shared',
            f'-D_GLIBCXX_USE_CXX11_ABI={abi}'] + inc + srcs +
           ['-o', REF_SO, f'-L{libdir}', f'-Wl,-rpath,{libdir}',
            '-ltorch', '-ltorch_cpu', '-lc10'])
    subprocess.check_call(cmd)
    return REF_SO


if __name__ == '__main__':
    print(build_oracle(force='--force' in sys.argv))
    print(build_ref(force='--force' in sys.argv))
