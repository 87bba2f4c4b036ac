"""Build variants of the library with extra -D flags here, time them on the GPU box.

    python tools/variants_probe.py build NAME "-DTORBI_KR=2 ..."     # here: tools/libtorbi_hip_NAME.so
    python tools/variants_probe.py run NAME [NAME ...]               # GPU box: tools/resident_probe.py per variant
                                                                     # ("base" = the in-tree library)
"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def lib(name):
    return os.path.join(ROOT, 'torbi_amd', 'libtorbi_hip.so') if name == 'base' else os.path.join(ROOT, 'tools', f'libtorbi_hip_{name}.so')

if sys.argv[1] == 'build':
    name, flags = sys.argv[2], sys.argv[3].split()
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-ffp-contract=off',
                           '-fno-slp-vectorize', '-Wno-pass-failed', '-pthread', f'-I{ROOT}/include', *flags, '-o', lib(name),
                           f'{ROOT}/torbi_amd/csrc/torbi_hip.hip'])
else:
    args = os.environ.get('PROBE_ARGS', '8 200').split()
    script = os.environ.get('PROBE_SCRIPT', 'resident_probe.py')        # e.g. uniform_probe.py (every line is shown)
    for rep in range(int(os.environ.get('PROBE_REPS', '2'))):
        for name in sys.argv[2:]:
            env = dict(os.environ, TORBI_HIP_LIBRARY=lib(name))
            out = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', script), *args], env=env,
                                 capture_output=True, text=True).stdout
            if script != 'resident_probe.py':
                print(f'{name}:\n{out}', flush=True)
                continue
            line = [l for l in out.splitlines() if l.startswith(f'resident x{args[0]}')]
            print(f'{name:12s}', line[-1] if line else out[-300:], flush=True)
