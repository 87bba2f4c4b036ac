#!/bin/bash
# AddressSanitizer + UBSan (and ThreadSanitizer for the reader / writer threads) over the host-side native code (CPU only; GPU sanitizers are not available on this pool):
# the file reader / writer threads (csrc/file_rows.hpp) through tools/file_rows_check.cpp, and the CPU twin
# (csrc/torbi_cpu.cpp) against the oracle.   bash tools/sanitize_host.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d)
cd "$R/tools"
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -o "$W/file_rows_check" file_rows_check.cpp
"$W/file_rows_check"
g++ -O1 -g -std=c++17 -pthread -fsanitize=thread -o "$W/file_rows_check_tsan" file_rows_check.cpp      # the reader / writer threads
"$W/file_rows_check_tsan"
rm -f /tmp/torbi_file_rows_check_*
g++ -O1 -g -fopenmp -fPIC -shared -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-omit-frame-pointer \
    -I"$R/include" -o "$W/libtorbi_cpu_asan.so" "$R/torbi_amd/csrc/torbi_cpu.cpp"
cd "$R"
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 TWIN="$W/libtorbi_cpu_asan.so" python3 - <<'PY'
import ctypes, os, sys
import numpy as np
sys.path.insert(0, '.')
import oracle
from torbi_amd import synth
lib = ctypes.CDLL(os.environ['TWIN'])
lib.torbi_cpu_viterbi_decode.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4
for (B, T, S) in [(1, 3, 3), (3, 40, 17), (20, 30, 70), (9, 25, 360), (17, 12, 129), (8, 5, 1)]:
    obs, trans, init = synth.problem(B, T, S, seed=5)
    frames = np.clip(synth.lengths(B, 1, T, seed=2), 1, T).astype(np.int32)
    frames[0] = T
    for threads in (1, 3, 8):
        out = np.empty((B, T), np.int32)
        rc = lib.torbi_cpu_viterbi_decode(obs.ctypes.data, frames.ctypes.data, trans.ctypes.data, init.ctypes.data,
                                          out.ctypes.data, B, T, S, threads)
        assert rc == 0 and np.array_equal(out, oracle.decode(obs, frames, trans, init, num_threads=2)), (B, T, S, threads)
    if S > 3:           # items that met a NaN / +inf are decoded again in the reference's order (int32 trellis in the history)
        obs[0, T // 2, 1] = np.nan
        obs[B - 1, 0, 0] = np.inf
        trans[2, 0] = -np.inf
        out = np.empty((B, T), np.int32)
        rc = lib.torbi_cpu_viterbi_decode(obs.ctypes.data, frames.ctypes.data, trans.ctypes.data, init.ctypes.data,
                                          out.ctypes.data, B, T, S, 3)
        assert rc == 0 and np.array_equal(out, oracle.decode(obs, frames, trans, init, num_threads=2)), (B, T, S, 'nan')
print('CPU twin under ASan/UBSan: clean, equal to the oracle')
PY
rm -rf "$W"
