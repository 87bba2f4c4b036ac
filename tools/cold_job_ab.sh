#!/bin/bash
# Cold many-file job with the ring of pinned chunks against whole-batch pinned slabs, each in a process of its own, the
# input files read once beforehand (GPU box):  bash tools/cold_job_ab.sh > gpurun_out/cold_job_ab.txt
# (arguments: "chunks:MB" pairs; 0 chunks = whole-batch slabs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pair in ${@:-6:256 0:256 6:256 0:256}; do
  echo "== TORBI_RING_CHUNKS=${pair%%:*} TORBI_RING_CHUNK_MB=${pair##*:}"
  PREREAD=1 TORBI_FILE_TIMINGS=1 TORBI_RING_CHUNKS=${pair%%:*} TORBI_RING_CHUNK_MB=${pair##*:} timeout 300 python3 tools/file_job_profile.py 4096 16 2>&1 | grep -E "^run|^  batch|^  slab|pre-read|cumulative|core.py|fastio.py|pipeline.py|viterbi.py|slabs.py|method|built-in"
done
