R=${GRAFT_REPO_ROOT:-/root/repo}
export TORBI_HIP_RESIDENT_KR=1
for c in 7; do TORBI_HIP_LIBRARY=$R/tools/libtorbi_hip_cap$c.so python tools/cap_check.py 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-160; done
PROBE_ARGS="8 200" PROBE_REPS=2 python tools/variants_probe.py run base cap6 cap7 cap8
