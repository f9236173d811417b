set -u
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out/r05z0
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 ) > gpurun_out/r05z0/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -12 gpurun_out/r05z0/pytest_gpu.log
bash tools/gpu_evidence_r05.sh
