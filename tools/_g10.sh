# GPU call 10 (round 3): the whole GPU suite, no -x, with durations
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
timeout 1700 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider --durations=15 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -60 > $O/pytest_gpu.log
tail -45 $O/pytest_gpu.log
