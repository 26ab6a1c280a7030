# GPU call 8 (round 3): the whole GPU suite
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider 2>&1 | tail -40 > $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
