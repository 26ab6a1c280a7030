cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3s; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 -p no:cacheprovider ) 2>&1 | grep -v "Warning\|warnings.warn" | tail -15 > $O/full_gpu.log
tail -15 $O/full_gpu.log
