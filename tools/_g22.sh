cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3t; mkdir -p $O
timeout 120 python tools/chol_trace.py 4096 float64 2>&1 | tail -8 | tee $O/chol_trace_4096.txt
timeout 120 python tools/chol_trace.py 2048 float32 2>&1 | tail -8 | tee $O/chol_trace_2048_f32.txt
