export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2k
mkdir -p $O
cd /tmp
for xl in 1 0; do for rows in 1 2 4; do PTHIP_GEMV_XLDS=$xl PTHIP_GEMV_ROWS=$rows timeout 200 python $R/tools/bench_gemv.py 2>&1 | grep -v Warn; done; done | tee $O/gemv_sweep.txt
cd $R; timeout 300 python -m pytest tests -m gpu -q -rs -k "collective or dist or torch" 2>&1 | tail -12
