cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_refsuite_index.py -q --timeout 120 --tb=short -p no:cacheprovider -k "test_boolean or test_grad_list or runtime_broadcast or w_2vec or with_broadcasting or index_broadcasting or AdvancedSubtensor_bool or blockwise_shape or (Blockwise__Cholesky and test_grad) or (MatrixInverse and test_grad) or (SolveVector and test_grad)" 2>&1 | grep -v "Warning\|warnings.warn" > $O/tb.log
wc -l $O/tb.log
