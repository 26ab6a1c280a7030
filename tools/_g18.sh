cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3q; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_refsuite_index.py -q --timeout 120 -rf --tb=line -p no:cacheprovider 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed|^/|Error" | cut -c1-330 > $O/refindex.log
grep -c "^FAILED" $O/refindex.log; tail -2 $O/refindex.log; grep -v "^FAILED" $O/refindex.log | sed 's/^\/[^ ]*\///' | sort | uniq -c | sort -rn | head -40
