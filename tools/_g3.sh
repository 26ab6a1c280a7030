# GPU call 3 (round 3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_dotew.py -q -x 2>&1 | tail -15 > $O/dotew_tests.log
(for ov in 0 1; do echo "SCAN_OVERLAP=$ov"; PTHIP_SCAN_OVERLAP=$ov timeout 200 python tools/bench_configs.py c5 --reps 3 --no-check 2>&1 | tail -2; done) > $O/c5_overlap.txt 2>&1
for pf in 9999 24; do for un in 1 2 4; do
  echo "prefetch_min_ops=$pf unroll=$un"
  PTHIP_EW_PREFETCH_MIN_OPS=$pf PTHIP_EW_UNROLL=$un timeout 120 python tools/bench_configs.py c2 --reps 10 --no-check 2>/dev/null | python -c "
import sys, json
for line in sys.stdin:
    line = line.strip()
    if not line.startswith('{'): continue
    v = json.loads(line)
    print('  ', v['key'], 'ms_device', round(v['ms_device'], 5), 'frac', round(v['frac'], 4), 'kernel_ms', round(v.get('kernel_ms', 0), 5), 'kernel_frac', round(v.get('kernel_frac', 0), 4))
"
done; done > $O/c2_prefetch.txt 2>&1
cd /tmp; rocprofv3 --kernel-trace --stats -d /tmp/pk_chol -o k -- python $GRAFT_REPO_ROOT/tools/bench_chol.py 4096 > $GRAFT_REPO_ROOT/$O/chol4096_under_rocprof.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $(find /tmp/pk_chol -name "*.db" | head -1) > $GRAFT_REPO_ROOT/$O/chol4096_kernel_stats.md 2>&1
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_refsuite_scan.py -q --timeout 120 --tb=long -p no:cacheprovider -k "pushforward or some_truncate or inner_grad" 2>&1 | tail -150 > $O/refscan_tb.log
cat $O/c5_overlap.txt; cat $O/c2_prefetch.txt; head -20 $O/chol4096_kernel_stats.md; tail -3 $O/dotew_tests.log; tail -5 $O/refscan_tb.log
