# GPU call 2 (round 3)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout 120 python tools/guard_probe.py > $O/guard_probe.txt 2>&1
timeout 400 python -m pytest tests/test_gpu_chol_blocked.py -q -x 2>&1 | tail -25 > $O/chol_tests.log
timeout 300 python tools/bench_chol.py 128 256 512 1024 2048 4096 > $O/chol_bench.txt 2>&1
timeout 200 python -m pytest tests/test_gpu_dotew.py -q -x 2>&1 | tail -15 > $O/dotew_tests.log
(echo "PACKA=0"; PTHIP_DOTEW_PACKA=0 timeout 120 python tools/dotew_variants.py 256 none; echo "PACKA=1"; PTHIP_DOTEW_PACKA=1 timeout 120 python tools/dotew_variants.py 256 none acc2) > $O/dotew_packa.txt 2>&1
for pf in 9999 24; do for un in 1 2 4; do
  echo "prefetch_min_ops=$pf unroll=$un"
  PTHIP_EW_PREFETCH_MIN_OPS=$pf PTHIP_EW_UNROLL=$un timeout 120 python tools/bench_configs.py c2 --reps 10 --no-check 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for k, v in d.items(): print('  ', k, 'ms_device', round(v['ms_device'], 5), 'frac', round(v['frac'], 4), 'kernel_ms', round(v.get('kernel_ms', 0), 5), 'kernel_frac', round(v.get('kernel_frac', 0), 4))
"
done; done > $O/c2_prefetch.txt 2>&1
timeout 500 python tools/pmc_kernels.py $O/c5_pmc.md dotew_ -- python tools/profile_c5_small.py > /dev/null 2> $O/c5_pmc.err
timeout 500 python tools/pmc_kernels.py $O/c2_pmc.md ew_ -- python tools/bench_configs.py c2 --reps 5 --no-check > /dev/null 2> $O/c2_pmc.err
timeout 300 python -m pytest tests/test_gpu_refsuite_scan.py -q --timeout 120 --tb=long -p no:cacheprovider -k "pushforward or some_truncate or inner_grad" 2>&1 | tail -150 > $O/refscan_tb.log
cat $O/guard_probe.txt; tail -4 $O/chol_tests.log; cat $O/chol_bench.txt; tail -3 $O/dotew_tests.log; cat $O/dotew_packa.txt; cat $O/c2_prefetch.txt
