#!/bin/bash
# Round-6 measurement set ("r7" files): MFMA-pipe utilisation by counters, the small-graph call-overhead table,
# the gchain sweep on the final tree, the wide_200 variants.   usage: bash tools/profile_r7_measure.sh <tag> [parts]
TAG=${1:-r7}
shift
PARTS=${@:-mfma calls sweep wide}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for part in $PARTS; do
case $part in
mfma)
  : > $O/mfma_pmc.jsonl
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pm1_$TAG -o x -- python $R/tools/bench_gemm.py 4 2 > $O/mfma_gemm.log 2>&1
  python $R/tools/pmc_mfma.py $(find /tmp/pm1_$TAG -name "*.db" | head -1) gemm >> $O/mfma_pmc.jsonl
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d /tmp/pm2_$TAG -o x -- python $R/tools/profile_c5_small.py > $O/mfma_c5.log 2>&1
  python $R/tools/pmc_mfma.py $(find /tmp/pm2_$TAG -name "*.db" | head -1) dotew,gemm >> $O/mfma_pmc.jsonl
  cat $O/mfma_pmc.jsonl
  ;;
calls)
  python $R/tools/bench_call_overhead.py > $O/call_overhead.md 2> $O/call_overhead.err
  cat $O/call_overhead.md
  ;;
sweep)
  python $R/tools/bench_gchain_sweep.py > $O/gchain_sweep.txt 2> $O/gchain_sweep.err
  cut -c1-330 $O/gchain_sweep.txt
  ;;
wide)
  python $R/tools/bench_configs.py wide200 wide200gemm --reps 10 > $O/wide200.jsonl 2> $O/wide200.err
  cut -c1-900 $O/wide200.jsonl
  ;;
esac
done
