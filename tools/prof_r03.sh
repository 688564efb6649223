# Round-3 evidence run (one MI355X): headline-only kernel summary, decode-only kernel trace, both checked against the
# figure bench.py printed in the same run.  usage: bash tools/prof_r03.sh [tag]
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-extra-legs --no-ceiling > $O/head_bench.log 2>&1
echo headline stats done
S=$(find $O/head -name "*kernel_stats.csv" | head -1)
cp $S $O/${TAG}_headline_kernel_stats.csv
python3 $R/tools/check_profile.py encoder $S $O/head_bench.log $O/${TAG}_headline_check.json > $O/head_check.log 2>&1 || echo "HEADLINE CHECK OUTSIDE 3%"
tail -5 $O/head_check.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python3 $R/bench.py --workload decode --decode-steps 64 > $O/dec_bench.log 2>&1
echo decode stats done
T=$(find $O/dec -name "*kernel_trace.csv" | head -1)
S2=$(find $O/dec -name "*kernel_stats.csv" | head -1)
cp $S2 $O/${TAG}_decode_kernel_stats.csv
python3 $R/tools/check_profile.py decode $T $O/dec_bench.log $O/${TAG}_decode_check.json > $O/dec_check.log 2>&1 || echo "DECODE CHECK OUTSIDE BAND"
tail -22 $O/dec_check.log
find $O -name "*kernel_trace.csv" -size +20M -delete
