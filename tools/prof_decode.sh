# decode-only kernel trace of bench.py's decode leg, checked against the figure the bench printed in the same run.
# usage: bash tools/prof_decode.sh TAG [ENV=VAL ...]
set -e
TAG=$1; shift || true
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec -- python3 $R/bench.py --workload decode --decode-steps 64 > $O/dec_bench.log 2>&1
T=$(find $O/dec -name "*kernel_trace.csv" | head -1)
S2=$(find $O/dec -name "*kernel_stats.csv" | head -1)
cp $S2 $O/${TAG}_decode_kernel_stats.csv
python3 $R/tools/check_profile.py decode $T $O/dec_bench.log $O/${TAG}_decode_check.json > $O/dec_check.log 2>&1 || echo "DECODE CHECK OUTSIDE BAND"
tail -30 $O/dec_check.log
find $O -name "*kernel_trace.csv" -size +20M -delete
