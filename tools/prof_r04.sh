# Round-4 evidence run (one MI355X).  usage: bash tools/prof_r04.sh part   (part = stats | pmc_enc | pmc_dec)
set -e
PART=${1:-stats}
TAG=r04
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
if [ "$PART" = "stats" ]; then
  # headline-only kernel summary, checked against the figure bench.py printed in the same run
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu --no-decode --no-extra-legs --no-ceiling > $O/head_bench.log 2>&1
  S=$(find $O/head -name "*kernel_stats.csv" | head -1); cp $S $O/${TAG}_headline_kernel_stats.csv
  python3 $R/tools/check_profile.py encoder $S $O/head_bench.log $O/${TAG}_headline_check.json > $O/head_check.log 2>&1 || echo "HEADLINE CHECK OUTSIDE 3%"
  tail -6 $O/head_check.log
  # decode-only kernel traces: bf16 and W8A16 at B = 8, bf16 at B = 16
  for cfg in "8 bf16 " "8 fp8 --decode-fp8" "16 bf16 "; do
    set -- $cfg; B=$1; NAME=$2; FLAG=$3
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/dec_${B}_$NAME -- python3 $R/bench.py --workload decode --decode-steps 64 --decode-batch $B $FLAG > $O/dec_${B}_$NAME.log 2>&1
    T=$(find $O/dec_${B}_$NAME -name "*kernel_trace.csv" | head -1); S2=$(find $O/dec_${B}_$NAME -name "*kernel_stats.csv" | head -1)
    cp $S2 $O/${TAG}_decode_b${B}_${NAME}_kernel_stats.csv
    python3 $R/tools/check_profile.py decode $T $O/dec_${B}_$NAME.log $O/${TAG}_decode_b${B}_${NAME}_check.json > $O/dec_${B}_${NAME}_check.log 2>&1 || echo "DECODE CHECK OUTSIDE BAND"
    tail -14 $O/dec_${B}_${NAME}_check.log
  done
  find $O -name "*kernel_trace.csv" -size +20M -delete
elif [ "$PART" = "pmc_enc" ]; then
  CMD="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-decode --no-extra-legs --no-ceiling"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/efetch -- $CMD > $O/efetch.log 2>&1; echo fetch done
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/ewrite -- $CMD > $O/ewrite.log 2>&1; echo write done
  F=$(find $O/efetch -name "*counter_collection.csv" | head -1); W=$(find $O/ewrite -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summarize.py $F $W $O/${TAG}_hbm_traffic_pmc.json kernels_$TAG "headline-only encoder step, B=32" > $O/etraffic.log 2>&1 || true
  cat $O/etraffic.log
  find $O -name "*.csv" -size +20M -delete
else
  for cfg in "8 bf16 PROBE_BF16_ONLY" "8 fp8 PROBE_FP8_ONLY" "16 bf16 PROBE_BF16_ONLY" "16 fp8 PROBE_FP8_ONLY"; do
    set -- $cfg; B=$1; NAME=$2; VAR=$3
    export $VAR=1 AFHIP_DECODE_GRAPH=0
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dfetch_${B}_$NAME -- python3 $R/tools/decode_probe.py $B 8 790 > $O/dfetch_${B}_$NAME.log 2>&1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dwrite_${B}_$NAME -- python3 $R/tools/decode_probe.py $B 8 790 > $O/dwrite_${B}_$NAME.log 2>&1
    unset $VAR
    F=$(find $O/dfetch_${B}_$NAME -name "*counter_collection.csv" | head -1); W=$(find $O/dwrite_${B}_$NAME -name "*counter_collection.csv" | head -1)
    python3 $R/tools/decode_traffic.py $F $W $O/${TAG}_decode_traffic_pmc.json B${B}_$NAME
    echo "B=$B $NAME done"
  done
  find $O -name "*.csv" -size +20M -delete
fi
