set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r02
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu > $O/stats_bench.log 2>&1
echo stats done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-decode --no-extra-legs > $O/fetch.log 2>&1
echo fetch done
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu --no-decode --no-extra-legs > $O/write.log 2>&1
echo write done
AFHIP_DECODE_GRAPH=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/dfetch -- python3 $R/tools/decode_probe.py 8 8 790 > $O/dfetch.log 2>&1
echo dfetch done
AFHIP_DECODE_GRAPH=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/dwrite -- python3 $R/tools/decode_probe.py 8 8 790 > $O/dwrite.log 2>&1
echo dwrite done
find $O -name "*.csv" | xargs ls -la | awk '{print $5, $9}'
# keep only what is small enough to merge back (<= 64 MiB)
find $O -name "*kernel_trace.csv" -size +20M -delete
