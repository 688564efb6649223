cd $GRAFT_REPO_ROOT
V=$GRAFT_REPO_ROOT/tools/_variants
for pass in 0 1 2; do for v in head rawep; do
  echo "== $v"; PROBE_LIB=$V/lib_$v.so timeout -k 10 200 python tools/decode_probe.py 8 96 790 2>&1 | grep "ms/step"
done; done
