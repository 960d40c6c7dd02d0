export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # tag, env...
  tag=$1; shift
  cd /tmp
  env "$@" MCI_KERNEL_CACHE=/tmp/kc_$tag timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c4x_$tag -o s -- bash -c "cd $R && python tools/c4_prof.py 32" > $R/gpurun_out/c4x_$tag.log 2>&1
  cd $R
  f=$(find gpurun_out/c4x_$tag -name "*.db" | head -1)
  echo "== $tag $@"; python - "$f" <<'PY'
import sqlite3, sys
for n, k, a in sqlite3.connect(sys.argv[1]).execute("select name, total_calls, average from top_kernels"):
    if n.startswith("mci_vegas"):
        print("   %-20s calls %s avg %.1f us" % (n[:20], k, a / 1e3))
PY
}
if [ $# -gt 0 ]; then
  i=0
  for spec in "$@"; do i=$((i+1)); run v$i $spec; done
  exit 0
fi
run base X=1
run ec80 MCI_EC_BUDGET=81920
run ec80_t8 MCI_EC_BUDGET=81920 MCI_HIST_TILE_BINS=8192
run t8 MCI_HIST_TILE_BINS=8192
