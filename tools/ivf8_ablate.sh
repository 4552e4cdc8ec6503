# A/B builds of the int8-MFMA scan (tools/build_variant.sh ivf8_<tag> ivfpq_mfma.hip -D...): kernel time per variant
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in $R/gnn-lm_amd/lib/libgnnlm_hip.so $R/gnn-lm_amd/build/exp/libivf8_*.so; do
  [ -f $lib ] || continue
  echo "$(basename $lib): $(GNNLM_LIB=$lib python3 $R/tools/ivfpq_bench.py 2>&1 | grep -E 'ivfpq_scan8|IVF-PQ search' | tr '\n' ' ')"
done
