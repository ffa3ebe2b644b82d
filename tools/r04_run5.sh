cd $GRAFT_REPO_ROOT
for w in cfg3_deepconn_electronics_e300 cfg4_narre_kindle cfg5_transnetpp_synthetic; do
  echo "== $w"
  bash tools/r04_ab.sh "pstr300 base pstr320" 2 --workload $w
done
