R=$PWD
for rep in 1 2 3; do for tree in r5 r6; do
  root=$R; [ $tree = r5 ] && root=$R/r5tree
  for mode in "ADAM=1 DIRECT=0" "ADAM=0 DIRECT=0" "ADAM=1 DIRECT=1"; do
    echo "$tree $mode unpinned: $(env $mode PKG_ROOT=$root PROFILE=0 BATCHES=5 N=200000 W=256 H=256 timeout 120 python $R/scripts/exp/map_iter.py 2>&1 | tail -1)"
  done
done; done
