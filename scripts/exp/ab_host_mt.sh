R=$PWD
for rep in 1 2 3; do for mt in 1 0; do for mode in "ADAM=1 DIRECT=0" "ADAM=0 DIRECT=0"; do
    echo "MT=$mt $mode: $(env $mode MT=$mt PROFILE=0 BATCHES=5 N=200000 W=256 H=256 timeout 120 python $R/scripts/exp/map_iter.py 2>&1 | tail -1)"
done; done; done
