# Development helper: build the library once more with extra compiler flags into activesplat_amd/libgsplat_hip_<name>.so (objects in /tmp), for the
# A/B scripts (ab_libs.sh, ab_generic.sh, ab_c4.sh), which swap libraries inside ONE box session.
#   bash scripts/exp/build_variant.sh c4096 -DGS_BIN_CHUNK=4096
R=$PWD; NAME=$1; shift; O=/tmp/gs_variant_$NAME; mkdir -p $O
cd $R/activesplat_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function $*"
for f in *.hip; do b=${f%.hip}; x=""; [ $b = preprocess ] && x="-ffp-contract=off"; [ $b = blend ] && x="-munsafe-fp-atomics -fno-slp-vectorize"
  /opt/rocm/bin/hipcc $F $x -c $f -o $O/$b.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libgsplat_hip_$NAME.so $O/*.o && echo built libgsplat_hip_$NAME.so
