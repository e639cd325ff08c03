# Builds a second copy of the library with extra blend flags for A/B runs on one GPU box:  tools/build_ab.sh "<flags>" [name]
# -> tools/ab/libgsrast_<name>.so (git-ignored, travels with gpurun).  Select with GSR_LIB_PATH.
set -e
R=$(cd $(dirname $0)/.. && pwd); N=${2:-b}
mkdir -p $R/tools/ab/obj_$N
cp $R/gs-sr_amd/csrc/*.hip $R/gs-sr_amd/csrc/*.h $R/gs-sr_amd/csrc/Makefile $R/tools/ab/obj_$N/
sed -i 's#\.\./\.\./include#'$R'/include#g' $R/tools/ab/obj_$N/*.h $R/tools/ab/obj_$N/*.hip $R/tools/ab/obj_$N/Makefile
make -s -C $R/tools/ab/obj_$N -j8 BLEND_EXTRA="$1" OUT=$R/tools/ab/libgsrast_$N.so
echo built $R/tools/ab/libgsrast_$N.so
