# usage: bash tools/build_variant.sh NAME "-DMACRO=1 ..."  -> lattigo_amd/variants/libhering_NAME.so (A/B builds, select with HERING_LIB)
set -e
R=$(cd $(dirname $0)/.. && pwd); N=$1; shift
D=/tmp/hering_variant_$N; rm -rf $D; mkdir -p $D $R/lattigo_amd/variants
cp -r $R/lattigo_amd/csrc $D/csrc; mkdir -p $D/include; cp $R/include/*.h $D/include/; mkdir -p $D/x; 
sed -i 's#\.\./\.\./include#../include#g; s#OUT      = ../libhering.so#OUT = ../libhering.so#' $D/csrc/Makefile
sed -i 's#"\.\./\.\./include/#"../include/#' $D/csrc/api.cpp $D/csrc/replay.cpp
make -C $D/csrc -s -j4 clean >/dev/null; make -C $D/csrc -s -j4 EXTRA="$*" 2>&1 | grep -E "error" || true
cp $D/libhering.so $R/lattigo_amd/variants/libhering_$N.so; ls -la $R/lattigo_amd/variants/libhering_$N.so
