#!/bin/bash
# developer tool: builds a variant of libmi355pt.so with extra -D flags into gpurun_ab/lib_<name>.so (objects under /tmp), leaving the in-tree library alone.
# usage: tools/build_variant.sh <name> [-DFLAG=V ...]
set -e
name=$1; shift
R=/root/repo; D=/tmp/variant_$name; rm -rf $D; mkdir -p $D/rtxpt_amd $D/include $R/gpurun_ab
cp -r $R/rtxpt_amd/csrc $D/rtxpt_amd/csrc; cp $R/include/*.h $D/include/; rm -f $D/rtxpt_amd/csrc/*.o
make -C $D/rtxpt_amd/csrc -j8 EXTRA="$*" OUT=$R/gpurun_ab/lib_$name.so $R/gpurun_ab/lib_$name.so > $D/build.log 2>&1 || { tail -20 $D/build.log; exit 1; }
echo "built gpurun_ab/lib_$name.so ($*)"
