# build a variant of the library with one source compiled with extra flags: bash tools/ab/build_variant.sh NAME file.hip -DFLAG ...
# -> tools/ab/libgsvc_NAME.so (use with GSVC_LIB_PATH)
NAME=$1; SRC=$2; shift 2
C=gsvc_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function "$@" -c $C/$SRC -o /tmp/variant_$NAME.o || exit 1
OBJS=$(ls $C/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/ab/libgsvc_$NAME.so $OBJS /tmp/variant_$NAME.o && echo built tools/ab/libgsvc_$NAME.so
