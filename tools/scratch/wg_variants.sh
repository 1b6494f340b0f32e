for v in NO_MFMA NO_DUP; do
  export GSVC_LIB_PATH=$PWD/tools/scratch/libgsvc_$v.so
  echo "variant=$v"; python tools/scratch/wgrad_scale.py | grep -E "K="
done
