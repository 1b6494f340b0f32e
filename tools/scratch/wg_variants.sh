for v in "$@"; do
  if [ "$v" = main ]; then unset GSVC_LIB_PATH; else export GSVC_LIB_PATH=$PWD/tools/scratch/libgsvc_$v.so; fi
  echo "variant=$v"; timeout -k 10 120 python tools/scratch/wgrad_scale.py | grep -E "K=" || exit 1
done
