for lib in "" turn128 minw3 noslp; do
  for layers in 32,32 32,32,32,32; do
    if [ -z "$lib" ]; then unset PACOH_LIB; else export PACOH_LIB=$PWD/meta_learning_pacoh_amd/lib/libpacoh_gp_$lib.so; fi
    echo "== lib=${lib:-base} layers=$layers"; python tools/mlp_time.py --quick --layers $layers --reps 100 2>/dev/null
  done
done
