# same-run A/B of builds and engine options on the bench workload (one process per arm, arms interleaved over rounds):
#   tools/ab_libs.sh ROUNDS "lib.so|opts" "lib.so|opts" ...      (lib "-" = the in-tree library; opts as for tools/opt_sweep.py)
rounds=$1; shift
for i in $(seq 1 $rounds); do
  for arm in "$@"; do
    lib=${arm%%|*}; opts=${arm#*|}
    if [ "$lib" = "-" ]; then python tools/opt_sweep.py "$opts" 2>&1 | grep poses | awk -v a="$arm" '{print a "  " $0}'
    else FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/$lib python tools/opt_sweep.py "$opts" 2>&1 | grep poses | awk -v a="$arm" '{print a "  " $0}'; fi
  done
done
