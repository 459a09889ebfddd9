for i in 1 2 3; do
  python tools/opt_sweep.py "" 2>&1 | grep poses | sed 's/^/new  /'
  FLOPE_AMD_LIB=$GRAFT_REPO_ROOT/build/prev/libprev.so python tools/opt_sweep.py "" 2>&1 | grep poses | sed 's/^/prev /'
done
