ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
for r in 1 2 3; do
 echo "new: $(python tools/bench_bake.py --no-cpu 2>/dev/null | tail -1 | cut -c95-112)"
 echo "old: $(T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_tex_old.so python tools/bench_bake.py --no-cpu 2>/dev/null | tail -1 | cut -c95-112)"
done
