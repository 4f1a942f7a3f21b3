# GPU box: the dense one-view scene (full lat-long grid, views 12 and 4) under the shipped library and under a build whose
# depth-segmented backward also takes launches of more than 8,192 tiles (lib_segany.so: kSegMaxTiles = 100000)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
echo "== shipped"; python tools/micro/dense_poles.py 2>/dev/null | grep "full" | cut -c1-330
echo "== segments for any one-view launch"; T4D_SORT_256=1 T4D_LIB=$ROOT/topo4d_amd/csrc/variants/lib_segany.so python tools/micro/dense_poles.py 2>/dev/null | grep "full" | cut -c1-330
