for r in 1 2; do
for lib in "" tools/diag/libsgc_prev.so; do
echo "== lib=$lib cfg4"; SGC_DIAG_LIB=$lib SGC_TILE_CONFIGS="27,30,3,3,1,0,1,0,0,0" timeout 300 python tools/tile_bench.py cfg4 64x80 ring 2>&1 | grep "tile bin"
echo "== lib=$lib cfg2"; SGC_DIAG_LIB=$lib SGC_TILE_CONFIGS="16,22,3,3,0,0,1,0,0,1" timeout 300 python tools/tile_bench.py cfg2 64x80 ring 2>&1 | grep "tile bin"
done; done
