export VS_BAND_BENCH_VARIANTS='VS_BAND2_ABLATE=0,31,63,95,127,128'
export VS_BAND2_WM=1
python3 tools/band_bench.py fwd "--only=256->256 @8 dec,256->256 @16 dec,64->64 @64 dec,512->512 @4 dec,64->64 @32 dec" 2>&1 | grep -v amdgpu.ids
