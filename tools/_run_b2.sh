export VS_BAND_BENCH_VARIANTS='VS_BAND_V2=0,1'
python3 tools/band_bench.py fwd --check 2>&1 | grep -v amdgpu.ids
python3 tools/band_bench.py k4 --check 2>&1 | grep -v amdgpu.ids
export VS_BAND_BENCH_VARIANTS='VS_BAND2_WM=1,2'
python3 tools/band_bench.py fwd --check 2>&1 | grep -v amdgpu.ids
python3 tools/band_bench.py k4 2>&1 | grep -v amdgpu.ids
export VS_BAND2_STAMP=1
export VS_BAND_BENCH_VARIANTS='VS_BAND2_WM=1'
python3 tools/band_bench.py fwd "--only=256->256 @8 dec" 2>&1 | grep stamp | head -8
