python3 tools/aten_trace.py taxibj > gpurun_out/r05_aten_taxibj.txt 2>gpurun_out/r05_aten.err
python3 tools/aten_trace.py sst > gpurun_out/r05_aten_sst.txt 2>>gpurun_out/r05_aten.err
python3 tools/aten_trace.py mnist_b128 > gpurun_out/r05_aten_mnist.txt 2>>gpurun_out/r05_aten.err
