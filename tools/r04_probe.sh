#!/bin/bash
# Round-4 calibration pass on a GPU box: the vendor-GEMM yardstick (tools/vendor_gemm.py), its kernel names, the L2 hit rate of the
# GEMM families and this box's headline step.   tools/r04_probe.sh [tag]
set -u
TAG="${1:-r04}"; OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
python3 bench.py --lean --steps 60 --warmup 15 > $OUT/${TAG}_box_bench.json 2> $OUT/${TAG}_box_bench.err; echo "bench rc=$?"
python3 tools/vendor_gemm.py > $OUT/${TAG}_vendor_gemm.txt 2> $OUT/${TAG}_vendor_gemm.err; echo "vendor rc=$?"
rm -rf $OUT/${TAG}_p_names
rocprofv3 --kernel-trace --output-format csv -d $OUT/${TAG}_p_names -- python3 tools/vendor_gemm.py --names > $OUT/${TAG}_names.log 2>&1; echo "names rc=$?"
python3 tools/vendor_gemm.py --parse "$(ls $OUT/${TAG}_p_names/*/*kernel_trace.csv | head -1)" > $OUT/${TAG}_vendor_gemm_kernels.txt 2>&1; echo "parse rc=$?"
rm -rf $OUT/${TAG}_p_tcc
RGQA_WGRAD_SERIAL=1 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d $OUT/${TAG}_p_tcc -- python3 bench.py --steps 5 --warmup 2 --lean > $OUT/${TAG}_pmc_tcc.log 2>&1; echo "tcc rc=$?"
python3 tools/tcc_hit.py "$(ls $OUT/${TAG}_p_tcc/*/*counter_collection.csv | head -1)" $OUT/${TAG}_pmc_tcc_hit.json "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum, RGQA_WGRAD_SERIAL=1 python3 bench.py --steps 5 --warmup 2 --lean" > $OUT/${TAG}_pmc_tcc_hit.txt 2>&1; echo "tcc summary rc=$?"
rm -rf $OUT/${TAG}_p_names $OUT/${TAG}_p_tcc
ls $OUT | grep "^${TAG}_"
