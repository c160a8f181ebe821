export TMPDIR=/tmp
OUT=gpurun_out
timeout 200 python -m pytest tests/test_gpu_conv_nhwc.py -q -x 2>&1 | tail -3 > $OUT/n8_tests.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/n8_pmc_f -o p -- python scripts/pmc_kernels.py run > /dev/null 2> $OUT/n8_pmc_f.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/n8_pmc_w -o p -- python scripts/pmc_kernels.py run > /dev/null 2> $OUT/n8_pmc_w.err
python scripts/pmc_kernels.py reduce $OUT/n8_pmc_f $OUT/n8_pmc_w > $OUT/n8_adapter_pmc.json 2> $OUT/n8_pmc_reduce.err
rm -rf $OUT/n8_pmc_f $OUT/n8_pmc_w
timeout 120 python scripts/kbench.py --what nhwc > $OUT/n8_kbench_nhwc.log 2>&1
cat $OUT/n8_tests.log
