O=gpurun_out/${1:-r05s}; mkdir -p $O
(timeout -k 10 ${4:-280} python tools/fuzz_parity_sweep.py $((5000+${3:-0})) ${2:-120} 2>&1 | tail -3) > $O/fuzz_counting.txt
(PRODUCT=1 timeout -k 10 ${4:-280} python tools/fuzz_parity_sweep.py $((6000+${3:-0})) ${2:-120} 2>&1 | tail -3) > $O/fuzz_product.txt
(MI355RT_NO_LDS_STAGING=1 timeout -k 10 ${4:-280} python tools/fuzz_parity_sweep.py $((7000+${3:-0})) ${2:-120} 2>&1 | tail -3) > $O/fuzz_counting_global.txt
(PRODUCT=1 MI355RT_NO_LDS_STAGING=1 timeout -k 10 ${4:-280} python tools/fuzz_parity_sweep.py $((8000+${3:-0})) ${2:-120} 2>&1 | tail -3) > $O/fuzz_product_global.txt
(PRODUCT=1 MI355RT_NO_LDS_STAGING=1 MI355RT_WF_RAYREG=1 timeout -k 10 ${4:-280} python tools/fuzz_parity_sweep.py $((9000+${3:-0})) ${2:-120} 2>&1 | tail -3) > $O/fuzz_product_global_rayreg.txt
for f in $O/fuzz_*.txt; do echo "== $f"; cat $f; done
