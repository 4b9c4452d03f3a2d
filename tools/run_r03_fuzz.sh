# round-3 randomised campaigns on the final code (copy the .txt files into profiles/)
mkdir -p gpurun_out/r03_fuzz
O=gpurun_out/r03_fuzz
timeout -k 10 900 python tests/tools/fuzz_p3p_large.py ${N_LARGE:-60} 93000 > $O/fuzz_p3p_large.txt 2>&1; rc=$?; tail -2 $O/fuzz_p3p_large.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tests/tools/fuzz_parity.py ${N_PARITY:-1500} 152000 > $O/fuzz_parity.txt 2>&1; rc=$?; tail -2 $O/fuzz_parity.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_sharded.py ${N_SHARDED:-300} 158000 > $O/fuzz_sharded.txt 2>&1; rc=$?; tail -2 $O/fuzz_sharded.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_gang.py ${N_GANG:-1000} 131000 > $O/fuzz_gang.txt 2>&1; rc=$?; tail -2 $O/fuzz_gang.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_akaze.py ${N_AKAZE:-1500} 164000 > $O/fuzz_akaze.txt 2>&1; rc=$?; tail -2 $O/fuzz_akaze.txt; [ $rc -eq 0 ] || exit $rc
SFMLOC_FUZZ_BATCH=1 timeout -k 10 600 python tests/tools/fuzz_akaze.py ${N_AKAZE_BATCH:-300} 171000 > $O/fuzz_akaze_batches.txt 2>&1; rc=$?; tail -2 $O/fuzz_akaze_batches.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_bow.py ${N_BOW:-600} > $O/fuzz_bow.txt 2>&1; rc=$?; tail -2 $O/fuzz_bow.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 300 ./tools/mfma_hamming_lab > $O/mfma_hamming_lab.jsonl 2>&1; cat $O/mfma_hamming_lab.jsonl
