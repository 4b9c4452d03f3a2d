# the forms the path takes while the GPU is shared, forced for whole campaigns (they are chosen from the load otherwise):
# K5's small round for every query (fallback to the full form when the set is larger), the coming round's hypotheses solved
# by the replaying workgroup, the shared-GPU round sizes, K3 with 4 waves per view
mkdir -p gpurun_out/r03_fuzz
O=gpurun_out/r03_fuzz
export SFMLOC_P3P_SMALL=2 SFMLOC_P3P_PREP_AHEAD=2 SFMLOC_P3P_ADAPTIVE=1 SFMLOC_K3_WAVES_ALONE=4 SFMLOC_K3_WAVES_SHARED=4
timeout -k 10 900 python tests/tools/fuzz_p3p_large.py 40 96000 > $O/fuzz_p3p_large_shared_forms.txt 2>&1; rc=$?; tail -1 $O/fuzz_p3p_large_shared_forms.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tests/tools/fuzz_parity.py 1000 182000 > $O/fuzz_parity_shared_forms.txt 2>&1; rc=$?; tail -1 $O/fuzz_parity_shared_forms.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_sharded.py 200 188000 > $O/fuzz_sharded_shared_forms.txt 2>&1; rc=$?; tail -1 $O/fuzz_sharded_shared_forms.txt; [ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/tools/fuzz_gang.py 600 191000 > $O/fuzz_gang_shared_forms.txt 2>&1; rc=$?; tail -1 $O/fuzz_gang_shared_forms.txt; [ $rc -eq 0 ] || exit $rc
