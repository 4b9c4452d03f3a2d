# The randomised parity campaigns on the code as it stands (one gpurun call; copy the .txt files into profiles/ under the
# round's prefix).  ROUND=r04 tags the output directory; SEED0 moves every campaign to fresh seeds; FORMS=shared repeats the
# localisation / sharded / gang / large-set campaigns with the forms the path takes while the GPU is shared forced for every
# query (K5 small rounds + prepared hypotheses + shared-GPU round sizes, K3 with 4 waves per view); FORMS=wide forces one
# workgroup per model for every query (SFMLOC_P3P_WIDE_ALONE=2); FORMS=seq the sequential form of K5.
R=${ROUND:-r04}; S=${SEED0:-0}; F=${FORMS:-default}
O=gpurun_out/${R}_fuzz; mkdir -p $O
sfx=""
case $F in
  shared) export SFMLOC_P3P_SMALL=2 SFMLOC_P3P_PREP_AHEAD=2 SFMLOC_P3P_ADAPTIVE=1 SFMLOC_K3_WAVES_ALONE=4 SFMLOC_K3_WAVES_SHARED=4; sfx=_shared_forms;;
  wide) export SFMLOC_P3P_WIDE_ALONE=2; sfx=_wide_forms;;
  seq) export SFMLOC_P3P_SEQ=2 SFMLOC_P3P_SEQ_WAVES=${SEQ_WAVES:-8}; sfx=_sequential_form;;
esac
run() { name=$1; lim=$2; shift 2
  timeout -k 10 $lim "$@" > $O/fuzz_${name}${sfx}.txt 2>&1; rc=$?; tail -1 $O/fuzz_${name}${sfx}.txt; [ $rc -eq 0 ] || exit $rc; }
run p3p_large 900 python tests/tools/fuzz_p3p_large.py ${N_LARGE:-60} $((93000 + S))
run parity 900 python tests/tools/fuzz_parity.py ${N_PARITY:-1500} $((152000 + S))
run sharded 600 python tests/tools/fuzz_sharded.py ${N_SHARDED:-300} $((158000 + S))
run gang 600 python tests/tools/fuzz_gang.py ${N_GANG:-1000} $((131000 + S))
[ $F = default ] || exit 0
run akaze 600 python tests/tools/fuzz_akaze.py ${N_AKAZE:-1500} $((164000 + S))
SFMLOC_FUZZ_BATCH=1 run akaze_batches 600 python tests/tools/fuzz_akaze.py ${N_AKAZE_BATCH:-300} $((171000 + S))
run bow 600 python tests/tools/fuzz_bow.py ${N_BOW:-600}
run mapside 600 python tests/tools/fuzz_mapside.py ${N_MAPSIDE:-300}
run undistort 600 python tests/tools/fuzz_undistort.py ${N_UNDIST:-300}
