# in-kernel time stamps of K3 / K5 on the headline workload (instrumented build of the library, tools only)
mkdir -p gpurun_out
timeout -k 10 200 env SFMLOC_LIB_PATH=$GRAFT_REPO_ROOT/sfmlocalization_amd/lib/libsfmloc_hip_stamps.so python tools/k35_stamps.py > gpurun_out/k35_stamps.txt 2>&1; rc=$?
tail -60 gpurun_out/k35_stamps.txt; exit $rc
