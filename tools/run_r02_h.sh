mkdir -p gpurun_out
rm -f gpurun_out/akaze_fuse_sweep.txt
for ks in 4 8 12 16; do
  echo "FUSE_SMALL=$ks" | tee -a gpurun_out/akaze_fuse_sweep.txt
  SFMLOC_AKAZE_FUSE_SMALL=$ks timeout -k 10 120 python tools/akaze_time.py 2>&1 | tee -a gpurun_out/akaze_fuse_sweep.txt
done
timeout -k 10 300 python tests/tools/fuzz_akaze.py 200 > gpurun_out/fuzz_akaze.txt 2>&1; tail -2 gpurun_out/fuzz_akaze.txt
SFMLOC_AKAZE_FUSE_SMALL=16 timeout -k 10 300 python tests/tools/fuzz_akaze.py 100 > gpurun_out/fuzz_akaze16.txt 2>&1; tail -1 gpurun_out/fuzz_akaze16.txt
