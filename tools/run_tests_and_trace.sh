bash tools/run_gpu_tests.sh && bash tools/run_latency_trace.sh 2>&1 | grep -E "k_fmatrix|k_p3p_eval|k_p3p_select" | head -8
