cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r05_c
timeout 1500 python -m pytest tests -m gpu -x -q -k "hall_of_10000 or 1536 or prebuilt" > gpurun_out/r05_c/sel.log 2>&1; echo "rc=$?" >> gpurun_out/r05_c/sel.log; tail -15 gpurun_out/r05_c/sel.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_parity.py::test_3000_particles_map_a_hall_of_10000_m2_with_in_place_resampling > gpurun_out/r05_c/rest.log 2>&1; echo "rc=$?" >> gpurun_out/r05_c/rest.log; tail -6 gpurun_out/r05_c/rest.log
